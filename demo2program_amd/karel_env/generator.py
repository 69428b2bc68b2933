"""Karel worlds, random programs and batches of (program, demonstrations) in the layout the
model is fed with.  Restates karel_env/state_generator.py:13-33 (world sampling: same RNG call
order, so a seed gives the reference's worlds -- pinned by tests/golden/karel_dsl.json),
the acceptance loop of karel_env/generator.py:76-108 (a program is kept once enough of its
executions succeed with an admissible length) and the padding / one-hot rules of
karel_env/dataset_karel.py:38-115.  The program sampler is this build's own (the reference
samples from its yacc grammar, dsl_prob.py); it only has to produce valid programs.
"""
import numpy as np

from .dsl import KarelVocab, parse
from .karel import Karel_world

_ACTIONS = ('move', 'turnLeft', 'turnRight', 'pickMarker', 'putMarker')
_CONDS = ('frontIsClear', 'leftIsClear', 'rightIsClear', 'markersPresent', 'noMarkersPresent')


class KarelStateGenerator(object):
    def __init__(self, seed=None):
        self.rng = np.random.RandomState(seed)

    def generate_single_state(self, h=8, w=8, wall_prob=0.1):
        """-> (state [h,w,16] bool, karel row, karel col, #walls, #markers)."""
        rng = self.rng
        s = np.zeros((h, w, 16), bool)
        s[:, :, 4] = rng.rand(h, w) > 1 - wall_prob
        s[0, :, 4] = s[h - 1, :, 4] = True
        s[:, 0, 4] = s[:, w - 1, 4] = True
        while True:
            y = rng.randint(0, h)
            x = rng.randint(0, w)
            if not s[y, x, 4]:
                s[y, x, rng.randint(0, 4)] = True
                break
        s[:, :, 6] = (rng.rand(h, w) > 0.9) & ~s[:, :, 4]
        s[:, :, 5] = ~s[:, :, 6:].any(axis=-1)
        return s, y, x, int(s[:, :, 4].sum()), int(s[:, :, 6].sum())


def random_program(rng, max_stmts=4, max_depth=2):
    """A syntactically valid program string (accepted by dsl.parse and reducing to `prog`)."""
    def cond():
        c = _CONDS[rng.randint(len(_CONDS))]
        return 'not c( %s c)' % c if rng.rand() < 0.2 else c

    def block(depth):
        out = []
        for _ in range(rng.randint(1, max_stmts + 1)):
            r = rng.rand()
            if depth >= max_depth or r < 0.6:
                out.append(_ACTIONS[rng.randint(len(_ACTIONS))])
            elif r < 0.7:
                out.append('IF c( %s c) i( %s i)' % (cond(), block(depth + 1)))
            elif r < 0.8:
                out.append('IFELSE c( %s c) i( %s i) ELSE e( %s e)' % (cond(), block(depth + 1), block(depth + 1)))
            elif r < 0.9:
                out.append('WHILE c( %s c) w( %s w)' % (cond(), block(depth + 1)))
            else:
                out.append('REPEAT R=%d r( %s r)' % (rng.randint(2, 6), block(depth + 1)))
        return ' '.join(out)

    return 'DEF run m( %s m)' % block(0)


def demonstrate(code, state, make_error=True):
    """Executes `code` from `state`; -> (success, world) with world.s_h / a_h / p_v_h filled."""
    world = Karel_world(state, make_error=make_error)
    _, _, ok = parse(code).run(world)
    return ok, world


def sample_batch(config, seed=123, min_demo_len=2, max_tries=200, programs=None):
    """One batch_chunk (numpy, keys and dtypes of karel_env/input_ops_karel.py:69-75 after
    dataset_karel.py:38-115) of REAL programs with k seen and test_k held-out demonstrations each:
    frames are executions of the program, `a_h_tokens` its actions + <e>, `per` the perception
    vector of every state.  `programs`: optional list of code strings to cycle through."""
    rng = np.random.RandomState(seed)
    vocab = KarelVocab()
    s_gen = KarelStateGenerator(seed=seed + 1)
    B, k, tk = config.batch_size, config.k, getattr(config, 'test_k', 5)
    T, L = config.max_demo_len, config.max_program_len
    h, w, depth = config.h, config.w, config.depth
    V, A, P = config.dim_program_token, config.action_space, config.per_dim
    assert depth == 16 and V == len(vocab.int2token) and A == 6 and P == 5, 'Karel shapes expected'
    nd = k + tk
    out = {
        'id': np.array(['synthetic_%d_%d' % (seed, i) for i in range(B)]),
        'program': np.zeros((B, V, L), np.float32), 'program_tokens': np.zeros((B, L), np.int32),
        'program_len': np.zeros((B, 1), np.float32),
        's_h': np.zeros((B, k, T, h, w, depth), np.float32), 'test_s_h': np.zeros((B, tk, T, h, w, depth), np.float32),
        'a_h': np.zeros((B, k, T, A), np.float32), 'test_a_h': np.zeros((B, tk, T, A), np.float32),
        'a_h_tokens': np.zeros((B, k, T), np.int32), 'test_a_h_tokens': np.zeros((B, tk, T), np.int32),
        'demo_len': np.zeros((B, k), np.float32), 'test_demo_len': np.zeros((B, tk), np.float32),
        'per': np.zeros((B, k, T, P), np.float32), 'test_per': np.zeros((B, tk, T, P), np.float32),
    }
    codes = []
    b = 0
    cursor = 0
    while b < B:
        if programs is not None:
            code = programs[cursor % len(programs)]
            cursor += 1
        else:
            code = random_program(rng)
        ids = vocab.str2intseq(code)
        if len(ids) > L:
            continue
        demos = []
        for _ in range(max_tries):
            state = s_gen.generate_single_state(h, w, 0.1)[0]
            ok, world = demonstrate(code, state)
            if ok and min_demo_len <= len(world.s_h) <= T:
                demos.append(world)
                if len(demos) == nd:
                    break
        if len(demos) < nd:
            if programs is not None:
                raise ValueError('program %r yields too few admissible demonstrations' % code)
            continue
        codes.append(code)
        out['program_tokens'][b, :len(ids)] = ids
        out['program'][b, ids, np.arange(len(ids))] = 1.0
        out['program_len'][b, 0] = len(ids)
        for d, world in enumerate(demos):
            pre, j = ('', d) if d < k else ('test_', d - k)
            n = len(world.s_h)
            out[pre + 's_h'][b, j, :n] = np.stack(world.s_h)
            out[pre + 'demo_len'][b, j] = n
            out[pre + 'a_h_tokens'][b, j, :n - 1] = world.a_h
            out[pre + 'a_h_tokens'][b, j, n - 1] = A - 1                     # <e>
            out[pre + 'a_h'][b, j, np.arange(n), out[pre + 'a_h_tokens'][b, j, :n]] = 1.0
            out[pre + 'per'][b, j, :n] = np.stack(world.p_v_h)
        b += 1
    out['codes'] = np.array(codes)
    return out
