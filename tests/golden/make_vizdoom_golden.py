#!/opt/conda/bin/python3.9
"""Builds the ViZDoom fixtures by RUNNING THE REFERENCE's own code (build container only: needs
/root/reference and /opt/conda/bin/python3.9 with h5py and numpy 1.26).

  tests/golden/vizdoom_dsl.json
      vocab tables of vizdoom_env/dsl/vocab.py for the perception types this build supports;
      per program string: accepted by vizdoom_env/dsl/dsl_parse.py parse()?  canonical program of
      vizdoom_env/dsl/dsl_enum_program.py parse() (length + SHA-1, full list when short);
      executions of the compiled closure on a SCRIPTED world (class ScriptWorld below -- the game
      engine is not installable here; the DSL only needs the four world methods): success flag,
      call counter, SHA-1 of the call log.
  tests/golden/vizdoom_dataset/  +  tests/golden/vizdoom_dataset_expected.npz
      a small data.hdf5 in the layout vizdoom_env/generator.py:243-285 writes (synthetic content,
      8x6 frames; the engine that renders real ones is absent), the reference reader's
      Dataset.get_data() / all_ids() results on it, and the converted copy
      (tools/convert_vizdoom_hdf5.py) this build's reader takes.

Shims, nothing edited or copied: sys.path for the Python-2 implicit-relative imports; a
list-returning `zip` in the namespace the two py2 parser files are exec'd in; a `colorlog` stub;
h5py's removed `Dataset.value`.
"""
import builtins
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

np.bool = bool
import h5py  # noqa: E402


def _value(self):
    v = self[()]
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray) and v.dtype.kind in 'SO':
        return np.array([s.decode() if isinstance(s, bytes) else s for s in v.tolist()])
    return v


h5py.Dataset.value = property(_value)
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REF, os.path.join(REF, 'vizdoom_env', 'dsl')]

cl = types.ModuleType('colorlog')


class _Fmt(object):
    def __init__(self, *a, **k):
        import logging
        self._f = logging.Formatter('%(message)s')

    def __getattr__(self, name):
        return getattr(self._f, name)


cl.ColoredFormatter = _Fmt
sys.modules['colorlog'] = cl


def load_py2(path, name):
    mod = types.ModuleType(name)
    mod.__dict__['zip'] = lambda *a: list(builtins.zip(*a))
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), mod.__dict__)
    return mod


dsl_parse = load_py2(os.path.join(REF, 'vizdoom_env', 'dsl', 'dsl_parse.py'), 'dsl_parse')
sys.modules['dsl_parse'] = dsl_parse                 # vocab.py: `from dsl_parse import ...`
dsl_enum = load_py2(os.path.join(REF, 'vizdoom_env', 'dsl', 'dsl_enum_program.py'), 'ref_dsl_enum')
from vocab import VizDoomDSLVocab  # noqa: E402  (reference)


class ScriptWorld(object):
    """Deterministic stand-in for Vizdoom_env: percepts answer from a seeded stream, actions fail
    once `fail_after` calls were logged.  tests/test_vizdoom_env.py holds the same class."""

    def __init__(self, seed, fail_after):
        self.rs = np.random.RandomState(seed)
        self.fail_after = fail_after
        self.log = []

    def state_transition(self, action):
        if len(self.log) >= self.fail_after:
            raise RuntimeError('dead')
        self.log.append(action)

    def _percept(self, *key):
        v = bool(self.rs.randint(2))
        self.log.append('?%s=%d' % (' '.join(key), v))
        return v

    def exist_actor_in_distance_horizontal(self, actor, dist, horz):
        return self._percept('exist', actor, dist, horz)

    def in_target(self, actor):
        return self._percept('in_target', actor)

    def is_there(self, actor):
        return self._percept('is_there', actor)


ACTIONS = ['MOVE_FORWARD', 'MOVE_BACKWARD', 'MOVE_LEFT', 'MOVE_RIGHT', 'TURN_LEFT', 'TURN_RIGHT', 'ATTACK',
           'SELECT_WEAPON1', 'SELECT_WEAPON3', 'SELECT_WEAPON4', 'SELECT_WEAPON5']
MONSTERS = ['Demon', 'HellKnight', 'Revenant']


def sample_program(rs, max_depth):
    def percept():
        r = rs.randint(10)
        if r < 4:
            return 'ISTHERE ' + MONSTERS[rs.randint(3)]
        if r < 8:
            return 'INTARGET ' + MONSTERS[rs.randint(3)]
        if r < 9:
            return 'ISTHERE MyAmmo'
        return 'EXIST %s IN %s %s' % (MONSTERS[rs.randint(3)], ['far', 'mid', 'close', 'doncare_dist'][rs.randint(4)],
                                      ['center', 'left', 'slight_right', 'doncare_horz'][rs.randint(4)])

    def cond():
        c = percept()
        while rs.rand() < 0.25:
            c = 'not c( %s c)' % c
        return c

    def stmt(depth):
        out = []
        for _ in range(rs.randint(1, 4)):
            r = rs.rand()
            if depth >= max_depth or r < 0.4:
                out.append(ACTIONS[rs.randint(len(ACTIONS))])
            elif r < 0.55:
                out.append('IF c( %s c) i( %s i)' % (cond(), stmt(depth + 1)))
            elif r < 0.7:
                out.append('IFELSE c( %s c) i( %s i) ELSE e( %s e)' % (cond(), stmt(depth + 1), stmt(depth + 1)))
            elif r < 0.85:
                out.append('WHILE c( %s c) w( %s w)' % (cond(), stmt(max_depth)))
            else:
                out.append('REPEAT R=%d r( %s r)' % (rs.randint(0, 7), stmt(depth + 1)))
        return ' '.join(out)

    return 'DEF run m( %s m)' % stmt(0)


def dsl_cases():
    rs = np.random.RandomState(11)
    programs, seen = [], set()
    while len(programs) < 80:
        code = sample_program(rs, int(rs.randint(1, 4)))
        if code in seen or len(code.split()) > 60:
            continue
        seen.add(code)
        programs.append(code)
    hand = [
        'DEF run m( ATTACK m)', 'ATTACK', 'MOVE_FORWARD TURN_LEFT', 'DEF run m( m)', 'DEF run m( ATTACK',
        'm) ATTACK m( run DEF', 'Demon', 'MyAmmo', 'far', 'center', 'S=3', 'R=5', 'ISTHERE Demon',
        'not c( INTARGET Revenant c)', 'EXIST Demon IN mid_far slight_left', 'EXIST MyAmmo IN far',
        'INTARGET far', 'ISTHERE', 'IF c( ISTHERE Demon c) i( ATTACK i)',
        'WHILE c( ISTHERE Demon c) w( ATTACK w)', 'REPEAT R=3 r( ATTACK r)', 'REPEAT R=3 r( ATTACK r) ATTACK',
        'DEF run m( SELECT_WEAPON2 m)', 'DEF run m( REPEAT R=19 r( TURN_LEFT r) m)',
        'DEF run m( REPEAT R=0 r( TURN_LEFT r) m)', 'DEF run m( REPEAT R=20 r( TURN_LEFT r) m)',
        'DEF run m( WHILE c( not c( not c( ISTHERE HellKnight c) c) c) w( MOVE_LEFT w) m)',
        'DEF run m( IFELSE c( INTARGET Demon c) i( ATTACK i) ELSE e( ATTACK e) m)',
        'DEF run m( IFELSE c( not c( INTARGET Demon c) c) i( ATTACK i) ELSE e( MOVE_LEFT e) m)',
        'DEF run m( IFELSE c( INTARGET Demon c) i( ATTACK i) ELSE e( MOVE_LEFT e) m)',
        'DEF run m( IF c( INTARGET Demon c) i( ATTACK i) IF c( not c( INTARGET Demon c) c) i( MOVE_LEFT i) m)',
        'DEF run m( IF c( EXIST Revenant IN close mid_right c) i( ATTACK i) m)',
        'DEF run m( WHILE c( ISTHERE Demon c) w( WHILE c( INTARGET Demon c) w( ATTACK w) w) m)',
        'DEF run m( ' + ' '.join(['ATTACK'] * 60) + ' m)',
        'DEF run m( ' + ' '.join(['MOVE_FORWARD'] * 120) + ' m)',
        'DEF run m( REPEAT R=6 r( REPEAT R=6 r( REPEAT R=6 r( ATTACK r) r) r) m)',
        'DEF run m( DEF run m( ATTACK m) m)', 'DEF run m( ATTACK m) DEF run m( ATTACK m)',
        'DEF run m( IF c( ISTHERE Demon c) i( i) m)', 'DEF run m( IF c( c) i( ATTACK i) m)',
        'DEF run m( WHILE c( ISTHERE Demon c) i( ATTACK i) m)', 'DEF run m( ELSE m)', 'DEF run m( not m)',
    ]
    soup = []
    vocab = VizDoomDSLVocab(perception_type='simple', level='not_simple').int2token
    for code in programs[:40]:
        toks = code.split()
        kind = rs.randint(3)
        if kind == 0:
            del toks[rs.randint(len(toks))]
        elif kind == 1:
            toks[rs.randint(len(toks))] = vocab[rs.randint(len(vocab))]
        else:
            i, j = rs.randint(len(toks)), rs.randint(len(toks))
            toks[i], toks[j] = toks[j], toks[i]
        soup.append(' '.join(toks))
    cases = []
    for code in programs + hand + soup:
        exe, ok = dsl_parse.parse(code)
        case = {'code': code, 'ok': bool(ok)}
        if ok:
            canon, _ = dsl_enum.parse(code)
            if isinstance(canon, list):
                blob = json.dumps(canon, separators=(',', ':')).encode()
                case['canonical_len'] = len(canon)
                case['canonical_sha1'] = hashlib.sha1(blob).hexdigest()
                if len(canon) <= 40:
                    case['canonical'] = canon
            else:
                case['canonical_len'] = None
            runs = []
            for seed, fail_after in ((0, 10 ** 9), (1, 10 ** 9), (2, 7), (3, 40)):
                world = ScriptWorld(seed, fail_after)
                try:
                    _, n, success = exe(world, 0)
                    runs.append({'seed': seed, 'fail_after': fail_after, 'success': bool(success), 'n': int(n),
                                 'calls': len(world.log),
                                 'log_sha1': hashlib.sha1('\n'.join(world.log).encode()).hexdigest()})
                except (TypeError, ValueError):          # roots that are not statements
                    runs.append({'seed': seed, 'fail_after': fail_after, 'raises': True})
            case['runs'] = runs
        cases.append(case)
    vocabs = {}
    for ptype, level in (('simple', 'not_simple'), ('more_simple', 'not_simple'), ('simple', 'simple'),
                         ('simple', None)):
        v = VizDoomDSLVocab(perception_type=ptype, level=level)
        vocabs['%s/%s' % (ptype, level)] = {'int2token': list(v.int2token), 'action_int2token': list(v.action_int2token)}
    with open(os.path.join(HERE, 'vizdoom_dsl.json'), 'w') as f:
        json.dump({'vocabs': vocabs, 'cases': cases}, f, indent=0, sort_keys=True)
    print('%d DSL cases, %d accepted' % (len(cases), sum(c['ok'] for c in cases)))


def write_hdf5(dir_name):
    """Synthetic content in the generator's layout (vizdoom_env/generator.py:182-285)."""
    rs = np.random.RandomState(5)
    vocab = VizDoomDSLVocab(perception_type='simple', level='not_simple')
    D, Dt, h, w, c, K = 4, 2, 6, 8, 3, 2
    percepts = ['ISTHERE ' + m for m in MONSTERS] + ['INTARGET ' + m for m in MONSTERS]
    pos_keys = ['player_pos', 'demon_pos']
    os.makedirs(dir_name)
    f = h5py.File(os.path.join(dir_name, 'data.hdf5'), 'w')
    ids = []
    max_demo_len, max_prog_len, max_pos = 0, 0, 0
    for count in range(9):
        code = sample_program(rs, 2)
        while any(t not in vocab.token2int for t in code.split()):      # EXIST / R=0 ... are not 'simple' tokens
            code = sample_program(rs, 2)
        program_seq = np.array(vocab.str2intseq(code), dtype=np.int8)
        num_demo = D + Dt
        len_s_h = rs.randint(2, 9, size=num_demo).astype(np.int16)
        demos_s_h = np.zeros([num_demo, np.max(len_s_h), h, w, c], dtype=np.int16)
        for i in range(num_demo):
            demos_s_h[i, :len_s_h[i]] = rs.randint(0, 256, size=(len_s_h[i], h, w, c))
        len_a_h = (len_s_h - 1).astype(np.int16)
        demos_a_h = np.zeros([num_demo, np.max(len_a_h)], dtype=np.int8)
        for i in range(num_demo):
            demos_a_h[i, :len_a_h[i]] = rs.randint(0, len(vocab.action_int2token), size=len_a_h[i])
        demos_p_v_h = np.zeros([num_demo, np.max(len_s_h), len(percepts)], dtype=bool)
        for i in range(num_demo):
            demos_p_v_h[i, :len_s_h[i]] = rs.randint(0, 2, size=(len_s_h[i], len(percepts)))
        pos_len = rs.randint(1, 4, size=(num_demo, K)).astype(np.int32)
        pos = np.zeros([num_demo, K, pos_len.max(), 2], dtype=np.int32)
        for i in range(num_demo):
            for p in range(K):
                pos[i, p, :pos_len[i, p]] = rs.randint(-500, 500, size=(pos_len[i, p], 2))
        max_demo_len = max(max_demo_len, int(np.max(len_s_h)))
        max_prog_len = max(max_prog_len, program_seq.shape[0])
        max_pos = max(max_pos, int(pos_len.max()))
        id_ = 'no_{}_prog_len_{}_max_s_h_len_{}'.format(count, program_seq.shape[0], np.max(len_s_h))
        ids.append(id_)
        grp = f.create_group(id_)
        grp['program'] = program_seq
        grp['s_h_len'], grp['s_h'] = len_s_h[:D], demos_s_h[:D]
        grp['a_h_len'], grp['a_h'], grp['p_v_h'] = len_a_h[:D], demos_a_h[:D], demos_p_v_h[:D]
        grp['test_s_h_len'], grp['test_s_h'] = len_s_h[D:], demos_s_h[D:]
        grp['test_a_h_len'], grp['test_a_h'], grp['test_p_v_h'] = len_a_h[D:], demos_a_h[D:], demos_p_v_h[D:]
        grp['vizdoom_init_pos'], grp['vizdoom_init_pos_len'] = pos[:D], pos_len[:D]
        grp['test_vizdoom_init_pos'], grp['test_vizdoom_init_pos_len'] = pos[D:], pos_len[D:]
    grp = f.create_group('data_info')
    grp['max_demo_length'] = max_demo_len + 1          # room for the <e> row past the widest a_h
    grp['max_program_length'] = max_prog_len
    grp['num_program_tokens'] = len(vocab.int2token)
    grp['num_demo_per_program'] = D
    grp['num_test_demo_per_program'] = Dt
    grp['num_action_tokens'] = len(vocab.action_int2token)
    grp['num_train'], grp['num_test'], grp['num_val'] = 5, 2, 2
    grp['s_h_h'], grp['s_h_w'], grp['s_h_c'] = h, w, c
    grp['percepts'] = [p.encode() for p in percepts]
    grp['vizdoom_pos_keys'] = [p.encode() for p in pos_keys]
    grp['vizdoom_max_init_pos_len'] = max_pos
    grp['perception_type'] = 'simple'
    f.close()
    with open(os.path.join(dir_name, 'id.txt'), 'w') as fp:
        fp.write(''.join(i + '\n' for i in ids))


def dataset_cases():
    tmp = tempfile.mkdtemp()
    src = os.path.join(tmp, 'vizdoom_dataset')
    write_hdf5(src)
    import vizdoom_env.dataset_vizdoom as ref_ds                 # reference reader
    splits = ref_ds.all_ids(src)
    expected = {'ids_train': np.array(splits[0]), 'ids_test': np.array(splits[1]), 'ids_val': np.array(splits[2])}
    for num_k in (4, 3):
        ds = ref_ds.Dataset(splits[0] + splits[1] + splits[2], src, name='all', num_k=num_k)
        for id_ in ds.ids:
            for j, arr in enumerate(ds.get_data(id_)):
                expected['k%d/%s/%d' % (num_k, id_, j)] = np.asarray(arr)
        expected['k%d/meta' % num_k] = np.array(json.dumps({
            'vizdoom_pos_keys': [str(s) for s in ds.vizdoom_pos_keys], 'perception_type': str(ds.perception_type),
            'level': str(ds.level), 'max_demo_len': ds.max_demo_len, 'k': ds.k, 'test_k': ds.test_k,
            'vizdoom_max_init_pos_len': ds.vizdoom_max_init_pos_len}))
    np.savez_compressed(os.path.join(HERE, 'vizdoom_dataset_expected.npz'), **expected)
    dst = os.path.join(HERE, 'vizdoom_dataset')
    shutil.rmtree(dst, ignore_errors=True)
    subprocess.check_call([sys.executable, os.path.join(REPO, 'tools', 'convert_vizdoom_hdf5.py'), src, dst])
    shutil.rmtree(tmp)
    print('dataset fixture: %d arrays' % len(expected))


if __name__ == '__main__':
    dsl_cases()
    dataset_cases()
