"""Host-side program metrics of the full model: syntax check, exact-program comparison and
execution accuracy.  In the reference these are tf.py_func callbacks inside the graph
(models/model_full.py:602-616 check_correct_syntax, :713-729 exact_program_compare_karel,
:745-780 generate_program_output_karel, :878-901 CompareDemoAndExecution); they are logging /
evaluation metrics, never part of the loss, so they stay on the host here too and work on the
arrays the GPU path hands back (argmax tokens, lengths, is_same_seq).

Syntax and exact-program metrics exist for both DSLs (`parser_for`).  Execution is built in for
Karel; the ViZDoom variant (model_full.py:789-848) drives whatever world object the caller's
`world_factory` returns -- the game engine (`vizdoom`, `cv2`) is not part of this build, and
without a factory `require_env` raises for it.
"""
import numpy as np

from ..karel_env import Karel_world, parse


def parser_for(dataset_type):
    """The stack parser the reference imports per dataset type (model_full.py:603-606)."""
    if dataset_type == 'karel':
        return parse
    if dataset_type == 'vizdoom':
        from ..vizdoom_env.dsl import parse as parse_vizdoom
        return parse_vizdoom
    raise ValueError('unknown dataset_type %r' % (dataset_type,))


def require_env(dataset_type, world_factory=None):
    """Execution metrics need an environment: Karel's is built in, ViZDoom's is the caller's."""
    if dataset_type != 'karel' and world_factory is None:
        raise NotImplementedError(
            'execution metrics for dataset_type=%r need the ViZDoom engine (vizdoom_env/vizdoom_env.py), '
            'which this build does not ship: pass a world_factory, or ask for syntax / exact-program '
            'metrics only' % (dataset_type,))


def check_correct_syntax(vocab, p_token, p_len, is_same_seq, parse=parse):
    """[B] float32: 1 where the predicted token string is accepted by the stack parser; rows that
    already equal the ground truth are 1 without parsing (model_full.py:608-610)."""
    p_len = np.asarray(p_len).reshape(-1)
    out = np.zeros(len(p_len), np.float32)
    for i in range(len(p_len)):
        if is_same_seq[i]:
            out[i] = 1.0
        else:
            out[i] = 1.0 if parse(vocab.intseq2str(p_token[i, :int(p_len[i])])).ok else 0.0
    return out


def exact_program_compare(vocab, p_token, p_len, is_correct_syntax, gt_token, gt_len, parse=parse):
    """[B] float32: canonical form of the prediction == canonical form of the ground truth, for
    rows with correct syntax (model_full.py:713-729)."""
    p_len, gt_len = np.asarray(p_len).reshape(-1), np.asarray(gt_len).reshape(-1)
    out = np.zeros(len(p_len), np.float32)
    for i in range(len(p_len)):
        if is_correct_syntax[i] == 1:
            p = parse(vocab.intseq2str(p_token[i, :int(p_len[i])])).canonical()
            g = parse(vocab.intseq2str(gt_token[i, :int(gt_len[i])])).canonical()
            out[i] = float(p is not None and g is not None and p == g)
    return out


def generate_program_output(vocab, initial_states, max_demo_len, p_token, p_len, is_correct_syntax,
                            is_same_seq, make_error=True):
    """Runs each row's predicted program from each of its demos' first frames.
    initial_states [B, k, h, w, depth] -> (execution [B, k, max_demo_len, h, w, depth] float32,
    execution_len [B, k] int32); zero length where the program is skipped (already identical,
    bad syntax) or fails (model_full.py:745-780)."""
    B, k = initial_states.shape[:2]
    h, w, depth = initial_states.shape[2:]
    p_len = np.asarray(p_len).reshape(-1)
    execution = np.zeros((B, k, max_demo_len, h, w, depth), np.float32)
    execution_len = np.zeros((B, k), np.int32)
    for i in range(B):
        if is_same_seq[i] != 0 or is_correct_syntax[i] != 1:
            continue
        prog = parse(vocab.intseq2str(p_token[i, :int(p_len[i])]))
        if not prog.ok:
            raise RuntimeError("s_exe couldn't be False here")
        for d in range(k):
            world = Karel_world(initial_states[i, d], make_error=make_error)
            _, _, ok = prog.run(world)
            if ok:
                hist = np.stack(world.s_h, axis=0)
                execution_len[i, d] = hist.shape[0]
                n = min(hist.shape[0], max_demo_len)
                execution[i, d, :n] = hist[:n]
    return execution, execution_len


def generate_program_output_vizdoom(vocab, world_factory, init_pos, init_pos_len, pos_keys, max_demo_len, demo_k,
                                    h, w, depth, p_token, p_len, is_correct_syntax, is_same_seq):
    """ViZDoom counterpart (model_full.py:789-848): for every row that is neither identical to
    the ground truth nor a syntax error, start an episode per demonstration from its initial
    positions and run the predicted program.

    `world_factory()` -> a world with the reference Vizdoom_env surface the metric uses:
    `new_episode(init_dict)` (init_dict[key] = squeezed init_pos[i, d, p, :len]), the four DSL
    methods, `s_h` (list of [h, w, depth] frames) and optionally `init_game()` / `end_game()`.
    Frames must already have the dataset's size (the reference shrinks with cv2 INTER_AREA,
    which this build does not ship).  -> (execution [B, demo_k, max_demo_len, h, w, depth]
    float32, execution_len [B, demo_k] int32)."""
    from ..vizdoom_env.dsl import parse as parse_vizdoom
    B = p_token.shape[0]
    p_len = np.asarray(p_len).reshape(-1)
    execution = np.zeros((B, demo_k, max_demo_len, h, w, depth), np.float32)
    execution_len = np.zeros((B, demo_k), np.int32)
    world = world_factory()
    if hasattr(world, 'init_game'):
        world.init_game()
    try:
        for i in range(B):
            if is_same_seq[i] != 0 or is_correct_syntax[i] != 1:
                continue
            prog = parse_vizdoom(vocab.intseq2str(p_token[i, :int(p_len[i])]))
            if not prog.ok:
                raise RuntimeError('Compile failure should not happen here')
            for d in range(demo_k):
                init_dict = {key: np.squeeze(init_pos[i, d, p][:int(init_pos_len[i, d, p])])
                             for p, key in enumerate(pos_keys)}
                world.new_episode(init_dict)
                _, _, ok = prog.run(world)
                if not ok:
                    continue
                frames = [np.asarray(s) for s in world.s_h]
                if frames and frames[0].shape != (h, w, depth):
                    raise ValueError('world frames are %s, the dataset has %s: resize them in the world'
                                     % (frames[0].shape, (h, w, depth)))
                execution_len[i, d] = len(frames)
                n = min(len(frames), max_demo_len)
                if n:
                    execution[i, d, :n] = np.stack(frames[:n], axis=0)
    finally:
        if hasattr(world, 'end_game'):
            world.end_game()
    return execution, execution_len


def compare_demo_and_execution(demo, demo_len, execution, execution_len, is_same_program):
    """demo / execution [B, k, T, h, w, depth], lengths [B, k], is_same_program [B] ->
    (num_correct_execution [B] float32, is_correct_execution [B, k] bool, histogram [k+1] float32
    over how many of the k demos each program reproduces) (model_full.py:878-901)."""
    demo = np.asarray(demo)
    B, k = demo.shape[:2]
    same_exec = (demo == execution).reshape(B, k, -1).all(axis=-1)
    same_len = np.asarray(demo_len).reshape(B, k) == np.asarray(execution_len).reshape(B, k)
    is_correct = (same_exec & same_len) | (np.asarray(is_same_program).reshape(B, 1) != 0)
    num_correct = is_correct.sum(axis=1).astype(np.float32)
    hist = np.array([(num_correct == i).sum() / float(B) for i in range(k + 1)], np.float32)
    return num_correct, is_correct, hist
