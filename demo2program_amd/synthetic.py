"""Seeded synthetic batches shaped like the reference's input pipeline output.

Keys, shapes and dtypes are those of ``batch_chunk`` consumed by Model.get_feed_dict
(models/model_full.py:185-206; karel_env/input_ops_karel.py:69-75,90-103); padding rules are
those of karel_env/dataset_karel.py:38-115; value distributions follow the dataset
generators (karel_env/generator.py:183-190, karel_env/state_generator.py:18-33).
There is no network for the real 13 GB dataset, so benchmarks and parity tests run on these.
"""
import numpy as np


def _karel_frames(rs, n, T, h, w, depth, lens):
    """Karel state tensors: channels 0-3 hero heading, 4 wall, 5 'no marker', 6.. markers
    (karel_env/karel.py:6-23).  Values in {0,1}; frames past the demo length are zero."""
    s = np.zeros((n, T, h, w, depth), np.float32)
    for d in range(n):
        wall = rs.rand(h, w) < 0.1
        wall[0, :] = wall[-1, :] = True
        wall[:, 0] = wall[:, -1] = True
        marker = (rs.rand(h, w) < 0.1) & ~wall
        free = np.argwhere(~wall)
        if len(free) == 0:
            free = np.array([[h // 2, w // 2]])
            wall[h // 2, w // 2] = False
        pos = free[rs.randint(len(free))].copy()
        heading = rs.randint(4)
        for t in range(int(lens[d])):
            f = s[d, t]
            if depth > 4:
                f[:, :, 4] = wall
            if depth > 6:
                f[:, :, 5] = ~marker & ~wall
                f[:, :, 6] = marker
            f[pos[0], pos[1], heading % min(4, depth)] = 1
            # one random action between consecutive frames
            a = rs.randint(5)
            if a == 0:
                step = [(-1, 0), (0, 1), (1, 0), (0, -1)][heading]
                nxt = pos + np.array(step)
                if 0 <= nxt[0] < h and 0 <= nxt[1] < w and not wall[nxt[0], nxt[1]]:
                    pos = nxt
            elif a == 1:
                heading = (heading + 1) % 4
            elif a == 2:
                heading = (heading + 3) % 4
            elif a == 3:
                marker[pos[0], pos[1]] = False
            else:
                marker[pos[0], pos[1]] = True
    return s


def make_batch(config, seed=123, frames='auto'):
    """One batch_chunk (numpy arrays).  frames: 'karel' / 'uint8' / 'auto' (by dataset_type)."""
    rs = np.random.RandomState(seed)
    B, k, T, L = config.batch_size, config.k, config.max_demo_len, config.max_program_len
    tk = getattr(config, 'test_k', 5)
    V, A, P = config.dim_program_token, config.action_space, config.per_dim
    h, w, depth = config.h, config.w, config.depth
    if frames == 'auto':
        frames = 'uint8' if config.dataset_type == 'vizdoom' else 'karel'
    lo_demo = min(8, max(2, T // 2)) if config.dataset_type == 'karel' else 2
    lo_prog = min(5, L)

    def demo_part(nd):
        lens = rs.randint(lo_demo, T + 1, size=(B, nd))
        if frames == 'karel':
            s = _karel_frames(rs, B * nd, T, h, w, depth, lens.reshape(-1)).reshape(B, nd, T, h, w, depth)
        else:
            s = rs.randint(0, 256, size=(B, nd, T, h, w, depth)).astype(np.float32)
            mask = (np.arange(T)[None, None, :] < lens[:, :, None])
            s *= mask[..., None, None, None]
        a_tok = np.zeros((B, nd, T), np.int32)
        a_h = np.zeros((B, nd, T, A), np.float32)
        per = np.zeros((B, nd, T, P), np.float32)
        for b in range(B):
            for i in range(nd):
                n = lens[b, i]
                a_tok[b, i, :n - 1] = rs.randint(0, A - 1, size=n - 1)
                a_tok[b, i, n - 1] = A - 1                       # <e> (dataset_karel.py:74-76)
                a_h[b, i, np.arange(n), a_tok[b, i, :n]] = 1
                per[b, i, :n] = rs.randint(0, 2, size=(n, P))
                if P >= 5:
                    per[b, i, :n, 3] = 1 - per[b, i, :n, 4]
        return s, a_h, a_tok, per, lens.astype(np.float32)

    s_h, a_h, a_tok, per, demo_len = demo_part(k)
    test_s_h, test_a_h, test_a_tok, test_per, test_demo_len = demo_part(tk)

    program_len = rs.randint(lo_prog, L + 1, size=(B, 1))
    program_tokens = np.zeros((B, L), np.int32)
    program = np.zeros((B, V, L), np.float32)
    for b in range(B):
        n = int(program_len[b, 0])
        toks = rs.randint(0, V, size=n)
        if n >= 4:
            toks[:3] = [0, 1, 2]                                  # DEF run m(
            toks[-1] = 3                                          # m)
        program_tokens[b, :n] = toks
        program[b, toks, np.arange(n)] = 1                        # zero columns past len
    return {
        'id': np.array(['synthetic_%06d' % (seed * 1000 + b) for b in range(B)]),
        'program': program, 'program_tokens': program_tokens,
        's_h': s_h, 'test_s_h': test_s_h,
        'a_h': a_h, 'a_h_tokens': a_tok, 'test_a_h': test_a_h, 'test_a_h_tokens': test_a_tok,
        'program_len': program_len.astype(np.float32), 'demo_len': demo_len,
        'test_demo_len': test_demo_len, 'per': per, 'test_per': test_per,
    }


def to_torch(batch, device=None):
    """numpy batch -> torch tensors (strings dropped), optionally moved to ``device``."""
    import torch
    out = {}
    for n, v in batch.items():
        if v.dtype.kind in 'US':
            continue
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[n] = t.to(device) if device is not None else t
    return out
