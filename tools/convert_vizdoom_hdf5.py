#!/usr/bin/env python
"""Converts a reference ViZDoom dataset (datasets/<name>/data.hdf5 + id.txt, written by
vizdoom_env/generator.py:243-285) into the h5py-free layout
demo2program_amd.vizdoom_env.dataset_vizdoom reads.  Needs h5py, so run it with an interpreter
that has it (e.g. /opt/conda/bin/python3.9 in the build image); the training interpreter does not.

usage: convert_vizdoom_hdf5.py <dataset_dir with data.hdf5> <output_dir>

The published datasets are hundreds of GB of int16 frames, so frames are stored RAGGED and as
uint8: only the first s_h_len frames of each demonstration (the generator leaves the rest of its
padded array zero -- checked here), written through a memory map in two passes.
  ids.txt, data_info.json
  program.npy [N, Lmax] int8              program_len.npy [N] int16
  frames.npy [total_frames, h, w, c] uint8
  frame_offset.npy [N, D+Dt] int64        first frame of each demonstration (seen, then held-out)
  s_h_len.npy [N, D] int16                test_s_h_len.npy [N, Dt]
  a_h.npy [N, D, Amax] int8               a_h_len.npy [N, D] int16      a_h_width.npy [N] int16
                                          (the stored, padded width: the reader's one-hot quirk)
  p_v_h.npy [N, D, Tmax, P] bool          s_h_width.npy [N] int16       (test_* likewise)
  init_pos.npy [N, D, K, Pmax, 2] int32   init_pos_len.npy [N, D, K] int32   (test_* likewise)
"""
import json
import os
import sys

import h5py
import numpy as np


def val(x):
    v = x[()]
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, np.ndarray) and v.dtype.kind in 'SO':
        return [s.decode() if isinstance(s, bytes) else str(s) for s in v.tolist()]
    return v.item() if hasattr(v, 'item') and np.ndim(v) == 0 else (v.tolist() if hasattr(v, 'tolist') else v)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    f = h5py.File(os.path.join(src, 'data.hdf5'), 'r')
    with open(os.path.join(src, 'id.txt')) as fp:
        ids = [s.strip() for s in fp.readlines() if s.strip()]
    info = {k: val(f['data_info'][k]) for k in f['data_info'].keys()}
    N = len(ids)
    g0 = f[ids[0]]
    h, w, c = (int(v) for v in g0['s_h'].shape[2:])
    D, Dt = int(g0['s_h'].shape[0]), int(g0['test_s_h'].shape[0])
    K = int(g0['vizdoom_init_pos'].shape[1])
    P = int(g0['p_v_h'].shape[2])

    # pass 1: lengths
    out = {'program_len': np.zeros(N, np.int16),
           's_h_len': np.zeros((N, D), np.int16), 'test_s_h_len': np.zeros((N, Dt), np.int16),
           'a_h_len': np.zeros((N, D), np.int16), 'test_a_h_len': np.zeros((N, Dt), np.int16),
           's_h_width': np.zeros(N, np.int16), 'test_s_h_width': np.zeros(N, np.int16),
           'a_h_width': np.zeros(N, np.int16), 'test_a_h_width': np.zeros(N, np.int16)}
    pos_w = 0
    for n, id_ in enumerate(ids):
        g = f[id_]
        out['program_len'][n] = g['program'].shape[0]
        for pre in ('', 'test_'):
            out[pre + 's_h_len'][n] = g[pre + 's_h_len'][()]
            out[pre + 'a_h_len'][n] = g[pre + 'a_h_len'][()]
            out[pre + 's_h_width'][n] = g[pre + 's_h'].shape[1]
            out[pre + 'a_h_width'][n] = g[pre + 'a_h'].shape[1]
            pos_w = max(pos_w, g[pre + 'vizdoom_init_pos'].shape[2])
    L = int(out['program_len'].max())
    Tmax = int(max(out['s_h_width'].max(), out['test_s_h_width'].max()))
    Amax = int(max(out['a_h_width'].max(), out['test_a_h_width'].max()))
    Pmax = max(int(info.get('vizdoom_max_init_pos_len', pos_w)), pos_w)
    lens = np.concatenate([out['s_h_len'], out['test_s_h_len']], axis=1).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens.reshape(-1))])
    out['frame_offset'] = offsets[:-1].reshape(N, D + Dt)
    total = int(offsets[-1])

    out.update({
        'program': np.zeros((N, L), np.int8),
        'a_h': np.zeros((N, D, Amax), np.int8), 'test_a_h': np.zeros((N, Dt, Amax), np.int8),
        'p_v_h': np.zeros((N, D, Tmax, P), bool), 'test_p_v_h': np.zeros((N, Dt, Tmax, P), bool),
        'init_pos': np.zeros((N, D, K, Pmax, 2), np.int32), 'init_pos_len': np.zeros((N, D, K), np.int32),
        'test_init_pos': np.zeros((N, Dt, K, Pmax, 2), np.int32),
        'test_init_pos_len': np.zeros((N, Dt, K), np.int32)})
    frames = np.lib.format.open_memmap(os.path.join(dst, 'frames.npy'), mode='w+', dtype=np.uint8,
                                       shape=(total, h, w, c))
    # pass 2: payload
    for n, id_ in enumerate(ids):
        g = f[id_]
        p = g['program'][()]
        out['program'][n, :len(p)] = p
        for pre, col0 in (('', 0), ('test_', D)):
            s = g[pre + 's_h'][()]
            if s.min() < 0 or s.max() > 255:
                raise ValueError('%s/%ss_h has values outside 0..255' % (id_, pre))
            for d in range(s.shape[0]):
                m = int(lens[n, col0 + d])
                if s[d, m:].any():
                    raise ValueError('%s/%ss_h[%d] is not zero past its length' % (id_, pre, d))
                o = int(out['frame_offset'][n, col0 + d])
                frames[o:o + m] = s[d, :m]
            a = g[pre + 'a_h'][()]
            out[pre + 'a_h'][n, :, :a.shape[1]] = a
            q = g[pre + 'p_v_h'][()]
            out[pre + 'p_v_h'][n, :, :q.shape[1]] = q
            ip = g[pre + 'vizdoom_init_pos'][()]
            out[pre + 'init_pos'][n, :, :, :ip.shape[2]] = ip
            out[pre + 'init_pos_len'][n] = g[pre + 'vizdoom_init_pos_len'][()]
    frames.flush()
    del frames
    for k, v in out.items():
        np.save(os.path.join(dst, k + '.npy'), v)
    with open(os.path.join(dst, 'ids.txt'), 'w') as fp:
        fp.write('\n'.join(ids) + '\n')
    info.update(height=h, width=w, depth=c, num_programs=N, dataset_type='vizdoom')
    with open(os.path.join(dst, 'data_info.json'), 'w') as fp:
        json.dump(info, fp, indent=1, sort_keys=True)
    print('converted %d programs, %d frames -> %s' % (N, total, dst))


if __name__ == '__main__':
    main()
