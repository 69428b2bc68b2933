#!/usr/bin/env python
"""Converts a reference Karel dataset (datasets/<name>/data.hdf5 + id.txt, written by
karel_env/generator.py:129-153, append_demonstration.py:130-143, add_per.py:41-55) into the
h5py-free layout demo2program_amd.karel_env.dataset_karel reads.  Needs h5py, so run it with an
interpreter that has it (e.g. /opt/conda/bin/python3.9 in the build image); the training
interpreter does not.

usage: convert_karel_hdf5.py <dataset_dir with data.hdf5> <output_dir>

Output: whole-dataset arrays, one row per program in id.txt order, ragged axes zero-padded to the
dataset maxima, booleans bit-packed:
  ids.txt, data_info.json
  program.npy [N, Lmax] int8        program_len.npy [N] int16
  s_h.npy [N, D, Tmax, h*w*16/8] uint8 (np.packbits)   s_h_len.npy [N, D] int16
  a_h.npy [N, D, Tmax-1] int8       a_h_len.npy [N, D] int16        (test_* likewise, D -> Dtest)
  per.npy [N, D, Tmax, 5] uint8     test_per.npy
"""
import json
import os
import sys

import h5py
import numpy as np


def val(x):
    v = x[()]
    return v.decode() if isinstance(v, bytes) else v


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    f = h5py.File(os.path.join(src, 'data.hdf5'), 'r')
    with open(os.path.join(src, 'id.txt')) as fp:
        ids = [s.strip() for s in fp.readlines() if s.strip()]
    info = {k: val(f['data_info'][k]) for k in f['data_info'].keys()}
    info = {k: (v.item() if hasattr(v, 'item') else v) for k, v in info.items()}
    N = len(ids)
    g0 = f[ids[0]]
    h, w, c = g0['s_h'].shape[2:]
    D = max(f[i]['s_h'].shape[0] for i in ids)
    Dt = max(f[i]['test_s_h'].shape[0] for i in ids) if 'test_s_h' in g0 else 0
    T = max(max(f[i]['s_h'].shape[1] for i in ids), max(f[i]['test_s_h'].shape[1] for i in ids) if Dt else 0)
    L = max(f[i]['program'].shape[0] for i in ids)
    nb = (h * w * c + 7) // 8
    out = {
        'program': np.zeros((N, L), np.int8), 'program_len': np.zeros(N, np.int16),
        's_h': np.zeros((N, D, T, nb), np.uint8), 's_h_len': np.zeros((N, D), np.int16),
        'a_h': np.zeros((N, D, T - 1), np.int8), 'a_h_len': np.zeros((N, D), np.int16),
        'per': np.zeros((N, D, T, 5), np.uint8),
    }
    if Dt:
        out.update({'test_s_h': np.zeros((N, Dt, T, nb), np.uint8), 'test_s_h_len': np.zeros((N, Dt), np.int16),
                    'test_a_h': np.zeros((N, Dt, T - 1), np.int8), 'test_a_h_len': np.zeros((N, Dt), np.int16),
                    'test_per': np.zeros((N, Dt, T, 5), np.uint8)})
    for n, id_ in enumerate(ids):
        g = f[id_]
        p = g['program'][()]
        out['program'][n, :len(p)] = p
        out['program_len'][n] = len(p)
        for pre in (('', 'test_') if Dt else ('',)):
            s = g[pre + 's_h'][()]
            out[pre + 's_h'][n, :s.shape[0], :s.shape[1]] = np.packbits(
                s.reshape(s.shape[0], s.shape[1], -1).astype(np.uint8), axis=-1)
            out[pre + 's_h_len'][n, :s.shape[0]] = g[pre + 's_h_len'][()]
            a = g[pre + 'a_h'][()]
            out[pre + 'a_h'][n, :a.shape[0], :a.shape[1]] = a
            out[pre + 'a_h_len'][n, :a.shape[0]] = g[pre + 'a_h_len'][()]
            key = pre + 'p_v_h' if (pre + 'p_v_h') in g else pre + 'per'
            if key in g:
                q = g[key][()]
                out[pre + 'per'][n, :q.shape[0], :q.shape[1]] = q.astype(np.uint8)
    for k, v in out.items():
        np.save(os.path.join(dst, k + '.npy'), v)
    with open(os.path.join(dst, 'ids.txt'), 'w') as fp:
        fp.write('\n'.join(ids) + '\n')
    info.update(height=int(h), width=int(w), depth=int(c), num_programs=N)
    with open(os.path.join(dst, 'data_info.json'), 'w') as fp:
        json.dump(info, fp, indent=1, sort_keys=True)
    print('converted %d programs -> %s' % (N, dst))


if __name__ == '__main__':
    main()
