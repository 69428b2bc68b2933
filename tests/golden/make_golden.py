"""Generates the committed golden fixtures under tests/golden/.

The reference itself cannot be executed here (TensorFlow 1.3 is not installable; see
oracle/__init__.py), so the vectors are produced by the CPU oracle in fp32 -- inputs come
from the seeded synthetic generator, weights from the seeded initialiser (+ perturbed
biases/gammas).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from helpers import run_oracle, small_case   # noqa: E402


def main():
    seed = 7
    cfg, params, batch = small_case('karel', seed=seed)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float32)
    blob = {'seed': np.asarray(seed), 'loss': out['loss'].numpy(),
            'program_loss': out['program_loss'].numpy(),
            'avg_action_loss': out['avg_action_loss'].numpy(),
            'avg_per_loss': out['avg_per_loss'].numpy(),
            'pred_program': out['pred_program'].numpy(),
            'pred_action': out['pred_action'].permute(0, 1, 3, 2).contiguous().numpy(),
            'pred_per': out['pred_per'].permute(0, 1, 3, 2).contiguous().numpy()}
    for n, v in params.items():
        blob['param/' + n] = v
        blob['grad/' + n] = grads[n].numpy()
    for n, v in batch.items():
        if v.dtype.kind not in 'US':
            blob['batch/' + n] = v
    path = os.path.join(ROOT, 'tests', 'golden', 'karel_small.npz')
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
