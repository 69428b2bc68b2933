#!/usr/bin/env python
"""Per-tensor gradient error of the HIP path against the fp64 oracle at BASELINE config 2's full size (the numbers
behind the tolerance of tests/test_model_gpu.py::test_headline_config_matches_oracle_at_full_size)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from helpers import run_oracle
from demo2program_amd.config import make_config
from demo2program_amd.karel_env.generator import sample_batch
from demo2program_amd.models.model_full import Model
from demo2program_amd.params import init_params
cfg = make_config('karel')
params = init_params(cfg, 123)
batch = sample_batch(cfg, seed=11)
model = Model(cfg, params=params)
loss = float(model.forward(model.get_feed_dict(batch)).item())
model.backward()
torch.set_num_threads(16)
out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
got = model.params.to_numpy('g')
for n in grads:
    ref = grads[n].numpy()
    e = np.abs(got[n] - ref).max(); s = np.abs(ref).max()
    print('%-24s max|g| %.3e  err %.3e  err/max %.2e' % (n, s, e, e / max(s, 1e-30)))
