"""The REAL data-parallel step on hardware with WORLD_SIZE = 2: two processes sharing the
one MI355X, Trainer.train_step on the HIP kernels, gradients + the step-status slot exchanged over a gloo group
(device buffers staged through host memory; RCCL refuses two ranks on one device), against SURVEY 8(e)'s definition
computed in ONE process: N independent steps on the ranks' batches from the same parameters, gradients averaged, one
clip(20) + Adam update; batch-norm statistics per rank (trainer.py:134-138 of the reference has nothing to match)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _reference(n_steps):
    """Two models in this process, one per rank's batches: forward + backward each, gradients summed, prescale 1/2,
    clip + Adam on model 0's parameters, copied to model 1 -- the definition the exchange step implements."""
    import math
    from demo2program_amd import kernels as K
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.trainer import ADAM_B1, ADAM_B2, ADAM_EPS, CLIP_GRADIENTS
    from dp_gpu_worker import case_config, rank_batches
    cfg, params = case_config()
    K.lstm_set_persistent(False)
    try:
        models = [Model(cfg, params=params) for _ in range(2)]
        feeds = [[m.get_feed_dict(b) for b in rank_batches(cfg, r)] for r, m in enumerate(models)]
        sumsq = torch.zeros(1, dtype=torch.float64, device='cuda')
        P0 = models[0].params
        for step in range(n_steps):
            for r, m in enumerate(models):
                m.forward(feeds[r][step])
                m.backward()
            P0.grad.add_(models[1].params.grad)
            K.l2norm_flat(P0.grad, 0.5, sumsq)
            t = step + 1
            lr_t = cfg.learning_rate * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
            K.adam_clip_flat(P0.flat, P0.grad, P0.m, P0.v, sumsq, 0.5, CLIP_GRADIENTS, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS)
            models[1].params.flat.copy_(P0.flat)
        torch.cuda.synchronize()
        return dict(flat=P0.flat.cpu().numpy(), m=P0.m.cpu().numpy(), v=P0.v.cpu().numpy(),
                    moving=[m.moving_flat.cpu().numpy() for m in models])
    finally:
        K.lstm_set_persistent(True)


@pytest.fixture(scope='module')
def reference():
    assert torch.cuda.is_available(), 'GPU tests need a real MI355X (run through gpurun)'
    from demo2program_amd import build
    build.build_library()
    from dp_gpu_worker import N_STEPS
    return _reference(N_STEPS)


def _run_case(case, tmp_path):
    from test_dp_gloo import _free_port
    port = _free_port()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), str(r), '2', str(port),
                               str(tmp_path), case], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=420)
            outs.append(out.decode(errors='replace'))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d of case %s failed:\n%s' % (r, case, outs[r][-4000:])
    return [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(2)], outs


@pytest.mark.parametrize('case', ['per_step', 'overlap', 'persistent', 'overlap_persistent', 'inject'])
def test_two_ranks_on_one_gpu_equal_averaged_independent_steps(case, tmp_path, reference):
    from dp_gpu_worker import N_STEPS
    r, outs = _run_case(case, tmp_path)
    # identical on both ranks, bit for bit: the same summed gradient, the same update
    for name in ('flat', 'm', 'v'):
        assert np.array_equal(r[0][name], r[1][name]), (case, name)
    for i in range(2):
        assert int(r[i]['global_step']) == N_STEPS and int(r[i]['adam_step']) == N_STEPS
        assert int(r[i]['applied']) == N_STEPS, (case, i, int(r[i]['applied']), int(r[i]['skipped']))
        assert np.all(np.isfinite(r[i]['flat']))
    # every rank saw the SAME number of failures (the status word travels through the all-reduce)
    assert int(r[0]['failures']) == int(r[1]['failures']) and int(r[0]['skipped']) == int(r[1]['skipped'])
    if case in ('per_step', 'overlap'):
        assert int(r[0]['failures']) == 0 and int(r[0]['skipped']) == 0
    if case == 'inject':
        # rank 1's word was set before its step 1: BOTH ranks skipped on the device (rank 0's own kernels were fine),
        # restored their moving statistics and re-ran
        assert int(r[0]['failures']) >= 1 and int(r[0]['skipped']) >= 1
        assert 'gave up a hand-off' in outs[0] and 'gave up a hand-off' in outs[1]
    # ... and equal to N independent steps with averaged gradients (per-step vs persistent kernels, the order of the
    # host-side sum: fp32 round-off -- the tolerance of the single-rank re-run test)
    scale = float(np.abs(reference['flat']).max())
    assert float(np.abs(r[0]['flat'] - reference['flat']).max()) <= 2e-5 * scale, case
    ms = float(np.abs(reference['m']).max())
    assert float(np.abs(r[0]['m'] - reference['m']).max()) <= 1e-4 * ms + 1e-9, case
    for i in range(2):      # batch-norm moving statistics stay per rank: each rank's equal its own batches'
        want = reference['moving'][i]
        assert float(np.abs(r[i]['moving'] - want).max()) <= 1e-5 * (1.0 + float(np.abs(want).max())), (case, i)
