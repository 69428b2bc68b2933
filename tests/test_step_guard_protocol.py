"""The guarded optimizer step across ranks, without a GPU: world_size 2 over gloo drives Trainer.train_step's exchange
step and the StepGuard protocol (demo2program_amd/trainer.py) with the kernels replaced by CPU stand-ins.

Rank 1's "persistent recurrent kernel" gives up a hand-off in its 4th step.  Its status word travels through the
gradient all-reduce, so BOTH ranks must skip that step and every later one (the word is sticky), detect the failure at the
same step index (the ring slot about to be reused), restore the batch-norm moving statistics of the first skipped step,
re-run exactly the skipped steps on the per-step kernels and end with parameters, Adam moments and moving statistics
bit-identical to an undisturbed run -- on both ranks.  (reference: trainer.py:102-109 has one
process and no failure path -- this is the protocol the data-parallel path adds around it.)"""
import math
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_STEPS, FAIL_AT, FAIL_RANK = 10, 3, 1


class FakeK(object):
    """CPU stand-ins for the entry points the exchange step and the guard call (demo2program_amd/kernels.py)."""

    def __init__(self):
        self.err = 0
        self.persistent = True
        self.applied_with = []            # (global value of the status slot, persistent?) per optimizer call

    def lstm_is_persistent(self):
        return self.persistent

    def lstm_set_persistent(self, on):
        self.persistent = bool(on)

    def lstm_persist_error(self, reset=False):
        v = self.err
        if reset:
            self.err = 0
        return v

    def step_status_publish(self, slot):
        slot.zero_()
        slot[0] = 1.0 if self.err else 0.0

    def l2norm_flat(self, grad, prescale, sumsq):
        sumsq[0] = float(((grad.double() * prescale) ** 2).sum())

    def adam_clip_flat(self, p, g, m, v, sumsq, prescale, clip, lr_t, b1=0.9, b2=0.999, eps=1e-8, lr_t_dev=None,
                       counters=None, fail_slot=None, mirror=None):
        failed = self.err != 0 or (fail_slot is not None and float(fail_slot[0]) != 0.0)
        self.applied_with.append((None if fail_slot is None else float(fail_slot[0]), self.persistent, failed))
        if counters is not None and failed:
            counters[1] += 1
            mirror.copy_(counters)
            return
        norm = math.sqrt(float(sumsq[0]))
        gg = g * (prescale * (clip / max(norm, clip)))
        m.mul_(b1).add_(gg, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(gg, gg, value=1.0 - b2)
        p.sub_(lr_t * m / (v.sqrt() + eps))
        if counters is not None:
            counters[0] += 1
            mirror.copy_(counters)


class FakeModel(object):
    """What Trainer.train_step touches of Model: a flat parameter set, moving statistics in one buffer, forward /
    backward.  The 'gradient' is a deterministic function of the parameters and the feed; the moving statistics move in
    EVERY forward pass, failed or not (as the conv encoder's do: it runs before the recurrences)."""
    fail_at = None                 # set per run: the index of the forward pass whose persistent kernel gives up

    def __init__(self, config, debug_information=False, global_step=None, **kw):
        from demo2program_amd.params import FlatParams
        self.params = FlatParams(config, seed=3, device='cpu')
        self.moving_flat = torch.zeros(64)
        self.moving = {'conv1': (self.moving_flat[:32], self.moving_flat[32:])}
        self.scheduled_sampling = False
        self.use_side_stream = True           # (so that the trainer takes the moving-statistics snapshots AHEAD, as on the GPU)
        self.calls = 0
        self.K = None

    def _side_stream(self):
        return None

    def decoder_grad_offset(self):
        return self.params.offsets['prog/embedding']

    def forward(self, feed, defer_loss=False):
        self.moving_flat.mul_(0.9).add_(feed['stat'], alpha=0.1)
        self._feed = feed
        failing = self.K.persistent and FakeModel.fail_at is not None and self.calls == FakeModel.fail_at
        self.calls += 1
        if failing:
            self.K.err = (0x7f << 24) | 0x800000
        self._garbage = self.K.err != 0 and self.K.persistent
        return torch.tensor([float(feed['stat'].sum())])

    def backward(self, split_cb=None):
        P = self.params
        if self._garbage:
            P.grad.fill_(float('nan'))             # a timed-out launch leaves invalid results behind
        else:
            P.grad.copy_(torch.sin(P.flat * 3.0 + self._feed['phase']) * 0.5 + 0.01 * P.flat)
        if split_cb is not None:
            split_cb()
        return P.grad


class Feeds(object):
    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.feeds = [dict(stat=torch.rand(64, generator=g), phase=float(torch.rand(1, generator=g)), n_prog=1, n_demo=1)
                      for _ in range(N_STEPS)]
        self.i = 0

    def next(self):
        f = self.feeds[self.i % len(self.feeds)]
        self.i += 1
        return f


class _Done(object):
    def query(self):
        return True

    def synchronize(self):
        pass


def _run(rank, dp, fail):
    from demo2program_amd import trainer as T
    from demo2program_amd.config import make_config
    fake = FakeK()
    T.K = fake
    T._new_event = lambda: _Done()
    T._device_sync = lambda: None
    T.Trainer.get_model_class = staticmethod(lambda name: FakeModel)
    FakeModel.fail_at = FAIL_AT if (fail and rank == FAIL_RANK) else None
    cfg = make_config('karel_tiny', num_lstm_cell_units=64)
    tr = T.Trainer(cfg, dataset=Feeds(rank), dataset_test=Feeds(rank), make_train_dir=False, dp=dp, use_graph=False)
    tr.model.K = fake
    assert tr.guard is not None and tr.guard.moving is tr.model.moving_flat
    src = tr.batch_train
    steps_seen = []
    for _ in range(N_STEPS):
        tr.train_step(src.next())
        steps_seen.append(tr.global_step)
    tr.settle()
    P = tr.model.params
    return dict(flat=P.flat.numpy().copy(), m=P.m.numpy().copy(), v=P.v.numpy().copy(),
                moving=tr.model.moving_flat.numpy().copy(), failures=tr.guard.failures, global_step=tr.global_step,
                adam_step=tr.adam_step, steps_seen=np.asarray(steps_seen),
                skipped=int(tr.guard.counters[1]), applied=int(tr.guard.counters[0]),
                log=np.asarray([[-1.0 if a is None else a, float(b), float(c)] for a, b, c in fake.applied_with]))


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), D2P_DP_OVERLAP='0', D2P_GRAPH='0')
    torch.set_num_threads(1)
    from demo2program_amd.dist import DataParallel
    dp = DataParallel.from_env(backend='gloo')
    clean = _run(rank, dp, fail=False)
    dp.barrier()
    hit = _run(rank, dp, fail=True)
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), **{'clean_' + k: v for k, v in clean.items()},
             **{'hit_' + k: v for k, v in hit.items()})
    dp.barrier()
    dp.shutdown()


def test_both_ranks_skip_and_rerun_the_same_steps(tmp_path):
    from test_dp_gloo import _free_port
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    for i in range(world):
        assert int(r[i]['clean_failures']) == 0 and int(r[i]['clean_skipped']) == 0
        assert int(r[i]['clean_applied']) == N_STEPS
        # the disturbed run: one failure on EVERY rank (rank 0's own kernels were fine), the four steps 3..6 skipped on
        # the device (the failed one and the three launched before the ring slot came round), all of them re-run
        assert int(r[i]['hit_failures']) == 1
        assert int(r[i]['hit_skipped']) == 4
        assert int(r[i]['hit_applied']) == N_STEPS
        assert int(r[i]['hit_global_step']) == N_STEPS and int(r[i]['hit_adam_step']) == N_STEPS
        # detected at the same step index everywhere: global_step drops back to 3 + 1 after the 8th call
        assert list(r[i]['hit_steps_seen']) == [1, 2, 3, 4, 5, 6, 7, 4 + 4, 9, 10]
        log = r[i]['hit_log']
        # the status slot every rank's optimizer saw: 0 (fine), then 1.0 = "one rank failed" for steps 3..6 on BOTH
        # ranks, then the re-runs on the per-step kernels and the rest
        assert list(log[:, 0]) == [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
        assert list(log[:, 2]) == [0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
        assert list(log[7:11, 1]) == [0, 0, 0, 0] and list(log[11:, 1]) == [1, 1, 1]      # per-step kernels for the re-runs only
        for name in ('flat', 'm', 'v', 'moving'):
            assert np.all(np.isfinite(r[i]['hit_' + name])), name
            assert np.array_equal(r[i]['hit_' + name], r[i]['clean_' + name]), (i, name)
    for name in ('flat', 'm', 'v'):
        assert np.array_equal(r[0]['hit_' + name], r[1]['hit_' + name]), name


def test_single_rank_recovery_restores_the_moving_statistics():
    """One rank, no process group: the failure is detected by the next step's poll of an arrived slot; the re-run starts
    from the moving statistics of the skipped step."""
    from demo2program_amd.dist import DataParallel
    os.environ.update(D2P_GRAPH='0')
    clean = _run(0, DataParallel(), fail=False)
    FakeModel.fail_at = None
    saved_rank = FAIL_RANK
    try:
        globals()['FAIL_RANK'] = 0
        hit = _run(0, DataParallel(), fail=True)
    finally:
        globals()['FAIL_RANK'] = saved_rank
    assert hit['failures'] == 1 and hit['skipped'] >= 1 and hit['applied'] == N_STEPS
    for name in ('flat', 'm', 'v', 'moving'):
        assert np.array_equal(hit[name], clean[name]), name
