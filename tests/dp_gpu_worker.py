"""One rank of tests/test_dp_two_ranks_gpu.py: the REAL HIP Trainer.train_step with WORLD_SIZE = 2, two processes
sharing the one MI355X of the box (gloo group; device buffers staged through host memory by DataParallel._reduce --
RCCL refuses two ranks on one device).

    python dp_gpu_worker.py <rank> <world> <port> <outdir> <case>

case: per_step   -- recurrences on the per-step kernels (two processes' persistent launches cannot both be resident)
      overlap    -- per_step + D2P_DP_OVERLAP=1 (the decoders' slice reduced at Model.backward's split point)
      persistent -- the persistent kernels left ON: whether a hand-off times out depends on how the two processes'
                    launches meet on the device; whatever happens, the guarded step / all-reduced status slot /
                    restore / re-run protocol must end at the same parameters
      overlap_persistent -- D2P_DP_OVERLAP=1 with the persistent kernels ON: the recurrences behind backward's split point are
                    planned for Trainer.dp_overlap_cus CUs (round 6) -- the same protocol guarantees as `persistent`
      inject     -- persistent kernels on, and rank 1's status word is set before its step 1 (as a timed-out hand-off
                    sets it): BOTH ranks must skip that step on the device and re-run it
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_STEPS = 3


def case_config():
    from helpers import small_case
    # Karel frames (the one-launch State_Encoder's geometry), 128 units: large enough for the persistent kernels
    cfg, params, _ = small_case('karel', seed=11, batch_size=4, k=4, max_demo_len=8, max_program_len=12,
                                num_lstm_cell_units=128)
    return cfg, params


def rank_batches(cfg, rank):
    from demo2program_amd.synthetic import make_batch
    return [make_batch(cfg, seed=500 + 10 * step + rank) for step in range(N_STEPS)]


def main():
    rank, world, port, outdir, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                      D2P_GRAPH='0', D2P_DP_OVERLAP='1' if case in ('overlap', 'overlap_persistent') else '0')
    import numpy as np
    import torch
    from demo2program_amd import kernels as K
    from demo2program_amd.dist import DataParallel
    from demo2program_amd.trainer import Trainer
    dp = DataParallel.from_env(backend='gloo')
    assert dp.world_size == world and dp.rank == rank and dp.active
    cfg, params = case_config()
    if case in ('per_step', 'overlap'):
        K.lstm_set_persistent(False)
    tr = Trainer(cfg, make_train_dir=False, dp=dp)
    if case == 'overlap_persistent':
        assert tr.dp_overlap and tr.dp_overlap_cus > 0
    if rank == 0:
        tr.model.params.load(params)           # rank 1 keeps its own initialiser: the broadcast must overwrite it
    dp.broadcast_params(tr.model.params.flat)
    feeds = [tr.model.get_feed_dict(b) for b in rank_batches(cfg, rank)]
    losses = []
    with tr.step_stream():                     # (the loop on the trainer's high-priority stream, as Trainer.train runs it)
        for step in range(N_STEPS):
            if case == 'inject' and rank == 1 and step == 1:
                torch.cuda.synchronize()
                K.lstm_persist_inject_error()
            losses.append(tr.train_step(feeds[step]))
        failures = tr.settle()
    torch.cuda.synchronize()
    P = tr.model.params
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), flat=P.flat.cpu().numpy(), m=P.m.cpu().numpy(),
             v=P.v.cpu().numpy(), moving=tr.model.moving_flat.cpu().numpy(), failures=failures,
             applied=int(tr.guard.counters[0]), skipped=int(tr.guard.counters[1]), global_step=tr.global_step,
             adam_step=tr.adam_step, persistent=int(K.lstm_is_persistent()),
             losses=np.asarray([float(l.item()) for l in losses]))
    dp.barrier()
    dp.shutdown()


if __name__ == '__main__':
    main()
