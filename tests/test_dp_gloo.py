"""Data-parallel path on CPU: world_size 2 over gloo (the GPU path uses the same code with
backend "nccl" == RCCL).  Checks the one exchange step of SURVEY 8(e): broadcast of the flat
parameter buffer, SUM all-reduce of the flat gradient buffer, 1/N folded into `prescale`,
identical results on every rank, and equality with the mean of the per-rank gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_grads(cfg, params, rank):
    from helpers import run_oracle
    from demo2program_amd.synthetic import make_batch
    batch = make_batch(cfg, seed=100 + rank)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    return float(out['loss']), grads


def _worker(rank, world, port, outdir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from helpers import perturbed_params, small_case
    from demo2program_amd.dist import DataParallel
    from demo2program_amd.params import FlatParams
    dp = DataParallel.from_env(backend='gloo')
    assert dp.world_size == world and dp.rank == rank and abs(dp.prescale - 1.0 / world) < 1e-12
    cfg, params, _ = small_case('karel', seed=7, batch_size=2)
    # rank 1 starts from different weights: the broadcast must overwrite them
    vals = params if rank == 0 else perturbed_params(cfg, 99)
    fp = FlatParams(cfg, values=vals, device='cpu')
    dp.broadcast_params(fp.flat)
    synced = fp.to_numpy('p')
    loss, grads = _rank_grads(cfg, synced, rank)
    for n, gview in fp.g.items():
        gview.copy_(grads[n].float())
    # the two-piece exchange of the trainer (decoders' tail started first, the rest finishing) must
    # give the same buffer as the one-message form, bit for bit, on every rank
    two = fp.grad.clone()
    dec = fp.offsets['prog/embedding']
    assert 0 < dec < two.numel()           # (57 % of the bytes at the headline configuration)
    dp.all_reduce_start(two[dec:])
    two[:dec].mul_(1.0)                    # "the rest of backward" between start and finish
    dp.all_reduce_finish(two[:dec])
    assert dp._pending == []
    dp.all_reduce_grads(fp.grad)
    assert torch.equal(two, fp.grad)
    t = dp.max_over_ranks(float(rank + 1))
    ids = dp.shard(list(range(10)))
    np.savez(os.path.join(outdir, 'rank%d.npz' % rank), grad=fp.grad.numpy() * dp.prescale,
             flat=fp.flat.numpy(), loss=loss, tmax=t, ids=np.asarray(ids))
    dp.barrier()
    dp.shutdown()


def test_two_rank_allreduce_equals_mean_of_rank_gradients(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % i)) for i in range(world)]
    # every rank ends with the same parameters (rank 0's) and the same averaged gradient
    assert np.array_equal(r[0]['flat'], r[1]['flat'])
    assert np.array_equal(r[0]['grad'], r[1]['grad'])
    assert r[0]['tmax'] == 2.0 and r[1]['tmax'] == 2.0
    assert list(r[0]['ids']) == [0, 2, 4, 6, 8] and list(r[1]['ids']) == [1, 3, 5, 7, 9]
    # reference: N independent oracle steps, gradients averaged (SURVEY 8(e) definition)
    from helpers import small_case
    from demo2program_amd.params import FlatParams
    cfg, params, _ = small_case('karel', seed=7, batch_size=2)
    fp = FlatParams(cfg, values=params, device='cpu')
    assert np.array_equal(fp.flat.numpy(), r[0]['flat'])
    acc = None
    for rank in range(world):
        _, grads = _rank_grads(cfg, params, rank)
        for n, gview in fp.g.items():
            gview.copy_(grads[n].float())
        acc = fp.grad.numpy().copy() if acc is None else acc + fp.grad.numpy()
    ref = acc / world
    assert np.abs(ref - r[0]['grad']).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_flat_param_layout_is_16_byte_aligned_and_roundtrips():
    from demo2program_amd.config import make_config
    from demo2program_amd.params import FlatParams, init_params, num_params
    cfg = make_config('karel_tiny', num_lstm_cell_units=64)
    vals = init_params(cfg, 5)
    fp = FlatParams(cfg, values=vals, device='cpu')
    assert all(o % 4 == 0 for o in fp.offsets.values())
    assert fp.size >= num_params(cfg)
    back = fp.to_numpy('p')
    for n in vals:
        assert np.array_equal(back[n], vals[n])
    # views alias the flat buffers
    fp.p['conv1/b'].fill_(3.0)
    o = fp.offsets['conv1/b']
    assert float(fp.flat[o]) == 3.0


def test_single_rank_dataparallel_is_a_noop():
    from demo2program_amd.dist import DataParallel
    dp = DataParallel()
    g = torch.arange(8, dtype=torch.float32)
    assert dp.prescale == 1.0 and dp.all_reduce_grads(g) is g and dp.shard([1, 2, 3]) == [1, 2, 3]
    assert not dp.active
    dp.all_reduce_start(g[4:])             # no process group: the two-piece form is a no-op as well
    dp.all_reduce_finish(g[:4])
    assert torch.equal(g, torch.arange(8, dtype=torch.float32))
    assert dp.max_over_ranks(1.5) == 1.5
