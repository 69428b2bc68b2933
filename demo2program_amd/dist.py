"""Data parallelism over the batch (program) axis: one process per GPU, RCCL over xGMI.

The reference is single-GPU (trainer.py:134-138) and defines no multi-GPU behaviour; this is
the one exchange step the sharded path needs:

  * each rank trains on B/N programs with ALL k demonstrations of each program (the
    summarizer mixes across k inside a program, never across programs);
  * the flat gradient buffer (~45 MB) is all-reduced (SUM, fp32) in TWO large pieces: the
    decoders' slice (the tail of the buffer, 57 % of the bytes) is final two thirds of the way
    through backward and is reduced on RCCL's stream WHILE the summarizer / encoder backward
    runs (`all_reduce_start` at Model.backward's split point); the rest follows after backward
    (`all_reduce_finish`).  xGMI is point-to-point, so a ring all-reduce is per-link bound and
    few large messages are the efficient shape (2*(N-1)/N*45 MB / 153 GB/s ~= 0.5 ms at N=8);
    `all_reduce_grads` is the one-message form (D2P_DP_OVERLAP=0, or a model on two streams);
  * the 1/N of the average is folded into the clip+Adam kernel's ``prescale`` -- the
    averaged gradient is never written back to HBM;
  * clip-by-global-norm runs AFTER the reduce, on every rank, as trainer.py:107 implies;
  * batch-norm statistics and the loss denominators stay per rank (no sync-BN): N ranks x
    B/N programs == N independent reference steps with averaged gradients.

``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used for the CPU tests -- and for the one multi-rank run
of the REAL HIP step that a one-GPU box allows: two processes sharing the device (RCCL refuses two ranks on one device),
their device buffers exchanged through host memory (`_reduce`: device -> host copy, gloo all-reduce, copy back; stream
ordered with the step like the RCCL call it stands in for).  tests/test_dp_two_ranks_gpu.py.
"""
import os
import sys

import torch


class DataParallel(object):

    def __init__(self, rank=0, world_size=1, initialized=False):
        self.rank = rank
        self.world_size = world_size
        self.initialized = initialized

    @classmethod
    def from_env(cls, backend=None, force_init=None):
        """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run).
        force_init (or D2P_FORCE_DIST=1): create the process group even for a single rank, so the
        RCCL calls of the exchange step run (as identities) on a one-GPU box -- the self-test of
        SURVEY 8(e)(iii)."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', str(rank)))
        if force_init is None:
            force_init = os.environ.get('D2P_FORCE_DIST', '0') == '1'
        if world <= 1 and not force_init:
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            return cls(0, 1, False)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        use_cuda = torch.cuda.is_available()
        if use_cuda:
            torch.cuda.set_device(local % torch.cuda.device_count())
        if backend is None:
            backend = 'nccl' if use_cuda else 'gloo'
        if not dist.is_initialized():
            # RCCL prints a version banner on STDOUT when its communicator comes up; stdout is the
            # bench's one-JSON-line channel, so the process-level fd 1 points at stderr until the
            # communicator exists (forced here by a first tiny all-reduce)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if backend == 'nccl':
                    dev = torch.device('cuda', torch.cuda.current_device())
                    dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=dev)
                    warm = torch.zeros(1, device=dev)
                    dist.all_reduce(warm)
                    torch.cuda.synchronize()
                else:
                    dist.init_process_group(backend=backend, rank=rank, world_size=world)
            finally:
                sys.stdout.flush()
                try:
                    # the banner sits in the C library's stdout buffer (block-buffered on a pipe) until exit:
                    # flush it while fd 1 still points at stderr
                    import ctypes
                    ctypes.CDLL(None).fflush(None)
                except (OSError, AttributeError):
                    pass
                os.dup2(saved, 1)
                os.close(saved)
        return cls(rank, world, True)

    @property
    def prescale(self):
        return 1.0 / self.world_size

    @staticmethod
    def _through_host(t, collective):
        """TEST-ONLY path: a gloo group handed a DEVICE buffer (two ranks sharing the one GPU of the test box,
        tests/test_dp_two_ranks_gpu.py) stages it through pageable host memory.  SYNCHRONOUS: the copy out waits for
        everything on the current stream, the host blocks in the collective, the copy back is stream-ordered in front of
        what follows -- there is no work handle and nothing overlaps.  Production ranks run RCCL (backend 'nccl') on the
        device buffer and never come here."""
        host = t.detach().cpu()
        collective(host)
        t.copy_(host)

    def _reduce(self, t, op, async_op=False):
        """dist.all_reduce of `t` in place; returns the work handle of an asynchronous RCCL reduce, else None.
        async_op is IGNORED on the gloo device path (_through_host: synchronous, test-only) -- D2P_DP_OVERLAP=1 on such a
        group is correct but serialises the host in the middle of backward; a warning says so once."""
        import torch.distributed as dist
        if t.is_cuda and dist.get_backend() != 'nccl':
            if async_op and not getattr(self, '_warned_sync', False):
                self._warned_sync = True
                print('[demo2program_amd] all_reduce_start on a %s group with a device buffer is synchronous (staged '
                      'through host memory): nothing overlaps -- the overlap schedule needs RCCL' % dist.get_backend(),
                      file=sys.stderr)
            self._through_host(t, lambda host: dist.all_reduce(host, op=op))
            return None
        return dist.all_reduce(t, op=op, async_op=async_op)

    def all_reduce_grads(self, flat_grad):
        """SUM the flat gradient buffer across ranks, in place (no-op for one rank without a
        process group)."""
        if self.world_size > 1 or self.initialized:
            import torch.distributed as dist
            self._reduce(flat_grad, dist.ReduceOp.SUM)
        return flat_grad

    @property
    def active(self):
        """True when the exchange step issues collectives (several ranks, or a forced one-rank group)."""
        return self.world_size > 1 or self.initialized

    def all_reduce_start(self, piece):
        """Begins the SUM all-reduce of one contiguous piece of the flat gradient buffer on the
        collective library's own stream: it starts when the work enqueued on the current stream SO
        FAR is done and overlaps whatever is enqueued afterwards."""
        if not self.active:
            return
        import torch.distributed as dist
        if not hasattr(self, '_pending'):
            self._pending = []
        w = self._reduce(piece, dist.ReduceOp.SUM, async_op=True)
        if w is not None:
            self._pending.append(w)

    def all_reduce_finish(self, rest=None):
        """All-reduces `rest` (the part of the buffer not started earlier) and makes the current stream
        wait for every piece: after this the whole buffer holds the sum over ranks."""
        if not self.active:
            return
        import torch.distributed as dist
        if rest is not None and rest.numel():
            self._reduce(rest, dist.ReduceOp.SUM)
        for w in getattr(self, '_pending', []):
            w.wait()
        self._pending = []

    def broadcast_params(self, flat_params, src=0):
        if self.world_size > 1 or self.initialized:
            import torch.distributed as dist
            if flat_params.is_cuda and dist.get_backend() != 'nccl':
                self._through_host(flat_params, lambda host: dist.broadcast(host, src=src))
            else:
                dist.broadcast(flat_params, src=src)
        return flat_params

    def shard(self, ids):
        """Per-rank id shard: ids[rank::N] (after the reference's seeded shuffle,
        karel_env/dataset_karel.py:156)."""
        return list(ids)[self.rank::self.world_size]

    def barrier(self):
        if self.world_size > 1 or self.initialized:
            import torch.distributed as dist
            dist.barrier()

    def ranks_seen(self):
        """Number of ranks that answer a SUM all-reduce of ones on the job's process group (RCCL on GPUs);
        None when no collective is issued (one rank, no group)."""
        if not self.active:
            return None
        import torch.distributed as dist
        dev = 'cuda' if torch.cuda.is_available() and dist.get_backend() == 'nccl' else 'cpu'
        t = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(round(float(t.item())))

    def max_over_ranks(self, value):
        """max of a python float across ranks (bench timing)."""
        if self.world_size <= 1 and not self.initialized:
            return value
        import torch.distributed as dist
        dev = 'cuda' if torch.cuda.is_available() and dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def shutdown(self):
        if self.initialized:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
