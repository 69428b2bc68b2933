#!/usr/bin/env python
"""Time of an RCCL all-reduce in a forced ONE-rank process group (a 10 us no-op: what D2P_FORCE_DIST=1 measurements include)."""
import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['D2P_FORCE_DIST'] = '1'
from demo2program_amd.dist import DataParallel
dp = DataParallel.from_env()
import torch.distributed as dist
for mb in (1, 8, 19, 26, 45):
    t = torch.zeros(mb * 1024 * 1024 // 4, device='cuda')
    for _ in range(3): dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dist.all_reduce(t)
    e1.record(); torch.cuda.synchronize()
    print('one-rank all_reduce %d MB: %.1f us' % (mb, e0.elapsed_time(e1) * 1e3 / 20), file=sys.stderr)
dp.shutdown()
