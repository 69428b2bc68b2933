import os, sys, torch
sys.path.insert(0, '/root/repo')
from demo2program_amd import kernels as K
def mk(M, T, U=512):
    g = torch.Generator().manual_seed(M)
    return dict(M=M, U=U, n_steps=T, z=(torch.rand(T*M, 4*U, generator=g)-0.5).cuda(), Wh=((torch.rand(U, 4*U, generator=g)-0.5)*0.1).cuda(),
                h0=(torch.rand(M, U, generator=g)-0.5).cuda(), c0=(torch.rand(M, U, generator=g)-0.5).cuda(),
                hout=torch.zeros(T, M, U, device='cuda'), cs=torch.zeros(T, M, U, device='cuda'))
def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for M in (32, 64, 96, 128, 144, 192, 256, 320, 400):
    f = mk(M, 20)
    t = min(timed(lambda: K.lstm_seq_fwd_multi([f])) for _ in range(3))
    print('fwd M=%d T=20: %.0f us, %.2f us/step (err %d)' % (M, t, t/20, K.lstm_persist_error(True)))
