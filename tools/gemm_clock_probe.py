#!/usr/bin/env python
"""Does the shader clock hold 2.4 GHz under a dense fp32-MFMA GEMM load?  Runs back-to-back GEMMs for a few
seconds while sampling rocm-smi (sclk, socket power) from the host; prints the samples and the GEMM rate per
window.  Compare with the same sampling under (a) an idle GPU and (b) a pure register-resident MFMA loop
(tools/micro/mfma_valu_contention.hip reaches 155 TF = 99 % of the 2.4 GHz peak)."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demo2program_amd import kernels as K  # noqa: E402
from demo2program_amd.lib import load  # noqa: E402


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return 'rocm-smi failed: %r' % (e,)
    keep = [l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'fclk' in l or 'mclk' in l]
    return ' | '.join(keep)


def main():
    lib = load()
    print('idle:', smi())
    for (m, n, k, tile) in [(4096, 4096, 4096, 1), (6400, 2048, 512, 0), (6400, 2048, 512, 9)]:
        A = torch.rand(m, k, device='cuda') - 0.5
        B = torch.rand(k, n, device='cuda') - 0.5
        C = torch.empty(m, n, device='cuda')
        lib.d2p_gemm_force_plan(tile, 1)
        stop = [False]
        samples = []

        def sampler():
            while not stop[0]:
                samples.append(smi())
                time.sleep(0.3)
        th = threading.Thread(target=sampler)
        th.start()
        t_end = time.time() + 4.0
        rates = []
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 200 if m * n * k < 2e10 else 20
            for _ in range(reps):
                K.matmul_nn(A, B, out=C)
            e1.record()
            torch.cuda.synchronize()
            rates.append(2.0 * m * n * k * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        stop[0] = True
        th.join()
        print('GEMM %dx%dx%d tile %d: TF per window first %.0f last %.0f min %.0f max %.0f' % (
            m, n, k, tile, rates[0], rates[-1], min(rates), max(rates)))
        for s in samples[:2] + samples[-3:]:
            print('   ', s)
    lib.d2p_gemm_force_plan(-1, 0)


if __name__ == '__main__':
    main()
