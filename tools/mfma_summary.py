#!/usr/bin/env python
"""MFMA-pipe utilisation per kernel from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE` pass (tools/profile_mfma.sh; rocpd sqlite).

utilisation = sum over SIMDs of MFMA-busy cycles / (kernel duration x shader clock x 1024 SIMDs)
(`MfmaUtil` of rocprofv3 -L with the duration from the kernel trace; 256 CUs x 4 SIMDs; the counter
counts cycles: 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32 -- MI355X_MICROARCH.md).
usage: tools/mfma_summary.py <pmc_MFMA_results.db> [clock_GHz=2.4] [out.json]"""
import json
import sqlite3
import sys

SIMDS = 256 * 4


def main():
    db = sys.argv[1]
    ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
    con = sqlite3.connect(db)
    rows = con.execute(
        "select kernel_name, dispatch_id, value, duration from counters_collection "
        "where counter_name = 'SQ_VALU_MFMA_BUSY_CYCLES'").fetchall()
    agg = {}
    for name, _, busy, dur in rows:
        short = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if 'gemm_mfma_kernel' in short:
            short = 'gemm_mfma_kernel<' + ('conv (Im2col/Dgrad)' if ('Im2col' in name or 'Dgrad' in name) else
                                           'OneHot' if 'OneHot' in name else 'dense') + '>'
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += busy
        a[2] += dur
    out = []
    for k, (n, busy, dur) in agg.items():
        if busy <= 0:
            continue
        out.append(dict(kernel=k, launches=n, mfma_busy_cycles=busy, total_us=dur / 1e3,
                        mfma_util=busy / (dur * ghz * SIMDS)))
    out.sort(key=lambda r: -r['total_us'])
    print('| kernel | launches | total us | MFMA-busy cycles (sum over SIMDs) | MFMA utilisation |')
    print('|---|---|---|---|---|')
    for r in out:
        print('| `%s` | %d | %.1f | %.3g | %.1f %% |' % (r['kernel'][:90], r['launches'], r['total_us'],
                                                       r['mfma_busy_cycles'], 100 * r['mfma_util']))
    tb, td = sum(r['mfma_busy_cycles'] for r in out), sum(a[2] for a in agg.values())
    print()
    print('all kernels: MFMA pipe busy %.1f %% of (GPU-busy time x %d SIMDs) at %.1f GHz' %
          (100 * tb / (td * ghz * SIMDS), SIMDS, ghz))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
