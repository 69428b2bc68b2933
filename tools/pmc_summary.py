#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite) per kernel.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated.
usage: tools/pmc_summary.py <dir with pmc_FETCH_SIZE_results.db, pmc_WRITE_SIZE_results.db> [out.json]"""
import json
import os
import sqlite3
import sys


def family(name):
    name = name.replace('(anonymous namespace)::', '')
    if 'lstm_persist_fwd' in name: return 'lstm_persist_fwd_kernel'
    if 'lstm_persist_bwd' in name: return 'lstm_persist_bwd_kernel'
    if 'lstm_step_fwd' in name: return 'lstm_step_fwd_kernel'
    if 'lstm_step_bwd' in name: return 'lstm_step_bwd_kernel'
    if 'gemm_tn_direct_kernel' in name: return 'gemm_mfma_kernel'      # (the GEMM family's register-direct A^T B form)
    if 'gemm_mfma_kernel' in name:
        if 'Im2col' in name or 'Dgrad' in name: return 'gemm_mfma_kernel<conv>'
        if 'OneHot' in name: return 'gemm_mfma_kernel<onehot>'
        return 'gemm_mfma_kernel'
    return name.split('(')[0].replace('void ', '')


def load(db, counter):
    con = sqlite3.connect(db)
    out = {}
    for name, value in con.execute(
            "select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        f = family(name)
        s = out.setdefault(f, [0, 0.0])
        s[0] += 1
        s[1] += value
    return out


def main():
    d = sys.argv[1]
    fetch = load(os.path.join(d, 'pmc_FETCH_SIZE_results.db'), 'FETCH_SIZE')
    write = load(os.path.join(d, 'pmc_WRITE_SIZE_results.db'), 'WRITE_SIZE')
    rows = []
    for f in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(f, [0, 0.0])
        nw, kw = write.get(f, [0, 0.0])
        n = max(nf, nw)
        fb = 2.0 * kf * 1024 / max(nf, 1)
        wb = kw * 1024 / max(nw, 1)
        rows.append(dict(kernel=f, launches=n, fetch_bytes_per_launch=fb, write_bytes_per_launch=wb,
                         hbm_bytes_per_launch=fb + wb, total_mb=(fb + wb) * n / 1e6))
    rows.sort(key=lambda r: -r['total_mb'])
    print('| kernel | launches | fetch MB/launch (2x FETCH_SIZE) | write MB/launch | HBM MB/launch | total MB |')
    print('|---|---|---|---|---|---|')
    for r in rows[:30]:
        print('| `%s` | %d | %.3f | %.3f | %.3f | %.1f |' % (
            r['kernel'][:80], r['launches'], r['fetch_bytes_per_launch'] / 1e6,
            r['write_bytes_per_launch'] / 1e6, r['hbm_bytes_per_launch'] / 1e6, r['total_mb']))
    if len(sys.argv) > 2:
        out = {r['kernel']: r for r in rows}
        # where the numbers come from: bench.py quotes this in roofline.traffic_source
        out['_meta'] = {'collected_by': 'tools/profile_pmc.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, '
                                        'one counter per pass, eager one-stream launches)',
                        'source_commit': os.environ.get('D2P_COMMIT', 'unknown'),
                        # optimizer steps the passes ran (warm-up + timed; the adam_clip launches count them)
                        'steps': max((r['launches'] for r in rows if r['kernel'].startswith('adam_clip')), default=0),
                        'correction': 'HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half '
                                      'of wide coalesced reads; WRITE_SIZE uncalibrated)'}
        json.dump(out, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
