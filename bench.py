#!/usr/bin/env python
"""Headline benchmark: training throughput of the full model on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one complete optimizer step of the reference's trainer (trainer.py:186-205):
forward + backward + [RCCL all-reduce of the flat gradient buffer] + global-norm clip +
Adam, on one synthetic Karel batch that is ALREADY RESIDENT IN HBM.  Workload (weak
scaling): BASELINE.json config 2 per GPU -- Karel full model, k=10, 8x8x16 frames, T=20,
L=50, batch 32 programs per GPU.  value = programs (instances) per second over the whole
job; an instance = one program with its k demonstrations (trainer.py:238).

Besides the contract fields the JSON line carries
  roofline     -- the kernel family with the largest summed device time of the step (forward and
                  backward recurrent kernels count as ONE family, like all GEMM instantiations do),
                  timed live with HIP events on the launch stream in an instrumented pass right
                  after the timed region; `furthest_below_roofline` names the family (>= 3 % of
                  the step) with the lowest fraction of its roofline; `kernel_table` lists all
  value_incl_h2d_prefetched -- the same step loop fed from HOST batches through the trainer's
                  prefetcher (the reference's step time spans fetch + feed + run, trainer.py:187-205)
  config4_vizdoom -- BASELINE config 4 (80x80x3 frames) timed in the same run: ms/step and the conv
                  encoder's fraction of the fp32 MFMA peak (N = 1 only); its headline numbers again at the TOP
                  level (config4_ms_per_step, config4_value, config4_conv_frac, config4_conv_bn_ms,
                  config4_traffic_ratio) and inside `config.also_measured_in_this_run`
  config5_per_rank_vizdoom_k25 -- config 5's per-rank shape (k = 25, B = 16), 5 steps: config5_rank_ms_per_step / _value
  cpu_baseline -- the CPU oracle (torch-CPU restatement, NOT TF1) timed on this box's host
                  cores on the same batch (rank 0, N=1 only): forward + backward WITHOUT the
                  optimizer, on <= 16 threads.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0          # HBM3E spec
MEASURED_HBM_GBS = 6290.0      # MI355X_MICROARCH.md: float4 copy, 79 % of spec
EMPTY_LAUNCH_US = 2.7          # a dependent empty launch on one stream

# kernel family -> key in profiles/*_pmc_traffic.json (tools/pmc_summary.py)
PMC_KEYS = {1: 'gemm_mfma_kernel', 2: 'gemm_mfma_kernel<conv>', 7: 'lstm_persist_fwd_kernel',
            8: 'lstm_persist_bwd_kernel', 3: 'lstm_gate_fwd_kernel', 4: 'lstm_gate_bwd_kernel'}
PMC_FILES = [os.path.join(ROOT, 'profiles', n) for n in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json',
                                                         'r01_pmc_traffic.json')]

# profiling key (include/d2p.h) -> (name, roofline that bounds it, reporting group)
PROF_FAMILIES = {
    1: ('gemm_mfma_kernel + gemm_tn_direct_kernel (dense fp32 MFMA GEMM, all instantiations)', 'mfma', 'gemm'),
    2: ('conv kernels (whole-frame / direct 16x16x4 MFMA / row-strip, implicit-GEMM fallback)', 'mfma', 'conv'),
    3: ('lstm_gate_fwd_kernel', 'hbm', 'gate'),
    4: ('lstm_gate_bwd_kernel', 'hbm', 'gate'),
    5: ('batch-norm launches (bn_partial / finalize / apply, statistics from partial sums, backward coefficients)', 'hbm', 'bn'),
    7: ('recurrent forward (lstm_persist_fwdw_kernel: one launch for up to three sequences; lstm_step_fwd_kernel per step '
        'for shapes it does not take)', 'mfma', 'recurrent'),
    8: ('recurrent backward (lstm_persist_bwd_kernel / lstm_step_bwd_kernel)', 'mfma', 'recurrent'),
}
GROUP_NAMES = {
    'gemm': 'gemm_mfma_kernel + gemm_tn_direct_kernel (dense fp32 MFMA GEMM, all instantiations)',
    'conv': 'conv kernels (forward + dgrad + wgrad of every encoder layer)',
    'recurrent': 'recurrent LSTM kernels, forward + backward (lstm_persist_fwdw_kernel + lstm_persist_bwd_kernel: '
                 'h.Wh / dz.Wh^T fp32 MFMA + gate math for all time steps of a sequence in one launch)',
    'gate': 'standalone LSTM gate kernels',
    'bn': 'batch-norm launches of every layer (work = the bytes the separate passes read and write)',
}


def pmc_traffic(family):
    """HBM bytes per launch of this kernel family from the committed PMC passes of this same
    command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the
    gfx950 correction; tools/profile_pmc.sh + tools/pmc_summary.py).  None if not collected."""
    for path in PMC_FILES:
        try:
            d = json.load(open(path))
            return round(d[PMC_KEYS[family]]['hbm_bytes_per_launch'], 1)
        except (OSError, KeyError, ValueError):
            continue
    return None


def pmc_family_traffic(patterns, tag):
    """HBM MB per STEP of the kernels whose name holds one of `patterns`, from the committed PMC passes of `bench.py
    --preset <tag>` (profiles/r06_pmc_traffic_<tag>.json, else round 5's; tools/profile_vizdoom.sh); None if not collected."""
    d = path = None
    for rnd in ('r06', 'r05'):
        path = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.json' % (rnd, tag))
        try:
            d = json.load(open(path))
            break
        except (OSError, ValueError):
            continue
    if d is None:
        return None
    steps = d.get('_meta', {}).get('steps')
    if not steps:
        return None
    tot = sum(v['total_mb'] for k, v in d.items() if k != '_meta' and any(p_ in k for p_ in patterns))
    return {'MB_per_step': round(tot / steps, 1), 'file': os.path.relpath(path, ROOT),
            'source_commit': d['_meta'].get('source_commit', 'unknown'),
            'note': 'forward + backward, every conv / batch-norm kernel; counters from separate rocprofv3 --pmc passes'}


def pmc_traffic_source():
    """`traffic` is NOT measured in this run (hardware counters need rocprofv3 around the process): it is read from
    the committed summary of separate PMC passes of this same command.  This names the file and the commit its
    passes were collected at, so that a kernel changed since then is visible in the line."""
    for path in PMC_FILES:
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        meta = d.get('_meta', {})
        return {'file': os.path.relpath(path, ROOT), 'source_commit': meta.get('source_commit', 'unknown'),
                'note': 'counters collected in separate rocprofv3 --pmc passes (tools/profile_pmc.sh), not in this run'}
    return None


def _rate(work, ms, bound):
    return work / (ms / 1e3) / (1e12 if bound == 'mfma' else 1e9)


def roofline_leg(trainer, feeds, steps=2, keep_side=False):
    """Re-runs `steps` training steps with per-launch HIP events enabled inside the library (events on
    the launch stream, eager launches on ONE stream so a bracket times that kernel alone).  Families
    are grouped before ranking: all GEMM instantiations are one family, and so are the forward and
    backward recurrent kernels.  -> (roofline of the group with the largest summed time,
    the group furthest below its roofline, per-family table)."""
    from demo2program_amd.lib import load
    lib = load()
    torch.cuda.synchronize()
    lib.d2p_prof_enable(1)
    lib._d2p_prof_on = True          # Trainer.train_step takes the eager (un-graphed) path
    side = trainer.model.use_side_stream
    # (--prof-keep-side, diagnostic: the two-stream schedule under the brackets -- a family's time is then what its
    #  launches take BESIDE the other stream's work; the reported roofline always uses one stream)
    if not keep_side:
        trainer.model.use_side_stream = False
    with trainer.step_stream():          # (the stream the timed region ran on)
        for i in range(steps):
            trainer.train_step(feeds[i % len(feeds)])
        torch.cuda.synchronize()
    rows = []
    for fam, (name, bound, group) in PROF_FAMILIES.items():
        for tag in (0, 1):
            cnt, ms, work = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
            lib.d2p_prof_read(fam * 8 + tag, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(work))
            if cnt.value:
                rows.append(dict(family=fam, tag=tag, name=name, bound=bound, group=group, launches=cnt.value,
                                 total_ms=ms.value, work=work.value))
    lib.d2p_prof_enable(0)
    lib._d2p_prof_on = False
    trainer.model.use_side_stream = side
    if not rows:
        return None, None, [], []
    total_ms = sum(r['total_ms'] for r in rows)
    groups = {}
    for r in rows:
        g = groups.setdefault(r['group'], dict(group=r['group'], bound=r['bound'], launches=0, total_ms=0.0,
                                               work=0.0, families=[]))
        g['launches'] += r['launches']
        g['total_ms'] += r['total_ms']
        g['work'] += r['work']
        g['families'].append(r)

    def describe(g):
        peak = PEAK_F32_MFMA_TFLOPS if g['bound'] == 'mfma' else PEAK_HBM_GBS
        achieved = _rate(g['work'], g['total_ms'], g['bound'])
        traffic = [pmc_traffic(r['family']) for r in g['families']]
        d = {'kernel': GROUP_NAMES[g['group']], 'bound': g['bound'], 'achieved': round(achieved, 3), 'peak': peak,
             'unit': 'TFLOP/s' if g['bound'] == 'mfma' else 'GB/s', 'frac': round(achieved / peak, 4),
             'traffic': traffic[0] if len(traffic) == 1 else (None if any(t is None for t in traffic) else
                                                              round(sum(t * r['launches'] for t, r in zip(traffic, g['families'])) / g['launches'], 1)),
             'traffic_source': pmc_traffic_source(),
             'launches_per_step': g['launches'] / steps,
             'avg_launch_us': round(g['total_ms'] * 1e3 / g['launches'], 3),
             'work_per_launch': g['work'] / g['launches'],
             'ms_per_step': round(g['total_ms'] / steps, 4),
             'share_of_instrumented_ms': round(g['total_ms'] / total_ms, 3)}
        if g['group'] == 'recurrent':
            cfgm = trainer.config
            flop_step = 2.0 * cfgm.batch_size * cfgm.k * 4 * cfgm.num_lstm_cell_units ** 2
            # `work` is what the launches EXECUTE: the length-sorted backward launches stop each row domain at its
            # longest row, so the row-steps past a sequence's length are mostly not multiplied any more.  Rounds 1-2
            # multiplied (and counted) every row at every step -- the same count for this round's launches:
            dense = sum(recurrent_dense_flops(cfgm, feeds[i % len(feeds)]) for i in range(steps))
            d['work_counts'] = ('executed row-steps (a row domain of a length-sorted launch runs only its longest row\'s steps: '
                                'both encoders and every backward recurrence; round 5: in a training step also the action / '
                                'perception decoders\' FORWARD recurrences -- 30 % fewer row-steps there in 14 % less time, so '
                                'this fraction fell from 0.434 to ~0.42 while the family went from 1.36 to 1.33 ms per step)')
            d['frac_dense_rows'] = round(_rate(dense, g['total_ms'], 'mfma') / PEAK_F32_MFMA_TFLOPS, 4)
            d['frac_dense_rows_note'] = ('every row at every decoded step, the count of rounds 1-2 '
                                         '(2*M*4U*U per time step): comparable with their roofline.frac')
            d['parts'] = [{'kernel': r['name'], 'launches_per_step': r['launches'] / steps,
                           'ms_per_step': round(r['total_ms'] / steps, 4),
                           'achieved': round(_rate(r['work'], r['total_ms'], 'mfma'), 2),
                           'frac': round(_rate(r['work'], r['total_ms'], 'mfma') / PEAK_F32_MFMA_TFLOPS, 4),
                           # every sequence of the step priced per time step of a B*k-row recurrence
                           'us_per_%d_row_time_step' % (cfgm.batch_size * cfgm.k):
                               round(r['total_ms'] * 1e3 / (r['work'] / flop_step), 3)} for r in g['families']]
        return d
    ranked = sorted(groups.values(), key=lambda g: -g['total_ms'])
    roof = describe(ranked[0])
    sizeable = [g for g in ranked if g['total_ms'] >= 0.03 * total_ms]
    worst = min(sizeable, key=lambda g: _rate(g['work'], g['total_ms'], g['bound']) /
                (PEAK_F32_MFMA_TFLOPS if g['bound'] == 'mfma' else PEAK_HBM_GBS))
    furthest = describe(worst)
    rows.sort(key=lambda r: -r['total_ms'])
    table = [dict(kernel=r['name'], group=r['group'], tag=r['tag'], launches_per_step=r['launches'] / steps,
                  ms_per_step=round(r['total_ms'] / steps, 4),
                  rate=round(_rate(r['work'], r['total_ms'], r['bound']), 2),
                  unit='TFLOP/s' if r['bound'] == 'mfma' else 'GB/s') for r in rows]
    # (the GEMM family and the recurrent family are within a few per cent of each other's summed time since round 4:
    #  which of them `roofline` names changes from run to run -- every sizeable group is described here)
    return roof, furthest, table, [describe(g) for g in sizeable]


def recurrent_dense_flops(config, feed):
    """fp32 flops of one step's recurrences when every row is multiplied at every decoded step (what dynamic_rnn /
    dynamic_decode execute, models/model_full.py:243-258,465-471): forward h.Wh, backward dz.Wh^T."""
    U = config.num_lstm_cell_units
    M, B = config.batch_size * config.k, config.batch_size
    T, n_d, n_p = config.max_demo_len, feed['n_demo'], feed['n_prog']
    per_row_step = 2.0 * 4 * U * U
    fwd = M * (T - 1) + M * T + 2 * M * n_d + B * n_p          # encoder 1 (zero state: no product in step 0), 2, decoders
    bwd = M * (T - 1) + M * T + 2 * M * n_d + B * n_p          # (encoder 1 has no dh0 pass; the others one each)
    return per_row_step * (fwd + bwd)


def conv_binding_roofline(config, measured_ms, launches):
    """What bounds the conv encoder at this size (hard part 4): per layer and pass the
    larger of its MFMA time at peak and its HBM time at the measured copy rate, and under all of it a
    floor of one dependent launch per kernel.  -> the bound and measured / bound."""
    from demo2program_amd.config import conv_shapes
    n = config.batch_size * config.k * config.max_demo_len
    t_mfma = t_hbm = flops = 0.0
    for l, (h, w, cin, cout, ho, wo) in enumerate(conv_shapes(config)):
        f = 2.0 * n * ho * wo * 9 * cin * cout
        x_bytes = n * h * w * cin * (1.0 if l == 0 else 4.0)          # uint8 frames, fp32 activations
        y_bytes = n * ho * wo * cout * 4.0
        passes = [(f, x_bytes + y_bytes), (f, x_bytes + y_bytes)]     # forward, weight gradient
        if l > 0:
            passes.append((f, x_bytes + y_bytes))                     # input gradient
        for pf, pb in passes:
            flops += pf
            t_mfma += pf / (PEAK_F32_MFMA_TFLOPS * 1e12)
            t_hbm += pb / (MEASURED_HBM_GBS * 1e9)
    t_floor = launches * EMPTY_LAUNCH_US * 1e-6
    per_pass = max(t_mfma, t_hbm)
    bound = max(per_pass, t_floor)
    return {'flops_per_step': flops, 'mfma_ms_at_peak': round(t_mfma * 1e3, 4),
            'hbm_ms_at_%d_GBs' % MEASURED_HBM_GBS: round(t_hbm * 1e3, 4),
            'launch_floor_ms': round(t_floor * 1e3, 4), 'binding': 'launch floor' if bound == t_floor else
            ('hbm' if t_hbm > t_mfma else 'mfma'), 'bound_ms': round(bound * 1e3, 4),
            'measured_ms': round(measured_ms, 4), 'frac_of_binding_roofline': round(bound * 1e3 / measured_ms, 4)}


def north_star_targets(config, table):
    """The two kernel-level targets BASELINE.json's north_star names, measured live:
    (1) the conv encoder's fraction of the fp32 MFMA peak (conv family of the instrumented pass:
        forward + dgrad + wgrad of every layer, 2*M*K*N flops each), and next to it the roofline that
        actually binds at this size (MFMA, HBM or the launch floor);
    (2) the LSTM gate kernel's fraction of the HBM roofline (14 336 B/row forward,
        26 624 B/row backward at U=512).  A STANDALONE MICRO-BENCHMARK: in the product path the gate math is
        the epilogue of the recurrent kernels and these d2p_lstm_gate_* kernels are never launched.
        Measured over a ROTATING working set of > 600 MB (fresh buffers every launch: the 256 MiB
        Infinity Cache cannot serve it), at 32 000 rows per launch and at the rows of one training
        step (B*k*T), against both the 8.0 TB/s spec and the 6.29 TB/s this chip sustains on a copy;
        and at the per-time-step granularity (B*k rows), where a 4.6 MB launch is launch-latency."""
    from demo2program_amd import kernels as K
    out = {}
    conv = [r for r in table if r['group'] == 'conv']
    if conv:
        out['conv_encoder'] = {'achieved': conv[0]['rate'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(conv[0]['rate'] / PEAK_F32_MFMA_TFLOPS, 4),
                               'ms_per_step': conv[0]['ms_per_step'],
                               'launches_per_step': conv[0]['launches_per_step'],
                               'binding_roofline': conv_binding_roofline(config, conv[0]['ms_per_step'],
                                                                         conv[0]['launches_per_step']),
                               'note': 'launches_per_step counts the instrumented entry points (Karel: d2p_karel_encoder_fwd and '
                                       'd2p_karel_encoder_bwd = one forward launch, one backward launch + its combine launch). '
                                       'north_star asks for 0.30 of the fp32 MFMA peak: 46 us for the 2.18 GFLOP of the three '
                                       'layers forward + backward.  What bounds the family at this size: the batch-norm '
                                       'statistics of a demonstration index mix every frame of the index, so each direction has '
                                       'three index-wide exchanges between its workgroups (about 5 us each by the kernels\' own '
                                       'clock stamps, profiles/r05_karel_encoder_bwd_time.log: 30 us of the family by themselves), '
                                       'and the products are chains of 16x16x4 MFMAs on one wave per SIMD of 160 of the 256 CUs '
                                       '(40 frames per workgroup).  The fused launches changed the step by less than the box-to-box '
                                       'spread: the backward chain they replace ran beside the side stream\'s weight-gradient GEMMs'}
    U = config.num_lstm_cell_units
    M = config.batch_size * config.k
    gate = {'note': 'standalone micro-benchmark of d2p_lstm_gate_fwd/_bwd; the training step never launches them '
                    '(gate math = epilogue of the recurrent kernels)'}
    for label, rows in (('rows_32000', 32000), ('rows_per_step_batched', M * config.max_demo_len),
                        ('rows_per_time_step', M)):
        per_set = rows * (4 * U + U + U + U + U + U + 4 * U) * 4.0       # bytes of one buffer set
        nset = max(2, int(640e6 / per_set) + 1)
        sets = []
        for _ in range(nset):
            sets.append(dict(z=torch.randn(rows, 4 * U, device='cuda'), c_prev=torch.randn(rows, U, device='cuda'),
                             c_out=torch.empty(rows, U, device='cuda'), h_out=torch.empty(rows, U, device='cuda'),
                             dh=torch.randn(rows, U, device='cuda'), dc=torch.randn(rows, U, device='cuda'),
                             dz=torch.empty(rows, 4 * U, device='cuda')))
        res = {'rows': rows, 'working_set_MB': round(nset * per_set / 1e6, 1)}

        def fwd(b):
            K.lstm_gate_fwd(b['z'], b['c_prev'], None, None, 0, b['c_out'], None, b['h_out'])

        def bwd(b):
            K.lstm_gate_bwd(b['z'], b['c_prev'], b['c_out'], b['dh'], None, None, 0, b['dc'], b['dz'], None)
        for name, fn, nbytes in (('fwd', fwd, rows * 7 * U * 4.0), ('bwd', bwd, rows * 13 * U * 4.0)):
            for b in sets[:3]:
                fn(b)
            reps = max(30, 2 * nset)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(reps):
                fn(sets[i % nset])
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / reps
            gbs = nbytes / t / 1e9
            res[name] = {'us': round(t * 1e6, 2), 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                         'frac': round(gbs / PEAK_HBM_GBS, 4),
                         'frac_of_measured_copy_rate': round(gbs / MEASURED_HBM_GBS, 4)}
        gate[label] = res
        del sets
        torch.cuda.empty_cache()
    out['lstm_gate_kernel'] = gate
    out['recurrent_step'] = recurrent_step_microbench(M, U, config.max_demo_len)
    return out


def recurrent_step_microbench(M, U, T):
    """The recurrent kernels on the encoder / action-decoder shape (M = B*k rows, T steps) outside the step:
    one persistent launch per sequence timed at T and 4T steps, so that the per-call cost (preparation launch,
    kernel prologue: weights into registers, launch latency) separates from the per-time-step rate.  flops per
    time step = 2 * M * 4U * U (the h.Wh / dz.Wh^T product; the input projection is a separate GEMM)."""
    from demo2program_amd import kernels as K
    g = torch.Generator().manual_seed(7)

    def seq(Tn):
        f = dict(M=M, U=U, n_steps=Tn, z=(torch.rand(Tn * M, 4 * U, generator=g) - 0.5).cuda(),
                 Wh=((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.1).cuda(),
                 h0=(torch.rand(M, U, generator=g) - 0.5).cuda(), c0=(torch.rand(M, U, generator=g) - 0.5).cuda(),
                 hout=torch.zeros(Tn, M, U, device='cuda'), cs=torch.zeros(Tn, M, U, device='cuda'))
        b = dict(M=M, U=U, n_steps=Tn, z=f['z'], Wh=f['Wh'], c0=f['c0'], cs=f['cs'],
                 dhout=(torch.rand(Tn, M, U, generator=g) - 0.5).cuda(), dz=torch.zeros(Tn * M, 4 * U, device='cuda'),
                 dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
        return f, b

    def timed(fn, reps=10):
        fn()
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps           # us per call
    res = {'rows': M, 'units': U, 'note': 'standalone: one persistent launch per sequence (+ its preparation launch) '
                                           'at T and 4T steps; peak = fp32 MFMA'}
    flop_step = 2.0 * M * 4 * U * U
    for name, key in (('forward', 0), ('backward', 1)):
        ts = []
        for Tn in (T, 4 * T):                 # a long second point: the slope is the steady per-step time
            q = seq(Tn)
            if key == 0:
                ts.append(min(timed(lambda: K.lstm_seq_fwd_multi([q[0]])) for _ in range(3)))
            else:
                K.lstm_seq_fwd_multi([q[0]])
                ts.append(min(timed(lambda: K.lstm_seq_bwd_multi([q[1]])) for _ in range(3)))
        steady = (ts[1] - ts[0]) / (3 * T)
        res[name] = {'us_per_call_at_T': round(ts[0], 1), 'T': T,
                     'us_per_time_step_at_T': round(ts[0] / T, 2),
                     'frac_at_T': round(flop_step / (ts[0] / T * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                     'us_per_time_step_steady': round(steady, 2),
                     'frac_steady': round(flop_step / (steady * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                     'us_per_call_fixed': round(ts[0] - steady * T, 1)}
    if K.lstm_persist_error(True):
        res['error'] = 'a persistent launch gave up a hand-off during this micro-benchmark'
    return res


def config4_leg(steps=10, warmup=3, preset='vizdoom'):
    """BASELINE config 4 (ViZDoom full model, k=10, 80x80x3 frames, B=32; preset 'vizdoom_k25': config 5's per-rank
    shape, k=25, B=16) in the same run: ms/step of the same optimizer step, inputs resident in HBM, and the conv
    encoder's fraction of the fp32 MFMA peak from an instrumented pass."""
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    cfg = make_config(preset)
    tr = Trainer(cfg, make_train_dir=False)
    batches = [make_batch(cfg, seed=321 + i) for i in range(2)]
    for b in batches:
        b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [tr.model.get_feed_dict(b) for b in batches]
    with tr.step_stream():
        for i in range(warmup):
            tr.train_step(feeds[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.train_step(feeds[i % 2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    res = {'workload': 'vizdoom full model, k=%d, %dx%dx%d uint8 frames, T=%d, batch=%d, inputs resident in HBM'
                       % (cfg.k, cfg.h, cfg.w, cfg.depth, cfg.max_demo_len, cfg.batch_size),
           'steps': steps, 'ms_per_step': round(dt * 1e3, 4), 'value': round(cfg.batch_size / dt, 2),
           'unit': 'instances/s'}
    _, _, table, _ = roofline_leg(tr, feeds)
    conv = [r for r in table if r['group'] == 'conv']
    if conv:
        br = conv_binding_roofline(cfg, conv[0]['ms_per_step'], conv[0]['launches_per_step'])
        # on ALGORITHMIC flops (3 input channels; the kernels multiply the zero-padded 4th channel of conv1 as well)
        alg = br['flops_per_step'] / (conv[0]['ms_per_step'] * 1e-3) / 1e12
        res['conv_encoder'] = {'achieved': round(alg, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(alg / PEAK_F32_MFMA_TFLOPS, 4),
                               'frac_counting_the_padded_channel': round(conv[0]['rate'] / PEAK_F32_MFMA_TFLOPS, 4),
                               'ms_per_step': conv[0]['ms_per_step'], 'launches_per_step': conv[0]['launches_per_step'],
                               'binding_roofline': br}
        # the State_Encoder as ONE family: conv launches + batch-norm launches (round 5: conv1 / conv2 carry their
        # batch-norm statistics, conv1's apply pass and its whole batch-norm backward inside the conv launches)
        bn = [r for r in table if r['group'] == 'bn']
        ms = conv[0]['ms_per_step'] + (bn[0]['ms_per_step'] if bn else 0.0)
        res['conv_bn_encoder'] = {
            'ms_per_step': round(ms, 4), 'conv_ms': conv[0]['ms_per_step'], 'bn_ms': bn[0]['ms_per_step'] if bn else 0.0,
            'launches_per_step': conv[0]['launches_per_step'] + (bn[0]['launches_per_step'] if bn else 0),
            'note': 'bn_ms includes the relation networks\' and the perception encoder\'s batch norms (small)',
            'algorithmic_fwd_MB': 2776, 'algorithmic_fwd_MB_source': 'SURVEY.md 8(d): conv stack with two-pass batch norm',
            'hbm_traffic': pmc_family_traffic(('conv_', 'bn_', '<conv>'), preset)}
    res['kernel_table'] = table
    del tr
    torch.cuda.empty_cache()
    return res


def cpu_baseline_leg(config, batch, params, steps=5):
    """The CPU oracle (torch-CPU fp32 restatement of the TF-1.3 graph + torch autograd) on the
    host cores of this box, same batch, same weights.  Reported, never the target."""
    import oracle
    from demo2program_amd.synthetic import to_torch
    # many small ops: more than ~16 threads only adds fork/join overhead (256-core host: >10x slower)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ocfg = oracle.OracleConfig(
        batch_size=config.batch_size, k=config.k, max_demo_len=config.max_demo_len,
        max_program_len=config.max_program_len, h=config.h, w=config.w, depth=config.depth,
        dim_program_token=config.dim_program_token, action_space=config.action_space,
        per_dim=config.per_dim, num_lstm_cell_units=config.num_lstm_cell_units,
        dataset_type=config.dataset_type)
    tb = to_torch(batch)
    tp = {n: torch.from_numpy(v) for n, v in params.items()}
    oracle.loss_and_grads(tp, tb, ocfg, dtype=torch.float32)       # warm-up
    samples = []
    for _ in range(steps):
        t0 = time.time()
        oracle.loss_and_grads(tp, tb, ocfg, dtype=torch.float32)
        samples.append(time.time() - t0)
    dt = sum(samples) / steps
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(config.batch_size / dt, 3), 'unit': 'instances/s',
            'cores': torch.get_num_threads(), 'host_cores': os.cpu_count(), 'kind': 'port',
            'samples_s': [round(x, 3) for x in samples],
            'thread_sweep': 'profiles/r03_cpu_thread_sweep.json (tools/cpu_baseline_sweep.py: 16 threads is the fastest '
                            'setting for this op mix on the 256-thread host)',
            'caveat': 'forward + backward only (no clip / Adam), %d of %d host threads, %d samples; a CPU restatement '
                      'in torch, not TensorFlow 1.3' % (torch.get_num_threads(), os.cpu_count() or 0, steps),
            'sample': '%d full forward+backward steps (no optimizer) of the torch-CPU oracle on the '
                      'same batch of %d programs x %d demos, fp32, %.2f s/step; CPU restatement, '
                      'not TF1 (TensorFlow 1.3 is not installable here)' %
                      (steps, config.batch_size, config.k, dt),
            'cpu_model': model}


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) through
    torch.distributed.run on 127.0.0.1 and a free port, pass the command line on unchanged, and exit with the
    launcher's status.  Rank 0 of the children prints the one JSON line on the inherited stdout.  The
    `python -m torch.distributed.run ... bench.py --gpus N` form keeps working: it sets WORLD_SIZE, so this is
    skipped."""
    import socket
    import subprocess
    from demo2program_amd import build
    build.build_library()                      # once, before N ranks ask for it
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != '--self-spawn']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    env['D2P_BENCH_SELF_SPAWNED'] = '1'
    if args.gpus == 1:
        env['D2P_FORCE_DIST'] = '1'            # a one-rank RCCL group: the exchange step runs, as an identity
    sys.stderr.write('[bench] no launcher: spawning %d rank(s): %s\n' % (args.gpus, ' '.join(cmd)))
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def log(msg):
    if os.environ.get('RANK', '0') == '0':
        sys.stderr.write('[bench %.1fs] %s\n' % (time.time() - _T0, msg))
        sys.stderr.flush()


_T0 = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--preset', default='karel', help='karel | vizdoom | vizdoom_k25 | karel_tiny')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--h2d', action='store_true', help='also report the sequential host-batch rate (fetch + H2D + step, '
                    'no prefetch); the prefetched PCIe-inclusive rate is always reported')
    ap.add_argument('--no-h2d', action='store_true', help='skip the PCIe-inclusive leg')
    ap.add_argument('--no-config4', action='store_true', help='skip the ViZDoom (BASELINE config 4) leg')
    ap.add_argument('--prof-keep-side', action='store_true',
                    help='diagnostic: the instrumented pass keeps the two-stream schedule (family times BESIDE the other queue)')
    ap.add_argument('--ablate', default='',
                    help='MEASUREMENT ONLY (tools/step_ablation.sh): comma-separated pieces of the step to leave out '
                         '(Model.set_ablation); the line is then marked invalid -- it is not a throughput of the model')
    ap.add_argument('--self-spawn', action='store_true',
                    help='start the N ranks from this process even for N = 1 (the path `python bench.py --gpus N` takes '
                         'for N > 1 when it was not launched by torch.distributed.run)')
    ap.add_argument('--frames', default='uint8', choices=['uint8', 'float32'],
                    help='precision the demonstration frames are staged in: uint8 = the dataset\'s own (Karel states are '
                         'booleans, ViZDoom frames bytes), widened on load inside conv1; float32 = the reference\'s feed dtype')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.self_spawn):
        self_spawn(args)

    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.dist import DataParallel
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer

    dp = DataParallel.from_env()
    if dp.world_size != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run '
                         '--nproc-per-node %d (or without a launcher: bench.py starts the ranks itself)'
                         % (args.gpus, dp.world_size, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
    if dp.rank == 0:
        build.build_library()
    dp.barrier()

    config = make_config(args.preset)
    log('building trainer')
    trainer = Trainer(config, make_train_dir=False, dp=dp)
    if args.ablate:
        trainer.model.set_ablation(args.ablate.split(','))
    log('trainer ready; making batches')
    # distinct per-rank synthetic batches, made resident in HBM before the timed region
    host_batches = [make_batch(config, seed=123 + 7919 * dp.rank + i) for i in range(4)]
    if args.frames == 'uint8':
        for b in host_batches:
            assert np.array_equal(b['s_h'].astype(np.uint8).astype(b['s_h'].dtype), b['s_h'])   # lossless
            b['s_h'] = b['s_h'].astype(np.uint8)
    feeds = [trainer.model.get_feed_dict(b) for b in host_batches]
    torch.cuda.synchronize()
    log('feeds resident; warmup')

    # (warm-up and timed region on the trainer's high-priority stream, as Trainer.train runs its loop: Trainer.step_stream)
    with trainer.step_stream():
        for i in range(args.warmup):
            trainer.train_step(feeds[i % len(feeds)])
        torch.cuda.synchronize()
        log('warmup done; timing %d steps' % args.steps)
        dp.barrier()
        torch.cuda.synchronize()
        waited0 = trainer.guard.waited
        t0 = time.perf_counter()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
        for i in range(args.steps):
            loss = trainer.train_step(feeds[i % len(feeds)])
            marks[i + 1].record()             # device-side step boundaries (no host sync inside the region)
        host_enqueue = time.perf_counter() - t0   # host time to ENQUEUE the K steps (device may still be running)
        # the step guard lets the host run at most StepGuard.DEPTH steps ahead: time it spent WAITING for the device is not
        # enqueue work
        host_enqueue -= trainer.guard.waited - waited0
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        elapsed = dp.max_over_ranks(time.perf_counter() - t0)
    final_loss = float(loss.item())
    log('timed region done: %.3f s' % elapsed)
    # a persistent LSTM launch that gave up a hand-off: the guarded step skipped it on the device and the trainer
    # re-ran it on the per-step kernels INSIDE the timed region (count reported; 0 on a dedicated GPU)
    persist_fallbacks = trainer.settle()

    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    step_stats = {'median': round(per_step[len(per_step) // 2], 4), 'p10': round(per_step[len(per_step) // 10], 4),
                  'p90': round(per_step[(len(per_step) * 9) // 10], 4), 'max': round(per_step[-1], 4)} if per_step else None
    global_batch = config.batch_size * dp.world_size
    value = global_batch * args.steps / elapsed
    out = {
        'metric': 'train instances/sec (batch x k demos) Karel full model' if args.preset == 'karel'
                  else 'train instances/sec (batch x k demos) %s full model' % args.preset,
        'value': round(value, 3), 'unit': 'instances/s', 'n_gpus': dp.world_size,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': '%s full model, k=%d, %dx%dx%d frames, T=%d, L=%d, batch=%d programs per GPU; '
                        'one step = forward + backward + %s + clip(20) + Adam; '
                        'inputs resident in HBM' %
                        (args.preset, config.k, config.h, config.w, config.depth, config.max_demo_len,
                         config.max_program_len, config.batch_size,
                         ('RCCL all-reduce of the flat gradient buffer over %d ranks' % dp.world_size) if dp.active
                         else 'no collective (one rank, no process group)'),
            'global_batch': global_batch, 'parallelism': 'dp%d' % dp.world_size,
            'frames': args.frames,
            'lstm_units': config.num_lstm_cell_units,
        },
        'demo_instances_per_sec': round(value * config.k, 1),
        'device_step_ms': step_stats,
        'host_enqueue_ms_per_step': round(host_enqueue * 1e3 / args.steps, 4),
        'final_loss': round(final_loss, 5),
        # ranks that took part in the gradient exchange: an all-reduce of ones over the job's process group
        # (None: one rank without a process group, no collective issued)
        'persistent_lstm_fallbacks': persist_fallbacks,
        'rccl_ranks_seen': dp.ranks_seen(),
        'launcher': 'self-spawned' if os.environ.get('D2P_BENCH_SELF_SPAWNED') == '1' else
                    ('torch.distributed.run' if 'TORCHELASTIC_RUN_ID' in os.environ or 'WORLD_SIZE' in os.environ
                     else 'none (one process)'),
    }

    if not args.no_h2d:
        # PCIe-inclusive rates: host numpy batch -> H2D -> step (never `value`)
        n = max(5, args.steps // 5)
        if args.h2d:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n):
                trainer.train_step(trainer.model.get_feed_dict(host_batches[i % len(host_batches)]))
            torch.cuda.synchronize()
            out['value_incl_h2d'] = round(global_batch * n / dp.max_over_ranks(time.perf_counter() - t1), 3)
        # the trainer's own loop: the next batch is pinned + copied on a copy stream while this step runs
        from demo2program_amd.trainer import FeedPrefetcher

        class _Cycle(object):
            def __init__(self, bs):
                self.bs, self.i = bs, 0

            def next(self):
                self.i += 1
                return self.bs[self.i % len(self.bs)]

        pf = FeedPrefetcher(trainer.model, _Cycle(host_batches))
        for _ in range(8):          # every staging set of the ring allocated (pinned) before timing
            trainer.train_step(pf.take())
            pf.stage()
        torch.cuda.synchronize()
        dp.barrier()
        t1 = time.perf_counter()
        for _ in range(n):
            trainer.train_step(pf.take())
            pf.stage()
        torch.cuda.synchronize()
        out['value_incl_h2d_prefetched'] = round(global_batch * n / dp.max_over_ranks(time.perf_counter() - t1), 3)
        pf.close()

    if not args.no_roofline:
        log('roofline leg')
        roof, furthest, table, every = roofline_leg(trainer, feeds, keep_side=args.prof_keep_side)
        out['roofline'] = roof
        out['roofline_every_group'] = every
        out['furthest_below_roofline'] = furthest
        out['kernel_table'] = table
        if dp.rank == 0:
            out['north_star_targets'] = north_star_targets(config, table)
    if dp.rank == 0 and dp.world_size == 1 and args.preset == 'karel' and not args.no_config4:
        log('config 4 (ViZDoom 80x80x3) leg')
        c4 = out['config4_vizdoom'] = config4_leg()
        # the numbers of that leg that matter, at the TOP level of the line ...
        out['config4_ms_per_step'], out['config4_value'] = c4['ms_per_step'], c4['value']
        if 'conv_encoder' in c4:
            out['config4_conv_frac'] = c4['conv_encoder']['frac']
            out['config4_conv_bn_ms'] = c4['conv_bn_encoder']['ms_per_step']
            out['config4_bn_launches'] = next((r['launches_per_step'] for r in c4['kernel_table'] if r['group'] == 'bn'), 0)
            tr_ = c4['conv_bn_encoder'].get('hbm_traffic')
            # measured conv + batch-norm HBM bytes per step / the algorithmic bytes (2 776 MB forward with a
            # two-pass batch norm; backward reads / writes each tensor about twice more -> x3)
            out['config4_traffic_ratio'] = round(tr_['MB_per_step'] / (3 * 2776.0), 3) if tr_ else None
        log('config 5 per-rank shape (ViZDoom k=25, B=16) leg')
        c5 = config4_leg(steps=5, warmup=2, preset='vizdoom_k25')
        c5.pop('kernel_table', None)
        out['config5_per_rank_vizdoom_k25'] = c5
        out['config5_rank_ms_per_step'], out['config5_rank_value'] = c5['ms_per_step'], c5['value']
        # ... and inside `config`, which the driver's record keeps whole
        out['config']['also_measured_in_this_run'] = {n: out.get(n) for n in (
            'config4_ms_per_step', 'config4_value', 'config4_conv_frac', 'config4_conv_bn_ms', 'config4_traffic_ratio',
            'config5_rank_ms_per_step', 'config5_rank_value')}
    if dp.rank == 0 and dp.world_size == 1 and not args.no_cpu_baseline:
        from demo2program_amd.params import init_params
        log('cpu baseline leg (%d host cores)' % (os.cpu_count() or 1))
        out['cpu_baseline'] = cpu_baseline_leg(config, host_batches[0], init_params(config, 123))
    if args.ablate:
        out['invalid'] = 'timing-only ablation: the step leaves out %s -- not a throughput of the model' % args.ablate
        out['value'] = None
    if dp.rank == 0:
        # the long tables first, the contract fields and the summary numbers LAST: a reader who keeps only the tail of
        # the line still sees them
        bulky = ('kernel_table', 'north_star_targets', 'config4_vizdoom', 'config5_per_rank_vizdoom_k25',
                 'roofline_every_group', 'furthest_below_roofline')
        ordered = {n: out[n] for n in bulky if n in out}
        ordered.update({n: v for n, v in out.items() if n not in bulky})
        print(json.dumps(ordered))
    dp.shutdown()


if __name__ == '__main__':
    main()
