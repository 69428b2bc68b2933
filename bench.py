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
  roofline     -- the dominant kernel family of the step, timed live with HIP events on the
                  launch stream in an instrumented pass right after the timed region
  cpu_baseline -- the CPU oracle (torch-CPU restatement, NOT TF1) timed on this box's host
                  cores on the same batch (rank 0, N=1 only).
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
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6290 GB/s measured copy)

# kernel family -> key in profiles/*_pmc_traffic.json (tools/pmc_summary.py)
PMC_KEYS = {1: 'gemm_mfma_kernel', 2: 'gemm_mfma_kernel<conv>', 7: 'lstm_step_fwd_kernel',
            8: 'lstm_step_bwd_kernel', 3: 'lstm_gate_fwd_kernel', 4: 'lstm_gate_bwd_kernel'}
PMC_FILE = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')

PROF_FAMILIES = {
    1: ('gemm_mfma_kernel (dense fp32 MFMA GEMM)', 'mfma'),
    2: ('conv kernels (whole-frame / direct 16x16x4 MFMA, implicit-GEMM fallback)', 'mfma'),
    3: ('lstm_gate_fwd_kernel', 'hbm'),
    4: ('lstm_gate_bwd_kernel', 'hbm'),
    7: ('lstm_step_fwd_kernel (fused recurrent GEMM + gates)', 'mfma'),
    8: ('lstm_step_bwd_kernel (fused recurrent GEMM + gate backward)', 'mfma'),
}


def pmc_traffic(family):
    """HBM bytes per launch of this kernel family from the committed PMC passes of this same
    command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per the
    gfx950 correction; tools/profile_pmc.sh + tools/pmc_summary.py).  None if not collected."""
    try:
        d = json.load(open(PMC_FILE))
        return round(d[PMC_KEYS[family]]['hbm_bytes_per_launch'], 1)
    except (OSError, KeyError, ValueError):
        return None


def roofline_leg(trainer, feeds, steps=2):
    """Re-runs `steps` training steps with per-launch HIP events enabled inside the library
    and reports the family/tag with the largest summed device time."""
    from demo2program_amd.lib import load
    lib = load()
    torch.cuda.synchronize()
    lib.d2p_prof_enable(1)
    lib._d2p_prof_on = True          # Trainer.train_step takes the eager (un-graphed) path
    # one stream, so that a launch's event bracket measures that kernel alone (in the timed
    # region independent GEMMs overlap the recurrent step kernels on a side stream)
    side = trainer.model.use_side_stream
    trainer.model.use_side_stream = False
    for i in range(steps):
        trainer.train_step(feeds[i % len(feeds)])
    torch.cuda.synchronize()
    rows = []
    for fam, (name, bound) in PROF_FAMILIES.items():
        for tag in (0, 1):
            cnt, ms, work = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
            lib.d2p_prof_read(fam * 8 + tag, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(work))
            if cnt.value:
                rows.append(dict(family=fam, tag=tag, name=name, bound=bound, launches=cnt.value,
                                 total_ms=ms.value, work=work.value))
    lib.d2p_prof_enable(0)
    lib._d2p_prof_on = False
    trainer.model.use_side_stream = side
    if not rows:
        return None, []
    rows.sort(key=lambda r: -r['total_ms'])
    top = rows[0]
    sec = top['total_ms'] / 1e3
    if top['bound'] == 'mfma':
        achieved, peak, unit = top['work'] / sec / 1e12, PEAK_F32_MFMA_TFLOPS, 'TFLOP/s'
    else:
        achieved, peak, unit = top['work'] / sec / 1e9, PEAK_HBM_GBS, 'GB/s'
    roof = {
        'kernel': top['name'] + (' [inside the recurrence]' if top['tag'] == 1 else ''),
        'bound': top['bound'], 'achieved': round(achieved, 3), 'peak': peak, 'unit': unit,
        'frac': round(achieved / peak, 4), 'traffic': pmc_traffic(top['family']),
        'launches_per_step': top['launches'] / steps,
        'avg_launch_us': round(top['total_ms'] * 1e3 / top['launches'], 3),
        'work_per_launch': top['work'] / top['launches'],
        'share_of_instrumented_ms': round(top['total_ms'] / sum(r['total_ms'] for r in rows), 3),
    }
    table = [dict(kernel=r['name'], tag=r['tag'], launches_per_step=r['launches'] / steps,
                  ms_per_step=round(r['total_ms'] / steps, 4),
                  rate=round(r['work'] / (r['total_ms'] / 1e3) / (1e12 if r['bound'] == 'mfma' else 1e9), 2),
                  unit='TFLOP/s' if r['bound'] == 'mfma' else 'GB/s') for r in rows]
    return roof, table


def north_star_targets(config, table):
    """The two kernel-level targets BASELINE.json's north_star names, measured live:
    (1) the conv encoder's fraction of the fp32 MFMA peak (conv family of the instrumented pass:
        forward + dgrad + wgrad of every layer, 2*M*K*N flops each);
    (2) the LSTM gate kernel's fraction of the HBM roofline (14 336 B/row forward,
        26 624 B/row backward at U=512), on the rows of one training step applied in one launch
        (B*k*T rows) and, for reference, at the per-time-step granularity (B*k rows), where a
        4.6 MB launch is latency-bound.  In the shipped path the gate math is the epilogue of
        the fused recurrent step kernels; these standalone kernels are the d2p_lstm_gate_* ABI
        entry points (and the unfused d2p_lstm_set_fused(0) path)."""
    from demo2program_amd import kernels as K
    out = {}
    conv = [r for r in table if r['kernel'].startswith('conv kernels')]
    if conv:
        out['conv_encoder'] = {'achieved': conv[0]['rate'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(conv[0]['rate'] / PEAK_F32_MFMA_TFLOPS, 4),
                               'ms_per_step': conv[0]['ms_per_step'],
                               'launches_per_step': conv[0]['launches_per_step']}
    U = config.num_lstm_cell_units
    M = config.batch_size * config.k
    gate = {}
    for label, rows in (('rows_per_step_batched', M * config.max_demo_len), ('rows_per_time_step', M)):
        z = torch.randn(rows, 4 * U, device='cuda')
        c_prev = torch.randn(rows, U, device='cuda')
        c_out, h_out = torch.empty(rows, U, device='cuda'), torch.empty(rows, U, device='cuda')
        dh, dc = torch.randn(rows, U, device='cuda'), torch.randn(rows, U, device='cuda')
        dz = torch.empty(rows, 4 * U, device='cuda')
        res = {'rows': rows}
        for name, fn, nbytes in (
                ('fwd', lambda: K.lstm_gate_fwd(z, c_prev, None, None, 0, c_out, None, h_out), rows * 7 * U * 4.0),
                ('bwd', lambda: K.lstm_gate_bwd(z, c_prev, c_out, dh, None, None, 0, dc, dz, None), rows * 13 * U * 4.0)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / 50
            res[name] = {'us': round(t * 1e6, 2), 'achieved': round(nbytes / t / 1e9, 1), 'peak': PEAK_HBM_GBS,
                         'unit': 'GB/s', 'frac': round(nbytes / t / 1e9 / PEAK_HBM_GBS, 4)}
        gate[label] = res
    out['lstm_gate_kernel'] = gate
    return out


def cpu_baseline_leg(config, batch, params, steps=2):
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
    t0 = time.time()
    for _ in range(steps):
        oracle.loss_and_grads(tp, tb, ocfg, dtype=torch.float32)
    dt = (time.time() - t0) / steps
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(config.batch_size / dt, 3), 'unit': 'instances/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d full forward+backward steps (no optimizer) of the torch-CPU oracle on the '
                      'same batch of %d programs x %d demos, fp32, %.2f s/step; CPU restatement, '
                      'not TF1 (TensorFlow 1.3 is not installable here)' %
                      (steps, config.batch_size, config.k, dt),
            'cpu_model': model}


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
    ap.add_argument('--h2d', action='store_true', help='also report the host-batch (PCIe-inclusive) rate')
    ap.add_argument('--frames', default='uint8', choices=['uint8', 'float32'],
                    help='precision the demonstration frames are staged in: uint8 = the dataset\'s own (Karel states are '
                         'booleans, ViZDoom frames bytes), widened on load inside conv1; float32 = the reference\'s feed dtype')
    args = ap.parse_args()

    from demo2program_amd import build
    from demo2program_amd.config import make_config
    from demo2program_amd.dist import DataParallel
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer

    dp = DataParallel.from_env()
    if dp.world_size != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run '
                         '--nproc-per-node %d' % (args.gpus, dp.world_size, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
    if dp.rank == 0:
        build.build_library()
    dp.barrier()

    config = make_config(args.preset)
    log('building trainer')
    trainer = Trainer(config, make_train_dir=False, dp=dp)
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

    for i in range(args.warmup):
        trainer.train_step(feeds[i % len(feeds)])
    torch.cuda.synchronize()
    log('warmup done; timing %d steps' % args.steps)
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for i in range(args.steps):
        loss = trainer.train_step(feeds[i % len(feeds)])
        marks[i + 1].record()             # device-side step boundaries (no host sync inside the region)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0)
    final_loss = float(loss.item())
    log('timed region done: %.3f s' % elapsed)

    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    step_stats = {'median': round(per_step[len(per_step) // 2], 4), 'p10': round(per_step[len(per_step) // 10], 4),
                  'p90': round(per_step[(len(per_step) * 9) // 10], 4)} if per_step else None
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
                        'one step = forward + backward + grad all-reduce + clip(20) + Adam; '
                        'inputs resident in HBM' %
                        (args.preset, config.k, config.h, config.w, config.depth, config.max_demo_len,
                         config.max_program_len, config.batch_size),
            'global_batch': global_batch, 'parallelism': 'dp%d' % dp.world_size,
            'frames': args.frames,
            'lstm_units': config.num_lstm_cell_units,
        },
        'demo_instances_per_sec': round(value * config.k, 1),
        'device_step_ms': step_stats,
        'final_loss': round(final_loss, 5),
    }

    if args.h2d:
        # PCIe-inclusive rate: host numpy batch -> H2D -> step (never `value`)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n = max(5, args.steps // 5)
        for i in range(n):
            trainer.train_step(trainer.model.get_feed_dict(host_batches[i % len(host_batches)]))
        torch.cuda.synchronize()
        out['value_incl_h2d'] = round(global_batch * n / dp.max_over_ranks(time.perf_counter() - t1), 3)
        # the same with the trainer's prefetcher: next batch pinned + copied on a side stream meanwhile
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
        t1 = time.perf_counter()
        for _ in range(n):
            trainer.train_step(pf.take())
            pf.stage()
        torch.cuda.synchronize()
        out['value_incl_h2d_prefetched'] = round(global_batch * n / dp.max_over_ranks(time.perf_counter() - t1), 3)
        pf.close()

    if not args.no_roofline:
        log('roofline leg')
        roof, table = roofline_leg(trainer, feeds)
        out['roofline'] = roof
        out['kernel_table'] = table
        if dp.rank == 0:
            out['north_star_targets'] = north_star_targets(config, table)
    if dp.rank == 0 and dp.world_size == 1 and not args.no_cpu_baseline:
        from demo2program_amd.params import init_params
        log('cpu baseline leg (%d host cores)' % (os.cpu_count() or 1))
        out['cpu_baseline'] = cpu_baseline_leg(config, host_batches[0], init_params(config, 123))
    if dp.rank == 0:
        print(json.dumps(out))
    dp.shutdown()


if __name__ == '__main__':
    main()
