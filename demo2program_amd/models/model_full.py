"""Full model (demonstration encoder -> summarizer -> program / action / perception decoders)
on MI355X, behind the reference's ``Model`` surface.

Mirrors ``models/model_full.py:22-1132`` of shaohua0116/demo2program: same constructor,
``get_feed_dict(batch_chunk)``, ``loss``, ``output``, ``pred_program`` ... attributes.  The
TF-1.3 graph is replaced by explicit ``forward`` / ``backward`` schedules of calls into
libd2p_hip.so (include/d2p.h).  PyTorch provides device memory, the stream and views only.

What is batched differently from the reference (results are identical):
  * the reference builds k copies of the Demo encoder, SecondPath encoder and the action /
    perception decoders, one per demonstration index (:373-398, :530-599).  Here all
    M = B*k sequences run as one batch; per-index batch-norm statistics are kept
    by the grouped BN kernel (group = demonstration index), per-index loss normalisation by
    the grouped cross-entropy kernels.
  * the LSTM input projections x·Wx + b are hoisted out of the recurrence (one GEMM over all
    steps); only h·Wh + gates run per step.
  * rn_pool's first layer is factorised, fc1([f_c || f_a]) = f_c·W1[:U] + f_a·W1[U:], so the
    [B*k*k, 2U] pair matrix (:335-341) is never materialised.
Row order everywhere: sequences m = b*k + i; recurrent tensors are time-major [T, M, *].
"""
import sys

import numpy as np
import torch

from .. import kernels as K
from ..config import conv_shapes, feature_dim, n_conv
from ..options import flag
from ..params import FlatParams


def pick_concurrent_stream(against=None, max_tries=12, verbose=False):
    """A stream whose work really runs beside the work of the streams in `against` (default: the current
    stream).  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation
    order, and two streams on one queue serialise: a fresh `torch.cuda.Stream()` is concurrent with the default
    stream in a plain process, but NOT after `init_process_group` (RCCL's own streams shift the assignment) --
    measured: the two-stream schedule then runs at the one-stream rate, 4.98 instead of 4.44 ms per step.  So
    candidates are PROBED: a spin kernel on every stream of `against`, a tiny fill on the candidate; the
    candidate is taken when its fill finishes long before the spins do."""
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream()
    main = torch.cuda.current_stream()
    against = [main] if against is None else list(against)
    probe = torch.zeros(64, device='cuda')
    for st in against:
        with torch.cuda.stream(st):
            torch.cuda._sleep(1000)        # load the spin kernel before anything is timed
    cand = None
    for i in range(max_tries):
        cand = torch.cuda.Stream()
        with torch.cuda.stream(cand):
            probe.fill_(0.0)               # first use of a stream costs the host milliseconds: not part of the probe
        torch.cuda.synchronize()
        e0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ends = []
        e0.record(main)
        for st in against:
            if st != main:
                st.wait_event(e0)
            with torch.cuda.stream(st):
                torch.cuda._sleep(2000000)
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                ends.append(e)
        with torch.cuda.stream(cand):
            probe.fill_(1.0)
            c1.record(cand)
        torch.cuda.synchronize()
        spin, side = min(e0.elapsed_time(e) for e in ends), e0.elapsed_time(c1)
        if verbose:
            print('[stream probe %d] spin %.3f ms, candidate done after %.3f ms' % (i, spin, side), file=sys.stderr)
        if side < 0.5 * spin:
            return cand
    print('[demo2program_amd] no stream concurrent with %d busy stream(s) in %d tries: that schedule will '
          'serialise' % (len(against), max_tries), file=sys.stderr)
    return cand


class Model(object):

    def __init__(self, config, debug_information=False, is_train=True, global_step=None,
                 params=None, seed=123):
        self.debug = debug_information
        self.global_step = global_step
        self.config = config
        self.is_train = is_train
        self.dataset_type = config.dataset_type
        self.scheduled_sampling = getattr(config, 'scheduled_sampling', False) or False
        self.scheduled_sampling_decay_steps = \
            getattr(config, 'scheduled_sampling_decay_steps', 5000) or 5000
        self.batch_size = config.batch_size
        self.encoder_rnn_type = config.encoder_rnn_type
        self.num_lstm_cell_units = config.num_lstm_cell_units
        self.demo_aggregation = config.demo_aggregation    # only the synthesis baseline reads it
        # 'full', or one of the program-synthesis ablations that reuse this graph: 'summarizer'
        # (models/baselines/model_summarizer.py: both encoder passes, relation network alone as
        # the summary, program loss only) and 'synthesis_baseline' (model_synthesis.py: one encoder
        # pass, avg / max pooling over the demonstrations, program loss only)
        self.variant = getattr(config, 'model', 'full') or 'full'
        if self.variant not in ('full', 'summarizer', 'synthesis_baseline'):
            raise NotImplementedError('%s: not built (the induction baseline is an attention decoder '
                                      'over the test demonstrations, a different graph)' % self.variant)
        if self.variant == 'synthesis_baseline' and self.demo_aggregation not in ('avgpool', 'maxpool'):
            # 'concat' hands a [B, k*U] state to a U-unit cell (model_synthesis.py:339-341,463-467)
            raise ValueError('Unknown demo aggregation type')
        self.multitask = self.variant == 'full'
        self.dim_program_token = config.dim_program_token
        self.max_program_len = config.max_program_len
        self.max_demo_len = config.max_demo_len
        self.max_action_len = self.max_demo_len
        self.k = config.k
        self.test_k = getattr(config, 'test_k', 5)
        self.h, self.w, self.depth = config.h, config.w, config.depth
        self.action_space = config.action_space
        self.per_dim = config.per_dim
        # models/model_full.py:70-77: the DSL vocabulary behind intseq2str / the 'm)' end token
        if self.dataset_type == 'karel':
            from ..karel_env import get_KarelDSL
            self.vocab = get_KarelDSL(dsl_type=getattr(config, 'dsl_type', 'prob'), seed=123)
        else:
            from ..vizdoom_env import VizDoomDSLVocab
            try:
                self.vocab = VizDoomDSLVocab(perception_type=getattr(config, 'perception_type', ''),
                                             level=getattr(config, 'level', None))
            except NotImplementedError:
                self.vocab = None       # py2-ordered vocabularies: training works, DSL metrics refuse
        # ViZDoom execution metrics run on the caller's engine: world_factory() -> world
        # (program_metrics.generate_program_output_vizdoom); None = syntax / exact-program only
        self.world_factory = getattr(config, 'world_factory', None)

        if self.scheduled_sampling and global_step is None:
            raise ValueError('scheduled sampling requires global_step')       # model_full.py:59-61
        if self.encoder_rnn_type != 'lstm':
            # the reference reads cell_state.h/.c (models/model_full.py:258), which only an
            # LSTMStateTuple has: 'rnn' / 'gru' cannot work there either
            raise ValueError('Unknown encoder rnn type')
        # is_train=False (evaler.py:61): every batch norm normalises with its moving statistics;
        # forward / greedy decoding / report only -- backward() refuses.
        if not torch.cuda.is_available():
            raise RuntimeError('demo2program_amd.Model needs an MI355X (torch.cuda unavailable); '
                               'there is no CPU fallback')

        self.params = FlatParams(config, values=params, seed=seed)
        self._bufs = {}
        self._feed = None
        self._ctx = None
        self._marks = None                    # (tools/step_marks.py: list of (name, event) while measuring)
        self.forward_count = 0                # forward passes so far (trainer.StepOutput: is a step's output still there?)
        self._conv = conv_shapes(config)
        self._fused_enc_ok = {}
        self._fused_rn_ok = {}
        self.fused_rn = True                  # (tests switch it off on a Model object: the nine + eleven separate launches)
        self.decoder_skip_past_len = True     # (tests switch it off on a Model object: the A side of their A/B)
        self._fused_enc_bwd_ok = {}
        self.feature_dim = feature_dim(config)
        # non-trainable BN moving statistics (updated inline, once per reference call): views of ONE buffer
        # (`moving_flat`), so that the trainer's step guard snapshots / restores all of them with a single device copy
        U = self.num_lstm_cell_units
        bn_scopes = {'full': ('rn_h/fc1', 'rn_h/fc2', 'rn_c/fc1', 'rn_c/fc2', 'per/fc'),
                     'summarizer': ('rn_h/fc1', 'rn_h/fc2', 'rn_c/fc1', 'rn_c/fc2'),
                     'synthesis_baseline': ()}[self.variant]
        self._alloc_moving([('conv%d' % l, cout) for l, (_, _, _, cout, _, _) in enumerate(self._conv, start=1)],
                           bn_scopes, U)
        self.track_moving = True
        # scheduled sampling state (device memory: read by kernels inside a captured graph)
        #   _ss_prob : probability of feeding the decoder its own sample instead of the ground truth
        #              = 1 - polynomial_decay(1.0 -> 0.1 over decay_steps) (model_full.py:62-67,420-422)
        #   _ss_rng  : {seed, step counter}; set_sampling_step() advances it once per training step
        self._ss_prob = torch.zeros(1, device='cuda')
        self._ss_rng = torch.tensor([seed * 2654435761 + 12345, 0], dtype=torch.int64, device='cuda')
        if self.scheduled_sampling:
            self.set_sampling_step(int(global_step) if not callable(global_step) else 0)
        # all three decoders advanced by ONE per-step launch (d2p_lstm_seq_*_multi on the per-step kernels): an
        # attribute, not a switch -- tests/test_model_gpu.py sets it to check that path against the default
        self.fuse_decoders = False
        # the second encoder's input projection and dX run over the rows inside their sequences only (the rows past
        # a demonstration's length are zeros in its input and in dz), and the weight gradients over K lists of those
        # rows; a graph-static feed carries no row lists (their lengths would join the graph key), so a captured step
        # multiplies the zero rows in
        self.compact_rows = True
        self.k_rows = True
        # perception decoder: its batch-normed fc features are never multiplied by Wx row by row (forward _per_xproj)
        self.per_factored = flag('D2P_PER_FACTORED') and config.per_dim <= 8
        self.per_cols = (config.k * (config.per_dim + 1) + 3) // 4 * 4
        # token-input decoders: project the embedding TABLE and gather, instead of projecting gathered rows
        self.token_projection = flag('D2P_TOKEN_PROJECTION')
        # a second stream for the work that does not depend on the recurrences (decoder input projections in
        # forward, the weight-gradient products in backward): beside the persistent recurrences the GEMMs fill the
        # matrix pipe while a recurrence waits for its hand-offs
        self.use_side_stream = flag('D2P_SIDE_STREAM')
        self.fused_encoder = flag('D2P_FUSED_ENCODER')
        # batch norm folded into the conv launches where the geometry has folding kernels (the ViZDoom-size layers): an
        # attribute, not a switch -- tests set it to compare with the separate launches
        self.fold_bn = True
        # the decoders' small gradient products grouped into one launch (an attribute, not a switch: tests compare)
        self.grouped_decoder_grads = True
        # an encoder's two kernel-gradient halves as one product (an attribute, not a switch: tests compare)
        self.paired_kernel_grads = True
        self.fused_loss = flag('D2P_FUSED_LOSS')
        # timing-only ablation (tools/step_ablation.py -> set_ablation): NEVER from the environment; Trainer.train and
        # Evaler refuse a model that carries one
        self._ablate = frozenset()
        self._abl_cache = {}
        self._reserve_scratch()

    def mark(self, name):
        """MEASUREMENT HOOK (tools/step_marks.py): when `self._marks` is a list, a timing event on the CURRENT stream is
        appended under `name` -- the device time at which everything enqueued on that stream so far has finished.  Off
        (None) in every product path: one attribute test per call."""
        if self._marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    def set_ablation(self, names):
        """MEASUREMENT HOOK (tools/step_ablation.py, bench.py --ablate): leaves the named pieces of the step out from the
        second forward pass on -- their outputs go stale, the results of such a step are INVALID; what a piece is
        worth in the two-queue schedule is the step time without it.  Trainer.train / Evaler raise on such a model."""
        self._ablate = frozenset(n for n in names if n)
        self._abl_cache = {}

    def _abl(self, name):
        return name in self._ablate and self._abl_cache.get('ready', False)

    # ------------------------------------------------------------------ plumbing
    def _alloc_moving(self, conv, scopes, U):
        """self.moving[name] = (mean [C], variance [C]) as views of self.moving_flat.  The two relation networks' batch
        norms run as two-problem launches: mean of rn_h / rn_c of a layer adjacent ([2, U]), then their variances
        (checkpoint loading copies in place)."""
        chunks = [(name, C) for name, C in conv]
        rn = [leaf for leaf in ('fc1', 'fc2') if 'rn_h/' + leaf in scopes]
        chunks += [(s_, U) for s_ in scopes if not s_.startswith('rn_')]
        total = sum(2 * C for _, C in chunks) + 4 * U * len(rn)
        total = (total + 3) // 4 * 4
        flat = torch.zeros(max(total, 4), device='cuda')
        self.moving_flat = flat
        self.moving = {}
        o = 0
        for name, C in chunks:
            flat[o + C:o + 2 * C].fill_(1.0)
            self.moving[name] = (flat[o:o + C], flat[o + C:o + 2 * C])
            o += 2 * C
        for leaf in rn:
            flat[o + 2 * U:o + 4 * U].fill_(1.0)
            self.moving['rn_h/' + leaf] = (flat[o:o + U], flat[o + 2 * U:o + 3 * U])
            self.moving['rn_c/' + leaf] = (flat[o + U:o + 2 * U], flat[o + 3 * U:o + 4 * U])
            o += 4 * U

    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        t = self._bufs.get(name)
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device='cuda')
            self._bufs[name] = t
        return t

    def _reserve_scratch(self):
        from ..lib import call
        c = self.config
        B, k, T, U = c.batch_size, c.k, c.max_demo_len, c.num_lstm_cell_units
        M = B * k
        need = [call.d2p_lstm_ws_bytes(M, U), call.d2p_xent_ws_bytes(k),
                call.d2p_l2norm_ws_bytes(self.params.size),
                call.d2p_bn_ws_bytes(B * k * k, U, 1), call.d2p_bn_ws_bytes(T * M, U, k),
                call.d2p_colsum_ws_bytes(T * M, 4 * U),
                call.d2p_greedy_ws_bytes(M, U, max(c.dim_program_token, c.action_space)),
                call.d2p_gemm_ws_bytes(U, 4 * U, T * M), call.d2p_gemm_ws_bytes(2 * U, 4 * U, T * M),
                call.d2p_embedding_scatter_ws_bytes(T * M, c.dim_program_token + 1, U)]
        for (h, w, cin, cout, ho, wo) in self._conv:
            need.append(call.d2p_conv_ws_bytes(M * T, h, w, cin, cout))
            need.append(call.d2p_bn_ws_bytes(M * T * ho * wo, cout, k))
        K.SCRATCH.reserve(max(need))

    # ------------------------------------------------------------------ feed
    FEED_KEYS = ('s_h', 'program', 'program_tokens', 'a_h', 'a_h_tokens', 'per', 'per_rows', 'per_gram', 'active_rows',
                 'program_len',
                 'demo_len')

    def alloc_feed(self, frames_dtype=torch.float32):
        """Empty device feed: the tensors of FEED_KEYS as 256-byte aligned views into one byte buffer
        (kept under '_flat')."""
        c = self.config
        B, k, T, L = c.batch_size, c.k, c.max_demo_len, c.max_program_len
        cp = (c.depth + 3) // 4 * 4
        spec = [('s_h', frames_dtype, (B * k * T, c.h, c.w, cp)),
                ('program', torch.float32, (B, c.dim_program_token, L)),
                ('program_tokens', torch.int32, (B, L)),
                ('a_h', torch.float32, (B, k, T, c.action_space)),
                ('a_h_tokens', torch.int32, (B * k, T)),
                ('per', torch.float32, (B, k, T, c.per_dim)),
                # the perception rows spread by demonstration index (time-major) and their Gram matrix: what the
                # factored perception decoder multiplies instead of [rows, U] features (d2p.h: d2p_per_affine_rows)
                ('per_rows', torch.float32, (T * B * k, self.per_cols)),
                ('per_gram', torch.float32, (self.per_cols, self.per_cols)),
                # time-major row indices t*M + m of the demonstration steps inside their sequence (t < demo_len[m]),
                # padded to whole 32-row K slabs with the index of a row PAST its sequence (zeros in every dz)
                ('active_rows', torch.int32, (T * B * k + 32,)),
                # the same for the steps t >= 1, and those indices minus M (row t-1 of the same sequence): the two
                # operand lists of dWh = sum_t h[t-1]^T dz[t]; then both for the program decoder's rows
                ('rows_t1', torch.int32, (T * B * k + 32,)), ('rows_t1_prev', torch.int32, (T * B * k + 32,)),
                ('prog_rows', torch.int32, (L * B + 32,)),
                ('prog_rows_t1', torch.int32, (L * B + 32,)), ('prog_rows_t1_prev', torch.int32, (L * B + 32,)),
                # mask counts of the 1 + 2k loss terms (program, action per demo index, perception per demo index):
                # the denominators of Sequence_Loss (models/model_full.py:656-657), known with the lengths
                ('loss_dens', torch.float32, (1 + 2 * k,)),
                # the demonstrations by decreasing length: the backward recurrences group rows of similar length into
                # their row domains and stop each domain at its longest row (d2p_lstm_bwd_desc.rowmap)
                ('demo_order', torch.int32, (B * k,)),
                ('program_len', torch.int32, (B,)), ('demo_len', torch.int32, (B * k,))]
        offs, total = [], 0
        for _, dt, shape in spec:
            offs.append(total)
            n = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
            total += (n + 255) // 256 * 256
        flat = torch.empty(max(total, 256), dtype=torch.uint8, device='cuda')
        feed = {'_flat': flat}
        for (name, dt, shape), o in zip(spec, offs):
            n = int(np.prod(shape)) * torch.empty(0, dtype=dt).element_size()
            feed[name] = flat[o:o + n].view(dt).view(shape)
        return feed

    def derive_per_rows(self, feed):
        """Fills feed['per_rows'] / feed['per_gram'] from feed['per'] (part of staging a batch, like the channel
        padding of the frames): per_rows[t*M + b*k + i, i*(P+1) + j] = per[b, i, t, j], column i*(P+1) + P = 1,
        zeros elsewhere; per_gram = per_rows^T per_rows."""
        c = self.config
        B, k, T, P_ = c.batch_size, c.k, c.max_demo_len, c.per_dim
        NC = k * (P_ + 1)
        rows = feed['per_rows']
        rows.zero_()
        blk = rows.view(T, B, k, self.per_cols)[..., :NC].view(T, B, k, k, P_ + 1)
        ar = torch.arange(k, device=rows.device)
        blk[:, :, ar, ar, :P_] = feed['per'].permute(2, 0, 1, 3)
        blk[:, :, ar, ar, P_] = 1.0
        K.matmul_tn(rows, rows, out=feed['per_gram'])

    @staticmethod
    def row_lists(lens, R, cap, n_steps):
        """Time-major indices t*R + r of the rows inside their sequences (t < lens[r]), all of them and those with
        t >= 1, each padded to a multiple of 32 with the index of a row of the decoded steps that is PAST its sequence
        (zero in every dz; t >= 1) -- the K lists of d2p_gemm_f32_tn_rows.  -> (rows, rows_t1, n_pad, n_t1_pad); a
        padded count of 0 means "no list" (nothing to skip, or no pad row and a count that is not a multiple of 32)."""
        lens = np.minimum(np.asarray(lens, np.int64), cap)
        tt = np.arange(cap, dtype=np.int64)[:, None]
        inside = tt < lens[None, :]
        t_idx, r_idx = np.nonzero(inside)
        rows = (t_idx * R + r_idx).astype(np.int32)
        rows_t1 = rows[t_idx >= 1]
        past = np.nonzero((~inside) & (tt < n_steps) & (tt >= 1))
        pad = int(past[0][-1] * R + past[1][-1]) if past[0].size else -1

        def padded(a):
            if a.size == 0 or pad < 0 and a.size % 32:
                return a, 0
            n = (a.size + 31) // 32 * 32
            if n > a.size:
                a = np.concatenate([a, np.full(n - a.size, pad, np.int32)])
            return a, int(n)
        if pad < 0:                              # every row runs to the last decoded step: nothing to skip
            return rows, rows_t1, 0, 0
        rows, n_pad = padded(rows)
        rows_t1, n_t1_pad = padded(rows_t1)
        return rows, rows_t1, n_pad, n_t1_pad

    def get_feed_dict(self, batch_chunk, step=None, is_training=True):
        """batch_chunk (numpy arrays or torch tensors, keys of models/model_full.py:185-206)
        -> device-resident feed.  Lengths arrive as float32 and are cast to int32 exactly as the
        reference does (:155-171).  test_* / init_pos* entries are only read by metric
        py_funcs in the reference and stay on the host."""
        c = self.config
        B, k, T, L = c.batch_size, c.k, c.max_demo_len, c.max_program_len

        def host_np(x):
            return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)

        s_h = batch_chunk['s_h']
        s_dtype = torch.uint8 if getattr(s_h, 'dtype', None) in (np.uint8, torch.uint8) else torch.float32
        # all device tensors of a batch are views into ONE allocation (alloc_feed): a consumer that
        # needs a private copy (the trainer's graph-static feed) moves the batch with a single copy
        feed = self.alloc_feed(s_dtype)

        def put(name, x):
            t = torch.as_tensor(x) if not torch.is_tensor(x) else x
            dst = feed[name]
            if t.dtype != dst.dtype:
                t = t.to(dst.dtype)
            dst.copy_(t.reshape(dst.shape), non_blocking=True)

        if c.depth % 4 != 0:
            # device layout of the frames is NHWC with the channel count rounded up to 4 (zeros):
            # part of staging the batch, like the H2D copy, not of the training step
            raw = torch.as_tensor(s_h) if not torch.is_tensor(s_h) else s_h
            if raw.dtype != s_dtype:
                raw = raw.to(s_dtype)
            raw = raw.to('cuda', non_blocking=True).contiguous().view(B * k * T, c.h, c.w, c.depth)
            K.pad_axis(raw, B * k * T * c.h * c.w, c.depth, feed['s_h'].shape[3], 1, feed['s_h'])
        else:
            put('s_h', s_h)
        for name in ('program', 'program_tokens', 'a_h', 'a_h_tokens', 'per'):
            put(name, batch_chunk[name])
        self.derive_per_rows(feed)
        plen = host_np(batch_chunk['program_len']).astype(np.int32).reshape(B)
        dlen = host_np(batch_chunk['demo_len']).astype(np.int32).reshape(B * k)
        put('program_len', plen)
        put('demo_len', dlen)
        per_index = np.minimum(dlen, T).clip(0).reshape(B, k).sum(axis=0).astype(np.float32)
        put('loss_dens', np.concatenate([[np.float32(np.minimum(plen, L).clip(0).sum())], per_index, per_index]))
        # dynamic_decode runs until the longest sequence of the batch
        feed['n_prog'] = int(min(int(plen.max()) if B else 0, L))
        feed['n_demo'] = int(min(int(dlen.max()) if B * k else 0, T))
        for key, lens_np, R, cap, n_steps in (('', dlen, B * k, T, feed['n_demo']), ('prog_', plen, B, L, feed['n_prog'])):
            rows, rows_t1, n_pad, n_t1_pad = self.row_lists(lens_np, R, cap, n_steps)
            if key == '':
                feed['n_active'] = int(np.minimum(lens_np, cap).clip(0).sum())       # (without the padding)
            feed[key + 'n_active_pad'] = n_pad
            if rows.size:
                feed['active_rows' if key == '' else 'prog_rows'][:rows.size].copy_(torch.from_numpy(rows),
                                                                                    non_blocking=True)
            feed[key + 'n_t1_pad'] = n_t1_pad
            if rows_t1.size:
                feed[key + 'rows_t1'][:rows_t1.size].copy_(torch.from_numpy(rows_t1), non_blocking=True)
                feed[key + 'rows_t1_prev'][:rows_t1.size].copy_(torch.from_numpy(rows_t1 - R), non_blocking=True)
        dl = np.minimum(dlen, T).clip(0).astype(np.int64)
        order = np.argsort(-dl, kind='stable').astype(np.int32)
        put('demo_order', order)
        feed['demo_slab_steps'] = np.ascontiguousarray(dl[order][::16].astype(np.int32))     # (host: longest row per 16)
        feed['id'] = batch_chunk.get('id') if hasattr(batch_chunk, 'get') else None
        feed['host'] = {n: batch_chunk[n] for n in ('test_s_h', 'test_demo_len', 'test_per', 'init_pos',
                                                    'init_pos_len', 'test_init_pos', 'test_init_pos_len')
                        if n in batch_chunk}
        return feed

    # ------------------------------------------------------------------ forward
    def forward(self, feed, defer_loss=False):
        """defer_loss (Trainer.train_step: a backward pass follows at once): the loss VALUE -- the cross-entropy sums
        and their assembly, which nothing in backward reads: its denominators are mask counts that come with the feed
        -- is left on the side stream and joined where backward joins the streams."""
        c, p = self.config, self.params.p
        B, k, T, L = c.batch_size, c.k, c.max_demo_len, c.max_program_len
        U, V, A, P = c.num_lstm_cell_units, c.dim_program_token, c.action_space, c.per_dim
        M, NF, F = B * k, B * k * T, self.feature_dim
        if self.multitask and self.per_factored and 'per_rows' not in feed:
            # a hand-built feed (get_feed_dict derives these with the batch): same derivation, buffers of the model
            feed = dict(feed)
            feed['per_rows'] = self._buf('feed/per_rows', (T * M, self.per_cols))
            feed['per_gram'] = self._buf('feed/per_gram', (self.per_cols, self.per_cols))
            self.derive_per_rows(feed)
        ctx = {'feed': feed}
        self.forward_count += 1
        lens_d, lens_p = feed['demo_len'], feed['program_len']
        n_p, n_d = feed['n_prog'], feed['n_demo']
        # the demonstrations by decreasing length (with the feed): the encoders' forward and every backward recurrence
        # group rows of similar length into their row domains and stop each domain at its longest row
        order = None
        if feed.get('demo_slab_steps') is not None:
            order = (feed['demo_order'], feed['demo_slab_steps'])
        fwd_order = order

        # ---- side stream: everything that depends only on the batch -- decoder input ids,
        #      embeddings, the perception encoder and the three hoisted decoder projections
        #      (~30 GFLOP of MFMA-bound GEMM) -- runs concurrently with the encoder recurrences,
        #      whose step kernels leave most of the matrix pipe idle.
        main = torch.cuda.current_stream()
        side = self._side_stream()

        # ---- State_Encoder: conv -> +bias -> lrelu -> BN(train), per demo-index statistics
        self.mark('fwd:start')
        x = feed['s_h']
        ctx['conv'] = []
        feats_tm = None
        in_aff = None                  # (scale, shift) [k, C] when x is a pre-norm activation read through its batch-norm apply
        if self._abl('conv_fwd') and 'conv_fused' in self._abl_cache:        # (timing experiment: the previous step's)
            ctx['conv'], feats_tm = self._abl_cache['conv_fused']
        elif self._fused_encoder(B, k, T) and not self._abl('conv_fwd'):
            feats_tm = self._encoder_fwd_fused(x, ctx, main, side)
            self._abl_cache['conv_fused'] = (ctx['conv'], feats_tm)
        for l, (h, w, cin, cout, ho, wo) in enumerate(self._conv, start=1):
            if feats_tm is not None:
                break
            if self._abl('conv_fwd'):
                ctx['conv'], x = self._abl_cache['conv']
                break
            Wl = p['conv%d/W' % l]
            if l == 1 and cin % 4 != 0:
                # 3-channel (ViZDoom) frames: frames (once per batch, in get_feed_dict) and
                # weights are zero-padded to 4 channels so every tap is one 16-byte (uint8x4:
                # 4-byte) access; the extra channel contributes 0
                cp = (cin + 3) // 4 * 4
                if x.shape[3] != cp:
                    x = K.pad_axis(x, NF * h * w, cin, cp, 1,
                                   self._buf('conv1/xpad', (NF, h, w, cp), x.dtype))
                Wl = K.pad_axis(Wl, 9, cin, cp, cout, self._buf('conv1/Wpad', (3, 3, cp, cout)))
            name = 'conv%d' % l
            S = K.conv_bn_slices(x.shape, cout, k, T) if (self.is_train and self.fold_bn) else 0
            nxt = self._conv[l] if l < len(self._conv) else None
            # (round 6: every layer behind the first -- the 48-channel layers' kernels of conv_wide.hip take the affine too)
            fold_next = (S > 0 and nxt is not None and K.conv_bn_affine_ok((NF, ho, wo, cout), nxt[3], k, T))
            # (read through the affine by the next layer: the k pad pixels sit right behind the activation)
            a_ext = self._buf(name + '/a', (NF * ho * wo * cout + (k * cout if fold_next else 0),))
            a = a_ext[:NF * ho * wo * cout].view(NF, ho, wo, cout)
            if S > 0:
                # (round 5, the ViZDoom-size layers) batch norm folded into the conv launches: this layer's statistics
                # come out of its own conv launch (no partial-sum pass over `a`), and where the NEXT layer's kernels
                # can take it, the apply pass is folded into their input staging -- they read `a` through the affine
                # (gamma * rstd, beta - mean * gamma * rstd) and the normalised tensor is never written
                stats = self._buf(name + '/bn_partial', (k * S * cout * 2,), torch.float64)
                K.conv_fwd_bn(x, Wl, p[name + '/b'], k, T, S, stats, act=1, out=a, in_affine=in_aff)
                mean, rstd = self._buf(name + '/bn_mean', (k, cout)), self._buf(name + '/bn_rstd', (k, cout))
                var = self._buf(name + '/bn_var', (k, cout))
                aff = (self._buf(name + '/bn_scale', (k, cout)), self._buf(name + '/bn_shift', (k, cout)),
                       a_ext[NF * ho * wo * cout:].view(k, cout)) if fold_next else None
                K.bn_stats_from_partials(stats, B * T * ho * wo, cout, k, S, p[name + '/gamma'], p[name + '/beta'], mean, rstd,
                                         var, affine=aff)
                if self.track_moving:
                    st = side if self.use_side_stream and not torch.cuda.is_current_stream_capturing() else main
                    if st is not main:
                        st.wait_stream(main)
                    with torch.cuda.stream(st):
                        K.bn_update_moving(mean, var, *self.moving[name])
                ctx['conv'].append((x, a, mean, rstd, in_aff))
                if fold_next:
                    x, in_aff = a, aff                      # (the next layer reads a through the affine)
                else:
                    y = K.bn_apply_fwd(a.view(NF * ho * wo, cout), p[name + '/gamma'], p[name + '/beta'], mean, rstd, k,
                                       T * ho * wo, y=self._buf(name + '/bn_y', (NF * ho * wo, cout)))
                    x, in_aff = y.view(NF, ho, wo, cout), None
                continue
            assert in_aff is None
            K.conv_fwd(x, Wl, p[name + '/b'], act=1, out=a)
            y, mean, rstd = self._bn_fwd(name, a.view(NF * ho * wo, cout),
                                         p[name + '/gamma'], p[name + '/beta'], k, T * ho * wo)
            ctx['conv'].append((x, a, mean, rstd, None))
            x = y.view(NF, ho, wo, cout)
        if feats_tm is None:
            self._abl_cache['conv'] = (ctx['conv'], x)
            feats = x.view(M, T, F)
            feats_tm = K.transpose_rt(feats, M, T, F, out=self._buf('feats_tm', (T, M, F)))

        # (rows past their sequence: neither their projection nor, in backward, their input gradient is computed --
        #  the recurrence selects around what it reads there)
        act_rows = None
        if self.compact_rows and feed.get('n_active') is not None and 0 < feed['n_active'] < T * M:
            act_rows = (feed['active_rows'], feed['n_active'])
        ctx['rows_e1'] = act_rows
        if act_rows is not None:
            z_e1 = self._buf('demo_lstm/z', (T * M, 4 * U), zero=True)
            K.gemm_rows('nn', act_rows[1], 4 * U, F, feats_tm.view(T * M, F), F, p['demo_lstm/kernel'][:F], 4 * U,
                        z_e1, 4 * U, act_rows[0], bias=p['demo_lstm/bias'])
        else:
            z_e1 = self._lstm_xproj('demo_lstm', feats_tm.view(T * M, F), F, M, T, T)
        self.mark('fwd:conv+xproj1')
        # (forked here, not at the start of the step: beside the chain of small conv / batch-norm launches
        #  these GEMMs only took the CUs the chain was waiting for -- 320 us instead of 100 for the chain;
        #  beside the first recurrence they fill matrix-pipe time its hand-offs leave)
        # (round 6) the fork is an EVENT and the first recurrence is enqueued BEFORE the side stream's dozen small launches:
        # the host needs ~8 us per launch, and with the side work enqueued first the main queue sat idle for 77 us in
        # front of the recurrence (profiles/r05g_streams_timeline_two_queues.txt @ 79-156 us)
        fork = None
        if side != main:
            fork = torch.cuda.Event()
            fork.record(main)
        # ---- Demo_Encoder LSTM (zero initial state, length-masked)
        e1_hc = self._buf('demo_lstm/hc_final', (2, M, U))
        e1 = self._lstm_fwd('demo_lstm', feats_tm.view(T * M, F), F, M, T, T, None, None, lens_d,
                            want_final=True, z=z_e1, final_out=(e1_hc[0], e1_hc[1]), row_order=fwd_order)
        e1['hc_final'] = e1_hc
        self.mark('fwd:enc1')
        if fork is not None:
            side.wait_event(fork)
        with torch.cuda.stream(side):
            # Token-input decoders: x = embedding[id], so x.Wx + b takes one of tok+2 values per row -- the
            # projected TABLE (a [tok+1, U] x [U, 4U] GEMM: 7 or 51 rows) is gathered instead of projecting
            # 6400 gathered rows (13.4 GFLOP per decoder and direction; backward: _lstm_bwd_weights).  The
            # sampling decoders (scheduled sampling) choose their inputs step by step and keep the old form.
            tokproj = self.token_projection and not (self.scheduled_sampling and self.is_train)
            ids_p = K.shift_tokens_tm(feed['program_tokens'], V + 1, out=self._buf('ids_p', (L, B), torch.int32))
            emb_p = emb_a = None
            if not tokproj:
                emb_p = K.embedding_gather(ids_p, p['prog/embedding'], out=self._buf('emb_p', (L * B, U)), n=n_p * B)
            if self.multitask:
                ids_a = K.shift_tokens_tm(feed['a_h_tokens'], A + 1, out=self._buf('ids_a', (T, M), torch.int32))
                if not tokproj:
                    emb_a = K.embedding_gather(ids_a, p['act/embedding'], out=self._buf('emb_a', (T * M, U)),
                                               n=n_d * M)
                # Perception decoders: Per_Encoder = fc (no activation) + BN per demo index
                per_tm = K.transpose_rt(feed['per'].view(M, T, P), M, T, P, out=self._buf('per_tm', (T, M, P)))
                if self.per_factored and self.is_train and feed.get('per_gram') is not None:
                    # (round 4) the factored form below needs this batch norm's STATISTICS only, and those follow
                    # from the Gram matrix that comes with the feed: no K = 5 product over 6 400 rows, no batch-norm
                    # launches over its result (d2p_per_fc_bn_stats; 41 us of forward side-stream time)
                    pe_a = pe = None
                    pe_mean = self._buf('per/fc/bn_mean', (k, U))
                    pe_rstd = self._buf('per/fc/bn_rstd', (k, U))
                    pe_var = self._buf('per/fc/bn_var', (k, U))
                    K.per_fc_bn_stats(k, P, T * B, p['per/fc/W'], p['per/fc/b'], feed['per_gram'], pe_mean, pe_rstd, pe_var)
                    if self.track_moving:
                        K.bn_update_moving(pe_mean, pe_var, *self.moving['per/fc'])
                else:
                    pe_a = K.matmul_nn(per_tm.view(T * M, P), p['per/fc/W'], out=self._buf('pe_a', (T * M, U)),
                                       bias=p['per/fc/b'], act=0)
                    pe, pe_mean, pe_rstd = self._bn_fwd('per/fc', pe_a, p['per/fc/gamma'], p['per/fc/beta'], k, 1)
                z_a = (self._token_xproj('act', ids_a, A, M, T, n_d) if tokproj
                       else self._lstm_xproj('act/lstm', emb_a, U, M, T, n_d))
                if self.per_factored and self.is_train:
                    # pe = rows . H (H from the fc weights, the batch-norm parameters and this batch's statistics):
                    # z = rows . (H . Wx) + b -- a 60-row GEMM and a K = 60 GEMM instead of 6400 x 2048 x 512
                    H = K.per_affine_rows(k, P, p['per/fc/W'], p['per/fc/b'], p['per/fc/gamma'], p['per/fc/beta'],
                                          pe_mean, pe_rstd, self._buf('per/H', (self.per_cols, U)))
                    HWx = K.matmul_nn(H, p['per/lstm/kernel'][:U], out=self._buf('per/HWx', (self.per_cols, 4 * U)))
                    z_q = self._buf('per/lstm/z', (T * M, 4 * U))
                    if n_d > 0 and not self._abl('zq'):          # (timing experiment: tools/step_ablation.sh)
                        if K.per_rows_tn_ok(n_d * M, k, P, 4 * U):
                            # (round 4) from the structure of `rows` (P + 1 non-zeros per row): a write of z
                            K.per_rows_nn(k, per_tm.view(T * M, P), HWx, p['per/lstm/bias'], z_q, n_d * M)
                        else:
                            K.gemm_raw('nn', n_d * M, 4 * U, self.per_cols, feed['per_rows'], self.per_cols, HWx, 4 * U,
                                       z_q, 4 * U, bias=p['per/lstm/bias'])
                    pe = None
                else:
                    z_q = self._lstm_xproj('per/lstm', pe, U, M, T, n_d)
            z_p = (self._token_xproj('prog', ids_p, V, B, L, n_p) if tokproj
                   else self._lstm_xproj('prog/lstm', emb_p, U, B, L, n_p))

        if self.variant == 'synthesis_baseline':
            # model_synthesis.py:324-358: no second pass; the program decoder starts from the
            # demonstrations' final states pooled over k
            init_h, init_c = self._buf('pool_h', (B, U)), self._buf('pool_c', (B, U))
            if self.demo_aggregation == 'avgpool':
                K.group_mean(e1['h_final'], B, k, U, init_h, None)
                K.group_mean(e1['c_final'], B, k, U, init_c, None)
            else:
                ctx['arg_h'] = self._buf('pool_arg_h', (B, U), torch.int32)
                ctx['arg_c'] = self._buf('pool_arg_c', (B, U), torch.int32)
                K.group_max(e1['h_final'], B, k, U, init_h, ctx['arg_h'])
                K.group_max(e1['c_final'], B, k, U, init_c, ctx['arg_c'])
            e2 = rn_h = rn_c = h0_2 = c0_2 = demo_h = demo_c = None
        else:
            # ---- summary = mean over k; broadcast as SecondPath initial state
            # (h and c final states share one [2, M, U] buffer: the row-wise kernels see 2B programs, one launch)
            sum_hc, hc0_2 = self._buf('sum_hc', (2, B, U)), self._buf('hc0_2', (2, M, U))
            K.group_mean(e1['hc_final'], 2 * B, k, U, sum_hc, hc0_2)
            h0_2, c0_2 = hc0_2[0], hc0_2[1]
            # ---- SecondPathEncoder over the step-1 outputs (zeros past len)
            # the final states of all demonstrations, h then c, in one buffer: the two relation networks
            # (separate weights, same shapes) then run as strided-batched launches
            demo_hc = self._buf('demo_hc', (2, M, U))
            z_e2 = None
            ctx['rows'] = None
            if self.compact_rows and feed.get('n_active') is not None and 0 < feed['n_active'] < T * M:
                ctx['rows'] = (feed['active_rows'], feed['n_active'])
                z_e2 = self._buf('second_lstm/z', (T * M, 4 * U))
                K.gemm_rows('nn', feed['n_active'], 4 * U, U, e1['hout'], U, p['second_lstm/kernel'][:U], 4 * U,
                            z_e2, 4 * U, feed['active_rows'], bias=p['second_lstm/bias'])
            self.mark('fwd:mean+xproj2')
            e2 = self._lstm_fwd('second_lstm', e1['hout'].view(T * M, U), U, M, T, T, h0_2, c0_2, lens_d,
                                want_final=True, final_out=(demo_hc[0], demo_hc[1]), z=z_e2, row_order=fwd_order)
            self.mark('fwd:enc2')
            demo_h, demo_c = demo_hc[0], demo_hc[1]
            # ---- SummarizeFeature('rn') = mean_k + rn_pool (the summarizer baseline: rn_pool alone)
            if self._abl('rn_fwd'):
                rn_h = rn_c = self._abl_cache['rn']
            else:
                rn_h = rn_c = self._abl_cache['rn'] = self._rn_fwd(demo_hc, B, k, U, add_mean=self.multitask)
            init_h, init_c = rn_h['out'][0], rn_h['out'][1]

        self.mark('fwd:rn')
        main.wait_stream(side)
        # the initial states in front of the saved outputs (hbuf[0]): second encoder, the three decoders
        stage = [('prog/lstm', L, B, init_h)]
        if e2 is not None:
            stage.append(('second_lstm', T, M, h0_2))
        if self.multitask:
            stage += [('act/lstm', T, M, demo_h), ('per/lstm', T, M, demo_h)]
        if not (self.scheduled_sampling and self.is_train) and self.is_train:
            self._stage_h0(ctx, stage)
        # ---- Program decoder (teacher forcing; <s> = out-of-range id -> zero vector),
        #      action decoders (all k in one batch), perception decoders
        #      The three decoders are independent LSTMs.  fuse_decoders=True advances them
        #      together (one launch per time step for all three, d2p_lstm_seq_fwd_multi); measured
        #      neutral-to-slower on MI355X (the step kernels are L2-bandwidth bound, DESIGN.md
        #      3.2), so the default keeps one call per decoder.
        specs = [('prog', emb_p, U, B, L, n_p, init_h, init_c, V, z_p)]
        if self.multitask:
            specs += [('act', emb_a, U, M, T, n_d, demo_h, demo_c, A, z_a),
                      ('per', pe, U, M, T, n_d, demo_h, demo_c, P, z_q)]
        da = dq = None
        side_loss = False
        if self.scheduled_sampling and self.is_train:
            # program and action decoders feed back their own samples (per uses TrainingHelper,
            # model_full.py:409); the hoisted z_p / z_a of the side stream are simply not used
            dp = self._decoder_fwd_sampled('prog', ids_p, B, L, n_p, init_h, init_c, V, 0)
            ids_p = dp['fed_ids']                               # embedding gradient goes to what was fed
            if self.multitask:
                da = self._decoder_fwd_sampled('act', ids_a, M, T, n_d, demo_h, demo_c, A, 4096)
                dq = self._decoders_fwd([specs[2]])[0]
                ids_a = da['fed_ids']
        elif self.fuse_decoders and self.multitask:
            dp, da, dq = self._decoders_fwd(specs)
        elif self.multitask:
            # all three decoders in ONE launch of the wide-tile persistent kernel (3 + 3 + 2 row domains, as their
            # backward recurrences)
            side_loss = self.use_side_stream and defer_loss and feed.get('loss_dens') is not None
            # a training step (a backward pass follows at once): the logits themselves are left to backward's first
            # launch -- d2p_xent_bwd_dhout_multi computes hout . proj in front of the loss backward -- and the loss
            # value follows it on the side stream: three skinny GEMM launches (17 us each, K = 512 walked by 50
            # workgroups) leave the critical path between the forward and the backward recurrences
            defer_logits = (side_loss and max(V, A, P) <= 64 and U % 128 == 0 and U <= 512 and self.is_train)
            ctx['logits_deferred'] = defer_logits
            # (round 5) in a training step (defer_loss: Trainer.train_step) nothing reads a decoder's output past a row's own length -- the loss and its
            # gradient mask those steps, the weight-gradient products run over the rows inside their sequences -- so the
            # action / perception recurrences skip them like the encoders' (length-sorted row domains that run only
            # their longest row's steps; hout is zero there, as TF's impute_finished would leave it).  forward() outside
            # a training step, evaluation and the baselines keep the reference's free-running outputs past a row's
            # length (BasicDecoder without impute_finished, models/model_full.py:465-471).
            # (whatever the STREAM schedule -- the one-stream instrumented pass takes it too.  EAGER launches only: a
            #  graph-static feed carries no demo_slab_steps, fwd_order is None and a captured step (D2P_GRAPH=1) runs every
            #  row to the last decoded step.  So ctx['da' | 'dq']['hout'] past a row's length is schedule-dependent -- zeros
            #  here, the free-running output there -- and nothing may consume it: loss, gradients and the weight-gradient
            #  products all mask it)
            skip = ({'act': (lens_d, fwd_order), 'per': (lens_d, fwd_order)}
                    if (defer_loss and self.is_train and fwd_order is not None and self.decoder_skip_past_len) else None)
            dp, da, dq = self._decoders_fwd(specs, logits=False, skip_past=skip)
            for e_ in (dp, da, dq):
                if self._abl('logits'):
                    e_['logits'] = self._bufs[e_['scope'] + '/logits']
                elif defer_logits:
                    e_['logits'] = self._buf(e_['scope'] + '/logits', (e_['T'], e_['M'], e_['token_dim']), zero=True)
                else:
                    self._decoder_logits(e_)
            if side_loss and not defer_logits:
                # the loss VALUE: nothing in backward reads it (its denominators come with the feed)
                nums, dens = self._buf('loss_nums', (1 + 2 * k,)), self._buf('loss_dens', (1 + 2 * k,))
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    K.xent_fwd('softmax', dp['logits'], feed['program'], 'bvl', lens_p, L, B, V, 1, n_p,
                               nums[0:1], dens[0:1])
                    K.xent_fwd('softmax', da['logits'], feed['a_h'], 'rtv', lens_d, T, M, A, k, n_d,
                               nums[1:1 + k], dens[1:1 + k])
        else:
            dp = self._decoders_fwd([specs[0]])[0]

        self.mark('fwd:decoders')
        # ---- losses: program + mean_k action + mean_k perception, each mask-count normalised
        #      (the baselines: the program term alone)
        nums = self._buf('loss_nums', (1 + 2 * k,))
        dens = self._buf('loss_dens', (1 + 2 * k,))
        loss = self._buf('loss', (1,))
        terms = self._buf('loss_terms', (3,), zero=True)
        if not side_loss:
            K.xent_fwd('softmax', dp['logits'], feed['program'], 'bvl', lens_p, L, B, V, 1, n_p,
                       nums[0:1], dens[0:1])
        if self.multitask:
            if not side_loss:
                K.xent_fwd('softmax', da['logits'], feed['a_h'], 'rtv', lens_d, T, M, A, k, n_d,
                           nums[1:1 + k], dens[1:1 + k])
            if ctx.get('logits_deferred'):
                ctx['loss_bufs'] = (nums, dens, loss, terms)  # (filled behind backward's first launch)
                dens = feed['loss_dens']
            elif side_loss and defer_loss and feed.get('loss_dens') is not None:
                side.wait_stream(main)                      # the perception decoder's logits
                with torch.cuda.stream(side):
                    K.xent_fwd('sigmoid', dq['logits'], feed['per'], 'rtv', lens_d, T, M, P, k, n_d,
                               nums[1 + k:], dens[1 + k:])
                    K.loss_assemble([1, k, k], nums, dens, loss, terms)
                dens = feed['loss_dens']                    # what backward divides by (the same counts)
            else:
                K.xent_fwd('sigmoid', dq['logits'], feed['per'], 'rtv', lens_d, T, M, P, k, n_d,
                           nums[1 + k:], dens[1 + k:])
                if side_loss:
                    main.wait_stream(side)
                K.loss_assemble([1, k, k], nums, dens, loss, terms)
            ctx.update(da=da, dq=dq, ids_a=ids_a, emb_a=emb_a, per_tm=per_tm, pe_a=pe_a, pe=pe,
                       pe_mean=pe_mean, pe_rstd=pe_rstd)
        else:
            K.loss_assemble([1], nums, dens, loss, terms)

        # K lists of the weight-gradient GEMMs (rows inside their sequences only); absent from the graph-static feed
        ctx['klists'] = {}
        if self.k_rows and feed.get('n_active_pad') is not None:
            if feed['n_active_pad'] and feed['n_t1_pad']:
                ctx['klists']['demo'] = (feed['active_rows'], feed['n_active_pad'], feed['rows_t1'],
                                         feed['rows_t1_prev'], feed['n_t1_pad'])
            if feed['prog_n_active_pad'] and feed['prog_n_t1_pad']:
                ctx['klists']['prog'] = (feed['prog_rows'], feed['prog_n_active_pad'], feed['prog_rows_t1'],
                                         feed['prog_rows_t1_prev'], feed['prog_n_t1_pad'])
        for e_, space in ((e1, 'demo'), (e2, 'demo'), (dp, 'prog'), (da, 'demo'), (dq, 'demo')):
            if e_ is not None:
                e_['rowspace'] = space
                e_['row_order'] = order if space == 'demo' else None
        ctx.update(e1=e1, e2=e2, rn_h=rn_h, rn_c=rn_c, dp=dp, feats_tm=feats_tm, ids_p=ids_p, emb_p=emb_p,
                   dens=dens, h0_2=h0_2, c0_2=c0_2, demo_h=demo_h, demo_c=demo_c, init_h=init_h, init_c=init_c)
        self._ctx = ctx
        self._feed = feed
        self._loss, self._terms = loss, terms
        self._abl_cache['ready'] = True
        return loss

    def _side_stream(self):
        if not self.use_side_stream:
            return torch.cuda.current_stream()
        if getattr(self, '_side', None) is None:
            self._side = pick_concurrent_stream()
        return self._side

    _KAREL_CONV = [(8, 8, 16, 16, 4, 4), (4, 4, 16, 32, 2, 2), (2, 2, 32, 48, 1, 1)]

    def _fused_encoder(self, B, k, T):
        """the State_Encoder's forward pass as ONE launch (d2p_karel_encoder_fwd): training mode, Karel's 8x8x16
        frames, a batch whose workgroups are co-resident (D2P_FUSED_ENCODER=0: the 13 separate launches)"""
        if not self.is_train or not self.fused_encoder:
            return False
        # its bounded spin reports through the persistent kernels' status word; a step that is re-run after a time-out
        # (Trainer._recover, run_test's redo: both switch the recurrences to the per-step kernels first) must not meet
        # the same barrier again
        if not K.lstm_is_persistent():
            return False
        if [tuple(c) for c in self._conv] != self._KAREL_CONV:
            return False
        key = (B, k, T)
        if key not in self._fused_enc_ok:
            self._fused_enc_ok[key] = K.karel_encoder_ok(B, k, T)
        return self._fused_enc_ok[key]

    def _encoder_fwd_fused(self, x, ctx, main, side):
        c, p = self.config, self.params.p
        B, k, T = c.batch_size, c.k, c.max_demo_len
        M, NF = B * k, B * k * T
        names = ['conv%d' % l for l in (1, 2, 3)]
        a = [self._buf(n + '/a', (NF, ho, wo, cout)) for n, (_, _, _, cout, ho, wo) in zip(names, self._conv)]
        y = [self._buf(n + '/bn_y', (NF * ho * wo, cout)) for n, (_, _, _, cout, ho, wo) in zip(names[:2], self._conv)]
        mean = [self._buf(n + '/bn_mean', (k, cv[3])) for n, cv in zip(names, self._conv)]
        rstd = [self._buf(n + '/bn_rstd', (k, cv[3])) for n, cv in zip(names, self._conv)]
        var = [self._buf(n + '/bn_var', (k, cv[3])) for n, cv in zip(names, self._conv)]
        feats_tm = self._buf('feats_tm', (T, M, self.feature_dim))
        ws = self._buf('enc/ws', (K._load_lib().d2p_karel_encoder_ws_bytes(B, k, T),), torch.uint8)
        K.karel_encoder_fwd(x, B, k, T, [p[n + '/W'] for n in names], [p[n + '/b'] for n in names],
                            [p[n + '/gamma'] for n in names], [p[n + '/beta'] for n in names], a, y, feats_tm,
                            mean, rstd, var, ws)
        if self.track_moving:
            # the k moving-average updates of each layer (one per reference BN call, in group order): nothing in the
            # step reads them -- beside the recurrences
            st = side if self.use_side_stream and not torch.cuda.is_current_stream_capturing() else main
            if st is not main:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                for n, m_, v_ in zip(names, mean, var):
                    K.bn_update_moving(m_, v_, *self.moving[n])
        xin = [x, y[0].view(NF, 4, 4, 16), y[1].view(NF, 2, 2, 32)]
        ctx['conv'] = [(xin[l], a[l], mean[l], rstd[l], None) for l in range(3)]
        ctx['enc_fused'] = True            # backward may take the one-launch form too (_encoder_bwd_fused)
        return feats_tm

    def _encoder_bwd_fused(self, ctx, d_feats_tm):
        """the State_Encoder's backward pass as one launch + one combine launch (d2p_karel_encoder_bwd), from the
        time-major feature gradient; False when this batch geometry does not fit (the separate launches run)"""
        c, p, g = self.config, self.params.p, self.params.g
        B, k, T = c.batch_size, c.k, c.max_demo_len
        key = (B, k, T)
        if key not in self._fused_enc_bwd_ok:
            self._fused_enc_bwd_ok[key] = K.karel_encoder_bwd_ok(B, k, T)
        if not self._fused_enc_bwd_ok[key] or not K.lstm_is_persistent():
            return False
        names = ['conv%d' % l for l in (1, 2, 3)]
        ws = self._buf('enc/ws_bwd', (K._load_lib().d2p_karel_encoder_bwd_ws_bytes(B, k, T),), torch.uint8)
        conv = ctx['conv']
        K.karel_encoder_bwd(conv[0][0], d_feats_tm, B, k, T, [p[n + '/W'] for n in names], [p[n + '/gamma'] for n in names],
                            [p[n + '/beta'] for n in names], [cv[1] for cv in conv], [cv[2] for cv in conv],
                            [cv[3] for cv in conv], [g[n + '/W'] for n in names], [g[n + '/b'] for n in names],
                            [g[n + '/gamma'] for n in names], [g[n + '/beta'] for n in names], ws)
        return True

    def _bn_fwd(self, name, x2d, gamma, beta, G, inner, y=None):
        R, C = x2d.shape
        if y is None:
            y = self._buf(name + '/bn_y', (R, C))
        if not self.is_train:
            mm, mv = self.moving[name]
            K.bn_inference(x2d, gamma, beta, mm, mv, y=y)
            return y, None, None
        mean = self._buf(name + '/bn_mean', (G, C))
        rstd = self._buf(name + '/bn_rstd', (G, C))
        # the statistics kernel also applies the G moving-average updates
        K.bn_fwd(x2d, gamma, beta, G, inner, y=y, mean=mean, rstd=rstd,
                 moving=self.moving[name] if self.track_moving else None)
        return y, mean, rstd

    def _hbuf(self, name, T, M):
        """Outputs of a recurrence with one more slab in front: hbuf [T+1, M, U], hout = hbuf[1:].  hbuf[t] is the
        state the step-t product multiplied (hbuf[0] = the initial state, zeros without one), so the recurrent
        weight gradient is ONE product over the rows of all steps, dWh = sum_t hbuf[t]^T dz[t] -- round 2 ran the
        initial state's term as a separate K = M GEMM (26 us alone, 160-260 us beside a persistent launch)."""
        hb = self._buf(name + '/hbuf', (T + 1, M, self.num_lstm_cell_units), zero=True)
        return hb, hb[1:]

    def _stage_h0(self, ctx, items):
        """hbuf[0] <- h0 for every (name, T, M, h0): device copies on the side stream, forked where the initial
        states exist (read by the weight-gradient GEMMs of backward, which run on that stream too)."""
        items = [it for it in items if it[3] is not None]
        if not items:
            return
        main = torch.cuda.current_stream()
        side = self._side_stream()
        # (buffers are created -- and zero-filled -- on the MAIN stream: a first-use allocation inside the side-stream
        #  block would run its fill there, racing with the recurrence that writes the other slabs)
        dst = [self._hbuf(name, T, M)[0][0] for name, T, M, _ in items]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for d, (name, _, _, h0) in zip(dst, items):
                d.copy_(h0)
                ctx.setdefault('h0_staged', set()).add(name)
            # the weight-gradient GEMMs that read hbuf[0] wait for this, whichever stream they are issued on
            ctx['h0_event'] = torch.cuda.Event()
            ctx['h0_event'].record(side)
            ctx['h0_stream'] = side

    def _lstm_xproj(self, name, x2d, I, M, T, n_steps):
        """Hoisted input projection z = x·Wx + b for all steps (one GEMM)."""
        p = self.params.p
        U = self.num_lstm_cell_units
        kernel, bias = p[name + '/kernel'], p[name + '/bias']
        z = self._buf(name + '/z', (T * M, 4 * U))
        if n_steps > 0:
            K.gemm_raw('nn', n_steps * M, 4 * U, I, x2d, x2d.stride(0), kernel[:I], 4 * U, z, 4 * U,
                       bias=bias)
        return z

    def _token_xproj(self, scope, ids, tok, R, T, n_steps):
        """Input projection of a token-input decoder by table: P[v] = embedding[v] . Wx + b for v <= tok, P[tok+1] =
        b (the <s> id is out of range for the [tok+1, U] table: TF-GPU gathers zeros there), then
        z[row] = P[ids[row]].  Same values as projecting the gathered embeddings, up to the summation order of
        a 7- or 51-row GEMM against a 6400-row one."""
        p = self.params.p
        U = self.num_lstm_cell_units
        name = scope + '/lstm'
        P = self._buf(scope + '/table_proj', (tok + 2, 4 * U))
        K.matmul_nn(p[scope + '/embedding'], p[name + '/kernel'][:U], out=P[:tok + 1], bias=p[name + '/bias'])
        P[tok + 1].copy_(p[name + '/bias'])
        z = self._buf(name + '/z', (T * R, 4 * U))
        if n_steps > 0:
            K.embedding_gather(ids, P, out=z, n=n_steps * R)
        return z

    def _lstm_fwd(self, name, x2d, I, M, T, n_steps, h0, c0, lens, want_final, z=None, final_out=None, row_order=None):
        """x2d: [T*M, I] time-major inputs.  Returns saved tensors for backward."""
        p = self.params.p
        U = self.num_lstm_cell_units
        kernel = p[name + '/kernel']
        Wx, Wh = kernel[:I], kernel[I:]
        if z is None:
            z = self._lstm_xproj(name, x2d, I, M, T, n_steps)
        hbuf, hout = self._hbuf(name, T, M)
        cs = self._buf(name + '/cs', (T, M, U))
        if final_out is not None:
            hf, cf = final_out
        else:
            hf = self._buf(name + '/h_final', (M, U)) if want_final else None
            cf = self._buf(name + '/c_final', (M, U)) if want_final else None
        if n_steps > 0:
            K.lstm_seq_fwd_multi([dict(M=M, U=U, n_steps=n_steps, z=z, Wh=Wh, h0=h0, c0=c0, lens=lens, hout=hout, cs=cs,
                                       h_final=hf, c_final=cf,
                                       row_order=row_order if n_steps == T else None)])
        else:
            K.lstm_seq_fwd(z, 4 * U, M * 4 * U, M, U, n_steps, Wh, h0, c0, lens, hout, cs, hf, cf)
        return dict(name=name, x=x2d, I=I, M=M, T=T, n=n_steps, h0=h0, c0=c0, lens=lens, z=z,
                    hout=hout, hbuf=hbuf, cs=cs, h_final=hf, c_final=cf, Wx=Wx, Wh=Wh)

    def _decoders_fwd(self, specs, logits=True, skip_past=None):
        """BasicDecoder + TrainingHelper + Dense(no bias) (models/model_full.py:440-490) for
        several independent decoders at once.  logits=False: the recurrences only (the caller projects with
        _decoder_logits, e.g. on the other stream).  skip_past: {scope: (lens, row_order)} -- the recurrence of that
        decoder does not run a row past its length (training step only: see forward())."""
        p = self.params.p
        U = self.num_lstm_cell_units
        es, seqs = [], []
        for (scope, x2d, I, R, T, n_steps, h0, c0, token_dim, z) in specs:
            name = scope + '/lstm'
            kernel = p[name + '/kernel']
            hbuf, hout = self._hbuf(name, T, R)
            cs = self._buf(name + '/cs', (T, R, U))
            e = dict(name=name, x=x2d, I=I, M=R, T=T, n=n_steps, h0=h0, c0=c0, lens=None, z=z,
                     hout=hout, hbuf=hbuf, cs=cs, h_final=None, c_final=None, Wx=kernel[:I], Wh=kernel[I:],
                     token_dim=token_dim, scope=scope)
            if x2d is None and scope in ('prog', 'act'):      # token-input decoder on the projected-table path
                e['token_ids'] = self._bufs['ids_p' if scope == 'prog' else 'ids_a']
            es.append(e)
            if n_steps > 0:
                q = dict(M=R, U=U, n_steps=n_steps, z=z, Wh=e['Wh'], h0=h0, c0=c0, hout=hout, cs=cs)
                if skip_past and scope in skip_past:
                    q['lens'], q['row_order'] = skip_past[scope]
                seqs.append(q)
        if seqs:
            K.lstm_seq_fwd_multi(seqs)
        if logits:
            for e in es:
                self._decoder_logits(e)
        return es

    def _decoder_logits(self, e):
        p = self.params.p
        U = self.num_lstm_cell_units
        scope, R, T, n_steps, token_dim = e['scope'], e['M'], e['T'], e['n'], e['token_dim']
        logits = self._buf(scope + '/logits', (T, R, token_dim), zero=True)
        if n_steps > 0:
            K.gemm_raw('nn', n_steps * R, token_dim, U, e['hout'], U, p[scope + '/proj'], token_dim,
                       logits, token_dim)
        if n_steps < T:
            logits[n_steps:].zero_()    # dynamic zero padding (:476-484); memset, no arithmetic
        e['logits'] = logits

    def sample_prob_at(self, step):
        """models/model_full.py:62-67: teacher-forcing probability, polynomial_decay(1.0 -> 0.1)
        over scheduled_sampling_decay_steps, power 1, no cycling."""
        d = float(self.scheduled_sampling_decay_steps)
        s_ = min(float(step), d)
        return (1.0 - 0.1) * (1.0 - s_ / d) + 0.1

    def set_sampling_step(self, step):
        """Host-side update of the sampling probability and the noise counter for this step
        (two tiny device writes outside any captured graph)."""
        self._ss_prob.fill_(1.0 - self.sample_prob_at(step))
        self._ss_rng[1] = int(step)

    def _decoder_fwd_sampled(self, scope, gt_ids_tm, R, T, n_steps, h0, c0, token_dim, salt):
        """BasicDecoder + ScheduledEmbeddingTrainingHelper + Dense (models/model_full.py:414-423,
        463-471).  The next input depends on this step's logits, so nothing can be hoisted: per
        step  z_t = table[fed_t] + h_{t-1} Wh  (table = embedding·Wx + b for every token, one row
        of b for the out-of-range <s>), gates, logits_t = h_t·proj, then one sampling decision
        per row.  Saves exactly what the teacher-forced path saves (pre-activations z, hout, cs)
        plus the ids actually fed, which the backward pass uses for the embedding gradient."""
        p = self.params.p
        U = self.num_lstm_cell_units
        name = scope + '/lstm'
        kernel, bias = p[name + '/kernel'], p[name + '/bias']
        emb = p[scope + '/embedding']                                     # [token_dim + 1, U]
        rows = emb.shape[0]
        table = self._buf(scope + '/ss_table', (rows + 1, 4 * U))
        K.matmul_nn(emb, kernel[:U], bias=bias, out=table[:rows])
        table[rows].copy_(bias)                                           # <s>: zero embedding (F9)
        z = self._buf(name + '/z', (T * R, 4 * U))
        hout = self._buf(name + '/hout', (T, R, U))
        cs = self._buf(name + '/cs', (T, R, U))
        logits = self._buf(scope + '/logits', (T, R, token_dim), zero=True)
        fed = self._buf(scope + '/fed_ids', (T, R), torch.int32)
        flags = self._buf(scope + '/ss_flags', (T, R), torch.int32)
        fed.copy_(gt_ids_tm)
        flags.zero_()
        Wh = kernel[U:]
        h_prev, c_prev = h0, c0
        for t in range(n_steps):
            z_t = z[t * R:(t + 1) * R]
            K.embedding_gather(fed[t], table, out=z_t, n=R)
            K.gemm_raw('nn', R, 4 * U, U, h_prev, U, Wh, 4 * U, z_t, 4 * U, accumulate=True)
            K.lstm_gate_fwd(z_t, c_prev, None, None, t, cs[t], None, hout[t])
            K.gemm_raw('nn', R, token_dim, U, hout[t], U, p[scope + '/proj'], token_dim, logits[t], token_dim)
            if t + 1 < n_steps:
                K.sched_sample(logits[t], gt_ids_tm[t + 1], self._ss_prob, self._ss_rng, t + salt, fed[t + 1],
                               flags[t + 1])
            h_prev, c_prev = hout[t], cs[t]
        if n_steps < T:
            logits[n_steps:].zero_()
        x2d = K.embedding_gather(fed, emb, out=self._buf(scope + '/ss_x', (T * R, U)), n=n_steps * R)
        return dict(name=name, x=x2d, I=U, M=R, T=T, n=n_steps, h0=h0, c0=c0, lens=None, z=z, hout=hout, cs=cs,
                    h_final=None, c_final=None, Wx=kernel[:U], Wh=Wh, token_dim=token_dim, scope=scope,
                    logits=logits, fed_ids=fed, sampled=flags)

    def _decoder_fwd(self, scope, x2d, I, R, T, n_steps, h0, c0, token_dim, z=None):
        """BasicDecoder + TrainingHelper + Dense(no bias): models/model_full.py:440-490."""
        p = self.params.p
        U = self.num_lstm_cell_units
        e = self._lstm_fwd(scope + '/lstm', x2d, I, R, T, n_steps, h0, c0, None, want_final=False, z=z)
        logits = self._buf(scope + '/logits', (T, R, token_dim), zero=True)
        if n_steps > 0:
            K.gemm_raw('nn', n_steps * R, token_dim, U, e['hout'], U, p[scope + '/proj'], token_dim,
                       logits, token_dim)
        if n_steps < T:
            logits[n_steps:].zero_()        # dynamic zero padding (:476-484); memset, no arithmetic
        e['logits'] = logits
        e['token_dim'] = token_dim
        e['scope'] = scope
        return e

    RN_SCOPES = ('rn_h', 'rn_c')

    def _rn_stride(self):
        """floats between a tensor of rn_h and the same tensor of rn_c in the flat parameter / gradient
        buffers (the two scopes have identical layouts, so one stride serves every tensor)."""
        p, g = self.params.p, self.params.g
        st = (p['rn_c/fc1/W'].data_ptr() - p['rn_h/fc1/W'].data_ptr()) // 4
        for leaf in ('fc1/W', 'fc1/b', 'fc2/W', 'fc2/b'):
            assert (p['rn_c/' + leaf].data_ptr() - p['rn_h/' + leaf].data_ptr()) // 4 == st
            assert (g['rn_c/' + leaf].data_ptr() - g['rn_h/' + leaf].data_ptr()) // 4 == st
        return st

    def _rn_fwd(self, feat, B, k, U, add_mean=True):
        """SummarizeFeature('rn') for BOTH summaries at once: feat [2, B*k, U] (h states, then c
        states) -> out [2, B, U] = mean over k + rn_pool (models/model_full.py:333-362); without the
        mean term it is the summarizer baseline's (model_summarizer.py:345-352).  rn_h and rn_c have
        separate weights but identical shapes: their GEMMs run as strided batches (one launch for
        the four fc1 half-projections, one for the two fc2 layers) and the row-wise kernels see 2B
        programs; batch norm stays one call per summary (own statistics, own parameters)."""
        p = self.params.p
        M, R = B * k, B * k * k
        ps = self._rn_stride()
        W1, W2 = p['rn_h/fc1/W'], p['rn_h/fc2/W']
        # PQ[half][scope]: P = feat . W1[:U], Q = feat . W1[U:]
        PQ = self._buf('rn/PQ', (2, 2, M, U))
        K.gemm_batched('nn', 2, 2, M, U, U, feat, U, (M * U, 0), W1, U, (ps, U * U), PQ, U, (M * U, 2 * M * U))
        y1a = self._buf('rn/y1a', (2, R, U))
        if self._rn_fused(B, k, U):
            # (round 5) the pointwise chains around the two GEMMs in two launches (d2p_rn_fc1_fwd / d2p_rn_fc2_fwd) instead of nine
            y1, y2a = self._buf('rn/y1', (2, R, U)), self._buf('rn/y2a', (2, R, U))
            ws = self._buf('rn/ws', (K._load_lib().d2p_rn_ws_bytes(B, k, U),), torch.uint8)
            st = {}
            for leaf in ('fc1', 'fc2'):
                st[leaf] = [self._buf('rn/%s/bn_%s' % (leaf, q), (2, 1, U)) for q in ('mean', 'rstd', 'var')]
                assert self.moving['rn_c/' + leaf][0].data_ptr() - self.moving['rn_h/' + leaf][0].data_ptr() == 4 * U
                for q in ('gamma', 'beta'):
                    assert (p['rn_c/%s/%s' % (leaf, q)].data_ptr() - p['rn_h/%s/%s' % (leaf, q)].data_ptr()) // 4 == ps
            mov = (lambda leaf: self.moving['rn_h/' + leaf] if self.track_moving else None)
            K.rn_fc1_fwd(PQ[0], PQ[1], p['rn_h/fc1/b'], p['rn_h/fc1/gamma'], p['rn_h/fc1/beta'], ps, B, k, U, y1a, y1,
                         st['fc1'][0], st['fc1'][1], st['fc1'][2], mov('fc1'), 0.9, ws)
            K.gemm_batched('nn', 2, 1, R, U, U, y1, U, (R * U, 0), W2, U, (ps, 0), y2a, U, (R * U, 0),
                           bias=p['rn_h/fc2/b'], sbias=(ps, 0), act=1)
            out, psum = self._buf('rn/out', (2, B, U)), self._buf('rn/psum', (2, B, U))
            K.rn_fc2_fwd(y2a, p['rn_h/fc2/gamma'], p['rn_h/fc2/beta'], ps, feat if add_mean else None, B, k, U, out, psum,
                         st['fc2'][0], st['fc2'][1], st['fc2'][2], mov('fc2'), 0.9, ws)
            return dict(feat=feat, y1a=y1a, y1=y1, y2a=y2a, out=out, add_mean=add_mean, ps=ps, psum=psum, ws=ws, fused=True,
                        st1=[(y1[i], st['fc1'][0][i], st['fc1'][1][i]) for i in range(2)],
                        st2=[(None, st['fc2'][0][i], st['fc2'][1][i]) for i in range(2)])
        K.rn_pair_fwd(PQ[0], PQ[1], p['rn_h/fc1/b'], y1a, 2 * B, k, U, scopes=2, bias_stride=ps)
        y1, y2a, y2 = self._buf('rn/y1', (2, R, U)), self._buf('rn/y2a', (2, R, U)), self._buf('rn/y2', (2, R, U))
        st1 = self._rn_bn_fwd('fc1', y1a, y1, ps)
        K.gemm_batched('nn', 2, 1, R, U, U, y1, U, (R * U, 0), W2, U, (ps, 0), y2a, U, (R * U, 0),
                       bias=p['rn_h/fc2/b'], sbias=(ps, 0), act=1)
        st2 = self._rn_bn_fwd('fc2', y2a, y2, ps)
        base = None
        if add_mean:
            base = self._buf('rn/base', (2, B, U))
            K.group_mean(feat, 2 * B, k, U, base, None)
        out = self._buf('rn/out', (2, B, U))
        K.pair_mean_fwd(y2, base, out, 2 * B, k * k, U)
        return dict(feat=feat, y1a=y1a, y1=y1, y2a=y2a, st1=st1, st2=st2, out=out, add_mean=add_mean, ps=ps)

    def _rn_fused(self, B, k, U):
        """training mode, a geometry d2p_rn_* takes, and the switch the tests flip on a Model object"""
        if not (self.is_train and self.fused_rn):
            return False
        # its exchange's bounded spin reports through the persistent kernels' status word: a step that is re-run after a
        # time-out (the recurrences on the per-step kernels by then) must not meet the same barrier again -- as _fused_encoder
        if not K.lstm_is_persistent():
            return False
        key = (B, k, U)
        if key not in self._fused_rn_ok:
            self._fused_rn_ok[key] = K.rn_ok(B, k, U)
        return self._fused_rn_ok[key]

    def _rn_bn_fwd(self, leaf, x, y, ps):
        """Batch norm of layer `leaf` of BOTH relation networks: x, y [2, R, U].  Training: one two-problem
        launch set (own statistics, own parameters at stride ps, own moving statistics); returns per scope
        (y, mean, rstd) like _bn_fwd."""
        p = self.params.p
        if not self.is_train:
            return [self._bn_fwd(sc + '/' + leaf, x[i], p[sc + '/' + leaf + '/gamma'], p[sc + '/' + leaf + '/beta'],
                                 1, 1, y=y[i]) for i, sc in enumerate(self.RN_SCOPES)]
        U = x.shape[2]
        for q in ('gamma', 'beta'):
            assert (p['rn_c/%s/%s' % (leaf, q)].data_ptr() - p['rn_h/%s/%s' % (leaf, q)].data_ptr()) // 4 == ps
        mean = self._buf('rn/%s/bn_mean' % leaf, (2, 1, U))
        rstd = self._buf('rn/%s/bn_rstd' % leaf, (2, 1, U))
        mm, mv = self.moving['rn_h/' + leaf]
        assert self.moving['rn_c/' + leaf][0].data_ptr() - mm.data_ptr() == 4 * U
        K.bn_fwd_batched(x, p['rn_h/%s/gamma' % leaf], p['rn_h/%s/beta' % leaf], ps, 1, 1, y, mean, rstd,
                         moving=(mm, mv) if self.track_moving else None, mstride=U)
        return [(y[i], mean[i], rstd[i]) for i in range(2)]

    # ------------------------------------------------------------------ backward
    def decoder_grad_offset(self):
        """First element of the flat parameter / gradient buffer that belongs to the decoders
        (params.py orders encoder, summarizer, then prog/ act/ per/): everything from here on is final
        when `backward` reaches its split point."""
        return self.params.offsets['prog/embedding']

    @staticmethod
    def _call_split(split_cb, main, side):
        """The split point of backward: every decoder gradient has been ENQUEUED -- the last ones on the side
        stream.  Eager two-stream schedule: the callback runs with the side stream current, so a collective
        started there waits for the side stream (which has waited for everything the main stream had enqueued
        when the last decoder's gradients were forked) and the main stream goes on with the encoder backward
        without waiting for anything.  One stream, or under graph capture (the callback ends the capture):
        join first."""
        if side == main or torch.cuda.is_current_stream_capturing():
            main.wait_stream(side)
            split_cb()
        else:
            with torch.cuda.stream(side):
                split_cb()

    def backward(self, loss_scale=1.0, split_cb=None):
        """Hand-written reverse schedule; writes every entry of params.grad exactly once.
        split_cb (data parallelism): called once, right after the last decoder gradient has been
        launched (57 % of the gradient bytes, two thirds of the way through) and before the summarizer /
        encoder backward -- the trainer starts the all-reduce of that slice there (or, under graph
        capture, closes the first graph and opens the second)."""
        if not self.is_train:
            raise RuntimeError('Model(is_train=False) is the evaluation graph: no backward pass')
        ctx, c, p, g = self._ctx, self.config, self.params.p, self.params.g
        feed = ctx['feed']
        B, k, T, L = c.batch_size, c.k, c.max_demo_len, c.max_program_len
        U, V, A, P = c.num_lstm_cell_units, c.dim_program_token, c.action_space, c.per_dim
        M, NF, F = B * k, B * k * T, self.feature_dim
        lens_d, lens_p = feed['demo_len'], feed['program_len']
        n_p, n_d = feed['n_prog'], feed['n_demo']
        dens = ctx['dens']

        # ---- losses -> dlogits (time-major, first n_steps*R rows)
        dl_p = self._buf('prog/dlogits', (L * B, V))
        # one launch for the loss backward of all decoders AND their dhout = dlogits . proj^T (vocabularies
        # of more than 64 tokens: one launch per loss + one K = V GEMM per decoder)
        fused_xb = max(V, A, P) <= 64
        ctx['fused_xb'] = fused_xb
        xb = [dict(mode='softmax', logits=ctx['dp']['logits'], labels=feed['program'], lab_kind='bvl', lens=lens_p, T=L,
                   R=B, V=V, G=1, n_steps=n_p, den=dens[0:1], scale=loss_scale, dlogits=dl_p, proj=p['prog/proj'],
                   dhout=self._buf('prog/dhout', (L * B, U)), U=U)]
        if not fused_xb:
            K.xent_bwd('softmax', ctx['dp']['logits'], feed['program'], 'bvl', lens_p, L, B, V, 1, n_p,
                       dens[0:1], loss_scale, dl_p)
        d_init = self._buf('d_rn', (2, B, U))
        d_init_h, d_init_c = d_init[0], d_init[1]
        main = torch.cuda.current_stream()
        side = self._side_stream()
        if self.multitask:
            dl_a = self._buf('act/dlogits', (T * M, A))
            dl_q = self._buf('per/dlogits', (T * M, P))
            if fused_xb:
                xb += [dict(mode='softmax', logits=ctx['da']['logits'], labels=feed['a_h'], lab_kind='rtv', lens=lens_d,
                            T=T, R=M, V=A, G=k, n_steps=n_d, den=dens[1:1 + k], scale=loss_scale, dlogits=dl_a,
                            proj=p['act/proj'], dhout=self._buf('act/dhout', (T * M, U)), U=U),
                       dict(mode='sigmoid', logits=ctx['dq']['logits'], labels=feed['per'], lab_kind='rtv', lens=lens_d,
                            T=T, R=M, V=P, G=k, n_steps=n_d, den=dens[1 + k:], scale=loss_scale, dlogits=dl_q,
                            proj=p['per/proj'], dhout=self._buf('per/dhout', (T * M, U)), U=U)]
                fused_loss = bool(ctx.get('logits_deferred')) and self.fused_loss and 1 + 2 * k <= 64
                if ctx.get('logits_deferred'):
                    for q_, e_ in zip(xb, (ctx['dp'], ctx['da'], ctx['dq'])):
                        q_['hout'] = e_['hout']                # logits <- hout . proj inside the launch
                        if fused_loss:                         # ... and the rows' loss values by loss group
                            q_['loss_part'] = self._buf(e_['scope'] + '/loss_part',
                                                        (K.xent_blocks(q_['n_steps'], q_['R']) * q_['G'],))
                K.xent_bwd_dhout_multi(xb)
                if ctx.get('logits_deferred'):
                    nums, dens_v, loss_, terms_ = ctx['loss_bufs']
                    side.wait_stream(main)
                    with torch.cuda.stream(side):              # the loss VALUE (joined where backward joins the streams)
                        # (nothing in backward reads the logits: their zero padding leaves the critical path too)
                        for e_ in (ctx['dp'], ctx['da'], ctx['dq']):
                            if e_['n'] < e_['T']:
                                e_['logits'][e_['n']:].zero_()      # dynamic zero padding (:476-484)
                        if fused_loss:
                            # (round 4: the loss-backward launch left the rows' loss values summed per workgroup: one
                            #  small launch instead of three partial-sum, three final and one assembly launch -- 50 us
                            #  of side-stream time; the counts are the feed's)
                            K.loss_from_partials([1, k, k], [K.xent_blocks(q_['n_steps'], q_['R']) for q_ in xb],
                                                 [q_['loss_part'] for q_ in xb], dens, nums, loss_, terms_)
                        else:
                            K.xent_fwd('softmax', ctx['dp']['logits'], feed['program'], 'bvl', lens_p, L, B, V, 1,
                                       n_p, nums[0:1], dens_v[0:1])
                            K.xent_fwd('softmax', ctx['da']['logits'], feed['a_h'], 'rtv', lens_d, T, M, A, k, n_d,
                                       nums[1:1 + k], dens_v[1:1 + k])
                            K.xent_fwd('sigmoid', ctx['dq']['logits'], feed['per'], 'rtv', lens_d, T, M, P, k, n_d,
                                       nums[1 + k:], dens_v[1 + k:])
                            K.loss_assemble([1, k, k], nums, dens_v, loss_, terms_)
            else:
                K.xent_bwd('softmax', ctx['da']['logits'], feed['a_h'], 'rtv', lens_d, T, M, A, k, n_d,
                           dens[1:1 + k], loss_scale, dl_a)
                K.xent_bwd('sigmoid', ctx['dq']['logits'], feed['per'], 'rtv', lens_d, T, M, P, k, n_d,
                           dens[1 + k:], loss_scale, dl_q)

            d_demo = self._buf('d_demo', (2, M, U))
            d_demo_h, d_demo_c = d_demo[0], d_demo[1]
            tmp_hc = self._buf('tmp_dhc', (2, M, U))
            tmp_h, tmp_c = tmp_hc[0], tmp_hc[1]

            # ---- decoder recurrences (main stream): projection grads, dz for every step, and the
            #      initial-state gradients that feed the summarizer / encoder backward
            bspecs = [(ctx['dp'], dl_p, d_init_h, d_init_c), (ctx['da'], dl_a, d_demo_h, d_demo_c),
                      (ctx['dq'], dl_q, tmp_h, tmp_c)]

            # ---- side stream: each decoder's weight / input gradients (three large GEMMs) are not needed
            #      by the rest of backward.  The action and perception decoders' are forked right after
            #      their own recurrence, so that they run beside the program decoder's recurrence -- 32
            #      rows: its persistent kernel occupies half of the CUs -- and the summarizer / encoder
            #      recurrences after it
            def prog_grads(dz):
                self._token_decoder_grads(ctx['dp'], dz, ctx['ids_p'], n_p * B)

            def act_grads(dz):
                self._token_decoder_grads(ctx['da'], dz, ctx['ids_a'], n_d * M)

            def per_grads(dz):
                if ctx['dq']['x'] is None:
                    return self._per_factored_grads(ctx['dq'], dz, feed, ctx['pe_mean'], ctx['pe_rstd'], n_d * M)
                dx_q = self._lstm_bwd_params(ctx['dq'], dz, True)
                if n_d < T:
                    dx_q[n_d * M:].zero_()
                d_pe_a = K.bn_bwd(ctx['pe_a'], dx_q, p['per/fc/gamma'], ctx['pe_mean'], ctx['pe_rstd'], k, 1,
                                  False, g['per/fc/gamma'], g['per/fc/beta'], dx=self._buf('d_pe_a', (T * M, U)),
                                  dbias=g['per/fc/b'])
                K.matmul_tn(ctx['per_tm'].view(T * M, P), d_pe_a, out=g['per/fc/W'])
            grads = (prog_grads, act_grads, per_grads)
            # all three backward recurrences as ONE launch (3 + 3 + 2 row domains; with fuse_decoders on the per-step
            # kernels: one launch per step for all three), the three decoders' gradient products forked behind it
            self.mark('bwd:loss')
            dzs = self._decoders_bwd_rec(bspecs if self.fuse_decoders else [bspecs[i] for i in (2, 1, 0)])
            self.mark('bwd:decoders')
            side.wait_stream(main)
            with torch.cuda.stream(side):
                order_ = (0, 1, 2) if self.fuse_decoders else (2, 1, 0)
                if not self._decoder_grads_grouped(dict(zip(order_, dzs)), feed, n_p * B, n_d * M):
                    for i, dz in zip(order_, dzs):
                        grads[i](dz)
            K.axpy(1.0, tmp_hc, d_demo)
            if split_cb is not None:
                self._call_split(split_cb, main, side)
        else:
            # baselines: the program decoder is the only one
            if fused_xb:
                K.xent_bwd_dhout_multi(xb)
            dz_p = self._decoders_bwd_rec([(ctx['dp'], dl_p, d_init_h, d_init_c)])[0]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._token_decoder_grads(ctx['dp'], dz_p, ctx['ids_p'], n_p * B)
            if split_cb is not None:
                self._call_split(split_cb, main, side)

        d_hc1f = self._buf('d_hc1f', (2, M, U))
        d_h1f, d_c1f = d_hc1f[0], d_hc1f[1]
        if self.variant == 'synthesis_baseline':
            # pooled final states of the only encoder pass; no gradient through its outputs
            if self.demo_aggregation == 'avgpool':
                K.group_mean_bwd(d_init_h, None, d_h1f, B, k, U, False)
                K.group_mean_bwd(d_init_c, None, d_c1f, B, k, U, False)
            else:
                K.group_max_bwd(d_init_h, ctx['arg_h'], d_h1f, B, k, U, False)
                K.group_max_bwd(d_init_c, ctx['arg_c'], d_c1f, B, k, U, False)
            d_hout1 = None
        else:
            if not self.multitask:
                d_demo = self._buf('d_demo', (2, M, U))
                d_demo_h, d_demo_c = d_demo[0], d_demo[1]
                d_demo.zero_()
            # ---- SummarizeFeature('rn') backward, both summaries (adds into d_demo)
            if not self._abl('rn_bwd'):
                self._rn_bwd(ctx['rn_h'], d_init, d_demo, B, k, U)

            # ---- SecondPathEncoder backward: only the final states carry gradient
            e2 = ctx['e2']
            dhc0_2 = self._buf('dhc0_2', (2, M, U))
            dh0_2, dc0_2 = dhc0_2[0], dhc0_2[1]
            # (its weight gradients go to the side stream: two large GEMMs beside the next recurrence)
            self.mark('bwd:rn')
            dz2 = self._lstm_bwd_rec(e2, None, d_demo_h, d_demo_c, dh0_2, dc0_2)
            self.mark('bwd:enc2')
            if ctx.get('rows') is not None:
                rows_idx, n_act = ctx['rows']
                # (no zero fill: the only reader is the first encoder's backward recurrence, which SELECTS dhout by the row's
                #  length -- rows past their sequence keep whatever an earlier batch left there;
                #  test_rows_past_a_sequence_are_selected_around_not_multiplied poisons them)
                d_hout1 = self._buf(e2['name'] + '/dx', (T * M, U))
                K.gemm_rows('nt', n_act, U, 4 * U, dz2, 4 * U, e2['Wx'], 4 * U, d_hout1, U, rows_idx)
            else:
                d_hout1 = self._lstm_bwd_dx(e2, dz2)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._lstm_bwd_weights(e2, dz2)
            # summary = mean_k(step-1 final states), broadcast to every demo of the program
            K.group_mean_bwd(None, dhc0_2, d_hc1f, 2 * B, k, U, False)
        # ---- Demo_Encoder LSTM backward
        self.mark('bwd:dx2+mean')
        dz1 = self._lstm_bwd_rec(ctx['e1'], d_hout1, d_h1f, d_c1f, None, None)
        self.mark('bwd:enc1')
        if ctx.get('rows_e1') is not None:
            e1_ = ctx['e1']
            d_feats_tm = self._buf(e1_['name'] + '/dx', (T * M, e1_['I']))
            d_feats_tm.zero_()                        # rows past their sequence: no gradient
            K.gemm_rows('nt', ctx['rows_e1'][1], e1_['I'], 4 * U, dz1, 4 * U, e1_['Wx'], 4 * U, d_feats_tm, e1_['I'],
                        ctx['rows_e1'][0])
        else:
            d_feats_tm = self._lstm_bwd_dx(ctx['e1'], dz1)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._lstm_bwd_weights(ctx['e1'], dz1)
        # ---- State_Encoder backward
        if ctx.get('enc_fused') and not self._abl('conv_bwd') and self._encoder_bwd_fused(ctx, d_feats_tm):
            self.mark('bwd:dx1+conv')
            main.wait_stream(side)
            self.mark('bwd:join')
            return self.params.grad
        d_feats = K.transpose_rt(d_feats_tm.view(T, M, F), T, M, F, out=self._buf('d_feats', (M, T, F)))
        dy = d_feats
        dy_sums = None                 # (stats, S): batch-norm-backward partial sums the producer of dy left behind
        for l in range(len(self._conv), 0, -1):
            if self._abl('conv_bwd'):
                break
            (h, w, cin, cout, ho, wo) = self._conv[l - 1]
            x_in, a, mean, rstd, x_aff = ctx['conv'][l - 1]
            if l == 1 and self.fold_bn and K.conv_bnbwd_ok(x_in.shape, cout):
                # (round 5) the first layer needs no input gradient, so its batch-norm backward never materialises the
                # conv's output gradient: the sums leave coefficients, the weight-gradient kernel forms da from (a, dy)
                # as it loads them and returns the bias gradient too (the apply pass -- a read of a and dy and a 655 MB
                # write at 80x80 frames -- is gone)
                coef = self._buf('conv1/bn_coef', (k, cout, 4))
                K.bn_bwd_coef(a.view(NF * ho * wo, cout), dy.view(NF * ho * wo, cout), p['conv1/gamma'], mean, rstd, k,
                              T * ho * wo, coef, g['conv1/gamma'], g['conv1/beta'], sums=dy_sums)
                dyv = dy.view(NF, ho, wo, cout)
                if x_in.shape[3] != cin:            # channel-padded conv1 input: unpad the gradient
                    cp = x_in.shape[3]
                    gpad = self._buf('conv1/gWpad', (3, 3, cp, cout))
                    K.conv_wgrad_bnbwd(x_in, a, dyv, coef, k, T, gpad, g['conv1/b'])
                    K.pad_axis(gpad, 9, cin, cp, cout, g['conv1/W'], unpad=True)
                else:
                    K.conv_wgrad_bnbwd(x_in, a, dyv, coef, k, T, g['conv1/W'], g['conv1/b'])
                continue
            # (dy_sums: the input-gradient launch of layer l + 1 left this batch norm's backward partial sums behind)
            da_ = K.bn_bwd(a.view(NF * ho * wo, cout), dy.view(NF * ho * wo, cout), p['conv%d/gamma' % l],
                           mean, rstd, k, T * ho * wo, True, g['conv%d/gamma' % l], g['conv%d/beta' % l],
                           dx=self._buf('conv%d/da' % l, (NF * ho * wo, cout)), dbias=g['conv%d/b' % l], sums=dy_sums)
            dy_sums = None
            # the weight gradient feeds nothing inside backward: on the side stream, beside this layer's data
            # gradient and the next layer's batch-norm backward (ViZDoom frames: 40 % of the conv backward)
            # (only where the layer is large enough to matter: at Karel's 8x8 frames the three weight-gradient
            #  launches are 40 us and moving them costs 0.04 ms per step, at 80x80 frames it saves 0.2 ms)
            wg_side = self.use_side_stream and l > 1 and NF * h * w * cin >= (1 << 24)
            if wg_side:
                side.wait_stream(main)
            with torch.cuda.stream(side if wg_side else main):
                if x_aff is not None:               # x_in is the previous layer's pre-norm activation + its batch-norm apply
                    K.conv_wgrad_bn(x_in, da_.view(NF, ho, wo, cout), g['conv%d/W' % l], k, T, x_aff[:2])
                elif x_in.shape[3] != cin:          # channel-padded conv1 input: unpad the gradient
                    cp = x_in.shape[3]
                    gpad = K.conv_wgrad(x_in, da_.view(NF, ho, wo, cout),
                                        self._buf('conv1/gWpad', (3, 3, cp, cout)))
                    K.pad_axis(gpad, 9, cin, cp, cout, g['conv%d/W' % l], unpad=True)
                else:
                    K.conv_wgrad(x_in, da_.view(NF, ho, wo, cout), g['conv%d/W' % l])
            if l > 1:
                dxb = self._buf('conv%d/dx' % l, (NF, h, w, cin))
                Sd = 0
                if self.fold_bn and (l > 2 or K.conv_bnbwd_ok(ctx['conv'][0][0].shape, cin)):
                    Sd = K.conv_dgrad_bn_slices((NF, h, w, cin), cout, k, T)
                if Sd > 0:
                    # the previous layer's batch-norm-backward partial sums come out of this launch (it writes the gradient
                    # they are sums of): no separate pass over (a, dy) of that layer (round 5: the first layer's, out of
                    # the 16 -> 32 layer's row-strip kernel; round 6: every 48-channel layer's launch leaves them)
                    _, a_prev, mean_prev, rstd_prev, _ = ctx['conv'][l - 2]
                    st_ = self._buf('conv%d/bn_bwd_partial' % (l - 1), (k * Sd * cin * 2,), torch.float64)
                    dy = K.conv_dgrad_bn(da_.view(NF, ho, wo, cout), p['conv%d/W' % l], (NF, h, w, cin), a_prev, mean_prev,
                                         rstd_prev, k, T, st_, Sd, dx=dxb)
                    dy_sums = (st_, Sd)
                else:
                    dy = K.conv_dgrad(da_.view(NF, ho, wo, cout), p['conv%d/W' % l], (NF, h, w, cin), dx=dxb)
        self.mark('bwd:dx1+conv')
        main.wait_stream(side)
        self.mark('bwd:join')
        return self.params.grad

    def _lstm_bwd_rec(self, e, dhout, dh_final, dc_final, dh0, dc0):
        """Backward through the recurrence only: dz for every step, dh0, dc0."""
        name, M, T, n = e['name'], e['M'], e['T'], e['n']
        U = self.num_lstm_cell_units
        dz = self._buf(name + '/dz', (T * M, 4 * U))
        gb = self.params.g[name + '/bias']
        if n > 0:
            # (the bias gradient -- the column sums of dz -- comes out of the same launch)
            K.lstm_seq_bwd_multi([dict(M=M, U=U, n_steps=n, z=e['z'], Wh=e['Wh'], c0=e['c0'], lens=e['lens'],
                                       cs=e['cs'], dhout=dhout, dh_final=dh_final, dc_final=dc_final, dz=dz,
                                       dh0=dh0, dc0=dc0, db=gb,
                                       row_order=e.get('row_order'))])
        else:
            K.lstm_seq_bwd(e['z'], 4 * U, M * 4 * U, M, U, n, e['Wh'], e['c0'], e['lens'], e['cs'],
                           dhout, dh_final, dc_final, dz, dh0, dc0)
            gb.zero_()
        e['db_done'] = True
        return dz

    def _lstm_bwd_params(self, e, dz, want_dx):
        """Kernel / bias gradients and dX [n*M, I] from dz (independent of the recurrence order:
        three large GEMMs that may run on a side stream)."""
        self._lstm_bwd_weights(e, dz)
        return self._lstm_bwd_dx(e, dz) if want_dx else None

    def _per_factored_grads(self, e, dz, feed, mean, rstd, rows):
        """Perception decoder: kernel / bias gradients of its LSTM and the fc + batch-norm gradients of its input
        encoder from dz, through the NC columns of feed['per_rows'] (d2p.h: d2p_per_fc_bn_bwd)."""
        c, p, g = self.config, self.params.p, self.params.g
        U, P, k = self.num_lstm_cell_units, c.per_dim, c.k
        name = e['name']
        gk = g[name + '/kernel']
        NCp = self.per_cols
        if rows > 0:
            per_tm = self._ctx.get('per_tm')
            if per_tm is not None and K.per_rows_tn_ok(rows, k, P, 4 * U):
                # (round 4) rows^T dZ from the structure of `rows` (P + 1 non-zeros per row): a read of dz
                S = K.per_rows_tn(k, per_tm.view(-1, P), dz, self._buf('per/S', (NCp, 4 * U)), rows)
            else:
                S = K.matmul_tn(feed['per_rows'][:rows], dz[:rows], out=self._buf('per/S', (NCp, 4 * U)))
            K.matmul_tn(self._bufs['per/H'], S, out=gk[:U])                       # dWx = H^T (rows^T dZ)
            Q = K.matmul_nt(S, p[name + '/kernel'][:U], out=self._buf('per/Q', (NCp, U)))
            K.per_fc_bn_bwd(k, P, c.batch_size * c.max_demo_len, p['per/fc/W'], p['per/fc/b'], p['per/fc/gamma'],
                            mean, rstd, Q, feed['per_gram'], g['per/fc/W'], g['per/fc/b'], g['per/fc/gamma'],
                            g['per/fc/beta'])
        else:
            for n in ('per/fc/W', 'per/fc/b', 'per/fc/gamma', 'per/fc/beta'):
                g[n].zero_()
            gk[:U].zero_()
        self._lstm_bwd_weights(e, dz)              # bias and recurrent-kernel gradients (e['factored_x'] skips dWx)

    def _decoder_grads_grouped(self, dz_of, feed, rows_p, rows_d):
        """(round 6) The three decoders' gradient products behind their backward recurrences with the SMALL ones grouped:
        each decoder's dz rows summed by input token / perception column (S: three launches, one read of dz each), then
        ONE launch for the six products G1 = A^T S (the input half of the kernel gradient) and G2 = S Wx^T (embedding
        gradient / the perception encoder's Q) -- d2p_small_pair_products --, the perception encoder's fc / batch-norm
        gradients, and the three large recurrent-kernel gradients.  False (nothing done) when a decoder is not on the
        projected-table / factored path or the sizes are not taken: the per-decoder launches run."""
        ctx, c, p, g = self._ctx, self.config, self.params.p, self.params.g
        U, P, k = self.num_lstm_cell_units, c.per_dim, c.k
        dp, da, dq = ctx['dp'], ctx['da'], ctx['dq']
        if self.grouped_decoder_grads is False or self._abl('wgrad') or self._abl('scatter'):
            return False
        if dp.get('token_ids') is None or da.get('token_ids') is None or dq['x'] is not None or 'per/H' not in self._bufs:
            return False
        if rows_p <= 0 or rows_d <= 0 or not K.pair_products_ok(max(dp['token_dim'] + 1, da['token_dim'] + 1, self.per_cols), U):
            return False
        per_tm = ctx.get('per_tm')
        if per_tm is None or not K.per_rows_tn_ok(rows_d, k, P, 4 * U):
            return False
        probs = []
        for e, dz, rows in ((dq, dz_of[2], rows_d), (da, dz_of[1], rows_d), (dp, dz_of[0], rows_p)):
            name, scope = e['name'], e['scope']
            gk, Wx = g[name + '/kernel'], p[name + '/kernel'][:U]
            if e is dq:
                S = K.per_rows_tn(k, per_tm.view(-1, P), dz, self._buf('per/S', (self.per_cols, 4 * U)), rows)
                R_, A_, G2 = self.per_cols, self._bufs['per/H'], self._buf('per/Q', (self.per_cols, U))
            else:
                tok = e['token_dim']
                S = self._buf(name + '/dz_by_token', (tok + 2, 4 * U))
                K.embedding_scatter_add(e['token_ids'], dz[:rows], S, n=rows)
                R_, A_, G2 = tok + 1, p[scope + '/embedding'], g[scope + '/embedding']
            probs.append((R_, U, S, A_, Wx, gk[:U], G2))
        K.small_pair_products(probs)
        K.per_fc_bn_bwd(k, P, c.batch_size * c.max_demo_len, p['per/fc/W'], p['per/fc/b'], p['per/fc/gamma'],
                        ctx['pe_mean'], ctx['pe_rstd'], self._bufs['per/Q'], feed['per_gram'], g['per/fc/W'], g['per/fc/b'],
                        g['per/fc/gamma'], g['per/fc/beta'])
        # the recurrent halves: the action and the perception decoder's as ONE launch of two products (same shape, same
        # row lists, their own states and dz: d2p_gemm_f32_tn_rows_x2), the program decoder's on its own
        if not self._lstm_bwd_weights_h_pair(dq, dz_of[2], da, dz_of[1]):
            self._lstm_bwd_weights(dq, dz_of[2], part='h')
            self._lstm_bwd_weights(da, dz_of[1], part='h')
        self._lstm_bwd_weights(dp, dz_of[0], part='h')
        return True

    def _lstm_bwd_weights_h_pair(self, e0, dz0, e1, dz1):
        """part='h' of _lstm_bwd_weights for two recurrences of one geometry at once; False: not taken (nothing written)."""
        if not self.paired_kernel_grads or self._abl('wgrad') or self._abl('wgrad:' + e0['name']) or self._abl('wgrad:' + e1['name']):
            return False
        staged = self._ctx.get('h0_staged', ())
        kl = self._ctx.get('klists', {}).get(e0.get('rowspace'))
        same = all(e0[k] == e1[k] for k in ('M', 'T', 'n', 'I')) and e0.get('rowspace') == e1.get('rowspace')
        if not (same and kl is not None and kl[1] and e0['n'] > 0 and e0['name'] in staged and e1['name'] in staged
                and e0.get('hbuf') is not None and e1.get('hbuf') is not None):
            return False
        g = self.params.g
        U, M, T, n, I = self.num_lstm_cell_units, e0['M'], e0['T'], e0['n'], e0['I']
        rows = n * M
        for e, dz in ((e0, dz0), (e1, dz1)):
            if not e.get('db_done'):
                K.colsum(dz[:rows], out=g[e['name'] + '/bias'], rows=rows)
        ev = self._ctx.get('h0_event')
        if ev is not None and torch.cuda.current_stream() != self._ctx.get('h0_stream'):
            torch.cuda.current_stream().wait_event(ev)
        K.gemm_tn_rows_x2(U, 4 * U, kl[1], e0['hbuf'].view((T + 1) * M, U), U, dz0, 4 * U, g[e0['name'] + '/kernel'][I:],
                          e1['hbuf'].view((T + 1) * M, U), U, dz1, 4 * U, g[e1['name'] + '/kernel'][I:], 4 * U, kl[0], kl[0])
        return True

    def _token_decoder_grads(self, e, dz, ids, rows):
        """Kernel / bias / embedding gradients of a token-input decoder from its dz."""
        if e.get('token_ids') is not None:
            self._lstm_bwd_weights(e, dz)          # embedding gradient included (projected-table path)
        else:
            dx = self._lstm_bwd_params(e, dz, True)
            K.embedding_scatter_add(ids, dx, self.params.g[e['scope'] + '/embedding'], n=rows)

    def _lstm_bwd_dx(self, e, dz):
        """dX [n*M, I] = dZ Wx^T: the only product of dz the rest of backward waits for."""
        name, M, T, n, I = e['name'], e['M'], e['T'], e['n'], e['I']
        U = self.num_lstm_cell_units
        rows = n * M
        dx = self._buf(name + '/dx', (T * M, I))
        if rows > 0:
            K.gemm_raw('nt', rows, I, 4 * U, dz[:rows], 4 * U, e['Wx'], 4 * U, dx, I)
        return dx

    def _lstm_bwd_weights(self, e, dz, part='all'):
        """Kernel / bias gradients from dz: nothing downstream reads them before the optimizer.  part='h': the recurrent
        half alone (the input half was written by _decoder_grads_grouped)."""
        if self._abl('wgrad') or self._abl('wgrad:' + e['name']):
            return
        g = self.params.g
        name, M, T, n, I = e['name'], e['M'], e['T'], e['n'], e['I']
        U = self.num_lstm_cell_units
        gk, gb = g[name + '/kernel'], g[name + '/bias']
        rows = n * M
        dz_n = dz[:rows] if rows > 0 else dz[:0]
        # the rows inside their sequences, as K lists (d2p_gemm_f32_tn_rows): the others are zeros in dz
        kl = self._ctx.get('klists', {}).get(e.get('rowspace')) if rows > 0 else None
        # both halves of the kernel gradient as ONE product [X | H]^T dZ (d2p_gemm_f32_tn_rows2: 256 tiles of 128 x 64 at
        # I = U = 512; the same values as the two products below, bit for bit): the input half reads x through the K list,
        # the recurrent half the staged states (hbuf[0] = h0) through the same list
        pair = (part != 'h' and e.get('token_ids') is None and e['x'] is not None and kl is not None and kl[1] and n > 0
                and e['name'] in self._ctx.get('h0_staged', ()) and e.get('hbuf') is not None and self.paired_kernel_grads)
        if pair:
            ev = self._ctx.get('h0_event')
            if ev is not None and torch.cuda.current_stream() != self._ctx.get('h0_stream'):
                torch.cuda.current_stream().wait_event(ev)
            K.gemm_tn_rows2(I, U, 4 * U, kl[1], e['x'], e['x'].stride(0), e['hbuf'].view((T + 1) * M, U), U, kl[0],
                            dz, 4 * U, kl[0], gk, 4 * U)
            if not e.get('db_done'):
                K.colsum(dz_n, out=gb, rows=rows)
            return
        # dWx = X^T dZ ; db = colsum(dZ)
        if part == 'h':
            pass
        elif e.get('token_ids') is not None:
            # x = embedding[id]: S[v] = sum of the dz rows whose input token was v (one-hot GEMM, tok+2 rows),
            # then dWx = embedding^T S and d embedding = S Wx^T -- three small products instead of two
            # 13.4 GFLOP GEMMs (dWx, dX) and the scatter of dX
            scope, tok = e['scope'], e['token_dim']
            p = self.params.p
            if rows > 0:
                S = self._buf(name + '/dz_by_token', (tok + 2, 4 * U))
                if not self._abl('scatter'):              # (timing experiment: tools/step_ablation.sh)
                    K.embedding_scatter_add(e['token_ids'], dz_n, S, n=rows)
                K.matmul_tn(p[scope + '/embedding'], S[:tok + 1], out=gk[:I])
                K.matmul_nt(S[:tok + 1], p[name + '/kernel'][:I], out=g[scope + '/embedding'])
            else:
                gk[:I].zero_()
                g[scope + '/embedding'].zero_()
        elif e['x'] is not None:
            if kl is not None and kl[1]:
                K.gemm_tn_rows(I, 4 * U, kl[1], e['x'], e['x'].stride(0), kl[0], dz, 4 * U, kl[0], gk[:I], 4 * U)
            else:
                K.gemm_raw('tn', I, 4 * U, rows, e['x'], e['x'].stride(0), dz_n, 4 * U, gk[:I], 4 * U)
        # (x None without token ids: the factored perception decoder -- its dWx is written by _per_factored_grads)
        if not e.get('db_done'):             # (the backward recurrence's launch normally leaves it behind)
            K.colsum(dz_n, out=gb, rows=rows) if rows > 0 else gb.zero_()
        # dWh = sum_t h_{t-1}^T dZ_t : h_{-1} = h0 (skipped when zero), then hout[t-1]
        hout2d = e['hout'].view(T * M, U)
        if e['name'] in self._ctx.get('h0_staged', ()) and e.get('hbuf') is not None and n > 0:
            # hbuf[t] = the state step t multiplied (hbuf[0] = h0): one product over the rows of all steps
            hb2d = e['hbuf'].view((T + 1) * M, U)
            ev = self._ctx.get('h0_event')
            if ev is not None and torch.cuda.current_stream() != self._ctx.get('h0_stream'):
                # (issued on another stream than the staging copies: wait for them; on the staging stream itself stream
                #  order already holds -- and a wait for an event of the same capturing stream crashed hipStreamEndCapture)
                torch.cuda.current_stream().wait_event(ev)
            if kl is not None and kl[1]:
                K.gemm_tn_rows(U, 4 * U, kl[1], hb2d, U, kl[0], dz, 4 * U, kl[0], gk[I:], 4 * U)
            else:
                K.gemm_raw('tn', U, 4 * U, rows, hb2d, U, dz_n, 4 * U, gk[I:], 4 * U)
            return

        def dwh(accumulate):
            if kl is not None and kl[4]:
                K.gemm_tn_rows(U, 4 * U, kl[4], hout2d, U, kl[3], dz, 4 * U, kl[2], gk[I:], 4 * U, accumulate=accumulate)
            else:
                K.gemm_raw('tn', U, 4 * U, (n - 1) * M, hout2d, U, dz[M:], 4 * U, gk[I:], 4 * U, accumulate=accumulate)
        if e['h0'] is not None and n > 0:
            K.gemm_raw('tn', U, 4 * U, M, e['h0'], U, dz, 4 * U, gk[I:], 4 * U)
            if n > 1:
                dwh(True)
        elif n > 1:
            dwh(False)
        else:
            gk[I:].zero_()

    def _lstm_bwd(self, e, dhout, dh_final, dc_final, dh0, dc0, want_dx):
        """Backward of _lstm_fwd.  Writes the kernel / bias gradients; returns dX [n*M, I]."""
        dz = self._lstm_bwd_rec(e, dhout, dh_final, dc_final, dh0, dc0)
        return self._lstm_bwd_params(e, dz, want_dx)

    def _decoders_bwd_rec(self, specs):
        """Projection gradients, then the backward recurrences of several independent decoders
        advancing together (one launch per step for all of them).  Returns their dz buffers."""
        p, g = self.params.p, self.params.g
        U = self.num_lstm_cell_units
        seqs, dzs, wproj = [], [], []
        for (e, dlogits, dh0, dc0) in specs:
            scope, R, T, n, V = e['scope'], e['M'], e['T'], e['n'], e['token_dim']
            rows = n * R
            dhout = self._buf(scope + '/dhout', (T * R, U))
            dz = self._buf(e['name'] + '/dz', (T * R, 4 * U))
            dzs.append(dz)
            if rows > 0:
                wproj.append((U, V, rows, e['hout'].view(T * R, U), dlogits, g[scope + '/proj']))
                if not self._ctx.get('fused_xb'):       # (else: written with dlogits by d2p_xent_bwd_dhout_multi)
                    K.gemm_raw('nt', rows, U, V, dlogits, V, p[scope + '/proj'], V, dhout, U)
                seqs.append(dict(M=R, U=U, n_steps=n, z=e['z'], Wh=e['Wh'], c0=e['c0'], cs=e['cs'],
                                 dhout=dhout, dz=dz, dh0=dh0, dc0=dc0, db=g[e['name'] + '/bias'],
                                 row_order=e.get('row_order')))
            else:
                g[scope + '/proj'].zero_()
                g[e['name'] + '/bias'].zero_()
                dh0.zero_()
                dc0.zero_()
            e['db_done'] = True
        # the projections' weight gradients (K = all rows: split-K launches + their combine passes) feed nothing in
        # backward: on the side stream, beside the recurrences
        main = torch.cuda.current_stream()
        side = self._side_stream()
        on_side = side != main
        if on_side and wproj:
            side.wait_stream(main)
        with torch.cuda.stream(side if on_side else main):
            for (U_, V, rows, hout2d, dlogits, gproj) in wproj:
                K.gemm_raw('tn', U_, V, rows, hout2d, U_, dlogits, V, gproj, V)
        if seqs:
            K.lstm_seq_bwd_multi(seqs)
        return dzs

    def _decoder_bwd_rec(self, e, dlogits, dh0, dc0):
        """Projection gradients + backward recurrence of one decoder; returns dz."""
        p, g = self.params.p, self.params.g
        scope, R, T, n, V = e['scope'], e['M'], e['T'], e['n'], e['token_dim']
        U = self.num_lstm_cell_units
        rows = n * R
        hout2d = e['hout'].view(T * R, U)
        dhout = self._buf(scope + '/dhout', (T * R, U))
        if rows > 0:
            K.gemm_raw('tn', U, V, rows, hout2d, U, dlogits, V, g[scope + '/proj'], V)
            K.gemm_raw('nt', rows, U, V, dlogits, V, p[scope + '/proj'], V, dhout, U)
        else:
            g[scope + '/proj'].zero_()
        return self._lstm_bwd_rec(e, dhout, None, None, dh0, dc0)

    def _rn_bwd(self, r, d_out, d_feat, B, k, U):
        """d_out [2, B, U]: gradient of mean_k(feat) + rn_pool(feat) for the h and the c summary;
        accumulates into d_feat [2, B*k, U]."""
        p, g = self.params.p, self.params.g
        M, R, ps = B * k, B * k * k, r['ps']
        W1, W2 = p['rn_h/fc1/W'], p['rn_h/fc2/W']
        feat = r['feat']
        if r['add_mean']:
            K.group_mean_bwd(d_out, None, d_feat, 2 * B, k, U, True)         # the avg-pool branch
        if r.get('fused'):
            # (round 5) d2p_rn_fc2_bwd / d2p_rn_fc1_bwd around the input-gradient GEMM of fc2 instead of eleven launches
            dy2a, dy1 = self._buf('rn/dy2a', (2, R, U)), self._buf('rn/dy1', (2, R, U))
            dPQ = self._buf('rn/dPQ', (2, 2, M, U))                          # [half][scope], as PQ
            mean2, rstd2 = self._bufs['rn/fc2/bn_mean'], self._bufs['rn/fc2/bn_rstd']
            mean1, rstd1 = self._bufs['rn/fc1/bn_mean'], self._bufs['rn/fc1/bn_rstd']
            K.rn_fc2_bwd(r['y2a'], d_out, r['psum'], p['rn_h/fc2/gamma'], ps, mean2, rstd2, B, k, U, dy2a,
                         g['rn_h/fc2/gamma'], g['rn_h/fc2/beta'], r['ws'])
            K.gemm_batched('nt', 2, 1, R, U, U, dy2a, U, (R * U, 0), W2, U, (ps, 0), dy1, U, (R * U, 0))
            K.rn_fc1_bwd(r['y1a'], dy1, p['rn_h/fc1/gamma'], ps, mean1, rstd1, B, k, U, dPQ[0], dPQ[1],
                         g['rn_h/fc1/gamma'], g['rn_h/fc1/beta'], g['rn_h/fc1/b'], g['rn_h/fc2/b'], r['ws'])
            self._rn_bwd_tail(r, d_feat, dPQ, dy2a, B, k, U)
            return
        dy2 = self._buf('rn/dy2', (2, R, U))
        K.pair_mean_bwd(d_out, dy2, 2 * B, k * k, U)
        dy2a, dy1, dy1a = self._buf('rn/dy2a', (2, R, U)), self._buf('rn/dy1', (2, R, U)), self._buf('rn/dy1a', (2, R, U))
        batched = True
        if batched:
            # st[i] = (y, mean, rstd) with mean / rstd views of one [2, 1, U] buffer each
            K.bn_bwd_batched(r['y2a'], dy2, p['rn_h/fc2/gamma'], ps, r['st2'][0][1], r['st2'][0][2], 1, 1, True,
                             g['rn_h/fc2/gamma'], g['rn_h/fc2/beta'], dy2a, dbias=g['rn_h/fc2/b'])
        for i, sc in enumerate(self.RN_SCOPES):
            if not batched:
                _, m2, r2 = r['st2'][i]
                K.bn_bwd(r['y2a'][i], dy2[i], p[sc + '/fc2/gamma'], m2, r2, 1, 1, True,
                         g[sc + '/fc2/gamma'], g[sc + '/fc2/beta'], dx=dy2a[i], dbias=g[sc + '/fc2/b'])
        K.gemm_batched('nt', 2, 1, R, U, U, dy2a, U, (R * U, 0), W2, U, (ps, 0), dy1, U, (R * U, 0))
        if batched:
            K.bn_bwd_batched(r['y1a'], dy1, p['rn_h/fc1/gamma'], ps, r['st1'][0][1], r['st1'][0][2], 1, 1, True,
                             g['rn_h/fc1/gamma'], g['rn_h/fc1/beta'], dy1a, dbias=g['rn_h/fc1/b'])
        else:
            for i, sc in enumerate(self.RN_SCOPES):
                _, m1, r1 = r['st1'][i]
                K.bn_bwd(r['y1a'][i], dy1[i], p[sc + '/fc1/gamma'], m1, r1, 1, 1, True,
                         g[sc + '/fc1/gamma'], g[sc + '/fc1/beta'], dx=dy1a[i], dbias=g[sc + '/fc1/b'])
        dPQ = self._buf('rn/dPQ', (2, 2, M, U))                          # [half][scope], as PQ
        K.rn_pair_bwd(dy1a, dPQ[0], dPQ[1], 2 * B, k, U)
        self._rn_bwd_tail(r, d_feat, dPQ, dy2a, B, k, U)

    def _rn_bwd_tail(self, r, d_feat, dPQ, dy2a, B, k, U):
        """what follows the pair backward: the feature gradient through fc1's two half-projections, and the weight gradients"""
        p, g = self.params.p, self.params.g
        M, ps = B * k, r['ps']
        W1 = p['rn_h/fc1/W']
        feat = r['feat']
        # d_feat += dP . W1[:U]^T, then += dQ . W1[U:]^T (two launches: both write d_feat)
        K.gemm_batched('nt', 2, 1, M, U, U, dPQ[0], U, (M * U, 0), W1, U, (ps, 0), d_feat, U, (M * U, 0),
                       accumulate=True)
        K.gemm_batched('nt', 2, 1, M, U, U, dPQ[1], U, (M * U, 0), W1[U:], U, (ps, 0), d_feat, U, (M * U, 0),
                       accumulate=True)
        # The weight gradients feed nothing inside backward: on the two-stream schedule they leave the chain
        # of small launches between the decoders' and the encoders' recurrences and run on the side stream
        # (their operands y1, dy2a, feat, dPQ are not rewritten before the streams join at the end of backward).
        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for i, sc in enumerate(self.RN_SCOPES):
                K.matmul_tn(r['y1'][i], dy2a[i], out=g[sc + '/fc2/W'])   # K = B*k*k: split-K, one call each
            # gW1[:U] = feat^T dP, gW1[U:] = feat^T dQ for both scopes: four problems, one launch
            K.gemm_batched('tn', 2, 2, U, U, M, feat, U, (M * U, 0), dPQ, U, (M * U, 2 * M * U),
                           g['rn_h/fc1/W'], U, (ps, U * U))

    # ------------------------------------------------------------------ greedy decoding (N1)
    PROGRAM_END_TOKEN = 3      # vocab.token2int['m)'] in the Karel and every ViZDoom vocabulary

    def greedy_decode(self):
        """Greedy twins of the program and action decoders (models/model_full.py:513-523,
        546-558): argmax feedback from the <s> row (id token_dim) until the end token, weights
        shared with the teacher-forced decoders.  Needs forward() on the current feed (uses
        its summarizer outputs as initial states).  Reached from run_test / evaluation only."""
        ctx, c, p = self._ctx, self.config, self.params.p
        B, k, T, L = c.batch_size, c.k, c.max_demo_len, c.max_program_len
        U, V, A = c.num_lstm_cell_units, c.dim_program_token, c.action_space
        M = B * k
        out = {}
        todo = [('prog', B, L, V, ctx['init_h'], ctx['init_c'], self.PROGRAM_END_TOKEN)]
        if self.multitask:
            todo.append(('act', M, T, A, ctx['demo_h'], ctx['demo_c'], A - 1))
        for scope, R, steps, tok, h0, c0, end in todo:
            kernel, bias = p[scope + '/lstm/kernel'], p[scope + '/lstm/bias']
            # input projection of every possible token, once: [tok+1, 4U]
            table_proj = K.matmul_nn(p[scope + '/embedding'], kernel[:U], bias=bias,
                                     out=self._buf(scope + '/greedy_table', (tok + 1, 4 * U)))
            logits = self._buf(scope + '/greedy_logits', (steps, R, tok))
            ids = self._buf(scope + '/greedy_ids', (steps, R), torch.int32)
            lens = self._buf(scope + '/greedy_len', (R,), torch.int32)
            K.greedy_decode(table_proj, kernel[U:], p[scope + '/proj'], h0, c0, tok, end, steps,
                            logits, ids, lens)
            out[scope] = (logits, ids, lens)
        lp, ip, np_ = out['prog']
        self._greedy = dict(
            greedy_pred_program=lp.permute(1, 2, 0),                  # [B, V, L]
            greedy_program_tokens=ip.permute(1, 0),                   # [B, L]
            greedy_pred_program_len=np_.view(B, 1))
        if self.multitask:
            la, ia, na = out['act']
            self._greedy.update(
                greedy_pred_action=la.view(T, B, k, A).permute(1, 2, 0, 3),   # [B, k, T, A]
                greedy_action_tokens=ia.view(T, B, k).permute(1, 2, 0),       # [B, k, T]
                greedy_pred_action_len=na.view(B, k))
        return self._greedy

    @staticmethod
    def sequence_stats(pred, gt, pred_len, gt_len):
        """Accuracy statistics of Sequence_Loss (models/model_full.py:626-683) on the host
        (these are logging metrics in the reference too).  pred, gt: [B, token_dim, max_len]
        arrays; lengths [B].  Returns token_acc, seq_acc, is_same_seq, pred_tokens."""
        pred, gt = np.asarray(pred, np.float32), np.asarray(gt, np.float32)
        pred_len, gt_len = np.asarray(pred_len).reshape(-1), np.asarray(gt_len).reshape(-1)
        max_len = pred.shape[2]
        pos = np.arange(max_len)[None, :]
        gt_mask = (pos < gt_len[:, None]).astype(np.float32)
        max_mask = (pos < np.maximum(pred_len, gt_len)[:, None]).astype(np.float32)
        min_mask = (pos < np.minimum(pred_len, gt_len)[:, None]).astype(np.float32)
        label_argmax = gt.argmax(axis=1)
        logit_argmax = pred.argmax(axis=1)
        eq = (label_argmax == logit_argmax).astype(np.float32)
        token_acc = float((eq * min_mask).sum() / max_mask.sum())
        seq_equal = (label_argmax * gt_mask) == (logit_argmax * gt_mask)
        is_same_seq = seq_equal.all(axis=1) & (gt_len == pred_len)
        return dict(token_acc=token_acc, seq_acc=float(is_same_seq.mean()),
                    is_same_seq=is_same_seq, pred_tokens=logit_argmax)

    def report(self, with_greedy=True, with_programs=None):
        """report_loss / report_accuracy / report_hist of the reference
        (models/model_full.py:1102-1132): losses, token / sequence accuracies of the
        teacher-forced and greedy program and action decoders, and -- `with_programs`, default on
        for Karel -- the DSL metrics: syntax accuracy, exact-program accuracy and the execution
        histograms on the seen and the held-out demonstrations (host-side, as in the reference's
        py_funcs).  Returns (loss, accuracy) dicts; the histograms and per-row results are kept
        on the model (`report_hist`, `program_is_correct_syntax`, ...)."""
        from . import program_metrics as PM
        c, f = self.config, self._feed
        B, k, T = c.batch_size, c.k, c.max_demo_len
        if with_programs is None:
            with_programs = self.vocab is not None
        if with_programs and self.vocab is None:
            raise NotImplementedError('no DSL vocabulary for perception_type=%r (vizdoom_env/dsl.py)'
                                      % (getattr(c, 'perception_type', None),))
        karel = self.dataset_type == 'karel'
        with_execution = with_programs and (karel or self.world_factory is not None)
        parse = PM.parser_for(self.dataset_type)
        gt_prog = f['program'].cpu().numpy()
        plen = f['program_len'].cpu().numpy()
        dlen = f['demo_len'].cpu().numpy().reshape(B, k)
        gt_act = f['a_h'].view(B, k, T, self.action_space).permute(0, 1, 3, 2).cpu().numpy()
        t = self._terms.cpu().numpy()
        loss = {'program_loss': float(t[0])}
        if self.multitask:
            loss.update({'avg_action_loss': float(t[1]), 'avg_per_loss': float(t[2])})
        acc, hist, rows = {}, {}, {}
        gt_tokens = gt_prog.argmax(axis=1)

        if with_programs:
            vocab = self.vocab
            make_error = getattr(c, 'env_type', None) != 'no_error'
            # the feed keeps depth%4 frames padded to NHWC4 (get_feed_dict); metrics see the real depth
            s_h = f['s_h'].view(B, k, T, c.h, c.w, -1)[..., :c.depth].float().cpu().numpy()
            host = f.get('host', {})
            test = None

            def host_np(name):
                v = host[name]
                return v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)

            if not karel and with_execution and 'init_pos' not in host:
                raise KeyError('ViZDoom execution metrics need init_pos / init_pos_len in the batch')
            if 'test_s_h' in host and 'test_demo_len' in host:
                ts = host['test_s_h']
                ts = ts.cpu().numpy() if torch.is_tensor(ts) else np.asarray(ts)
                tl = host['test_demo_len']
                tl = tl.cpu().numpy() if torch.is_tensor(tl) else np.asarray(tl)
                test = (ts.astype(np.float32), tl.astype(np.int32).reshape(B, -1))

        def program_block(prefix, st, p_len):
            """syntax / exact / execution metrics of one decoded program set"""
            syn = PM.check_correct_syntax(vocab, st['pred_tokens'], p_len, st['is_same_seq'], parse=parse)
            exact = PM.exact_program_compare(vocab, st['pred_tokens'], p_len, syn, gt_tokens, plen, parse=parse)
            rows[prefix + 'is_correct_syntax'] = syn
            rows[prefix + 'exact_program_correct'] = exact
            out = {'syntax_acc': float(syn.mean()), 'exact_acc': float(exact.mean())}
            if not with_execution:
                return out

            def execute(first_frames, which, demo_k):
                if karel:
                    return PM.generate_program_output(vocab, first_frames, T, st['pred_tokens'], p_len, syn,
                                                      st['is_same_seq'], make_error)
                return PM.generate_program_output_vizdoom(
                    vocab, self.world_factory, host_np(which + 'init_pos'), host_np(which + 'init_pos_len'),
                    c.vizdoom_pos_keys, T, demo_k, c.h, c.w, c.depth, st['pred_tokens'], p_len, syn,
                    st['is_same_seq'])

            exe, exe_len = execute(s_h[:, :, 0], '', k)
            num, ok, h_ = PM.compare_demo_and_execution(s_h, dlen, exe, exe_len, st['is_same_seq'])
            rows[prefix + 'num_execution_correct'], rows[prefix + 'is_correct_execution'] = num, ok
            out['hist'] = h_
            if test is not None and (karel or 'test_init_pos' in host):
                exe, exe_len = execute(test[0][:, :, 0], 'test_', test[0].shape[1])
                num, ok, h_ = PM.compare_demo_and_execution(test[0], test[1], exe, exe_len, st['is_same_seq'])
                rows['test_' + prefix + 'num_execution_correct'] = num
                rows['test_' + prefix + 'is_correct_execution'] = ok
                out['test_hist'] = h_
            return out

        st = self.sequence_stats(self.pred_program.cpu().numpy(), gt_prog, plen, plen)
        acc['program_token_acc'], acc['program_seq_acc'] = st['token_acc'], st['seq_acc']
        if with_programs:
            r = program_block('program_', st, plen)
            acc['program_syntax_acc'] = r['syntax_acc']
            acc['pred_exact_program_accuracy'] = r['exact_acc']
            if 'hist' in r:
                hist['program_execution_acc_hist'] = r['hist']
            if 'test_hist' in r:
                hist['test_program_execution_acc_hist'] = r['test_hist']
        if self.multitask:
            pa = self.pred_action.permute(0, 1, 3, 2).cpu().numpy()
            sts = [self.sequence_stats(pa[:, i], gt_act[:, i], dlen[:, i], dlen[:, i]) for i in range(k)]
            acc['avg_action_token_acc'] = float(np.mean([s_['token_acc'] for s_ in sts]))
            acc['avg_action_seq_acc'] = float(np.mean([s_['seq_acc'] for s_ in sts]))
        if with_greedy:
            g = self.greedy_decode()
            glen = g['greedy_pred_program_len'].cpu().numpy().reshape(-1)
            st = self.sequence_stats(g['greedy_pred_program'].cpu().numpy(), gt_prog, glen, plen)
            acc['greedy_program_token_acc'], acc['greedy_program_seq_acc'] = st['token_acc'], st['seq_acc']
            if with_programs:
                r = program_block('greedy_', st, glen)
                acc['greedy_program_syntax_acc'] = r['syntax_acc']
                acc['greedy_exact_program_accuracy'] = r['exact_acc']
                if 'hist' in r:
                    hist['greedy_program_execution_acc_hist'] = r['hist']
                if 'test_hist' in r:
                    hist['test_greedy_program_execution_acc_hist'] = r['test_hist']
            if self.multitask:
                ga = g['greedy_pred_action'].permute(0, 1, 3, 2).cpu().numpy()
                gl = g['greedy_pred_action_len'].cpu().numpy()
                sts = [self.sequence_stats(ga[:, i], gt_act[:, i], gl[:, i], dlen[:, i]) for i in range(k)]
                acc['greedy_avg_action_token_acc'] = float(np.mean([s_['token_acc'] for s_ in sts]))
                acc['greedy_avg_action_seq_acc'] = float(np.mean([s_['seq_acc'] for s_ in sts]))
        self.report_accuracy, self.report_hist, self._program_rows = acc, hist, rows
        return loss, acc

    # per-row results of the last report(), under the attribute names evaler.py:264-278 reads
    def _row(self, name):
        rows = getattr(self, '_program_rows', None)
        if not rows or name not in rows:
            raise AttributeError('%s: call Model.report(with_programs=True) first' % name)
        return rows[name]

    @property
    def program_is_correct_syntax(self):
        return self._row('program_is_correct_syntax')

    @property
    def greedy_program_is_correct_syntax(self):
        return self._row('greedy_is_correct_syntax')

    @property
    def program_num_execution_correct(self):
        return self._row('program_num_execution_correct')

    @property
    def program_is_correct_execution(self):
        return self._row('program_is_correct_execution')

    @property
    def greedy_num_execution_correct(self):
        return self._row('greedy_num_execution_correct')

    @property
    def greedy_is_correct_execution(self):
        return self._row('greedy_is_correct_execution')

    @property
    def test_greedy_num_execution_correct(self):
        return self._row('test_greedy_num_execution_correct')

    @property
    def test_greedy_is_correct_execution(self):
        return self._row('test_greedy_is_correct_execution')

    @property
    def greedy_pred_program(self):
        return self._greedy['greedy_pred_program']

    @property
    def greedy_pred_program_len(self):
        return self._greedy['greedy_pred_program_len']

    # ------------------------------------------------------------------ reference attributes
    @property
    def loss(self):
        return self._loss

    @property
    def report_loss(self):
        t = self._terms
        return {'program_loss': t[0], 'avg_action_loss': t[1], 'avg_per_loss': t[2]}

    @property
    def pred_program(self):
        """[B, dim_program_token, max_program_len] logits, zero past max(program_len)."""
        return self._ctx['dp']['logits'].permute(1, 2, 0)

    @property
    def ground_truth_program(self):
        return self._feed['program']

    @property
    def program_len(self):
        return self._feed['program_len'].view(-1, 1)

    def _per_demo_logits(self, e, token_dim):
        c = self.config
        B, k, T = c.batch_size, c.k, c.max_demo_len
        M = B * k
        out = e['logits'].clone()
        K.zero_past_group_steps(out, self._feed['demo_len'], T, M, token_dim, k)   # per-call padding
        return out.view(T, B, k, token_dim).permute(1, 2, 0, 3)                  # [B,k,T,token_dim]

    @property
    def pred_action(self):
        """[B, k, T, action_space] (models/model_full.py:559-560)."""
        return self._per_demo_logits(self._ctx['da'], self.action_space)

    @property
    def pred_per(self):
        return self._per_demo_logits(self._ctx['dq'], self.per_dim)

    @property
    def output(self):
        """[gt, pred] pairs: program, then per demo action, then per demo perception
        (models/model_full.py:919-933,1034,1077)."""
        f = self._feed
        c = self.config
        B, k, T = c.batch_size, c.k, c.max_demo_len
        out = [f['program'], self.pred_program]
        if not self.multitask:                # the baselines: the program pair alone (model_synthesis.py / model_summarizer.py)
            return out
        pa, pq = self.pred_action, self.pred_per
        gt_a = f['a_h'].view(B, k, T, self.action_space)
        gt_q = f['per'].view(B, k, T, self.per_dim)
        for i in range(k):
            out.extend([gt_a[:, i].permute(0, 2, 1), pa[:, i].permute(0, 2, 1)])
        for i in range(k):
            out.extend([gt_q[:, i].permute(0, 2, 1), pq[:, i].permute(0, 2, 1)])
        return out
