"""Training driver behind the reference's ``Trainer`` surface (trainer.py:16-240 of
shaohua0116/demo2program): same constructor, ``train`` / ``run_single_step`` / ``run_test`` /
``log_step_message``, the same CLI flags and log line.

The TF session + queue runners are replaced by: a host batch source -> ``Model.get_feed_dict``
(H2D) -> ``Model.forward`` / ``Model.backward`` (HIP kernels) -> optional RCCL all-reduce of
the flat gradient buffer -> fused global-norm-clip + Adam kernel (trainer.py:102-109).
One process per GPU; ``torch.distributed`` (backend "nccl" == RCCL) is used only for that
all-reduce.
"""
import argparse
import contextlib
import math
import os
import time

import numpy as np
import torch

from . import kernels as K
from .config import config_from_dataset, dataset_module, has_dataset, input_ops_module, make_config
from .dist import DataParallel
from .options import flag
from .synthetic import make_batch

CLIP_GRADIENTS = 20.0        # trainer.py:107
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # [TF-1.3] AdamOptimizer defaults


def learning_rate_at(config, step):
    """trainer.py:82-93: constant, or exponential_decay(0.5 / 10000 steps, staircase)."""
    if getattr(config, 'lr_weight_decay', False):
        return config.learning_rate * 0.5 ** (step // 10000)
    return config.learning_rate


class SyntheticBatches(object):
    """Stand-in for create_input_ops (karel_env/input_ops_karel.py:24-125) when no HDF5
    dataset is present: an endless stream of seeded synthetic batch_chunks."""

    def __init__(self, config, seed=123, n_distinct=8, rank=0):
        self.batches = [make_batch(config, seed=seed + 7919 * rank + i) for i in range(n_distinct)]
        self.i = 0

    def next(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return b


class FeedPrefetcher(object):
    """Stages the NEXT batch while the current step runs on the GPU: after a step has been
    launched (asynchronously), `stage()` takes the next host batch_chunk from the input pipeline
    and issues its host-to-device copies (and the NHWC4 padding of 3-channel frames) on a copy
    stream; `take()` hands out that device-resident feed
    after making the training stream wait for the copy event.  This is the reference's
    queue-runner + feed_dict step (trainer.py:187-199) taken off the critical path; with uint8
    frames one Karel batch is 6.5 MB of PCIe traffic.

    Single-threaded on purpose: a loader thread doing this work fights the training thread for
    the GIL (measured 3x slower than no prefetching at all); all that is needed is that the
    staging happens between a step's launch and the next synchronisation point."""

    def __init__(self, model, batches):
        self.model, self.batches = model, batches
        # the copy stream must not share a hardware queue with the compute streams (it would queue behind them)
        from .models.model_full import pick_concurrent_stream
        busy = [torch.cuda.current_stream()]
        if getattr(model, 'use_side_stream', False):
            busy.append(model._side_stream())
        self.stream = pick_concurrent_stream(against=busy)
        self._staged = None
        self.stage()

    def stage(self):
        """Host batch -> device feed on the copy stream.  The copies come from pageable memory, so the
        host blocks in them -- but it does so AFTER the current step was launched, i.e. while the GPU
        is busy, which is all the overlap that is needed.  (Copying into hipHostMalloc'ed staging
        buffers first was measured 2x slower end to end: CPU writes into that memory are slow here.)"""
        chunk = self.batches.next()
        with torch.cuda.stream(self.stream):
            feed = self.model.get_feed_dict(chunk)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._staged = (feed, ev)

    def take(self):
        feed, ev = self._staged
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in feed.values():           # allocated on the copy stream, consumed on this one
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return feed

    def next(self):
        """take() + stage() for callers that do not separate launch and staging (no overlap)."""
        feed = self.take()
        self.stage()
        return feed

    def close(self):
        pass


def _new_event():
    """An event recorded on the current stream (tests of the multi-rank protocol replace this and _device_sync: the
    protocol logic below then runs on CPU tensors)."""
    ev = torch.cuda.Event()
    ev.record()
    return ev


def _device_sync():
    torch.cuda.synchronize()


class StepOutput(object):
    """The `output` slot of run_single_step / run_test (trainer.py:191-192,217-219 fetch `self.model.output`: the
    [ground truth, prediction] pairs of the program, the k action and the k perception decoders,
    models/model_full.py:919-933,1034,1077) as a LAZY list of numpy arrays: nothing is computed or copied until an
    element is read -- the reference's own train loop never reads it (trainer.py:163-166) -- and it must be read before
    the next forward pass of the model (afterwards the buffers hold another batch: RuntimeError).

    After a TRAINING step the action / perception predictions are zero past a row's own length where the reference's
    decoders are free-running up to the call's max(len) (BasicDecoder without impute_finished,
    models/model_full.py:465-471): Trainer.train_step does not run those steps (nothing in the step reads them).
    `model.decoder_skip_past_len = False` before the step, or `model.forward(feed)` outside a step -- what run_test
    does -- gives the free-running values."""

    def __init__(self, model):
        self._model, self._pass, self._items = model, model.forward_count, None

    def _get(self):
        if self._items is None:
            if self._model.forward_count != self._pass:
                raise RuntimeError('this step\'s output was not read before the next forward pass of the model')
            self._items = [t.detach().cpu().numpy() for t in self._model.output]
        return self._items

    def __len__(self):
        return 2 * (1 + 2 * self._model.k) if self._model.multitask else 2

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class StepGuard(object):
    """Host side of the guarded optimizer step (include/d2p.h: d2p_adam_clip_flat_guarded).

    The persistent recurrent kernels never hang: a hand-off that does not arrive within a bounded wait (a
    workgroup that was not resident -- a shared device, a collective's kernels holding CUs) sets a sticky device
    word and the launch runs to completion with invalid results.  The clip + Adam kernel reads that word ON THE
    DEVICE and skips its update while it is set, counting applied / skipped steps in `counters`; nothing has to
    synchronise for the parameters to stay intact.  The kernel also writes both counters into a slot of a pinned
    host ring (one slot per step in flight, no copy operation on the stream) and the feed is remembered; `poll`
    looks at the slots whose step has finished (an event recorded behind the kernel).  When one shows a skipped step the trainer synchronises, resets the word, switches
    the recurrences to the per-step kernels and re-runs exactly the skipped steps (they are the LAST ones: the
    word is sticky) -- see Trainer._recover.

    The ring is DEPTH steps deep: before a slot is reused the host waits for its copy, i.e. it runs at most DEPTH
    steps ahead of the device (the device never idles for that: DEPTH - 1 steps are still queued).  With several
    ranks only that oldest copy is consulted, so that every rank detects a failure at the same step index (the
    skip decision itself is identical on all ranks: it travels through the gradient all-reduce)."""
    DEPTH = 4

    def __init__(self, device='cuda', moving=None):
        self.counters = torch.zeros(2, dtype=torch.int64, device=device)      # applied, skipped
        self.host = torch.zeros(self.DEPTH, 2, dtype=torch.int64)
        if str(device).startswith('cuda'):
            self.host = self.host.pin_memory()
        # the batch-norm moving statistics as they were BEFORE each step in flight (one row per ring slot; `moving`: the
        # model's flat buffer of them).  A skipped step's forward pass has already moved them -- the conv encoder's
        # run before the recurrences, and on the ranks whose kernels did not fail every layer's do -- so a re-run
        # starts from the snapshot of the first skipped step
        self.moving = moving
        # (DEPTH + 1 rows: `snapshot_ahead` fills step n + 1's row during step n, BEFORE the poll in front of step n + 1
        #  has looked at step n + 1 - DEPTH -- whose row must survive until then)
        self.moving_ring = None if moving is None else torch.zeros(self.DEPTH + 1, moving.numel(), dtype=moving.dtype,
                                                                   device=moving.device)
        self.ahead = -1                           # step index whose slot `snapshot_ahead` has filled
        self.events = [None] * self.DEPTH
        self.records = [None] * self.DEPTH        # (feed, global_step, adam_step) of the step in that slot
        self.n = 0                                # guarded steps launched
        self.handled = 0                          # skipped steps already re-run
        self.failures = 0
        self.waited = 0.0                         # seconds the host spent waiting for a ring slot (it was DEPTH steps ahead)

    def mirror(self):
        """The pinned slot the NEXT guarded step writes its counters into."""
        return self.host[self.n % self.DEPTH]

    def snapshot(self):
        """Before the forward pass of the next guarded step: its slot keeps the moving statistics of this moment (one
        device copy of a few kilobytes, stream-ordered).  Skipped when `snapshot_ahead` already filled the slot."""
        if self.moving_ring is not None and self.ahead != self.n:
            self.moving_ring[self.n % (self.DEPTH + 1)].copy_(self.moving, non_blocking=True)

    def snapshot_ahead(self, stream):
        """The NEXT step's snapshot, taken on `stream` (the model's side stream, behind this step's forward pass: nothing
        moves the statistics until the next forward pass) instead of as a 4-us copy in front of the next step's first
        kernel; the caller joins `stream` before that step starts (Model.backward does)."""
        if self.moving_ring is None:
            return
        row = self.moving_ring[(self.n + 1) % (self.DEPTH + 1)]
        if stream is None:                   # (the CPU protocol test: no streams)
            row.copy_(self.moving)
        else:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                row.copy_(self.moving, non_blocking=True)
        self.ahead = self.n + 1

    def restore(self, k):
        """The moving statistics as they were before the k-th last launched step."""
        if self.moving_ring is not None:
            self.moving.copy_(self.moving_ring[(self.n - k) % (self.DEPTH + 1)])
        self.ahead = -1                      # (a snapshot taken ahead saw the statistics this call just replaced)

    def launched(self, record):
        slot = self.n % self.DEPTH
        self.events[slot] = _new_event()
        self.records[slot] = record
        self.n += 1

    def poll(self, deterministic, wait_all=False):
        """True when a copy that has arrived shows skipped steps that were not re-run yet.  Waits for the copy of
        the slot the next step will reuse; `deterministic` (several ranks): looks at nothing else."""
        slot = self.n % self.DEPTH
        worst = 0
        for i, ev in enumerate(self.events):
            if ev is None:
                continue
            if i == slot or wait_all:
                if not ev.query():
                    t0 = time.perf_counter()
                    ev.synchronize()
                    self.waited += time.perf_counter() - t0
            elif deterministic or not ev.query():
                continue
            worst = max(worst, int(self.host[i, 1]))
        return worst > self.handled

    def last(self, k):
        return [self.records[(self.n - k + i) % self.DEPTH] for i in range(k)]

    def forget(self):
        self.events = [None] * self.DEPTH


class Trainer(object):

    @staticmethod
    def get_model_class(model_name):
        """trainer.py:18-30.  'summarizer' and 'synthesis_baseline' are ablations of the full
        graph and run as variants of the same Model (it reads config.model); the induction
        baseline is a different graph (attention over the test demonstrations) and is not built."""
        if model_name in ('full', 'summarizer', 'synthesis_baseline'):
            from .models.model_full import Model
        elif model_name == 'induction_baseline':
            raise NotImplementedError('induction_baseline is outside the MI355X hot path')
        else:
            raise ValueError(model_name)
        return Model

    def __init__(self, config, dataset=None, dataset_test=None, make_train_dir=True, dp=None,
                 use_graph=None):
        self.config = config
        if use_graph is None:
            # eager launches by default since round 2: a step is ~215 launches (the recurrences are one
            # launch per sequence), the host keeps ahead of the device, and the two-stream schedule runs
            # faster eagerly (4.82 ms) than as a captured two-queue graph (4.94) or one stream (5.02).
            # D2P_GRAPH=1 captures forward + backward per (n_prog, n_demo) as before (frees the host).
            use_graph = flag('D2P_GRAPH')
        self.use_graph = bool(use_graph)
        # overlap of the decoders' all-reduce with the encoder backward: built and tested, OFF by default --
        # the persistent LSTM kernels of that part of backward need every CU, so a collective kernel running
        # beside them only delays their start (one rank, forced RCCL group: 5.20 vs 5.02 ms/step); no
        # multi-GPU box was available to show a gain at N > 1
        self.dp_overlap = flag('D2P_DP_OVERLAP')
        # the training loop on a priority -1 stream of the trainer's own (step_stream)
        self.priority_stream = flag('D2P_PRIORITY_STREAM')
        self._prio = None
        # CUs the recurrences behind the split point are planned for while a collective runs beside them (round 6): a
        # four-wave workgroup -- a collective's kernel -- never becomes resident on a CU that holds a persistent
        # recurrence's workgroup, so a launch over all 256 CUs and the collective only take turns; 7 row domains of 32
        # column tiles instead of 8 cost the step 0.2 % (profiles/r06b_cu_budget_karel.log) and leave 32 CUs free.
        # 0: the round-3 form (the recurrences behind the split run on the per-step kernels, which need no co-residency)
        self.dp_overlap_cus = max(0, torch.cuda.get_device_properties(0).multi_processor_count - 32) \
            if torch.cuda.is_available() else 0
        self._graphs = {}
        self._static_feed = None
        hyper_parameter_str = 'bs_{}_lr_{}_{}_cell_{}'.format(
            config.batch_size, config.learning_rate, config.encoder_rnn_type,
            config.num_lstm_cell_units)
        if config.scheduled_sampling:
            hyper_parameter_str += '_sd_{}'.format(config.scheduled_sampling_decay_steps)
        hyper_parameter_str += '_k_{}'.format(config.num_k)
        self.train_dir = './train_dir/%s-%s-%s-%s-%s-%s' % (
            config.dataset_type, '_'.join(config.dataset_path.split('/')), config.model,
            config.prefix, hyper_parameter_str, time.strftime("%Y%m%d-%H%M%S"))
        if make_train_dir and not os.path.exists(self.train_dir):
            os.makedirs(self.train_dir)

        self.batch_size = config.batch_size
        self.dp = dp if dp is not None else DataParallel()
        self.batch_train = self._batches(dataset, config, 123, True)
        self.batch_test = self._batches(dataset_test, config, 321, False)

        self.global_step = 0
        self.adam_step = 0          # steps the Adam moments have accumulated (beta powers of TF's Adam)
        Model = self.get_model_class(config.model)
        self.model = Model(config, debug_information=config.debug, global_step=self.global_step)
        self.dp.broadcast_params(self.model.params.flat)

        self.log_step = config.log_step
        self.test_sample_step = config.test_sample_step
        self.write_summary_step = config.write_summary_step
        dev = self.model.params.flat.device
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        # guarded optimizer step (StepGuard): always on -- the unguarded form applied invalid gradients until a host-side
        # status check noticed
        self.guard = StepGuard(self.model.params.flat.device, getattr(self.model, 'moving_flat', None))
        self._recovering = False

        if config.checkpoint is not None:
            self.load_checkpoint(config.checkpoint)

    # ------------------------------------------------------------------ one optimizer step
    def _batches(self, dataset, config, seed, is_training):
        """A reference-style Dataset (has get_data) gets the input pipeline of trainer.py:43-48
        (create_input_ops; with data parallelism each rank reads ids[rank::world]);
        anything with .next() is used as is; None -> synthetic batches."""
        if dataset is None:
            return SyntheticBatches(config, seed, 8 if is_training else 2, rank=self.dp.rank)
        if hasattr(dataset, 'get_data'):
            create_input_ops = input_ops_module(config.dataset_type).create_input_ops
            ids = dataset.ids[self.dp.rank::self.dp.world_size] if self.dp.world_size > 1 else dataset.ids
            # frames stay uint8 end to end (the HDF5 stores booleans / 0..255 pixels): a quarter of the H2D bytes and of
            # conv1's HBM reads; the kernels widen on load, results identical to the float32 feed
            _, batch = create_input_ops(dataset, self.batch_size, is_training=is_training, data_id=ids,
                                        shuffle=is_training, seed=seed + self.dp.rank, frames_dtype=np.uint8)
            return batch
        return dataset

    @contextlib.contextmanager
    def step_stream(self):
        """`with trainer.step_stream():` -- the steps enqueued inside run on a HIGH-PRIORITY stream of the trainer's own
        (D2P_PRIORITY_STREAM=1, the default; eager schedule only; `Trainer.train` and bench.py enter it around their
        loops): the step's main queue then takes a CU the moment one frees, in front of the side queue's weight-gradient
        workgroups -- a persistent recurrence needs ALL its workgroups resident before its first hand-off completes, and
        its last ones entered 17-27 us late behind ~68 us workgroups of the other queue
        (profiles/r06_launch_stamps_step.log).  2.605 -> 2.582 ms per step at config 2, 5.22 -> 5.18 at config 4; a
        priority-0 stream of its own gains nothing (profiles/r06_stream_priority_ab*.log).  Ordered with the caller's
        current stream at entry and exit -- ONCE per loop: the same two waits around every single step cost more than the
        priority gains (2.60 -> 2.63 ms, profiles/r06_ab_priority_stream_bench.log), so `train_step` itself stays on
        whatever stream is current."""
        st = self._priority_stream()
        cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
        if st is None or cur == st:
            yield cur
            return
        st.wait_stream(cur)
        try:
            with torch.cuda.stream(st):
                yield st
        finally:
            cur.wait_stream(st)

    def _priority_stream(self):
        if not self.priority_stream or self.use_graph or not torch.cuda.is_available() \
                or torch.cuda.is_current_stream_capturing():
            return None
        if self._prio is None:
            self._prio = torch.cuda.Stream(priority=-1)
        return self._prio

    def train_step(self, feed):
        """forward + backward + (all-reduce) + clip + Adam on a device-resident feed.
        Asynchronous: returns the device loss tensor without synchronising.

        With use_graph (D2P_GRAPH=1) forward+backward are captured once per (n_prog, n_demo) --
        the two batch-dependent step counts of dynamic_decode -- into a hipGraph reading a
        static copy of the feed, and replayed: ~500 launches per step collapse into one
        graph launch.  The all-reduce, norm and Adam stay outside the graph (RCCL call; the
        Adam rate changes per step and is read from device memory)."""
        m = self.model
        g = self.guard
        if not self._recovering and g.poll(self.dp.active):
            self._recover()
        if m.scheduled_sampling:
            m.set_sampling_step(self.global_step)     # sampling probability + noise counter of this step
        g.snapshot()
        P = m.params
        # data parallelism: the decoders' gradients (the tail of the flat buffer) are all-reduced while
        # the rest of backward runs; one message for everything when that is switched off
        overlap = self.dp.active and self.dp_overlap and not (m.use_side_stream and self.use_graph)
        dec = m.decoder_grad_offset() if overlap else 0
        start = None
        persist_was = K.lstm_is_persistent()
        # the encoder recurrences that follow the split run beside the collective's kernels.  The persistent kernels need
        # every workgroup resident at once: they are planned for dp_overlap_cus CUs there, so that the collective's
        # workgroups find CUs of their own (should it take more than that leaves -- a hand-off then times out -- the
        # guarded step is skipped on every rank and re-run on the per-step kernels like any other hand-off failure).
        # dp_overlap_cus = 0: per-step launches, which need no co-residency.  (A captured graph holds the launches it
        # was captured with: both toggles act on eager launches and at capture time.)
        budget_after_split = overlap and persist_was and self.dp_overlap_cus > 0
        per_step_after_split = overlap and persist_was and not budget_after_split
        if overlap:
            def start():
                self.dp.all_reduce_start(P.grad[dec:])
                self._after_split(per_step_after_split, budget_after_split)
        try:
            if self.use_graph and not self._profiling():
                loss = self._graphed_forward_backward(feed, start)
            else:
                loss = m.forward(feed, defer_loss=True)
                if getattr(m, 'use_side_stream', False) and not self._recovering:
                    g.snapshot_ahead(m._side_stream())          # the next step's moving-statistics snapshot, off the critical path
                m.backward(split_cb=start)
        finally:
            if per_step_after_split:
                K.lstm_set_persistent(True)       # (also when backward or the collective raised)
            if budget_after_split:
                K.lstm_set_cu_budget(0)
        slot = None
        if self.dp.active:
            # this rank's status word joins the exchange: after the SUM every rank skips the step together
            slot = P.status_slot
            K.step_status_publish(slot)
        if overlap:
            self.dp.all_reduce_finish(P.grad[:dec])
            if slot is not None:
                self.dp.all_reduce_grads(slot)
        else:
            # SUM over ranks; mean folded into prescale (grad_all = grad + the status slot behind it)
            self.dp.all_reduce_grads(P.grad_all if slot is not None else P.grad)
        pre = self.dp.prescale
        if getattr(m, '_abl', None) is not None and m._abl('adam'):      # (timing experiment: tools/step_ablation.py)
            self.adam_step += 1
            self.global_step += 1
            return loss
        K.l2norm_flat(P.grad, pre, self._sumsq)
        # bias correction from the number of steps the MOMENTS have seen, not from global_step: the
        # two differ when parameters are loaded without optimizer state (zero moments, restarted
        # beta powers -- TF's Adam keeps beta1_power / beta2_power next to its slots)
        t = self.adam_step + 1
        lr = learning_rate_at(self.config, self.global_step)
        lr_t = lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        K.adam_clip_flat(P.flat, P.grad, P.m, P.v, self._sumsq, pre, CLIP_GRADIENTS, lr_t,
                         ADAM_B1, ADAM_B2, ADAM_EPS, counters=g.counters, fail_slot=slot, mirror=g.mirror())
        g.launched((feed, self.global_step, self.adam_step))
        self.adam_step = t
        self.global_step += 1
        return loss

    def _after_split(self, per_step, budget):
        """what changes for the recurrences behind backward's split point while the decoders' all-reduce is in flight"""
        if per_step:
            K.lstm_set_persistent(False)
        if budget:
            K.lstm_set_cu_budget(self.dp_overlap_cus)

    MAX_PERSIST_FAILURES = 2      # after that many the run stays on the per-step recurrent kernels

    def _recover(self):
        """A guarded step was skipped on the device: the persistent recurrent kernels gave up a hand-off.  The
        parameters and the Adam moments are as they were before the first skipped step (the device skipped every
        step since: the word is sticky, and with several ranks it travels through the all-reduce); the batch-norm
        moving statistics are NOT -- a forward pass moves them before anything fails -- and are restored from the
        snapshot taken in front of that step (StepGuard.snapshot).  Synchronise, reset the word, re-run the skipped
        steps on the per-step kernels (lstm_step.hip)."""
        import sys
        g = self.guard
        _device_sync()
        applied, skipped = (int(v) for v in g.counters.tolist())
        k = skipped - g.handled
        err = K.lstm_persist_error(reset=True)
        g.handled = skipped
        g.forget()
        if k <= 0:
            return None
        g.failures += 1
        records = g.last(min(k, g.DEPTH))
        print('[demo2program_amd] persistent LSTM kernel gave up a hand-off (status 0x%08x): %d step(s) skipped on the '
              'device, re-running them on the per-step kernels%s' %
              (err & 0xffffffff, k, '' if g.failures < self.MAX_PERSIST_FAILURES else
               '; staying on the per-step kernels for the rest of the run'), file=sys.stderr)
        if k > g.DEPTH:          # cannot happen while poll() runs before every step; refuse to guess
            raise RuntimeError('%d skipped steps but only %d feeds kept' % (k, g.DEPTH))
        K.lstm_set_persistent(False)
        self._graphs.clear()     # captured graphs hold the persistent launches
        self.global_step, self.adam_step = records[0][1], records[0][2]
        # the skipped steps' forward passes moved the batch-norm moving statistics (every layer on the ranks whose own
        # kernels were fine, the layers in front of the recurrences everywhere): back to the first skipped step's
        g.restore(len(records))
        self._recovering = True
        loss = None
        try:
            for feed, _, _ in records:
                loss = self.train_step(feed)
            _device_sync()
        finally:
            self._recovering = False
        if g.failures < self.MAX_PERSIST_FAILURES:
            K.lstm_set_persistent(True)
            self._graphs.clear()
        # nothing can be skipped on the re-run: the per-step recurrent kernels and the separate conv / batch-norm
        # launches (Model._fused_encoder) have no bounded waits.  If the device still counted a skip, an optimizer step
        # would be lost silently while global_step / adam_step advance -- refuse
        skipped_after = int(g.counters[1].item())
        if skipped_after != skipped:
            raise RuntimeError('%d step(s) were skipped again while re-running skipped steps on the per-step kernels '
                               '(status 0x%08x): refusing to drop optimizer steps'
                               % (skipped_after - skipped, K.lstm_persist_error(reset=False) & 0xffffffff))
        g.handled = skipped_after
        return loss

    def settle(self):
        """Synchronising: waits for every launched step and re-runs skipped ones.  Returns the number of
        hand-off failures recovered so far."""
        g = self.guard
        if not self._recovering and g.poll(self.dp.active, wait_all=True):
            self._recover()
        return g.failures

    @staticmethod
    def _profiling():
        from .lib import load
        return bool(getattr(load(), '_d2p_prof_on', False))

    MAX_GRAPHS = 128      # distinct (n_prog, n_demo) pairs kept as instantiated graphs
    _STATIC_KEYS = ('s_h', 'program', 'program_tokens', 'a_h', 'a_h_tokens', 'per', 'per_rows', 'per_gram', 'active_rows',
                    'program_len',
                    'demo_len')

    def _graphed_forward_backward(self, feed, split_cb=None):
        """split_cb: with data parallelism the step is captured as TWO graphs cut at Model.backward's
        split point; split_cb (the start of the decoders' all-reduce) is called between their launches."""
        m = self.model
        sf = self._static_feed
        if sf is None or sf['s_h'].dtype != feed['s_h'].dtype:    # e.g. uint8 frames after float32 frames
            sf = self._static_feed = m.alloc_feed(feed['s_h'].dtype)
            self._graphs.clear()
        if '_flat' in feed and feed['_flat'].numel() == sf['_flat'].numel():
            sf['_flat'].copy_(feed['_flat'], non_blocking=True)   # the whole batch: one device copy
        else:                                                      # a hand-built feed
            for k in self._STATIC_KEYS:
                if k in feed:
                    sf[k].copy_(feed[k].reshape(sf[k].shape), non_blocking=True)
            if 'per_rows' not in feed:
                m.derive_per_rows(sf)
        key = (feed['n_prog'], feed['n_demo'])
        static = dict({k: sf[k] for k in self._STATIC_KEYS}, n_prog=key[0], n_demo=key[1], id=feed.get('id'),
                      host=feed.get('host'))
        key = key + (split_cb is not None,)
        g = self._graphs.get(key)
        if g is None and len(self._graphs) >= self.MAX_GRAPHS:
            # an unusually ragged dataset: stop instantiating graphs, run further new shapes eagerly
            loss = m.forward(static)
            m.backward(split_cb=split_cb)
            return loss
        if g is None:
            # one eager pass sizes every buffer / scratch, then capture the same schedule; the
            # BN moving statistics are restored so the warm-up does not count as a step
            saved = {n: (a.clone(), b.clone()) for n, (a, b) in m.moving.items()}
            m.forward(static)
            m.backward()
            for n, (a, b) in saved.items():
                m.moving[n][0].copy_(a)
                m.moving[n][1].copy_(b)
            torch.cuda.synchronize()
            if split_cb is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    m.forward(static)
                    m.backward()
            else:
                # two graphs sharing one memory pool, cut inside backward
                g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):
                    g1.capture_begin()

                    persist = K.lstm_is_persistent()
                    budget = persist and self.dp_overlap_cus > 0

                    def cut():
                        g1.capture_end()
                        g2.capture_begin(pool=g1.pool())
                        self._after_split(persist and not budget, budget)   # (the launches behind the split, as eager steps plan them)
                    m.forward(static)
                    try:
                        m.backward(split_cb=cut)
                    finally:
                        K.lstm_set_persistent(persist)
                        K.lstm_set_cu_budget(0)
                    g2.capture_end()
                torch.cuda.current_stream().wait_stream(cap)
                g = (g1, g2)
            self._graphs[key] = g
        else:
            m._ctx['feed'] = static
            m._feed = static
        if split_cb is None:
            g.replay()
        else:
            g[0].replay()
            split_cb()
            g[1].replay()
        return m.loss

    def run_single_step(self, batch, step=None, is_train=True):
        """trainer.py:186-205: step_time spans batch fetch + feed + run.  A FeedPrefetcher hands
        over feeds that are already on the device (the fetch + feed of the next batch overlaps
        this step)."""
        _start_time = time.time()
        if isinstance(batch, FeedPrefetcher):
            feed = batch.take()
            loss = self.train_step(feed)
            batch.stage()                 # next batch: host work + H2D while this step runs
        else:
            batch_chunk = batch.next()
            feed = self.model.get_feed_dict(batch_chunk, step=step, is_training=is_train)
            loss = self.train_step(feed)
        loss_value = float(loss.item())             # the reference fetches the loss every step
        if not self.dp.active and self.guard.poll(False, wait_all=True):
            # this step was skipped on the device (single rank: detected right here; several ranks detect it
            # together StepGuard.DEPTH steps later): re-run it, report the re-run's loss
            redo = self._recover()
            if redo is not None:
                loss_value = float(redo.item())
        _end_time = time.time()
        # (summary: the reference's serialized 'train' collection; here Model.report() builds it on write steps only)
        return self.global_step, None, loss_value, StepOutput(self.model), (_end_time - _start_time)

    def run_test(self, batch):
        """trainer.py:207-225: the training-mode graph on a test batch, no parameter update
        (SURVEY F7).  The batch-norm moving statistics DO move: the reference builds its layers with
        a Python is_train=True and updates_collections=None (models/ops.py:20-23), so the moving-
        average update is part of every forward of that graph, test batches included."""
        _start_time = time.time()
        batch_chunk = batch.next()
        feed = self.model.get_feed_dict(batch_chunk, is_training=False)
        guard = self.guard
        self.settle()                           # (a pending failure belongs to a training step, not to this batch)
        guard.ahead = -1                        # (this forward pass moves the statistics: the next step snapshots afresh)
        guard.snapshot()                        # (slot of the NEXT training step: rewritten before it is used)
        loss = self.model.forward(feed)
        loss_value = float(loss.item())
        # the 'test' summaries of the reference: greedy decoders + accuracies, and for Karel the
        # syntax / exact-program / execution metrics (models/model_full.py:1102-1177)
        self.last_test_report = self.model.report(with_greedy=True)
        if K.lstm_persist_error(reset=True):
            # a hand-off failure inside this forward pass: garbage must not reach the log / the event file -- the batch
            # again on the per-step kernels, from the moving statistics this pass started with (as evaler.py does)
            import sys
            print('[demo2program_amd] persistent LSTM kernel gave up a hand-off in a test batch: batch redone on the '
                  'per-step kernels', file=sys.stderr)
            guard.restore(0)
            was = K.lstm_is_persistent()
            K.lstm_set_persistent(False)
            try:
                loss_value = float(self.model.forward(feed).item())
                self.last_test_report = self.model.report(with_greedy=True)
            finally:
                K.lstm_set_persistent(was)
        _end_time = time.time()
        return self.global_step, self.last_test_report, loss_value, StepOutput(self.model), (_end_time - _start_time)

    def train(self, max_steps=1000000, prefetch=True):
        if getattr(self.model, '_ablate', None):
            raise RuntimeError('this model carries a timing-only ablation (%s): its steps leave work out and their '
                               'results are invalid -- Trainer.train refuses (tools/step_ablation.py times single '
                               'steps through Trainer.train_step)' % ','.join(sorted(self.model._ablate)))
        ckpt_save_step = 1000
        source = FeedPrefetcher(self.model, self.batch_train) if prefetch else self.batch_train
        # scalar summaries as TensorBoard event files in train_dir (trainer.py:116,170-178), rank 0 only
        writer = None
        if self.dp.rank == 0 and os.path.isdir(self.train_dir):
            from .summary import SummaryWriter
            writer = SummaryWriter(self.train_dir)
        with self.step_stream():
            self._train_loop(source, max_steps, writer, ckpt_save_step)
        self.settle()                          # (up to StepGuard.DEPTH trailing steps may still be skipped ones)
        if writer is not None:
            writer.close()

    def _train_loop(self, source, max_steps, writer, ckpt_save_step):
        for s in range(max_steps):
            step, train_summary, loss, output, step_time = \
                self.run_single_step(source, step=s, is_train=True)
            if s % self.log_step == 0:
                self.log_step_message(step, loss, step_time)
            if writer is not None and s % self.write_summary_step == 0:
                # the 'train' collection (models/model_full.py:1144-1173) of the step that just ran
                tl, ta = self.model.report(with_greedy=False)
                writer.add_scalars(dict({'loss/loss': loss}, **{'loss/' + n: v for n, v in list(tl.items()) + list(ta.items())
                                                                  if isinstance(v, float)}), step)
            if s % self.test_sample_step == 0:
                test_step, test_report, test_loss, _, test_step_time = self.run_test(self.batch_test)
                self.log_step_message(step, test_loss, test_step_time, is_train=False)
                if writer is not None:
                    tl, ta = test_report
                    writer.add_scalars(dict({'test_loss/loss': test_loss},
                                            **{'test_loss/' + n: v for n, v in list(tl.items()) + list(ta.items())
                                               if isinstance(v, float)}), step)
                    writer.flush()
            if s % ckpt_save_step == 0:
                # every rank settles (re-runs steps the device skipped) before rank 0 writes the parameters
                self.check_device_status()
                if self.dp.rank == 0:
                    self.save_checkpoint(os.path.join(self.train_dir, 'model-%d.npz' % step))

    def check_device_status(self):
        """The persistent LSTM kernels give up a hand-off after a bounded wait instead of hanging (a workgroup that was
        not resident, e.g. a shared device) and record it; the guarded optimizer step skipped those steps on the device.
        Synchronising: skipped steps are re-run here."""
        self.settle()

    def log_step_message(self, step, loss, step_time, is_train=True):
        """The reference's log line, verbatim (trainer.py:227-240)."""
        if step_time == 0:
            step_time = 0.001
        if self.dp.rank != 0:
            return
        print((" [{split_mode:5s} step {step:4d}] " +
               "Loss: {loss:.5f} " +
               "({sec_per_batch:.3f} sec/batch, {instance_per_sec:.3f} " +
               "instances/sec) "
               ).format(split_mode=(is_train and 'train' or 'val'),
                        step=step, loss=loss, sec_per_batch=step_time,
                        instance_per_sec=self.batch_size * self.dp.world_size / step_time))

    # ------------------------------------------------------------------ checkpoints (own format)
    def save_checkpoint(self, path):
        P = self.model.params
        blob = {'p/' + n: v for n, v in P.to_numpy('p').items()}
        blob.update({'m/' + n: v for n, v in P.to_numpy('m').items()})
        blob.update({'v/' + n: v for n, v in P.to_numpy('v').items()})
        for n, (mm, mv) in self.model.moving.items():
            blob['moving_mean/' + n] = mm.cpu().numpy()
            blob['moving_var/' + n] = mv.cpu().numpy()
        blob['global_step'] = np.asarray(self.global_step)
        blob['adam_step'] = np.asarray(self.adam_step)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, **blob)

    def load_checkpoint(self, path):
        """This build's .npz, or a TensorFlow V2 checkpoint prefix of the reference (`model-<step>`
        next to `model-<step>.index`): parameters and moving statistics only, as the reference's
        pretrain_saver restores trainable variables (trainer.py:115,142-147)."""
        from . import tf_checkpoint
        if tf_checkpoint.is_tf_checkpoint(path):
            # like the reference's pretrain_saver (trainer.py:100,115): trainable variables only --
            # global_step, the learning-rate / sampling schedules and Adam all restart at 0
            tf_checkpoint.import_checkpoint(path, self.model)
            self._restart_optimizer()
            return
        z = np.load(path)
        P = self.model.params
        P.load({n: z['p/' + n] for n in P.shapes})
        has_moments = 'm/' + next(iter(P.shapes)) in z
        if has_moments:
            for which, buf in (('m', P.m), ('v', P.v)):
                host = np.zeros(P.size, np.float32)
                for n in P.shapes:
                    a = z[which + '/' + n].reshape(-1)
                    host[P.offsets[n]:P.offsets[n] + a.size] = a
                buf.copy_(torch.from_numpy(host))
        for n in self.model.moving:
            if 'moving_mean/' + n in z:
                self.model.moving[n][0].copy_(torch.from_numpy(z['moving_mean/' + n]))
                self.model.moving[n][1].copy_(torch.from_numpy(z['moving_var/' + n]))
        if has_moments and 'global_step' in z:
            # a full resume of this build's own checkpoint; a parameters-only file restarts the schedules
            self.global_step = int(z['global_step'])
            self.adam_step = int(z['adam_step']) if 'adam_step' in z else self.global_step
        elif not has_moments:
            self._restart_optimizer()

    def _restart_optimizer(self):
        """Parameters came without optimizer state (a parameters-only file, a TensorFlow checkpoint): zero moments,
        restarted beta powers and schedules -- also when this trainer has already stepped."""
        P = self.model.params
        P.m.zero_()
        P.v.zero_()
        self.global_step = self.adam_step = 0


def build_arg_parser():
    """The reference's flags and defaults, verbatim (trainer.py:245-289), plus build-only
    --max_steps / --synthetic (trainer.py:153 hard-codes 1000000 steps)."""
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument('--debug', action='store_true', default=False)
    parser.add_argument('--prefix', type=str, default='default')
    parser.add_argument('--model', type=str, default='full',
                        choices=['synthesis_baseline', 'induction_baseline', 'summarizer', 'full'])
    parser.add_argument('--dataset_type', type=str, default='karel', choices=['karel', 'vizdoom'])
    parser.add_argument('--dataset_path', type=str, default='datasets/karel_dataset')
    parser.add_argument('--checkpoint', type=str, default=None)
    parser.add_argument('--log_step', type=int, default=10)
    parser.add_argument('--write_summary_step', type=int, default=100)
    parser.add_argument('--test_sample_step', type=int, default=100)
    parser.add_argument('--num_k', type=int, default=10)
    parser.add_argument('--batch_size', type=int, default=32)
    parser.add_argument('--learning_rate', type=float, default=0.001)
    parser.add_argument('--lr_weight_decay', action='store_true', default=False)
    parser.add_argument('--scheduled_sampling', action='store_true', default=False)
    parser.add_argument('--scheduled_sampling_decay_steps', type=int, default=20000)
    parser.add_argument('--encoder_rnn_type', default='lstm', choices=['lstm', 'rnn', 'gru'])
    parser.add_argument('--num_lstm_cell_units', type=int, default=512)
    parser.add_argument('--demo_aggregation', type=str, default='avgpool',
                        choices=['concat', 'avgpool', 'maxpool'])
    parser.add_argument('--max_steps', type=int, default=1000000, help='(build-only)')
    return parser


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    preset = 'karel' if args.dataset_type == 'karel' else 'vizdoom'
    flags = {k: v for k, v in vars(args).items() if k != 'max_steps'}
    flags['k'] = args.num_k
    config = make_config(preset, **flags)
    dp = DataParallel.from_env()
    dataset_train = dataset_test = None
    if has_dataset(config.dataset_path):
        dataset_train, dataset_test, _ = dataset_module(config.dataset_type).create_default_splits(
            config.dataset_path, num_k=config.num_k)
        config_from_dataset(config, dataset_train)          # trainer.py:306-335
    else:
        print('no dataset under %s: training on synthetic %s-shaped batches' % (config.dataset_path, preset))
    trainer = Trainer(config, dataset_train, dataset_test, dp=dp)
    trainer.train(max_steps=args.max_steps)


if __name__ == '__main__':
    main()
