"""Whole-path parity on a real MI355X: Model.forward / backward / Trainer.train_step (HIP
kernels through the C ABI) against the CPU oracle on the same seeded inputs and weights.

Tolerances (north_star): fp32 logits within 1e-4 absolute of the oracle; loss within 1e-5
relative; gradients within 2e-4 of each tensor's max |g| (fp32 accumulation over
thousands of terms vs the fp64 oracle)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import oracle_config, run_oracle, small_case

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    assert torch.cuda.is_available(), 'GPU tests need a real MI355X (run through gpurun)'
    from demo2program_amd import build
    build.build_library()


def _maxerr(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def _check_against_oracle(cfg, params, batch, out, grads, logit_tol=1e-4):
    from demo2program_amd.models.model_full import Model
    model = Model(cfg, params=params)
    feed = model.get_feed_dict(batch)
    loss = model.forward(feed)
    model.backward()
    torch.cuda.synchronize()
    ref_loss = float(out['loss'])
    assert abs(float(loss.item()) - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-6, (float(loss.item()), ref_loss)
    rl = model.report_loss
    assert abs(float(rl['program_loss'].item()) - float(out['program_loss'])) < 1e-4
    assert abs(float(rl['avg_action_loss'].item()) - float(out['avg_action_loss'])) < 1e-4
    assert abs(float(rl['avg_per_loss'].item()) - float(out['avg_per_loss'])) < 1e-4
    # logits: [B,V,L], [B,k,T,A] (oracle: [B,k,A,T]), [B,k,T,P]
    assert _maxerr(model.pred_program, out['pred_program']) <= logit_tol
    assert _maxerr(model.pred_action, out['pred_action'].permute(0, 1, 3, 2)) <= logit_tol
    assert _maxerr(model.pred_per, out['pred_per'].permute(0, 1, 3, 2)) <= logit_tol
    g = model.params.to_numpy('g')
    worst = []
    for n, ref in grads.items():
        ref = ref.double().numpy()
        err = np.abs(g[n].astype(np.float64) - ref).max()
        scale = np.abs(ref).max()
        worst.append((err / (scale + 1e-12), n, err, scale))
        assert err <= 2e-4 * scale + 1e-6, 'grad %s: err %.3e vs max|g| %.3e' % (n, err, scale)
    return model, sorted(worst)[-3:]


@pytest.mark.parametrize('kind', ['karel', 'vizdoom'])
def test_forward_backward_matches_oracle_small(kind):
    cfg, params, batch = small_case(kind)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    _check_against_oracle(cfg, params, batch, out, grads)


@pytest.mark.parametrize('kind,model,agg', [
    ('karel', 'summarizer', 'avgpool'), ('karel', 'synthesis_baseline', 'avgpool'),
    ('karel', 'synthesis_baseline', 'maxpool'), ('vizdoom', 'summarizer', 'avgpool'),
    ('vizdoom', 'synthesis_baseline', 'maxpool')])
def test_baseline_variants_match_oracle(kind, model, agg):
    """models/baselines/model_summarizer.py and model_synthesis.py as variants of the same graph:
    loss, program logits and every gradient against the oracle's restatement of them; the greedy
    program decoder too."""
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import to_torch
    cfg, params, batch = small_case(kind, seed=23, model=model, demo_aggregation=agg)
    assert not any(n.startswith(('act/', 'per/')) for n in params)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    m = Model(cfg, params=params)
    loss = m.forward(m.get_feed_dict(batch))
    m.backward()
    ref = float(out['loss'])
    assert abs(float(loss.item()) - ref) <= 1e-5 * abs(ref) + 1e-6
    assert _maxerr(m.pred_program, out['pred_program']) <= 1e-4
    g = m.params.to_numpy('g')
    assert set(g) == set(grads)
    for n, r in grads.items():
        r = r.double().numpy()
        assert np.abs(g[n] - r).max() <= 2e-4 * np.abs(r).max() + 1e-6, n
    rl, acc = m.report(with_greedy=True)
    assert set(rl) == {'program_loss'} and 'avg_action_token_acc' not in acc
    assert 'greedy_program_syntax_acc' in acc
    p64 = {n: torch.from_numpy(v).double() for n, v in params.items()}
    tb = to_torch(batch)
    fwd = oracle.forward(p64, tb, oracle_config(cfg))
    gref = oracle.greedy_program_and_actions(p64, tb, oracle_config(cfg), fwd)
    assert torch.equal(m.greedy_pred_program_len.view(-1).cpu().long(), gref['greedy_pred_program_len'])
    assert _maxerr(m.greedy_pred_program, gref['greedy_pred_program']) <= 1e-4


def test_config1_karel_synthesis_baseline_at_its_own_shape():
    """BASELINE.json configs[0] at its own size: Karel synthesis_baseline, k = 2 demonstrations, batch 4, U = 512,
    T = 20, L = 50 (the reference's CPU-runnable case) -- loss, logits and every gradient against the fp64 oracle."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('karel', seed=41, model='synthesis_baseline', demo_aggregation='avgpool',
                                    batch_size=4, k=2, max_demo_len=20, max_program_len=50, num_lstm_cell_units=512)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    m = Model(cfg, params=params)
    loss = m.forward(m.get_feed_dict(batch))
    m.backward()
    ref = float(out['loss'])
    assert abs(float(loss.item()) - ref) <= 1e-5 * abs(ref) + 1e-6
    assert _maxerr(m.pred_program, out['pred_program']) <= 1e-4
    g = m.params.to_numpy('g')
    assert set(g) == set(grads)
    for n, r in grads.items():
        r = r.double().numpy()
        assert np.abs(g[n] - r).max() <= 2e-4 * np.abs(r).max() + 1e-6, n


def test_baseline_trains_through_the_trainer():
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=31, model='summarizer')
    tr = Trainer(cfg, make_train_dir=False)
    tr.model.params.load(params)
    feed = tr.model.get_feed_dict(batch)
    losses = [float(tr.train_step(feed).item()) for _ in range(30)]
    assert losses[-1] < 0.7 * losses[0], losses[::6]
    with pytest.raises(ValueError):
        Trainer(small_case('karel', model='synthesis_baseline', demo_aggregation='concat')[0], make_train_dir=False)


def test_forward_backward_matches_oracle_u512_karel():
    # full-width cells (U=512, the shipped configuration) on a small batch
    cfg, params, batch = small_case('karel', seed=11, batch_size=2, k=2, max_demo_len=8,
                                    max_program_len=12, num_lstm_cell_units=512)
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    _check_against_oracle(cfg, params, batch, out, grads)


def test_matches_committed_golden_fixture():
    """tests/golden/karel_small.npz was produced by tests/golden/make_golden.py (fp32 oracle)."""
    z = np.load(os.path.join(GOLDEN, 'karel_small.npz'))
    cfg, params, batch = small_case('karel', seed=int(z['seed']))
    for n in params:
        assert np.array_equal(params[n], z['param/' + n]), n
    for n, v in batch.items():
        if v.dtype.kind not in 'US':
            assert np.array_equal(v, z['batch/' + n]), n
    from demo2program_amd.models.model_full import Model
    model = Model(cfg, params=params)
    loss = model.forward(model.get_feed_dict(batch))
    model.backward()
    assert abs(float(loss.item()) - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    assert np.abs(model.pred_program.cpu().numpy() - z['pred_program']).max() <= 1e-4
    assert np.abs(model.pred_action.cpu().numpy() - z['pred_action']).max() <= 1e-4
    assert np.abs(model.pred_per.cpu().numpy() - z['pred_per']).max() <= 1e-4
    g = model.params.to_numpy('g')
    for n in params:
        ref = z['grad/' + n]
        assert np.abs(g[n] - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-6, n


def test_uint8_frames_give_the_same_result_as_float_frames():
    """ViZDoom frames are bytes 0..255 (vizdoom_env/generator.py:184): feeding them as uint8
    (widened inside the conv loader, 4x less H2D traffic) must not change anything; also
    exercises the 3 -> 4 channel padding of conv1."""
    from demo2program_amd.models.model_full import Model
    for kind in ('vizdoom', 'karel'):
        cfg, params, batch = small_case(kind, seed=13)
        m1 = Model(cfg, params=params)
        l1 = float(m1.forward(m1.get_feed_dict(batch)).item())
        m1.backward()
        g1 = m1.params.grad.clone()
        b8 = dict(batch)
        b8['s_h'] = batch['s_h'].astype(np.uint8)
        assert np.array_equal(b8['s_h'].astype(np.float32), batch['s_h'])
        m2 = Model(cfg, params=params)
        l2 = float(m2.forward(m2.get_feed_dict(b8)).item())
        m2.backward()
        assert abs(l1 - l2) <= 1e-6 * abs(l1), (kind, l1, l2)
        assert (g1 - m2.params.grad).abs().max().item() <= 1e-5 * g1.abs().max().item()


def test_greedy_decode_matches_oracle():
    """greedy program / action decoders (GreedyEmbeddingHelper semantics):
    decoded token ids and lengths EXACTLY equal to the oracle, logits within 1e-4, and the
    Sequence_Loss accuracy statistics equal."""
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import to_torch
    for seed in (7, 19):
        cfg, params, batch = small_case('karel', seed=seed)
        # bias the projections so that the end token is actually produced by some rows
        params['prog/proj'][:, 3] += 0.02 * np.sign(params['prog/proj'][:, 3])
        m = Model(cfg, params=params)
        m.forward(m.get_feed_dict(batch))
        g = m.greedy_decode()
        p = {n: torch.from_numpy(v).double() for n, v in params.items()}
        tb = to_torch(batch)
        fwd = oracle.forward(p, tb, oracle_config(cfg))
        ref = oracle.greedy_program_and_actions(p, tb, oracle_config(cfg), fwd)
        assert torch.equal(g['greedy_pred_program_len'].view(-1).cpu().long(), ref['greedy_pred_program_len'])
        assert torch.equal(g['greedy_program_tokens'].cpu().long(), ref['greedy_program_ids'])
        assert _maxerr(g['greedy_pred_program'], ref['greedy_pred_program']) <= 1e-4
        assert torch.equal(g['greedy_pred_action_len'].cpu().long(), ref['greedy_pred_action_len'])
        assert torch.equal(g['greedy_action_tokens'].cpu().long(), ref['greedy_action_ids'])
        assert _maxerr(g['greedy_pred_action'], ref['greedy_pred_action'].permute(0, 1, 3, 2)) <= 1e-4
        # accuracy statistics
        loss, acc = m.report(with_greedy=True)
        plen = tb['program_len'].long().reshape(-1)
        st = oracle.sequence_stats(ref['greedy_pred_program'], tb['program'].double(),
                                   ref['greedy_pred_program_len'], plen, cfg.max_program_len,
                                   cfg.dim_program_token)
        assert abs(acc['greedy_program_token_acc'] - float(st['token_acc'])) < 1e-6
        assert abs(acc['greedy_program_seq_acc'] - float(st['seq_acc'])) < 1e-6
        st = oracle.sequence_stats(fwd['pred_program'], tb['program'].double(), plen, plen,
                                   cfg.max_program_len, cfg.dim_program_token)
        assert abs(acc['program_token_acc'] - float(st['token_acc'])) < 1e-6


def test_report_program_metrics_on_real_programs():
    """Model.report on a batch of real Karel programs with executed demonstrations: the DSL
    metrics of models/model_full.py:1102-1132 are present, consistent with the host functions
    applied to the model's own argmax tokens, and a model that predicts the ground truth gets
    full syntax / exact / execution accuracy."""
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.models import program_metrics as PM
    from demo2program_amd.models.model_full import Model
    cfg, params, _ = small_case('karel', seed=5)
    batch = sample_batch(cfg, seed=21)
    m = Model(cfg, params=params)
    m.forward(m.get_feed_dict(batch))
    loss, acc = m.report(with_greedy=True)
    for key in ('program_syntax_acc', 'pred_exact_program_accuracy', 'greedy_program_syntax_acc',
                'greedy_exact_program_accuracy', 'program_token_acc', 'greedy_avg_action_seq_acc'):
        assert 0.0 <= acc[key] <= 1.0, key
    for key in ('program_execution_acc_hist', 'greedy_program_execution_acc_hist',
                'test_program_execution_acc_hist', 'test_greedy_program_execution_acc_hist'):
        h = m.report_hist[key]
        assert h.shape == ((cfg.test_k if key.startswith('test_') else cfg.k) + 1,) and abs(h.sum() - 1) < 1e-6
    B = cfg.batch_size
    plen = batch['program_len'].reshape(-1).astype(np.int64)
    st = Model.sequence_stats(m.pred_program.cpu().numpy(), batch['program'], plen, plen)
    syn = PM.check_correct_syntax(m.vocab, st['pred_tokens'], plen, st['is_same_seq'])
    assert np.array_equal(syn, m.program_is_correct_syntax) and abs(acc['program_syntax_acc'] - syn.mean()) < 1e-7
    assert m.program_is_correct_execution.shape == (B, cfg.k)
    assert m.test_greedy_is_correct_execution.shape == (B, cfg.test_k)
    # a "perfect" decoder: the teacher-forced logits replaced by the one-hot ground truth
    onehot = torch.from_numpy(batch['program']).cuda().permute(2, 0, 1).contiguous() * 10.0
    m._ctx['dp']['logits'].copy_(onehot)
    _, acc2 = m.report(with_greedy=False)
    assert acc2['program_seq_acc'] == 1.0 and acc2['program_syntax_acc'] == 1.0
    assert acc2['pred_exact_program_accuracy'] == 1.0
    assert m.report_hist['program_execution_acc_hist'][cfg.k] == 1.0
    assert m.report_hist['test_program_execution_acc_hist'][cfg.test_k] == 1.0


def test_greedy_decode_length_and_padding_invariants():
    """Size-independent properties of dynamic_decode + GreedyEmbeddingHelper on the GPU outputs
    alone: length = 1 + first position of the end token (L if absent); logits / ids past the
    longest decoded sequence are exactly zero; all-zero logits decode token 0 for L steps."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('karel', seed=23)
    L, B = cfg.max_program_len, cfg.batch_size
    zero = {n: v.copy() for n, v in params.items()}
    zero['prog/proj'][:] = 0.0
    m = Model(cfg, params=zero)
    m.forward(m.get_feed_dict(batch))
    g = m.greedy_decode()
    assert g['greedy_pred_program_len'].cpu().numpy().reshape(-1).tolist() == [L] * B
    assert int(g['greedy_program_tokens'].abs().max()) == 0
    for boost in (0.0, 0.05, 0.5):
        pp = {n: v.copy() for n, v in params.items()}
        pp['prog/proj'][:, 3] += boost * np.sign(pp['prog/proj'][:, 3])
        pp['act/proj'][:, cfg.action_space - 1] += boost * np.sign(pp['act/proj'][:, cfg.action_space - 1])
        m = Model(cfg, params=pp)
        m.forward(m.get_feed_dict(batch))
        g = m.greedy_decode()
        for toks, lens, logits, end, steps in (
                (g['greedy_program_tokens'].cpu().numpy(), g['greedy_pred_program_len'].cpu().numpy().reshape(-1),
                 g['greedy_pred_program'].cpu().numpy(), 3, L),
                (g['greedy_action_tokens'].reshape(-1, cfg.max_demo_len).cpu().numpy(),
                 g['greedy_pred_action_len'].cpu().numpy().reshape(-1),
                 g['greedy_pred_action'].permute(0, 1, 3, 2).reshape(-1, cfg.action_space, cfg.max_demo_len).cpu().numpy(),
                 cfg.action_space - 1, cfg.max_demo_len)):
            n_run = int(lens.max())
            for r in range(toks.shape[0]):
                hits = np.nonzero(toks[r, :n_run] == end)[0]
                assert lens[r] == (hits[0] + 1 if len(hits) else steps)
            if n_run < steps:
                assert np.abs(logits[:, :, n_run:]).max() == 0 and np.abs(toks[:, n_run:]).max() == 0
            assert np.array_equal(logits[:, :, :n_run].argmax(axis=1), toks[:, :n_run])


def test_fused_decoders_option_gives_identical_results():
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('karel', seed=17)
    res = []
    for fuse in (False, True):
        m = Model(cfg, params=params)
        m.fuse_decoders = fuse
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        torch.cuda.synchronize()
        res.append((loss, m.params.grad.clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


def test_output_list_and_dynamic_padding():
    cfg, params, batch = small_case('karel', seed=5)
    # shorten every program / demo so that dynamic_decode stops early
    batch['program_len'][:] = np.minimum(batch['program_len'], 4)
    batch['demo_len'][:] = np.minimum(batch['demo_len'], 3)
    for b in range(cfg.batch_size):
        batch['program'][b, :, int(batch['program_len'][b, 0]):] = 0
    from demo2program_amd.models.model_full import Model
    model = Model(cfg, params=params)
    model.forward(model.get_feed_dict(batch))
    pp = model.pred_program.cpu()
    n = int(batch['program_len'].max())
    assert pp[:, :, n:].abs().max().item() == 0 and pp[:, :, :n].abs().max().item() > 0
    # rows shorter than the longest keep producing non-zero logits (impute_finished=False)
    short = int(np.argmin(batch['program_len'][:, 0]))
    if batch['program_len'][short, 0] < n:
        assert pp[short, :, n - 1].abs().max().item() > 0
    out = model.output
    assert len(out) == 2 + 2 * cfg.k + 2 * cfg.k
    assert tuple(out[1].shape) == (cfg.batch_size, cfg.dim_program_token, cfg.max_program_len)
    assert tuple(out[3].shape) == (cfg.batch_size, cfg.action_space, cfg.max_demo_len)
    out2, _ = run_oracle(cfg, params, batch, dtype=torch.float64)
    assert _maxerr(model.pred_action, out2['pred_action'].permute(0, 1, 3, 2)) <= 1e-4


def test_run_single_step_and_run_test_return_the_output_pairs():
    """trainer.py:186-225: both return (step, summary, loss, output, step_time) with output = the fetched
    `model.output`.  Here the slot is a lazy list (StepOutput): read after a TRAINING step it holds the step's own
    (pre-update) predictions -- the oracle's inside every row's length, zeros past it for the action / perception
    decoders (documented deviation: the step does not run a row past its length) --, after run_test the reference's
    free-running values everywhere; reading it after a later forward pass raises."""
    from demo2program_amd.trainer import Trainer, StepOutput
    cfg, params, batch = small_case('karel', seed=5, num_lstm_cell_units=128)
    batch['demo_len'][0, 0] = 2                               # (some row shorter than its demonstration index's longest)

    class One(object):
        def next(self):
            return batch

    k, T = cfg.k, cfg.max_demo_len
    ref, _ = run_oracle(cfg, params, batch, dtype=torch.float64)
    tr = Trainer(cfg, dataset=One(), dataset_test=One(), make_train_dir=False)
    tr.model.params.load(params)
    step, summary, loss, output, dt = tr.run_single_step(One())
    assert isinstance(output, StepOutput) and len(output) == 2 + 4 * k and step == 1
    got = list(output)
    assert all(isinstance(a, np.ndarray) for a in got)
    assert np.abs(got[1] - ref['pred_program'].numpy()).max() <= 1e-4
    dlen = np.asarray(batch['demo_len']).reshape(cfg.batch_size, k).astype(int)
    inside = np.arange(T)[None, None, :] < dlen[:, :, None]                      # [B, k, T]
    for i in range(k):
        assert np.array_equal(got[2 + 2 * i], np.asarray(batch['a_h'])[:, i].transpose(0, 2, 1))
        for j, name in ((2 + 2 * i + 1, 'pred_action'), (2 + 2 * k + 2 * i + 1, 'pred_per')):
            want = ref[name][:, i].numpy()                                        # [B, A, T]
            m_in = inside[:, i][:, None, :]
            assert np.abs(np.where(m_in, got[j] - want, 0)).max() <= 1e-4, name
            assert np.abs(np.where(m_in, 0, got[j])).max() == 0, name             # zeros past a row's own length
    # run_test: the training-mode graph without the skip -> the reference's values everywhere (parameters have moved
    # by one step: against a fresh oracle pass on them)
    after = tr.model.params.to_numpy('p')
    ref2, _ = run_oracle(cfg, after, batch, dtype=torch.float64)
    _, report, _, out_t, _ = tr.run_test(One())
    got_t = list(out_t)
    for i in range(k):
        assert np.abs(got_t[2 + 2 * i + 1] - ref2['pred_action'][:, i].numpy()).max() <= 1e-4
        assert np.abs(got_t[2 + 2 * k + 2 * i + 1] - ref2['pred_per'][:, i].numpy()).max() <= 1e-4
    assert dlen[:, 0].max() > 2 and np.abs(got_t[3][0, :, 2:dlen[:, 0].max()]).max() > 0     # free-running past row 0's length
    # stale: an output not read before the next forward pass
    _, _, _, stale, _ = tr.run_single_step(One())
    tr.run_single_step(One())
    with pytest.raises(RuntimeError):
        stale[0]


def test_trainer_steps_match_oracle_adam():
    """Three optimizer steps (clip 20 + Adam) against the oracle's optimizer, and the loss
    goes down on a repeated batch."""
    from demo2program_amd.trainer import Trainer, SyntheticBatches
    cfg, params, batch = small_case('karel', seed=3)

    class One(object):
        def next(self):
            return batch

    tr = Trainer(cfg, dataset=One(), dataset_test=One(), make_train_dir=False)
    tr.model.params.load(params)
    p = {n: torch.from_numpy(v).double() for n, v in params.items()}
    m = {n: torch.zeros_like(v) for n, v in p.items()}
    v = {n: torch.zeros_like(v_) for n, v_ in p.items()}
    losses = []
    for step in (1, 2, 3):
        out, grads = run_oracle(cfg, {n: t.numpy() for n, t in p.items()}, batch)
        oracle.adam_clip_step(p, grads, m, v, step, cfg.learning_rate)
        _, _, loss, _, _ = tr.run_single_step(One())
        losses.append(loss)
        assert abs(loss - float(out['loss'])) <= 2e-4 * abs(float(out['loss']))
    got = tr.model.params.to_numpy('p')
    # Adam's update is sign-like (m/sqrt(v)): an element whose gradient is at fp32-noise level
    # can move +-lr in either direction, so compare the bulk, not the max (the Adam kernel
    # itself is checked exactly against the oracle in test_kernels_gpu.py).
    for n in p:
        diff = np.abs(got[n] - p[n].numpy())
        assert (diff > 2e-4).sum() <= max(2, 0.02 * diff.size), (n, int((diff > 2e-4).sum()), diff.size)
        assert diff.max() <= 3 * 1.01e-3 * 2, n
    assert losses[2] < losses[0]
    assert tr.global_step == 3


def test_graph_replay_survives_scratch_growth_and_new_shapes():
    """Regression: a captured hipGraph keeps the workspace pointers it recorded.  Growing the
    scratch buffer (a later batch with longer sequences, or greedy decoding between steps) used
    to free that memory under the graph -> memory access fault on replay.  Outgrown buffers are
    now retired, not freed; replayed steps must equal eager steps on interleaved batch shapes."""
    from demo2program_amd import kernels as K
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.trainer import Trainer
    s = K.Scratch()
    p1, n1 = s.get(1 << 20)
    p2, n2 = s.get((1 << 20) + 1)
    assert s.retired and s.retired[0].data_ptr() == p1 and n2 >= 2 * n1 and p2 != p1
    cfg, params, _ = small_case('karel', seed=3)
    batches = [sample_batch(cfg, seed=40 + i) for i in range(3)]

    def run(use_graph):
        tr = Trainer(cfg, make_train_dir=False, use_graph=use_graph)
        tr.model.params.load(params)
        feeds = [tr.model.get_feed_dict(b) for b in batches]
        assert len({(f['n_prog'], f['n_demo']) for f in feeds}) > 1     # several graph keys
        losses = []
        for i in (0, 1, 2, 0, 1, 2, 0):
            losses.append(float(tr.train_step(feeds[i]).item()))
            if use_graph and i == 1:
                tr.model.report(with_greedy=True)                       # eager work between replays
                K.SCRATCH.reserve(4 * K.SCRATCH._cur().buf.numel())     # force a growth
                junk = torch.full((K.SCRATCH._cur().retired[-1].numel() // 4,), 7.0, device='cuda')
                del junk
        return losses

    eager, graphed = run(False), run(True)
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 1e-4 * abs(a), (eager, graphed)
    assert graphed[-1] < graphed[0]


def test_eval_mode_matches_oracle_and_evaler_runs(tmp_path):
    """Model(is_train=False) (evaler.py:61): forward with moving-average batch norm equals the
    oracle's inference graph; backward is refused; Evaler restores a Trainer checkpoint, runs
    batches of generated programs and writes the reference's report files."""
    from demo2program_amd.evaler import Evaler, GeneratedKarelBatches
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import to_torch
    from demo2program_amd.trainer import Trainer
    cfg, params, _ = small_case('karel', seed=11)
    batch = sample_batch(cfg, seed=77)
    m = Model(cfg, params=params, is_train=False)
    g = torch.Generator().manual_seed(5)
    moving = {}
    for n, (mm, mv) in m.moving.items():
        a = torch.randn(mm.shape, generator=g) * 0.1
        b = torch.rand(mv.shape, generator=g) + 0.5
        mm.copy_(a)
        mv.copy_(b)
        moving[n] = (a.double(), b.double())
    loss = float(m.forward(m.get_feed_dict(batch)).item())
    p = {n: torch.from_numpy(v).double() for n, v in params.items()}
    tb = {n: v for n, v in to_torch(batch).items()}
    ref = oracle.forward(p, tb, oracle_config(cfg), moving=moving)
    assert abs(loss - float(ref['loss'])) <= 1e-4 * abs(float(ref['loss']))
    assert _maxerr(m.pred_program, ref['pred_program']) <= 1e-4
    for n, (mm, mv) in m.moving.items():                      # evaluation never updates them
        assert torch.equal(mm.cpu().double(), moving[n][0].float().double())
    with pytest.raises(RuntimeError):
        m.backward()

    tr = Trainer(cfg, make_train_dir=False)
    tr.model.params.load(params)
    tr.train_step(tr.model.get_feed_dict(batch))
    ck = str(tmp_path / 'model-1.npz')
    tr.save_checkpoint(ck)
    cfg.checkpoint, cfg.train_dir, cfg.output_dir = ck, '', str(tmp_path)
    cfg.max_steps, cfg.pred_program, cfg.quiet, cfg.write_summary = 2, True, False, True
    cfg.dataset_split, cfg.no_loss = 'test', False
    ev = Evaler(cfg, GeneratedKarelBatches(cfg, seed=5))
    ev.eval_run()
    assert ev.global_step == 1
    assert set(ev.final['acc']) >= {'program_syntax_acc', 'greedy_exact_program_accuracy', 'program_token_acc'}
    assert abs(sum(ev.final['hist']['test_greedy_program_execution_acc_hist']) - 1.0) < 1e-5
    txt = open(str(tmp_path / 'out_model-1.npz_test.txt')).read()
    assert txt.count('[id: ') == 2 * cfg.batch_size and 'gt: DEF run m(' in txt
    assert os.path.exists(str(tmp_path / 'out_model-1.npz_test.json'))
    assert '[Final Avg Report]' in open(ev.summary_file).read()


def test_trainer_on_the_converted_reference_dataset():
    """End to end on data written by the reference's own generator: Dataset -> create_input_ops ->
    get_feed_dict -> train steps -> run_test with the program metrics (tests/golden/karel_dataset,
    12 programs, 3 seen + 2 held-out demonstrations each)."""
    from demo2program_amd.config import make_config
    from demo2program_amd.karel_env import dataset_karel as DK
    from demo2program_amd.trainer import Trainer
    path = os.path.join(GOLDEN, 'karel_dataset')
    DK.rs = np.random.RandomState(123)
    tr_ds, te_ds, _ = DK.create_default_splits(path, num_k=3)
    program, _, s_h, test_s_h, a_h, _, _, _, _, _, _, per, _ = tr_ds.get_data(tr_ds.ids[0])
    cfg = make_config('karel', batch_size=4, k=int(s_h.shape[0]), num_k=3, test_k=int(test_s_h.shape[0]),
                      max_demo_len=int(s_h.shape[1]), max_program_len=int(program.shape[1]),
                      num_lstm_cell_units=64, dataset_path=path)
    tr = Trainer(cfg, tr_ds, te_ds, make_train_dir=False)
    losses = [tr.run_single_step(tr.batch_train)[2] for _ in range(12)]
    assert all(np.isfinite(losses)) and min(losses[-4:]) < losses[0]
    step, report, loss, _, _ = tr.run_test(tr.batch_test)
    assert step == 12 and np.isfinite(loss)
    _, acc = report
    assert 'greedy_program_syntax_acc' in acc and 'test_greedy_program_execution_acc_hist' in tr.model.report_hist
    tr.batch_train.close()
    tr.batch_test.close()


def test_trainer_and_evaler_command_lines(tmp_path, monkeypatch, capsys):
    """The two entry points with the reference's flags: `trainer` trains on the converted reference
    dataset, logs the reference's line, saves model-<step>.npz; `evaler --train_dir` picks the
    newest checkpoint, evaluates generated programs and writes the summary file."""
    import glob
    from demo2program_amd import evaler, trainer
    monkeypatch.chdir(tmp_path)
    ds = os.path.join(GOLDEN, 'karel_dataset')
    trainer.main(['--dataset_path', ds, '--batch_size', '4', '--num_k', '3', '--num_lstm_cell_units', '64',
                  '--max_steps', '12', '--prefix', 'clitest'])
    out = capsys.readouterr().out
    assert ' [train step' in out and 'instances/sec' in out and ' [val   step' in out
    dirs = glob.glob(str(tmp_path / 'train_dir' / 'karel-*clitest*'))
    assert len(dirs) == 1 and os.path.exists(os.path.join(dirs[0], 'model-1.npz'))
    # scalar summaries under the reference's tags, as a TensorBoard event file in train_dir (trainer.py:116,170-178)
    from demo2program_amd.summary import read_events
    ev_files = glob.glob(os.path.join(dirs[0], 'events.out.tfevents.*'))
    assert len(ev_files) == 1
    tags = {}
    for step, sc in read_events(ev_files[0]):
        for t, v in sc.items():
            tags.setdefault(t, []).append((step, v))
    for t in ('loss/loss', 'loss/program_loss', 'loss/program_token_acc', 'loss/program_syntax_acc', 'loss/avg_action_loss',
              'loss/avg_per_loss', 'test_loss/loss', 'test_loss/greedy_program_token_acc',
              'test_loss/greedy_avg_action_seq_acc'):
        assert t in tags and all(np.isfinite(v) for _, v in tags[t]), t
    evaler.main(['--train_dir', dirs[0], '--batch_size', '4', '--num_k', '3', '--num_lstm_cell_units', '64',
                 '--max_steps', '2', '--output_dir', str(tmp_path / 'eval')])
    out = capsys.readouterr().out
    assert 'Loaded from checkpoint!' in out and '[Final Avg Report]' in out and 'greedy_exact_program_accuracy' in out
    assert glob.glob(os.path.join(dirs[0], 'model-*_report_testdata8_num_k3.txt'))
    # the dataset's own split, one pass (evaler.py:431-450): max_steps = len(split) // batch_size
    evaler.main(['--train_dir', dirs[0], '--dataset_path', ds, '--dataset_split', 'train', '--batch_size', '2',
                 '--num_k', '3', '--num_lstm_cell_units', '64', '--output_dir', str(tmp_path / 'eval2'),
                 '--pred_program', '--result_data', '--result_data_path', str(tmp_path / 'result.hdf5')])
    out = capsys.readouterr().out
    assert '[Final Avg Report]' in out and 'test_greedy_program_execution_acc_hist' in out
    res = np.load(str(tmp_path / 'result.npz'))
    ids = sorted({n.split('/')[0] for n in res.files})
    assert len(ids) >= 2 and res[ids[0] + '/pred_program'].shape == res[ids[0] + '/program'].shape
    assert res[ids[0] + '/s_h'].shape[0] == 3 and res[ids[0] + '/test_s_h'].ndim == 5
    listing = glob.glob(str(tmp_path / 'eval2' / 'out_*_train.txt'))
    assert len(listing) == 1 and '[id: ' in open(listing[0]).read()
    # a baseline through the same two entry points (trainer.py:18-30 model switch)
    base = ['--model', 'synthesis_baseline', '--demo_aggregation', 'maxpool', '--dataset_path', ds, '--batch_size', '4',
            '--num_k', '3', '--num_lstm_cell_units', '64']
    trainer.main(base + ['--max_steps', '4', '--prefix', 'clibase'])
    dirs = glob.glob(str(tmp_path / 'train_dir' / 'karel-*synthesis_baseline*clibase*'))
    assert len(dirs) == 1, os.listdir(str(tmp_path / 'train_dir'))
    capsys.readouterr()
    evaler.main(base + ['--train_dir', dirs[0], '--dataset_split', 'train', '--max_steps', '1'])
    out = capsys.readouterr().out
    assert 'greedy_exact_program_accuracy' in out and 'avg_action_loss' not in out


def test_vizdoom_command_lines_and_metrics(tmp_path, monkeypatch, capsys):
    """dataset_type=vizdoom end to end on the converted fixture dataset (5-conv encoder on its 6x8
    frames, 17-tuple reader, 42-token vocabulary): trainer, then evaler with syntax / exact-program
    metrics; execution metrics appear once a world factory is supplied."""
    import glob
    from demo2program_amd import evaler, trainer
    from demo2program_amd.config import config_from_dataset, dataset_module, make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.vizdoom_env.input_ops_vizdoom import create_input_ops
    monkeypatch.chdir(tmp_path)
    ds = os.path.join(GOLDEN, 'vizdoom_dataset')
    common = ['--dataset_type', 'vizdoom', '--dataset_path', ds, '--batch_size', '2', '--num_k', '3',
              '--num_lstm_cell_units', '64']
    trainer.main(common + ['--max_steps', '6', '--prefix', 'viztest'])
    out = capsys.readouterr().out
    assert ' [train step' in out and ' [val   step' in out
    dirs = glob.glob(str(tmp_path / 'train_dir' / 'vizdoom-*viztest*'))
    assert len(dirs) == 1
    evaler.main(common + ['--train_dir', dirs[0], '--dataset_split', 'train', '--output_dir', str(tmp_path / 'ev')])
    out = capsys.readouterr().out
    assert 'greedy_program_syntax_acc' in out and 'greedy_exact_program_accuracy' in out
    assert 'execution_acc_hist' not in out              # no engine, no execution histogram

    class World(object):                                 # frames of the right size; every action works
        def new_episode(self, init_dict):
            self.s_h = [np.full((6, 8, 3), float(np.atleast_2d(init_dict['player_pos'])[0, 0]))]

        def state_transition(self, action):
            self.s_h.append(self.s_h[-1] + 1)

        def is_there(self, actor):
            return False

        in_target = is_there

    train, _, _ = dataset_module('vizdoom').create_default_splits(ds, num_k=3)
    cfg = config_from_dataset(make_config('vizdoom', batch_size=2, k=3, num_lstm_cell_units=64, dataset_path=ds), train)
    cfg.world_factory = World
    m = Model(cfg, is_train=False)
    assert m.vocab.token_dim() == cfg.dim_program_token == 42
    _, batch = create_input_ops(train, 2, shuffle=False, frames_dtype=np.uint8)
    m.forward(m.get_feed_dict(batch.next_sync(), is_training=False))
    loss, acc = m.report(with_greedy=True)
    k1 = cfg.k + 1
    assert m.report_hist['greedy_program_execution_acc_hist'].shape == (k1,)
    assert m.report_hist['test_greedy_program_execution_acc_hist'].shape == (cfg.test_k + 1,)
    assert abs(float(m.report_hist['program_execution_acc_hist'].sum()) - 1.0) < 1e-6
    assert m.greedy_is_correct_execution.shape == (2, cfg.k)
    assert 0.0 <= acc['greedy_program_syntax_acc'] <= 1.0


def test_feed_prefetcher_equals_direct_feeding():
    """FeedPrefetcher (loader thread, pinned buffers, copy stream) hands out the same feeds in the
    same order as feeding host batches directly: identical losses step by step."""
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.trainer import FeedPrefetcher, Trainer
    cfg, params, _ = small_case('karel', seed=29)
    batches = [sample_batch(cfg, seed=60 + i) for i in range(4)]

    class Cycle(object):
        def __init__(self):
            self.i = 0

        def next(self):
            b = batches[self.i % 4]
            self.i += 1
            return b

    def run(prefetch):
        tr = Trainer(cfg, make_train_dir=False)
        tr.model.params.load(params)
        src = FeedPrefetcher(tr.model, Cycle()) if prefetch else Cycle()
        out = [tr.run_single_step(src)[2] for _ in range(7)]
        if prefetch:
            src.close()
        return out

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert abs(x - y) <= 1e-6 * abs(x), (a, b)


def test_scheduled_sampling_decoders():
    """models/model_full.py:59-67,414-423.  (1) sampling probability 0 (global_step 0: teacher
    forcing probability 1.0) reproduces the teacher-forced path exactly; (2) with sampling on, the
    ids fed are ground truth where no draw was taken and a valid token elsewhere, <s> first;
    (3) GIVEN those fed ids, loss, logits and every gradient equal the oracle's
    (ScheduledEmbeddingTrainingHelper does not differentiate through the draw); (4) the decay
    schedule is polynomial_decay(1.0 -> 0.1)."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('karel', seed=13)
    base = Model(cfg, params=params)
    l0 = float(base.forward(base.get_feed_dict(batch)).item())
    base.backward()
    g0 = base.params.grad.clone()

    cfg_ss = small_case('karel', seed=13)[0]
    cfg_ss.scheduled_sampling = True
    cfg_ss.scheduled_sampling_decay_steps = 1000
    with pytest.raises(ValueError):
        Model(cfg_ss, params=params)                              # needs global_step
    m = Model(cfg_ss, params=params, global_step=0)
    for step in (0, 250, 1000, 5000):
        assert abs(m.sample_prob_at(step) - oracle.polynomial_decay(1.0, step, 1000, 0.1)) < 1e-12
    l1 = float(m.forward(m.get_feed_dict(batch)).item())          # step 0: p(sample) = 0
    m.backward()
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert (m.params.grad - g0).abs().max().item() <= 1e-5 * g0.abs().max().item()
    assert int(m._ctx['dp']['sampled'].sum()) == 0

    m.set_sampling_step(600)                                      # p(sample) = 0.54
    feed = m.get_feed_dict(batch)
    loss = float(m.forward(feed).item())
    m.backward()
    B, k, T, L = cfg.batch_size, cfg.k, cfg.max_demo_len, cfg.max_program_len
    V, A = cfg.dim_program_token, cfg.action_space
    n_p, n_d = feed['n_prog'], feed['n_demo']
    fed_p = m._ctx['dp']['fed_ids'].cpu().numpy()                 # [L, B]
    fl_p = m._ctx['dp']['sampled'].cpu().numpy()
    fed_a = m._ctx['da']['fed_ids'].cpu().numpy()                 # [T, B*k]
    fl_a = m._ctx['da']['sampled'].cpu().numpy()
    gt_p = np.concatenate([np.full((1, B), V + 1), batch['program_tokens'].T[:-1]], 0)
    gt_a = np.concatenate([np.full((1, B * k), A + 1), batch['a_h_tokens'].reshape(B * k, T).T[:-1]], 0)
    assert (fed_p[0] == V + 1).all() and (fed_a[0] == A + 1).all()
    assert (fed_p[fl_p == 0] == gt_p[fl_p == 0]).all() and (fed_a[fl_a == 0] == gt_a[fl_a == 0]).all()
    assert ((fed_p[fl_p == 1] >= 0) & (fed_p[fl_p == 1] < V)).all()
    assert 0.15 < fl_p[1:n_p].mean() < 0.9 and 0.2 < fl_a[1:n_d].mean() < 0.85   # rates: test_sched_sample_statistics
    fed = {'prog': torch.from_numpy(fed_p.T.copy()).long(),
           'act': torch.from_numpy(fed_a.T.reshape(B, k, T).copy()).long()}
    out, grads = run_oracle(cfg, params, batch, fed_ids=fed)
    assert abs(loss - float(out['loss'])) <= 2e-5 * abs(float(out['loss']))
    assert _maxerr(m.pred_program, out['pred_program']) <= 1e-4
    got = m.params.to_numpy('g')
    for n in grads:
        ref = grads[n].numpy()
        assert np.abs(got[n] - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-3), n
    # a different step draws different noise
    m.set_sampling_step(601)
    m.forward(feed)
    assert (m._ctx['da']['fed_ids'].cpu().numpy() != fed_a).any()


def test_scheduled_sampling_training_graph_equals_eager():
    """Trainer with --scheduled_sampling: the sampling probability and the noise counter live in
    device memory, so hipGraph replays follow the schedule and draw fresh noise; with the same
    seed the graphed and the eager trainer take identical steps."""
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=17)
    cfg.scheduled_sampling, cfg.scheduled_sampling_decay_steps = True, 10

    def run(use_graph):
        tr = Trainer(cfg, make_train_dir=False, use_graph=use_graph)
        tr.model.params.load(params)
        feed = tr.model.get_feed_dict(batch)
        losses, fed = [], []
        for _ in range(6):
            losses.append(float(tr.train_step(feed).item()))
            fed.append(tr.model._ctx['da']['fed_ids'].cpu().numpy().copy())
        return losses, fed

    (le, fe), (lg, fg) = run(False), run(True)
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-4 * abs(a), (le, lg)
    for a, b in zip(fe, fg):
        assert (a == b).all()
    assert any((fe[i] != fe[i + 1]).any() for i in range(1, 5))     # fresh draws per step


def test_bn_moving_statistics_follow_reference_updates():
    cfg, params, batch = small_case('karel', seed=9)
    from demo2program_amd.models.model_full import Model
    model = Model(cfg, params=params)
    model.forward(model.get_feed_dict(batch))
    out, _ = run_oracle(cfg, params, batch, dtype=torch.float64)
    mm, mv = torch.zeros(16, dtype=torch.float64), torch.ones(16, dtype=torch.float64)
    for (mean, var) in out['bn_stats']['conv1']:        # k calls -> k updates
        mm = 0.9 * mm + 0.1 * mean
        mv = 0.9 * mv + 0.1 * var
    assert _maxerr(model.moving['conv1'][0], mm) < 1e-5
    assert _maxerr(model.moving['conv1'][1], mv) < 1e-5


def test_headline_config_matches_oracle_at_full_size():
    """BASELINE config 2 at its FULL size (Karel, B=32, k=10, T=20, L=50, U=512), on a batch of
    real generated programs: loss, the three kinds of logits and every gradient tensor against
    the fp64 CPU oracle (one oracle forward+backward, ~10-20 s of host time)."""
    from demo2program_amd.config import make_config
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.params import init_params
    cfg = make_config('karel')
    params = init_params(cfg, 123)
    batch = sample_batch(cfg, seed=11)
    model = Model(cfg, params=params)
    loss = float(model.forward(model.get_feed_dict(batch)).item())
    model.backward()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    assert abs(loss - float(out['loss'])) <= 1e-5 * abs(float(out['loss']))
    assert _maxerr(model.pred_program, out['pred_program']) <= 1e-4
    assert _maxerr(model.pred_action, out['pred_action'].permute(0, 1, 3, 2)) <= 1e-4
    assert _maxerr(model.pred_per, out['pred_per'].permute(0, 1, 3, 2)) <= 1e-4
    got = model.params.to_numpy('g')
    # every gradient tensor within 2e-4 of its own scale (the bound of the small cases) plus an absolute floor of 5e-7:
    # the floor is what fp32 sums over 3 200 pair rows / 6 400 step rows leave (measured 2e-7 at most,
    # tools/grad_error_full_size.py) and matters only for the two tensors whose true gradient is ~1e-4 by
    # cancellation (rn_h/fc2/W, rn_h/fc2/b: 3.6e-4 / 7.3e-4 of their scale) and for per/fc/b (exactly 0 in exact
    # arithmetic: a bias under a batch norm); everything else is within 8e-5 of its scale
    for n in grads:
        ref = grads[n].numpy()
        assert np.abs(got[n] - ref).max() <= 2e-4 * np.abs(ref).max() + 5e-7, (n, float(np.abs(ref).max()))


def _one_training_step_against_oracle(cfg, params, batch):
    """ONE Trainer.train_step -- the schedule bench.py times: two streams, logits deferred into the loss-backward launch,
    the loss value out of that launch, action / perception decoders stopped at a row's own length, one-launch
    State_Encoder / relation networks where the geometry has them -- from given parameters on a given batch; its loss
    and every gradient tensor it leaves in params.grad against the fp64 oracle, then the parameters after the step
    against the oracle's clip(20) + Adam."""
    from demo2program_amd import kernels as K
    from demo2program_amd.trainer import Trainer
    tr = Trainer(cfg, make_train_dir=False, use_graph=False)
    m = tr.model
    m.params.load(params)
    feed = m.get_feed_dict(batch)
    loss = float(tr.train_step(feed).item())
    torch.cuda.synchronize()
    assert tr.settle() == 0 and K.lstm_persist_error() == 0           # nothing fell back to the per-step kernels
    # the step really took the timed schedule
    ctx = m._ctx
    assert m.use_side_stream and m._side_stream() != torch.cuda.current_stream()
    assert ctx['logits_deferred'] and m.fused_loss and m.decoder_skip_past_len and K.lstm_is_persistent()
    assert ctx['da']['row_order'] is not None
    if feed['n_active_pad'] and feed['n_t1_pad']:
        assert ctx['klists'].get('demo') is not None
    assert int(tr.guard.counters[0].item()) == 1 and int(tr.guard.counters[1].item()) == 0
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    out, grads = run_oracle(cfg, params, batch, dtype=torch.float64)
    ref = float(out['loss'])
    assert abs(loss - ref) <= 1e-5 * abs(ref) + 1e-6, (loss, ref)
    got = m.params.to_numpy('g')
    assert set(got) == set(grads)
    for n in grads:
        r = grads[n].double().numpy()
        assert np.abs(got[n] - r).max() <= 2e-4 * np.abs(r).max() + 5e-7, (n, float(np.abs(r).max()))
    # ... and the optimizer step it applied: the oracle's clip + Adam from the oracle's gradients.  Adam's first update
    # is lr * g / (|g| + eps) -- sign-like -- so an element whose gradient is at fp32-noise level may move by up to lr
    # either way: the bulk within 2e-5, every element within 2 lr
    p64 = {n: torch.from_numpy(v).double() for n, v in params.items()}
    mom = {n: torch.zeros_like(v) for n, v in p64.items()}
    var = {n: torch.zeros_like(v) for n, v in p64.items()}
    oracle.adam_clip_step(p64, grads, mom, var, 1, cfg.learning_rate)
    after = m.params.to_numpy('p')
    for n in p64:
        diff = np.abs(after[n] - p64[n].numpy())
        assert diff.max() <= 2.02 * cfg.learning_rate, n
        assert (diff > 2e-5).sum() <= max(2, 0.02 * diff.size), (n, int((diff > 2e-5).sum()), diff.size)
    return tr


def test_one_training_step_matches_oracle_at_full_size():
    """BASELINE config 2 at its FULL size through Trainer.train_step (what bench.py times), not Model.forward +
    backward: loss 1e-5 rel, every gradient 2e-4 max|g| + 5e-7 -- the bounds of
    test_headline_config_matches_oracle_at_full_size."""
    from demo2program_amd.config import make_config
    from demo2program_amd.karel_env.generator import sample_batch
    from demo2program_amd.params import init_params
    cfg = make_config('karel')
    tr = _one_training_step_against_oracle(cfg, init_params(cfg, 123), sample_batch(cfg, seed=11))
    assert tr.model._ctx.get('enc_fused') and tr.model._ctx['rn_h'].get('fused')     # the one-launch forms ran


def test_one_training_step_matches_oracle_at_the_vizdoom_geometry():
    """The same at BASELINE config 4's frame geometry (80x80x3, five conv layers with the batch norm folded into the
    conv launches), at a batch the fp64 oracle finishes in seconds."""
    cfg, params, batch = small_case('vizdoom', seed=53, h=80, w=80, batch_size=2, k=3, max_demo_len=4, max_program_len=6,
                                    num_lstm_cell_units=128)
    tr = _one_training_step_against_oracle(cfg, params, batch)
    assert 'conv1/bn_partial' in tr.model._bufs


def test_headline_config_properties():
    """BASELINE config 2 (Karel, B=32, k=10): too slow for a per-element oracle comparison in a
    unit test budget, so check size-independent properties: finite loss near the
    uniform-prediction value, determinism across two runs, zero-padded logits, gradient norm
    finite, and per-term losses summing to the total."""
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel')
    batch = make_batch(cfg, seed=123)
    model = Model(cfg, seed=123)
    feed = model.get_feed_dict(batch)
    l1 = float(model.forward(feed).item())
    model.backward()
    g1 = model.params.grad.clone()
    l2 = float(model.forward(feed).item())
    model.backward()
    assert l1 == l2 and torch.equal(g1, model.params.grad)          # deterministic reductions
    assert abs(l1 - (math.log(50) + math.log(6) + math.log(2))) < 0.5
    t = model.report_loss
    s = sum(float(t[n].item()) for n in ('program_loss', 'avg_action_loss', 'avg_per_loss'))
    assert abs(s - l1) < 1e-5
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    n = feed['n_prog']
    assert model.pred_program[:, :, n:].abs().max().item() == 0 if n < cfg.max_program_len else True
    # the backward recurrences run over rows sorted by length and stop each row domain at its longest row: every dz row
    # past its sequence's length is still an exact zero, and the gradients are those of the launches that run every row
    # for all steps (d2p_lstm_persist_set_sorted(0)) up to the summation order of the bias gradients
    from demo2program_amd import kernels as K
    T, M, U = cfg.max_demo_len, cfg.batch_size * cfg.k, cfg.num_lstm_cell_units
    lens = feed['demo_len'].view(1, M).long()
    past = (torch.arange(T, device='cuda').view(T, 1) >= lens).view(T * M)
    assert bool(past.any())
    for name in ('demo_lstm', 'second_lstm', 'act/lstm', 'per/lstm'):
        dz = model._bufs[name + '/dz'].view(T * M, 4 * U)
        assert dz[past].abs().max().item() == 0, name
        assert dz[~past].abs().max().item() > 0, name
    K.lstm_set_sorted(False)
    try:
        l3 = float(model.forward(feed).item())
        model.backward()
        g3 = model.params.grad.clone()
    finally:
        K.lstm_set_sorted(True)
    assert l3 == l1
    P = model.params
    names = [n_ for n_ in P.shapes if n_.endswith('/bias') and 'lstm' in n_]
    for n_, sh in P.shapes.items():
        o, cnt = P.offsets[n_], int(np.prod(sh))
        a, b = g1[o:o + cnt], g3[o:o + cnt]
        if n_ in names:
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, float(b.abs().max())), n_
        else:
            assert torch.equal(a, b), n_


def test_vizdoom_80x80_frames_match_oracle():
    """BASELINE config 4's frame geometry (80x80x3, five conv layers: the row-strip kernels of
    conv_rows.hip for conv1 / conv2, the direct and implicit-GEMM back ends for the rest) through
    Model.forward / backward against the oracle, at a batch the fp64 oracle finishes in seconds."""
    cfg, params, batch = small_case('vizdoom', seed=51, h=80, w=80, batch_size=2, k=2, max_demo_len=3,
                                    max_program_len=6)
    out, grads = run_oracle(cfg, params, batch)
    _check_against_oracle(cfg, params, batch, out, grads)


def test_batch_norm_folded_into_the_conv_launches_equals_the_separate_launches():
    """Model.fold_bn (round 5): conv1 / conv2 of the 80x80 geometry with their batch-norm statistics out of the conv
    launch and conv1's normalised output never written (conv2's forward and weight gradient read it through the affine)
    -- same loss, gradients and moving statistics as the separate conv / batch-norm launches."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('vizdoom', seed=61, h=80, w=80, batch_size=2, k=3, max_demo_len=4, max_program_len=6)
    res = []
    for fold in (False, True):
        m = Model(cfg, params=params)
        m.fold_bn = fold
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        torch.cuda.synchronize()
        assert ('conv1/bn_partial' in m._bufs) == fold and ('conv1/bn_scale' in m._bufs) == fold
        res.append((loss, {n: t.clone() for n, t in m.params.g.items()}, m.moving_flat.clone()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[0][0]), (res[0][0], res[1][0])
    for n in res[0][1]:
        scale = float(res[0][1][n].abs().max()) + 1e-12
        assert float((res[0][1][n] - res[1][1][n]).abs().max()) <= 5e-5 * scale, n
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-5, atol=1e-6)


def test_k25_demonstrations_match_oracle():
    """BASELINE config 5's k = 25: 25 batch-norm groups per layer, 625 relation-network pairs per
    program, 25 action / perception decoders -- against the oracle's per-demonstration loops."""
    cfg, params, batch = small_case('vizdoom', seed=52, k=25, batch_size=2, max_demo_len=4, max_program_len=6)
    out, grads = run_oracle(cfg, params, batch)
    _check_against_oracle(cfg, params, batch, out, grads)


@pytest.mark.parametrize('preset', ['vizdoom', 'vizdoom_k25'])
def test_vizdoom_full_size_properties(preset):
    """BASELINE configs 4 (80x80x3, B=32, k=10) and 5 per rank (k=25, B=16) at their full workload:
    size-independent properties as for the headline config -- finite loss near the uniform-prediction
    value, bitwise determinism of loss and gradient across two runs, loss terms summing to the total,
    zero-padded logits past the batch's longest program, and one optimizer step that lowers the loss
    on the same batch."""
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    cfg = make_config(preset)
    batch = make_batch(cfg, seed=77)
    tr = Trainer(cfg, make_train_dir=False)
    model = tr.model
    feed = model.get_feed_dict(batch)
    l1 = float(model.forward(feed).item())
    model.backward()
    g1 = model.params.grad.clone()
    l2 = float(model.forward(feed).item())
    model.backward()
    assert l1 == l2 and torch.equal(g1, model.params.grad)
    uniform = math.log(cfg.dim_program_token) + math.log(cfg.action_space) + math.log(2)
    assert abs(l1 - uniform) < 0.6, (l1, uniform)
    t = model.report_loss
    assert abs(sum(float(t[n].item()) for n in ('program_loss', 'avg_action_loss', 'avg_per_loss')) - l1) < 1e-5
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    n = feed['n_prog']
    if n < cfg.max_program_len:
        assert model.pred_program[:, :, n:].abs().max().item() == 0
    assert K_persist_ok()
    for _ in range(3):
        tr.train_step(feed)
    l3 = float(model.forward(feed).item())
    assert l3 < l1, (l3, l1)


def K_persist_ok():
    from demo2program_amd import kernels as K
    return K.lstm_persist_error() == 0


def test_rccl_single_rank_self_test(monkeypatch):
    """the exchange step's RCCL calls (broadcast of the flat parameters,
    all-reduce of the flat gradient, the bench's MAX reduce and barrier) on a one-rank process
    group, through Trainer.train_step: same losses as without a process group."""
    import socket
    from demo2program_amd.dist import DataParallel
    from demo2program_amd.trainer import Trainer
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    for n, v in (('RANK', '0'), ('LOCAL_RANK', '0'), ('WORLD_SIZE', '1'), ('MASTER_ADDR', '127.0.0.1'),
                 ('MASTER_PORT', str(port))):
        monkeypatch.setenv(n, v)
    cfg, params, batch = small_case('karel', seed=37)

    def run(dp):
        tr = Trainer(cfg, make_train_dir=False, dp=dp)
        tr.model.params.load(params)
        if dp is not None:
            dp.broadcast_params(tr.model.params.flat)
        feed = tr.model.get_feed_dict(batch)
        return [float(tr.train_step(feed).item()) for _ in range(4)]

    plain = run(None)
    dp = DataParallel.from_env(force_init=True)
    try:
        import torch.distributed as dist
        assert dp.initialized and dist.get_backend() == 'nccl' and dp.world_size == 1
        one_message = run(dp)                               # default: one message after backward
        # D2P_DP_OVERLAP=1 (one compute stream): the decoders' slice all-reduced on RCCL's stream from
        # Model.backward's split point, the rest after backward -- eager launches ...
        monkeypatch.setenv('D2P_DP_OVERLAP', '1')
        two_streams = run(dp)                               # default schedule: the collective hangs off the side stream
        monkeypatch.setenv('D2P_SIDE_STREAM', '0')
        eager = run(dp)
        # ... and as two graphs cut at the split point
        monkeypatch.setenv('D2P_GRAPH', '1')
        with_group = run(dp)
        assert dp.max_over_ranks(1.25) == 1.25
        dp.barrier()
    finally:
        dp.shutdown()
    # the one-message form runs the very same kernels; the overlap forms plan the encoder recurrences behind the split for
    # Trainer.dp_overlap_cus CUs (seven row domains instead of eight: another summation order of the bias gradients) and
    # the graph form multiplies the padded rows in
    # its weight gradients (no row lists in a captured step): equal to fp32 round-off, not bit for bit
    assert one_message == plain
    for other in (with_group, eager, two_streams):
        assert all(abs(a - b) <= 2e-5 * abs(b) for a, b in zip(other, plain)), (other, plain)


def test_step_on_the_priority_stream_equals_the_step_on_the_callers_stream(monkeypatch):
    """`with trainer.step_stream():` runs the steps on a priority -1 stream of the trainer's own (D2P_PRIORITY_STREAM=1,
    default; Trainer.train and bench.py enter it around their loops), ordered with the caller's stream at entry and exit:
    same kernels, same order within each queue -- the same losses and parameters bit for bit as on the caller's stream
    ('0'), with the caller writing the parameters right before entry and reading them right after exit, unsynchronised."""
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=43)

    def run(on):
        monkeypatch.setenv('D2P_PRIORITY_STREAM', on)
        tr = Trainer(cfg, make_train_dir=False)
        feed = tr.model.get_feed_dict(batch)
        caller = torch.cuda.current_stream()
        losses = []
        for rep in range(2):
            tr.model.params.load(params)               # on the caller's stream, right in front of the entry
            tr.adam_step = tr.global_step = 0
            tr.model.params.m.zero_()
            tr.model.params.v.zero_()
            with tr.step_stream() as st:
                assert torch.cuda.current_stream() == st
                if on == '1':
                    assert st != caller and st.priority < 0
                else:
                    assert st == caller
                for i in range(3):
                    losses.append(tr.train_step(feed))
            assert torch.cuda.current_stream() == caller
            after = tr.model.params.flat.clone()       # on the caller's stream, right behind the exit
        assert tr.settle() == 0
        return [float(v.item()) for v in losses], after

    la, pa = run('1')
    lb, pb = run('0')
    assert la == lb and la[:3] == la[3:]
    assert torch.equal(pa, pb)


def test_tf_checkpoint_export_import_round_trip(tmp_path):
    """tf_checkpoint.export_checkpoint writes parameters + moving statistics under the reference's
    variable names as a TF V2 bundle; Trainer.load_checkpoint / Evaler recognise such a prefix and
    restore an identical model (format unverified against TensorFlow itself, see the module)."""
    from demo2program_amd import tf_checkpoint
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=41)
    m = Model(cfg, params=params)
    for _ in range(2):                                   # move the moving statistics off their initial values
        m.forward(m.get_feed_dict(batch))
    prefix = str(tmp_path / 'model-77')
    written = tf_checkpoint.export_checkpoint(m, prefix, global_step=77)
    assert 'Demo_Encoder/State_Encoder/conv1/Conv/weights' in written and 'global_step' in written
    tr = Trainer(cfg, make_train_dir=False)
    tr.load_checkpoint(prefix)
    # like the reference's pretrain_saver (trainer.py:100,115): trainable variables only -- global_step,
    # the schedules and Adam's beta powers restart
    assert tr.global_step == 0 and tr.adam_step == 0
    a, b = m.params.to_numpy('p'), tr.model.params.to_numpy('p')
    for n in a:
        assert np.array_equal(a[n], b[n]), n
    for n in m.moving:
        assert torch.equal(m.moving[n][0], tr.model.moving[n][0]) and torch.equal(m.moving[n][1], tr.model.moving[n][1]), n
    # a file whose names do not match is reported, not half-loaded
    tensors = tf_checkpoint.read_bundle(prefix)
    tensors['Demo_Encoder/State_Encoder/conv1/Conv/kernel'] = tensors.pop('Demo_Encoder/State_Encoder/conv1/Conv/weights')
    tf_checkpoint.write_bundle(str(tmp_path / 'odd'), tensors)
    with pytest.raises(KeyError, match='conv1/Conv/kernel'):
        tf_checkpoint.import_checkpoint(str(tmp_path / 'odd'), tr.model)
    tf_checkpoint.import_checkpoint(str(tmp_path / 'odd'), tr.model,
                                    name_map={'conv1/W': 'Demo_Encoder/State_Encoder/conv1/Conv/kernel'})


def test_parameters_only_checkpoint_restarts_adam_and_schedules(tmp_path):
    """loading parameters without optimizer state (a TF checkpoint of the reference, or an
    .npz without moments) must not apply a late-step bias correction to zero moments: the first update
    after such a load equals the first update of a fresh optimizer on the same parameters, and a full
    .npz resume continues the step counters."""
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=43)
    tr = Trainer(cfg, make_train_dir=False)
    tr.model.params.load(params)
    feed = tr.model.get_feed_dict(batch)
    for _ in range(3):
        tr.train_step(feed)
    full = str(tmp_path / 'model-3.npz')
    tr.save_checkpoint(full)
    # parameters-only file
    z = dict(np.load(full))
    bare = str(tmp_path / 'bare.npz')
    np.savez(bare, **{k: v for k, v in z.items() if k.startswith('p/') or k.startswith('moving_')})
    a = Trainer(cfg, make_train_dir=False)
    a.load_checkpoint(bare)
    assert a.global_step == 0 and a.adam_step == 0
    b = Trainer(cfg, make_train_dir=False)          # fresh optimizer on the same parameters / statistics
    b.load_checkpoint(bare)
    b.global_step, b.adam_step = 0, 0
    a.train_step(a.model.get_feed_dict(batch))
    b.train_step(b.model.get_feed_dict(batch))
    pa, pb = a.model.params.to_numpy('p'), b.model.params.to_numpy('p')
    for n in pa:
        assert np.array_equal(pa[n], pb[n]), n
    # the step size of that first update is the learning rate (Adam's first step), not several times it
    before = {k[2:]: v for k, v in z.items() if k.startswith('p/')}
    step = max(np.abs(pa[n] - before[n]).max() for n in pa)
    assert step <= 1.001 * cfg.learning_rate, step
    # full resume keeps both counters
    c = Trainer(cfg, make_train_dir=False)
    c.load_checkpoint(full)
    assert c.global_step == 3 and c.adam_step == 3
    # ... and a parameters-only file loaded into a trainer that HAS stepped leaves no stale moments or counters
    # behind: same first update as the fresh trainer's
    c.train_step(c.model.get_feed_dict(batch))
    c.load_checkpoint(bare)
    assert c.global_step == 0 and c.adam_step == 0
    assert float(c.model.params.m.abs().max()) == 0.0 and float(c.model.params.v.abs().max()) == 0.0
    c.train_step(c.model.get_feed_dict(batch))
    pc = c.model.params.to_numpy('p')
    for n in pa:
        assert np.array_equal(pa[n], pc[n]), n


def test_run_test_moves_the_batch_norm_moving_statistics():
    """models/ops.py:20-23 (updates_collections=None, Python is_train=True): the moving-average update
    is part of every forward of the training graph, so the reference's run_test (trainer.py:207-225)
    moves the statistics without touching the parameters."""
    from demo2program_amd.trainer import Trainer

    class One(object):
        def __init__(self, b):
            self.b = b

        def next(self):
            return self.b
    cfg, params, batch = small_case('karel', seed=44)
    tr = Trainer(cfg, make_train_dir=False)
    tr.model.params.load(params)
    p0 = tr.model.params.to_numpy('p')
    mv0 = {n: (a.clone(), b.clone()) for n, (a, b) in tr.model.moving.items()}
    tr.run_test(One(batch))
    p1 = tr.model.params.to_numpy('p')
    for n in p0:
        assert np.array_equal(p0[n], p1[n]), n
    moved = [n for n, (a, b) in tr.model.moving.items() if not torch.equal(a, mv0[n][0]) or not torch.equal(b, mv0[n][1])]
    assert set(moved) == set(tr.model.moving), (moved, list(tr.model.moving))


def test_side_stream_really_runs_beside_the_main_stream():
    """HIP maps streams onto a few hardware queues; two streams on one queue serialise (a fresh stream does after
    init_process_group: RCCL's streams shift the assignment).  The stream the model's two-stream schedule and
    the feed prefetcher use is probed for concurrency: a spin on the current stream must not delay it."""
    from demo2program_amd.models.model_full import pick_concurrent_stream
    main = torch.cuda.current_stream()
    side = pick_concurrent_stream()
    assert side != main
    x = torch.zeros(64, device='cuda')
    with torch.cuda.stream(side):
        x.fill_(0.0)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0, e1, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(main)
    torch.cuda._sleep(4000000)
    e1.record(main)
    with torch.cuda.stream(side):
        x.fill_(1.0)
        c1.record(side)
    torch.cuda.synchronize()
    assert e0.elapsed_time(c1) < 0.5 * e0.elapsed_time(e1)
    third = pick_concurrent_stream(against=[main, side])
    assert third != main and third != side


@pytest.mark.parametrize('kind,over', [('karel', {}), ('vizdoom', {}),
                                       ('karel', dict(batch_size=32, k=10, max_demo_len=20, max_program_len=50,
                                                      num_lstm_cell_units=512))])
def test_token_decoders_project_the_table_not_the_rows(kind, over, monkeypatch):
    """The program and action decoders read embedding[id]: their input projection is a gather from the projected
    TABLE ([tok+2, 4U]) and their dWx / embedding gradients come from dz summed by token -- same loss, logits and
    gradients as projecting the gathered rows (D2P_TOKEN_PROJECTION=0), up to summation order."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case(kind, seed=53, **over)
    runs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('D2P_TOKEN_PROJECTION', flag)
        m = Model(cfg, params=params)
        assert m.token_projection == (flag == '1')
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        runs.append((loss, m._ctx['dp']['logits'].clone(), m._ctx['da']['logits'].clone(),
                     {n: t.clone() for n, t in m.params.g.items()}))
    (l1, p1, a1, g1), (l0, p0, a0, g0) = runs
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert (p1 - p0).abs().max().item() <= 2e-5 and (a1 - a0).abs().max().item() <= 2e-5
    for n in g0:
        scale = g0[n].abs().max().item() + 1e-12
        assert (g1[n] - g0[n]).abs().max().item() <= 5e-5 * scale, n


@pytest.mark.parametrize('kind,over', [('karel', {}), ('vizdoom', {}),
                                       ('karel', dict(batch_size=32, k=10, max_demo_len=20, max_program_len=50,
                                                      num_lstm_cell_units=512))])
def test_perception_decoder_factored_input_equals_row_wise_form(kind, over, monkeypatch):
    """pe = BN(per . W + b) is rows . H with rows = the 5 perception bits spread by demonstration index: the
    decoder's input projection, its weight gradient and the whole fc + batch-norm backward go through the 60
    columns of `per_rows` (d2p_per_affine_rows / d2p_per_fc_bn_bwd) -- same loss, logits and gradients as the
    row-wise form (D2P_PER_FACTORED=0), up to summation order; the fc bias in front of the batch norm gets an
    exact zero gradient where the row-wise form leaves round-off."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case(kind, seed=59, **over)
    runs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('D2P_PER_FACTORED', flag)
        m = Model(cfg, params=params)
        assert m.per_factored == (flag == '1')
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        runs.append((loss, m._ctx['dq']['logits'].clone(), {n: t.clone() for n, t in m.params.g.items()}))
    (l1, q1, g1), (l0, q0, g0) = runs
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert (q1 - q0).abs().max().item() <= 5e-5
    assert g1['per/fc/b'].abs().max().item() == 0.0
    ref = max(g0[n].abs().max().item() for n in ('per/fc/W', 'per/fc/gamma', 'per/fc/beta'))
    assert g0['per/fc/b'].abs().max().item() <= 1e-4 * ref              # round-off of an exact zero
    for n in g0:
        if n == 'per/fc/b':
            continue
        scale = g0[n].abs().max().item() + 1e-12
        assert (g1[n] - g0[n]).abs().max().item() <= 1e-4 * scale, n


@pytest.mark.gpu
def test_greedy_exact_match_on_1k_generated_programs():
    """north_star: "decoded program-token exact-match equal to the reference on a fixed 1k-example shard".
    32 batches x 32 generated Karel programs at BASELINE config 2's full size (k = 10, U = 512), weights after
    300 optimizer steps: the HIP greedy decoder against the fp64 oracle's (GreedyEmbeddingHelper semantics,
    models/model_full.py:424-435,513-523; evaler.py:444-449 scores these tokens).  Every row must be token- and
    length-exact unless the ORACLE's top-2 logit gap at the first differing step is below 1e-4 (an fp32 / fp64
    argmax tie; such rows are listed).  "Reference" = the CPU oracle: TF-1.3 cannot run here."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import exact_match_1k
    res = exact_match_1k.run(n_batches=32, train_steps=300, verbose=False)
    assert res['programs'] == 1024
    assert res['persistent_lstm_status'] == 0
    assert res['unexcused_mismatches'] == 0, res['mismatches']
    assert res['token_exact_rows'] + res['mismatches_excused_as_fp_ties'] == 1024
    assert res['token_exact_rows'] >= 1014                 # ties are rare: more than 1 % of them is a bug
    assert res['max_abs_logit_err_on_exact_rows'] <= 1e-4
    assert res['rows_that_emit_the_end_token'] > 512       # the trained decoder ends its programs: a non-trivial argmax
    assert res['action_token_exact_sequences'] >= res['action_sequences'] - 10


@pytest.mark.gpu
def test_greedy_exact_match_at_the_vizdoom_geometry():
    """The same exact-match check at BASELINE config 4's geometry (ViZDoom full model, k = 10, 80x80x3 frames, B = 32,
    U = 512): 4 synthetic batches = 128 programs and 1 280 action sequences, weights after 40 optimizer steps, HIP
    greedy decoders against the fp64 oracle's."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import exact_match_1k
    res = exact_match_1k.run(n_batches=4, train_steps=40, n_train_batches=4, verbose=False, preset='vizdoom')
    assert res['programs'] == 128
    assert res['persistent_lstm_status'] == 0
    assert res['unexcused_mismatches'] == 0, res['mismatches']
    assert res['token_exact_rows'] + res['mismatches_excused_as_fp_ties'] == 128
    assert res['max_abs_logit_err_on_exact_rows'] <= 1e-4
    assert res['action_token_exact_sequences'] >= res['action_sequences'] - 13      # (1 % may be fp argmax ties)


@pytest.mark.gpu
def test_step_guard_skips_and_reruns_a_failed_step(monkeypatch):
    """The guarded optimizer step: with the persistent kernels' status word set (injected, as a timed-out hand-off
    sets it) the clip + Adam kernel skips its update ON THE DEVICE -- parameters, moments and the batch-norm moving
    statistics stay as they were -- and the trainer then resets the word, re-runs exactly the skipped steps on
    the per-step kernels and ends where an undisturbed run ends."""
    from demo2program_amd import kernels as K
    from demo2program_amd.trainer import Trainer
    cfg, params, batch = small_case('karel', seed=41)
    batches = [batch, small_case('karel', seed=42)[2], small_case('karel', seed=43)[2]]

    def fresh():
        tr = Trainer(cfg, make_train_dir=False)
        tr.model.params.load(params)
        return tr, [tr.model.get_feed_dict(b) for b in batches]

    # undisturbed: 6 steps
    tr, feeds = fresh()
    for i in range(6):
        tr.train_step(feeds[i % 3])
    assert tr.settle() == 0
    want = tr.model.params.flat.clone()
    want_m = tr.model.params.m.clone()
    want_mov = {n: (a.clone(), b.clone()) for n, (a, b) in tr.model.moving.items()}
    # disturbed: the word is set before step 2; the device skips steps 2.. until the host notices
    tr, feeds = fresh()
    for i in range(2):
        tr.train_step(feeds[i % 3])
    torch.cuda.synchronize()
    before = tr.model.params.flat.clone()
    mov_before = {n: (a.clone(), b.clone()) for n, (a, b) in tr.model.moving.items()}
    K.lstm_persist_inject_error()
    tr.train_step(feeds[2])
    torch.cuda.synchronize()
    assert torch.equal(tr.model.params.flat, before)                     # skipped on the device
    assert tr.guard.counters.tolist() == [2, 1]
    for n in ('rn_h/fc1', 'rn_c/fc2'):                                    # downstream of the recurrences: untouched
        assert torch.equal(tr.model.moving[n][0], mov_before[n][0]) and torch.equal(tr.model.moving[n][1], mov_before[n][1])
    tr.train_step(feeds[0])          # detects (single rank: any arrived copy), re-runs step 2, then runs step 3
    for i in (4, 5):
        tr.train_step(feeds[i % 3])
    assert tr.settle() == 1
    assert K.lstm_is_persistent()                                         # one failure: persistent kernels back on
    assert tr.global_step == 6 and tr.adam_step == 6
    assert K.lstm_persist_error(reset=True) == 0
    # the re-run used the per-step kernels (forward bit-identical, backward equal to fp32 round-off)
    assert (tr.model.params.flat - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    assert (tr.model.params.m - want_m).abs().max().item() <= 1e-4 * want_m.abs().max().item() + 1e-9
    # every moving statistic saw each batch once: the conv layers' too, which had moved in the skipped step's forward pass
    # and were restored from the guard's snapshot before the re-run (round 4)
    for n in want_mov:
        for q in (0, 1):
            assert (tr.model.moving[n][q] - want_mov[n][q]).abs().max().item() <= 1e-5 * (1.0 + want_mov[n][q].abs().max().item()), n
    # run_single_step: detected in the same call, and the loss it reports is the re-run's
    class Src(object):
        def next(self):
            return batches[0]
    tr3, _ = fresh()
    clean_loss = tr3.run_single_step(Src())[2]
    tr4, _ = fresh()
    K.lstm_persist_inject_error()
    hurt_loss = tr4.run_single_step(Src())[2]
    assert tr4.guard.failures == 1 and tr4.global_step == 1
    assert abs(hurt_loss - clean_loss) <= 1e-6 * abs(clean_loss)
    assert (tr4.model.params.flat - tr3.model.params.flat).abs().max().item() <= 2e-5
    K.lstm_set_persistent(True)


@pytest.mark.gpu
def test_evaler_redoes_a_batch_on_the_per_step_kernels(tmp_path):
    """A persistent launch that gave up a hand-off during evaluation: the batch is redone on the per-step kernels
    and the reported numbers equal an undisturbed evaluation (the sticky word used to corrupt every
    later batch silently)."""
    from demo2program_amd import kernels as K
    from demo2program_amd.evaler import Evaler, GeneratedKarelBatches
    cfg, params, _ = small_case('karel', seed=47)
    cfg.checkpoint, cfg.train_dir, cfg.output_dir = '', '', str(tmp_path)
    cfg.max_steps, cfg.pred_program, cfg.quiet, cfg.write_summary = 3, False, True, False
    cfg.dataset_split, cfg.no_loss = 'test', False

    def run(inject):
        ev = Evaler(cfg, GeneratedKarelBatches(cfg, seed=5))
        ev.model.params.load(params)
        if inject:
            K.lstm_persist_inject_error()
        ev.eval_run()
        return ev
    try:
        clean = run(False)
        hurt = run(True)
        assert getattr(hurt, 'persist_fallbacks', 0) == 1
        assert K.lstm_persist_error(reset=True) == 0
        for kk, v in clean.final['loss'].items():
            assert abs(hurt.final['loss'][kk] - v) <= 1e-6 * max(1.0, abs(v))
        assert hurt.final['acc'] == clean.final['acc']
    finally:
        K.lstm_set_persistent(True)
        K.lstm_persist_error(reset=True)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without torch.distributed.run must start the N ranks itself (the driver's
    multi-GPU command may have either shape).  N = 1 through that path: a one-rank RCCL group, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--self-spawn', '--steps', '4',
                          '--warmup', '2', '--no-cpu-baseline', '--no-roofline', '--no-h2d', '--no-config4'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['value'] > 0
    assert d['rccl_ranks_seen'] == 1 and d['launcher'] == 'self-spawned'


@pytest.mark.gpu
def test_hand_built_feed_without_the_derived_perception_rows():
    """A feed assembled by hand (the tensors of the reference's placeholders only) goes through Model.forward /
    backward in eager mode like one from get_feed_dict: the factored perception decoder derives its row matrix and
    Gram matrix itself (it used to raise KeyError)."""
    from demo2program_amd.models.model_full import Model
    cfg, params, batch = small_case('karel', seed=51)
    m = Model(cfg, params=params)
    full = m.get_feed_dict(batch)
    l1 = float(m.forward(full).item())
    m.backward()
    g1 = m.params.grad.clone()
    bare = {k_: v for k_, v in full.items() if k_ not in ('per_rows', 'per_gram', '_flat')}
    l2 = float(m.forward(bare).item())
    m.backward()
    assert l1 == l2 and torch.equal(g1, m.params.grad)


def test_one_launch_state_encoder_equals_the_separate_launches(monkeypatch):
    """D2P_FUSED_ENCODER (d2p_karel_encoder_fwd: the three conv -> batch-norm layers of the Karel State_Encoder in one
    launch) against the 13 separate launches at the headline geometry: same loss and gradients to fp32 rounding of the
    batch statistics, same moving statistics."""
    from demo2program_amd import kernels as K
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel', batch_size=32, k=10, num_lstm_cell_units=128)
    if not K.karel_encoder_ok(32, 10, cfg.max_demo_len):
        pytest.skip('geometry not taken by the one-launch kernel on this device')
    batch = make_batch(cfg, seed=3)
    res = []
    m = Model(cfg, seed=5)
    mov0 = m.moving_flat.clone()
    for fused in ('0', '1'):
        m.fused_encoder = fused == '1'          # (what D2P_FUSED_ENCODER sets at construction)
        m.moving_flat.copy_(mov0)
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        torch.cuda.synchronize()
        assert K.lstm_persist_error() == 0
        res.append((loss, m.params.grad.clone(), m.moving_flat.clone()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[0][0])
    scale = float(res[0][1].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * scale
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('model_kind', ['full', 'summarizer', 'full_k25', 'full_k3', 'full_k28'])
def test_relation_networks_in_four_launches_equal_the_separate_launches(model_kind):
    """Round 5: the relation networks' pointwise chains around their two GEMMs as d2p_rn_fc1_fwd / _fc2_fwd / _fc2_bwd /
    _fc1_bwd (batch-norm sums from recomputed pair values, batch norm commuted with the mean over a program's pairs,
    closed-form sums in fc2's batch-norm backward, the pair backward from registers) against the 9 + 11 separate
    launches (rn_pool of models/model_full.py:333-349 under tf.gradients): same summaries, loss, gradients and moving
    statistics to fp32 rounding of the batch statistics; the summarizer baseline runs it without the avg-pool branch."""
    from demo2program_amd import kernels as K
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    kw = dict(batch_size=32, k=10, num_lstm_cell_units=128)
    if model_kind == 'summarizer':
        kw['model'] = 'summarizer'
    if model_kind == 'full_k25':          # (k not known at compile time, odd: the halves split 13 / 12; config 5's k)
        kw.update(batch_size=8, k=25)
    if model_kind == 'full_k3':
        kw.update(batch_size=5, k=3)
    if model_kind == 'full_k28':          # (rn_fc1_fwd asks for > 32 KB of LDS from k = 27 on: the residency bound of
        kw.update(batch_size=6, k=28)     #  rn_geom_ok counts workgroups per CU from the real request -- ADVICE round 5)
    cfg = make_config('karel', **kw)
    if not K.rn_ok(kw['batch_size'], kw['k'], 128):
        pytest.skip('geometry not taken by the four-launch form on this device')
    batch = make_batch(cfg, seed=13)
    m = Model(cfg, seed=7)
    mov0 = m.moving_flat.clone()
    feed = m.get_feed_dict(batch)
    res = []
    for fused in (False, True):
        m.fused_rn = fused
        m.moving_flat.copy_(mov0)
        for rep in range(2):                                   # (twice: the arrival tickets of a second generation)
            loss = float(m.forward(feed).item())
            m.backward()
        torch.cuda.synchronize()
        assert K.lstm_persist_error() == 0
        assert bool(m._ctx['rn_h'].get('fused')) == fused
        res.append((loss, m.params.grad.clone(), m.moving_flat.clone(), m._ctx['rn_h']['out'].clone()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[0][0])
    torch.testing.assert_close(res[0][3], res[1][3], rtol=1e-5, atol=1e-6)
    scale = float(res[0][1].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * scale
    # every relation-network gradient tensor on its own scale (the bound of the oracle parity tests)
    m.params.grad.copy_(res[0][1]); g_sep = m.params.to_numpy('g')
    m.params.grad.copy_(res[1][1]); g_fus = m.params.to_numpy('g')
    for n in g_sep:
        if n.startswith('rn_'):
            s_ = np.abs(g_sep[n]).max()
            assert np.abs(g_sep[n] - g_fus[n]).max() <= 2e-4 * s_ + 1e-6, n
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('kind', ['karel', 'vizdoom'])
def test_decoder_small_gradient_products_grouped_equal_the_separate_launches(kind):
    """Round 6: the six products behind the decoders' dz rows summed by token / perception column in one launch
    (d2p_small_pair_products), against the per-decoder GEMM launches: same gradients to fp32 summation order (the oracle
    parity tests run with the grouped form)."""
    from demo2program_amd import kernels as K
    from demo2program_amd.models.model_full import Model
    over = dict(num_lstm_cell_units=128)
    if kind == 'vizdoom':
        over.update(h=20, w=20)
    cfg, params, batch = small_case(kind, seed=71, **over)
    res = []
    for grouped in (False, True):
        m = Model(cfg, params=params)
        m.grouped_decoder_grads = grouped
        loss = float(m.forward(m.get_feed_dict(batch)).item())
        m.backward()
        torch.cuda.synchronize()
        res.append((loss, m.params.to_numpy('g')))
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        a, b = res[0][1][n], res[1][1][n]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max() + 1e-7, n
    changed = [n for n in res[0][1] if not np.array_equal(res[0][1][n], res[1][1][n])]
    assert any('embedding' in n or n.endswith('lstm/kernel') for n in changed), changed      # (the other path really ran)


def test_encoder_kernel_gradient_halves_as_one_product(monkeypatch):
    """Round 6: the second encoder's kernel gradient -- dWx = X^T dZ and dWh = H^T dZ, adjacent row blocks of one tensor, the
    same dZ rows through the same list (its states are staged behind their initial state) -- as ONE product [X | H]^T dZ
    (d2p_gemm_f32_tn_rows2, 128 x 64 tiles at I = U = 512): the same gradients bit for bit as the two products (the oracle
    parity tests run with the paired form).  (The first encoder's input half is 48 / 432 rows wide: two products.)  Likewise the
    action and the perception decoder's recurrent halves: two products of one shape in one launch (d2p_gemm_f32_tn_rows_x2)."""
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel', batch_size=8, k=10)                       # U = 512: 256 tiles of 128 x 64
    batch = make_batch(cfg, seed=13)
    from demo2program_amd import kernels as K
    res, calls = [], []
    orig, orig_x2 = K.gemm_tn_rows2, K.gemm_tn_rows_x2

    def counted(*a, **kw):
        calls[-1] += 1
        return orig(*a, **kw)

    def counted_x2(*a, **kw):
        calls[-1] += 10
        return orig_x2(*a, **kw)
    monkeypatch.setattr(K, 'gemm_tn_rows2', counted)
    monkeypatch.setattr(K, 'gemm_tn_rows_x2', counted_x2)
    for paired in (False, True):
        m = Model(cfg, seed=5)
        m.paired_kernel_grads = paired
        feed = m.get_feed_dict(batch)
        calls.append(0)
        loss = m.forward(feed, defer_loss=True)
        m.backward()
        torch.cuda.synchronize()
        res.append((float(loss.item()), m.params.grad.clone()))
    assert calls == [0, 11]          # (the paired forms really ran: the second encoder's halves, the two decoders' recurrent halves)
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


def test_training_step_decoders_skip_the_steps_past_a_rows_length():
    """Round 5: in a training step (deferred logits) the action / perception decoders' forward recurrences run
    length-sorted and do not compute a row past its own length -- nothing reads those outputs there (the loss and its
    gradient mask them: models/model_full.py:1018-1079; the weight gradients run over the rows inside their sequences).
    Same loss and gradients as the free-running recurrences up to fp32 summation order, hout zero past a row's length;
    forward() outside a training step keeps the reference's free-running outputs (BasicDecoder without
    impute_finished, models/model_full.py:465-471)."""
    from demo2program_amd import kernels as K
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel', batch_size=32, k=10, num_lstm_cell_units=128)
    batch = make_batch(cfg, seed=11)
    m = Model(cfg, seed=5)
    feed = m.get_feed_dict(batch)
    if feed.get('demo_slab_steps') is None or feed.get('loss_dens') is None:
        pytest.skip('this feed carries no row order / mask counts')
    res = []
    for skip in (False, True):
        m.decoder_skip_past_len = skip
        loss = m.forward(feed, defer_loss=True)
        m.backward()
        torch.cuda.synchronize()
        assert K.lstm_persist_error() == 0
        assert m._ctx.get('logits_deferred'), 'the deferred-logits path of a training step did not run'
        res.append((float(loss.item()), m.params.grad.clone(), m._ctx['da']['hout'].clone(), m._ctx['dq']['hout'].clone()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[0][0]), (res[0][0], res[1][0])
    scale = float(res[0][1].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * scale
    T, M = cfg.max_demo_len, cfg.batch_size * cfg.k
    lens = feed['demo_len'].view(-1).cpu()
    past = (torch.arange(T).view(T, 1) >= lens.view(1, M)).cuda()          # [T, M]: step t is past row m's length
    inside = ~past
    for i in (2, 3):
        free, skipped = res[0][i], res[1][i]
        assert float(skipped[past].abs().max()) == 0.0                         # zeros, as impute_finished would leave
        assert float(free[past].abs().max()) > 0.0                             # (the free-running cell keeps going)
        assert float((free[inside] - skipped[inside]).abs().max()) <= 1e-6
    # outside a training step: the free-running outputs, whatever the switch says
    m.decoder_skip_past_len = True
    m.forward(feed)
    assert float(m._ctx['da']['hout'][past].abs().max()) > 0.0


def test_one_launch_state_encoder_under_graph_replay():
    """the one-launch encoder's cross-workgroup barrier under hipGraph replay.  Its arrival counters were
    chosen per launch on the HOST (baked into the captured node: every replay after the first fell through the barrier
    and normalised with stale partial sums); they are monotonic device-side tickets now.  S = 16 slices per
    demonstration index, five replays of one graph against five eager steps: same losses, same moving statistics."""
    from demo2program_amd import kernels as K
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    cfg = make_config('karel', batch_size=32, k=10, num_lstm_cell_units=128)
    if not K.karel_encoder_ok(32, 10, cfg.max_demo_len):
        pytest.skip('geometry not taken by the one-launch kernel on this device')
    batches = [make_batch(cfg, seed=3), make_batch(cfg, seed=4)]
    res = []
    for use_graph in (False, True):
        tr = Trainer(cfg, make_train_dir=False, use_graph=use_graph)
        assert tr.model._fused_encoder(32, 10, cfg.max_demo_len)
        feeds = [tr.model.get_feed_dict(b) for b in batches]
        # (the same (n_prog, n_demo) key for both batches would be luck: replay ONE graph, feeding batch 0 throughout,
        #  after one step on batch 1 so that stale partial sums of another batch sit in the workspace)
        losses = [float(tr.train_step(feeds[1]).item())]
        for _ in range(5):
            losses.append(float(tr.train_step(feeds[0]).item()))
        assert tr.settle() == 0 and K.lstm_persist_error() == 0
        res.append((losses, tr.model.moving_flat.clone(), tr.model._bufs['conv1/bn_mean'].clone(),
                    tr.model._bufs['conv3/bn_rstd'].clone()))
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 1e-4 * abs(a), res
    # the conv layers' statistics (what the barrier protects) tightly; the statistics downstream of the recurrences follow
    # six optimizer steps whose weight gradients a captured step sums in another order (no row lists): loosely
    nconv = 2 * (16 + 32 + 48)
    torch.testing.assert_close(res[0][1][:nconv], res[1][1][:nconv], rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-2, atol=2e-4)
    for q in (2, 3):
        torch.testing.assert_close(res[0][q], res[1][q], rtol=2e-4, atol=1e-6)


def test_loss_value_from_the_loss_backward_launch(monkeypatch):
    """D2P_FUSED_LOSS: a training step's loss value comes from the per-workgroup sums the loss-backward launch leaves
    behind (d2p_xent_bwd_desc.loss_part + d2p_loss_from_partials), not from forward loss launches: the same value up to
    the order of an fp32 sum, the same gradients bit for bit."""
    from demo2program_amd.config import make_config
    from demo2program_amd.synthetic import make_batch
    from demo2program_amd.trainer import Trainer
    cfg = make_config('karel', batch_size=8, k=4, num_lstm_cell_units=128)
    batch = make_batch(cfg, seed=77)
    res = []
    for fused in ('0', '1'):
        monkeypatch.setenv('D2P_FUSED_LOSS', fused)
        tr = Trainer(cfg, make_train_dir=False)
        feed = tr.model.get_feed_dict(batch)
        loss = tr.train_step(feed)
        tr.settle()
        res.append((float(loss.item()), tr.model.params.grad.clone(), tr.model._buf('loss_terms', (3,)).clone()))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[0][0]), (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][2], res[1][2], rtol=2e-6, atol=1e-7)


def test_rows_past_a_sequence_are_selected_around_not_multiplied():
    """The encoders' input projections are computed over the rows inside their sequences only (d2p_gemm_f32_rows): the
    other rows of z keep whatever an earlier batch left there.  Poisoned with NaN, loss and every gradient must come out
    bit-identical -- the recurrences select around those rows, they do not multiply them by a mask."""
    from demo2program_amd.config import make_config
    from demo2program_amd.models.model_full import Model
    from demo2program_amd.synthetic import make_batch
    cfg = make_config('karel', batch_size=8, k=4, num_lstm_cell_units=128)
    m = Model(cfg, seed=21)
    feed = m.get_feed_dict(make_batch(cfg, seed=22))
    assert feed.get('n_active') is not None and 0 < feed['n_active'] < cfg.max_demo_len * cfg.batch_size * cfg.k
    loss0 = float(m.forward(feed).item())
    m.backward()
    torch.cuda.synchronize()
    g0 = m.params.grad.clone()
    poisoned = 0
    for name in ('demo_lstm/z', 'second_lstm/z', 'second_lstm/dx', 'demo_lstm/dx'):
        if name in m._bufs:
            m._bufs[name].fill_(float('nan'))
            poisoned += 1
    assert poisoned >= 2
    loss1 = float(m.forward(feed).item())
    m.backward()
    torch.cuda.synchronize()
    assert loss1 == loss0
    assert not torch.isnan(m.params.grad).any()
    assert torch.equal(m.params.grad, g0)
