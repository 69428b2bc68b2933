"""Host-side logic that needs no GPU: config presets, synthetic batch contract, trainer CLI,
learning-rate schedule, and that the product refuses to run without its HIP path."""
import math

import numpy as np
import pytest
import torch

from demo2program_amd.config import conv_shapes, feature_dim, make_config
from demo2program_amd.synthetic import make_batch


def test_presets_match_baseline_configs():
    c2 = make_config('karel')
    assert (c2.batch_size, c2.k, c2.max_demo_len, c2.max_program_len) == (32, 10, 20, 50)
    assert (c2.h, c2.w, c2.depth, c2.dim_program_token, c2.action_space, c2.per_dim) == (8, 8, 16, 50, 6, 5)
    assert feature_dim(c2) == 48
    c4 = make_config('vizdoom')
    assert (c4.h, c4.w, c4.depth, c4.dim_program_token, c4.action_space, c4.per_dim) == (80, 80, 3, 42, 12, 6)
    assert feature_dim(c4) == 432
    assert [s[4] for s in conv_shapes(c4)] == [40, 20, 10, 5, 3]
    c5 = make_config('vizdoom_k25')
    assert (c5.batch_size, c5.k) == (16, 25)


def test_synthetic_batch_follows_the_reference_padding_rules():
    cfg = make_config('karel', batch_size=5, k=3)
    b = make_batch(cfg, seed=1)
    B, k, T, L, V, A, P = 5, 3, cfg.max_demo_len, cfg.max_program_len, 50, 6, 5
    assert b['s_h'].shape == (B, k, T, 8, 8, 16) and b['s_h'].dtype == np.float32
    assert b['program'].shape == (B, V, L) and b['program_tokens'].shape == (B, L)
    assert b['a_h'].shape == (B, k, T, A) and b['a_h_tokens'].dtype == np.int32
    assert b['program_len'].shape == (B, 1) and b['program_len'].dtype == np.float32
    assert b['demo_len'].shape == (B, k) and b['test_s_h'].shape[1] == cfg.test_k
    for bi in range(B):
        n = int(b['program_len'][bi, 0])
        assert b['program'][bi, :, :n].sum() == n and b['program'][bi, :, n:].sum() == 0
        assert list(b['program_tokens'][bi, :3]) == [0, 1, 2] and b['program_tokens'][bi, n - 1] == 3
        assert np.array_equal(b['program'][bi].argmax(0)[:n], b['program_tokens'][bi, :n])
        for i in range(k):
            m = int(b['demo_len'][bi, i])
            assert b['s_h'][bi, i, m:].sum() == 0 and b['s_h'][bi, i, :m].sum() > 0
            assert b['a_h_tokens'][bi, i, m - 1] == A - 1                 # <e> closes the demo
            assert b['a_h'][bi, i, :m].sum() == m and b['a_h'][bi, i, m:].sum() == 0
            assert b['per'][bi, i, m:].sum() == 0
    assert set(np.unique(b['s_h'])) <= {0.0, 1.0}
    again = make_batch(cfg, seed=1)
    assert all(np.array_equal(b[n], again[n]) for n in b)


def test_vizdoom_frames_are_unnormalised_bytes():
    cfg = make_config('vizdoom', batch_size=1, k=2, h=8, w=8)
    b = make_batch(cfg, seed=2)
    assert b['s_h'].max() > 200 and b['s_h'].min() == 0 and b['s_h'].dtype == np.float32


def test_cli_flags_and_defaults_match_the_reference():
    from demo2program_amd.trainer import build_arg_parser, learning_rate_at
    a = build_arg_parser().parse_args([])
    assert (a.model, a.dataset_type, a.num_k, a.batch_size, a.learning_rate) == ('full', 'karel', 10, 32, 0.001)
    assert (a.log_step, a.write_summary_step, a.test_sample_step) == (10, 100, 100)
    assert a.encoder_rnn_type == 'lstm' and a.num_lstm_cell_units == 512
    assert a.scheduled_sampling is False and a.scheduled_sampling_decay_steps == 20000
    assert a.demo_aggregation == 'avgpool' and a.lr_weight_decay is False
    with pytest.raises(SystemExit):
        build_arg_parser().parse_args(['--model', 'nope'])
    cfg = make_config('karel', lr_weight_decay=True)
    assert learning_rate_at(cfg, 9999) == 1e-3 and learning_rate_at(cfg, 10000) == 5e-4
    assert learning_rate_at(make_config('karel'), 50000) == 1e-3


def test_model_class_lookup_and_errors():
    from demo2program_amd.trainer import Trainer
    with pytest.raises(ValueError):
        Trainer.get_model_class('bogus')
    with pytest.raises(NotImplementedError):
        Trainer.get_model_class('induction_baseline')
    for name in ('full', 'summarizer', 'synthesis_baseline'):      # one graph, three variants
        assert Trainer.get_model_class(name).__name__ == 'Model'


def test_baseline_parameter_sets():
    from demo2program_amd.params import param_shapes
    import oracle
    from helpers import oracle_config
    full = set(param_shapes(make_config('karel')))
    summ = set(param_shapes(make_config('karel', model='summarizer')))
    synt = set(param_shapes(make_config('karel', model='synthesis_baseline')))
    assert synt < summ < full
    assert not any(n.startswith(('act/', 'per/')) for n in summ) and 'rn_h/fc1/W' in summ
    assert not any(n.startswith(('second_lstm/', 'rn_')) for n in synt) and 'demo_lstm/kernel' in synt
    for model in ('full', 'summarizer', 'synthesis_baseline'):
        cfg = make_config('karel_tiny', model=model)
        assert list(param_shapes(cfg).items()) == list(oracle.param_shapes(oracle_config(cfg)).items())


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_fails_loudly_without_a_gpu():
    from demo2program_amd import kernels
    from demo2program_amd.models.model_full import Model
    with pytest.raises(RuntimeError):
        Model(make_config('karel_tiny'))
    with pytest.raises(RuntimeError):
        kernels.matmul_nn(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_never_imports_the_oracle():
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'demo2program_amd')
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_bench_without_a_launcher_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus 8` (no torch.distributed.run around it, no WORLD_SIZE) must not die: it re-launches
    itself under torch.distributed.run with N ranks on 127.0.0.1 and passes its flags through."""
    import argparse
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from demo2program_amd import build
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(build, 'build_library', lambda *a, **k: None)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.self_spawn(argparse.Namespace(gpus=8))
    assert e.value.code == 7                                  # the launcher's status is the bench's
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'D2P_FORCE_DIST' not in seen['env']


def test_scalar_summaries_are_tensorboard_event_files(tmp_path):
    """demo2program_amd/summary.py: TFRecord-framed Event messages with masked CRC-32C (the format of
    tf.summary.FileWriter, trainer.py:116): the CRC's published check value, the framing of a known record, and a
    round trip of scalar values under the reference's tags (models/model_full.py:1144-1173)."""
    import struct
    from demo2program_amd import summary as S
    assert S.crc32c(b'123456789') == 0xE3069283                       # CRC-32C check value (RFC 3720, B.4)
    assert S.crc32c(bytes(32)) == 0x8A9136AA                          # 32 zero bytes (RFC 3720, B.4)
    rec = S.frame_record(b'abc')
    assert rec[:8] == struct.pack('<Q', 3) and rec[12:15] == b'abc' and len(rec) == 8 + 4 + 3 + 4
    w = S.SummaryWriter(str(tmp_path))
    w.add_scalars({'loss/loss': 1.5, 'loss/program_token_acc': 0.25}, 7)
    w.add_scalars({'test_loss/greedy_program_seq_acc': 0.125}, 300)
    w.close()
    ev = S.read_events(w.path)
    assert ev == [(7, {'loss/loss': 1.5, 'loss/program_token_acc': 0.25}), (300, {'test_loss/greedy_program_seq_acc': 0.125})]
    # the first record is the version header TensorBoard looks for
    with open(w.path, 'rb') as f:
        head = f.read(64)
    assert b'brain.Event:2' in head


def test_every_environment_switch_is_declared():
    """demo2program_amd/options.py: the product reads no D2P_* environment name that is not in its table, every A/B
    switch names a test that exists, and no switch leaves work out of a step."""
    import os
    import re
    from demo2program_amd import options
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for dirpath, _, files in os.walk(os.path.join(root, 'demo2program_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                if f.endswith('.py'):
                    seen |= set(re.findall(r"environ[^\n]*?['\"](D2P_[A-Z0-9_]+)['\"]", text))
                    seen |= set(re.findall(r"flag\(['\"](D2P_[A-Z0-9_]+)['\"]\)", text))
                else:
                    seen |= set(re.findall(r'getenv\("(D2P_[A-Z0-9_]+)"\)', text))
    assert seen <= set(options.SWITCHES), sorted(seen - set(options.SWITCHES))
    assert 'D2P_ABLATE' not in seen
    for name, (default, _, test) in options.SWITCHES.items():
        if default is None:
            continue
        path, _, fn = test.partition('::')
        text = open(os.path.join(root, path)).read()
        assert ('def %s(' % fn) in text, (name, test)
        assert name in text or fn in ('test_bench_starts_its_own_ranks',), (name, test)
    # INTEGRATION.md names exactly the declared switches (it listed six deleted ones)
    doc = set(re.findall(r'D2P_[A-Z][A-Z0-9_]+', open(os.path.join(root, 'INTEGRATION.md')).read()))
    doc -= {n for n in doc if n.startswith('D2P_E')}            # the C ABI's error codes (D2P_EINVAL, ...)
    assert doc == set(options.SWITCHES), (sorted(doc - set(options.SWITCHES)), sorted(set(options.SWITCHES) - doc))


def test_trainer_and_evaler_refuse_an_ablated_model():
    """Model.set_ablation is a timing hook of tools/step_ablation.sh: Trainer.train and Evaler must refuse such a model."""
    from demo2program_amd.evaler import Evaler
    from demo2program_amd.trainer import Trainer

    class M(object):
        _ablate = frozenset(['conv_fwd'])

    tr = Trainer.__new__(Trainer)
    tr.model = M()
    with pytest.raises(RuntimeError, match='ablation'):
        tr.train(max_steps=1)
    ev = Evaler.__new__(Evaler)
    ev.model = M()

    class B(object):
        def next(self):
            return {}
    with pytest.raises(RuntimeError, match='ablation'):
        ev.run_single_step(B())
