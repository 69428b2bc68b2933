"""Host-side program metrics (models/model_full.py:602-616,713-729,745-780,878-901 restated in
demo2program_amd/models/program_metrics.py) on hand-built cases.  The parser / interpreter
underneath is pinned to the reference separately (tests/test_karel_dsl.py)."""
import json
import os

import numpy as np
import pytest

from demo2program_amd.karel_env import KarelVocab, Karel_world, parse
from demo2program_amd.models import program_metrics as PM

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'karel_dsl.json')))
WORLDS = [np.unpackbits(np.frombuffer(bytes.fromhex(h), dtype=np.uint8))[:8 * 8 * 16].reshape(8, 8, 16).astype(bool)
          for h in GOLD['worlds_hex']]
T, L = 20, 24


def demos_of(code, worlds):
    s_h = np.zeros((len(worlds), T, 8, 8, 16), np.float32)
    lens = np.zeros(len(worlds), np.int32)
    for d, w in enumerate(worlds):
        world = Karel_world(w.copy(), make_error=True)
        _, _, ok = parse(code).run(world)
        assert ok, (code, d)
        hist = np.stack(world.s_h)
        lens[d] = len(hist)
        s_h[d, :len(hist)] = hist
    return s_h, lens


def tokens_of(vocab, codes):
    tok = np.zeros((len(codes), L), np.int64)
    lens = np.zeros(len(codes), np.int64)
    for i, c in enumerate(codes):
        ids = vocab.str2intseq(c)
        tok[i, :len(ids)] = ids
        lens[i] = len(ids)
    return tok, lens


def test_program_metrics_end_to_end():
    vocab = KarelVocab()
    gt = 'DEF run m( turnLeft turnLeft turnRight m)'
    worlds = [WORLDS[0], WORLDS[2], WORLDS[4]]
    s_h1, len1 = demos_of(gt, worlds)
    preds = [
        gt,                                                                   # identical sequence
        'DEF run m( IFELSE c( frontIsClear c) i( turnLeft i) ELSE e( turnLeft e) turnLeft turnRight m)',  # same canonical form
        'DEF run m( turnLeft m)',                                             # same END state? no: one turn short
        'DEF run m( turnLeft turnLeft turnRight',                             # syntax error
        'DEF run m( turnRight turnLeft turnLeft m)',                          # different program, different trace
    ]
    B, k = len(preds), len(worlds)
    s_h = np.broadcast_to(s_h1, (B,) + s_h1.shape).copy()
    demo_len = np.broadcast_to(len1, (B, k)).copy()
    p_tok, p_len = tokens_of(vocab, preds)
    gt_tok, gt_len = tokens_of(vocab, [gt] * B)
    is_same = np.array([(p_len[i] == gt_len[i]) and np.array_equal(p_tok[i], gt_tok[i]) for i in range(B)], np.float32)
    assert is_same.tolist() == [1, 0, 0, 0, 0]

    syn = PM.check_correct_syntax(vocab, p_tok, p_len, is_same)
    assert syn.tolist() == [1, 1, 1, 0, 1]
    exact = PM.exact_program_compare(vocab, p_tok, p_len, syn, gt_tok, gt_len)
    assert exact.tolist() == [1, 1, 0, 0, 0]
    exe, exe_len = PM.generate_program_output(vocab, s_h[:, :, 0], T, p_tok, p_len, syn, is_same, make_error=True)
    assert exe.shape == (B, k, T, 8, 8, 16) and exe.dtype == np.float32
    assert exe_len[0].tolist() == [0] * k            # identical rows are not executed
    assert exe_len[1].tolist() == len1.tolist()
    assert exe_len[2].tolist() == [2] * k and exe_len[3].tolist() == [0] * k
    num, ok, hist = PM.compare_demo_and_execution(s_h, demo_len, exe, exe_len, is_same)
    assert num.tolist() == [k, k, 0, 0, 0]
    assert ok[0].all() and ok[1].all() and not ok[2:].any()
    assert hist.shape == (k + 1,) and abs(hist.sum() - 1.0) < 1e-6
    assert abs(hist[0] - 3 / 5) < 1e-6 and abs(hist[k] - 2 / 5) < 1e-6


def test_failed_and_overlong_executions():
    vocab = KarelVocab()
    # make_error: walking into the border fails the execution; without it Karel turns around
    code = 'DEF run m( ' + ' '.join(['move'] * 9) + ' m)'
    tok, ln = tokens_of(vocab, [code])
    init = WORLDS[1][None, None].astype(np.float32)
    ones, zeros = np.ones(1, np.float32), np.zeros(1, np.float32)
    _, l_err = PM.generate_program_output(vocab, init, T, tok, ln, ones, zeros, make_error=True)
    _, l_ok = PM.generate_program_output(vocab, init, T, tok, ln, ones, zeros, make_error=False)
    assert l_err[0, 0] == 0 and l_ok[0, 0] == 10
    # a trace longer than max_demo_len keeps its true length but only max_demo_len frames
    code = 'DEF run m( REPEAT R=19 r( turnLeft r) REPEAT R=5 r( turnLeft r) m)'
    tok2 = np.zeros((1, 40), np.int64)
    ids = vocab.str2intseq(code)
    tok2[0, :len(ids)] = ids
    exe, ln2 = PM.generate_program_output(vocab, init, T, tok2, np.array([len(ids)]), ones, zeros)
    assert ln2[0, 0] == 25 and exe.shape[2] == T and exe[0, 0, T - 1].any()


def test_vizdoom_metrics_are_refused():
    with pytest.raises(NotImplementedError):
        PM.require_env('vizdoom')
    PM.require_env('karel')
