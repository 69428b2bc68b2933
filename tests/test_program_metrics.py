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


def test_vizdoom_execution_needs_a_world_factory():
    with pytest.raises(NotImplementedError):
        PM.require_env('vizdoom')
    PM.require_env('vizdoom', world_factory=lambda: None)
    PM.require_env('karel')


class _FrameWorld(object):
    """Tiny stand-in for the game: the frame is a counter image, ATTACK fails when no Demon is there."""

    def __init__(self):
        self.episodes = []

    def new_episode(self, init_dict):
        self.episodes.append(init_dict)
        self.demon = bool(init_dict['demon_pos'].size and int(np.atleast_2d(init_dict['demon_pos'])[0, 0]) > 0)
        self.s_h = [np.zeros((2, 2, 3))]

    def state_transition(self, action):
        if action == 'ATTACK' and not self.demon:
            raise RuntimeError('nothing to attack')
        self.s_h.append(self.s_h[-1] + 1)

    def is_there(self, actor):
        return self.demon and actor == 'Demon'

    def in_target(self, actor):
        return False

    def exist_actor_in_distance_horizontal(self, actor, dist, horz):
        return False


def test_vizdoom_program_metrics():
    from demo2program_amd.vizdoom_env import VizDoomDSLVocab
    vocab = VizDoomDSLVocab('simple', 'not_simple')
    parse = PM.parser_for('vizdoom')
    codes = ['DEF run m( IF c( ISTHERE Demon c) i( ATTACK i) MOVE_LEFT m)',     # runs everywhere
             'DEF run m( ATTACK m)',                                              # fails without a demon
             'DEF run m( ATTACK ATTACK',                                          # syntax error
             'DEF run m( MOVE_LEFT m)']                                           # identical to the ground truth
    gts = ['DEF run m( IFELSE c( ISTHERE Demon c) i( ATTACK MOVE_LEFT i) ELSE e( MOVE_LEFT e) m)',
           'DEF run m( REPEAT R=2 r( ATTACK r) m)', 'DEF run m( ATTACK m)', 'DEF run m( MOVE_LEFT m)']
    L = 24
    tok, ln = np.zeros((4, L), np.int64), np.zeros((4, 1), np.int64)
    gtok, gln = np.zeros((4, L), np.int64), np.zeros((4, 1), np.int64)
    for i, (c, g) in enumerate(zip(codes, gts)):
        ids, gids = vocab.str2intseq(c), vocab.str2intseq(g)
        tok[i, :len(ids)], ln[i] = ids, len(ids)
        gtok[i, :len(gids)], gln[i] = gids, len(gids)
    same = np.array([0, 0, 0, 1])
    syn = PM.check_correct_syntax(vocab, tok, ln, same, parse=parse)
    assert syn.tolist() == [1, 1, 0, 1]
    exact = PM.exact_program_compare(vocab, tok, ln, syn, gtok, gln, parse=parse)
    assert exact.tolist() == [0, 0, 0, 1]
    # 'IF c ATTACK; MOVE_LEFT' != 'IFELSE c (ATTACK MOVE_LEFT) else (MOVE_LEFT)' in canonical form, but
    # REPEAT unrolls: ATTACK ATTACK == REPEAT R=2 ATTACK
    two = vocab.str2intseq('DEF run m( ATTACK ATTACK m)')
    t2 = np.zeros((1, L), np.int64)
    t2[0, :len(two)] = two
    assert PM.exact_program_compare(vocab, t2, np.array([[len(two)]]), np.ones(1), gtok[1:2], gln[1:2],
                                    parse=parse).tolist() == [1]
    # execution: 2 demos, demon present in the first only
    init_pos = np.zeros((4, 2, 2, 3, 2), np.int32)
    init_pos[:, 0, 1, 0] = (5, 5)
    init_pos_len = np.ones((4, 2, 2), np.int32)
    world = _FrameWorld()
    exe, exe_len = PM.generate_program_output_vizdoom(vocab, lambda: world, init_pos, init_pos_len,
                                                      ['player_pos', 'demon_pos'], 5, 2, 2, 2, 3, tok, ln, syn, same)
    assert exe.shape == (4, 2, 5, 2, 2, 3)
    assert exe_len.tolist() == [[3, 2], [2, 0], [0, 0], [0, 0]]
    assert exe[0, 0, 2].max() == 2 and exe[0, 0, 3].max() == 0
    assert len(world.episodes) == 4 and world.episodes[0]['demon_pos'].tolist() == [5, 5]
    demo = exe.copy()
    num, ok, hist = PM.compare_demo_and_execution(demo, exe_len, exe, exe_len, same)
    assert num.tolist() == [2, 2, 2, 2]
