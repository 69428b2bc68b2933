"""Karel DSL / world against tests/golden/karel_dsl.json, which was produced by
running the reference's own parser, canonicaliser and interpreter (make_karel_dsl_golden.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from demo2program_amd.karel_env import KarelVocab, Karel_world, get_KarelDSL, parse

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'karel_dsl.json')))
WORLDS = [np.unpackbits(np.frombuffer(bytes.fromhex(h), dtype=np.uint8))[:8 * 8 * 16].reshape(8, 8, 16).astype(bool)
          for h in GOLD['worlds_hex']]


def pack(hist):
    return np.packbits(np.asarray(hist, dtype=np.uint8).reshape(-1)).tobytes()


def test_vocab_table():
    v = get_KarelDSL(dsl_type='prob', seed=123)
    assert v.int2token == GOLD['vocab'] and len(v.int2token) == 50
    assert v.token2int['m)'] == 3 and v.token2int['DEF'] == 0
    code = 'DEF run m( REPEAT R=3 r( move r) m)'
    assert v.intseq2str(v.str2intseq(code)) == code
    with pytest.raises(ValueError):
        get_KarelDSL(dsl_type='other')


def test_syntax_verdicts_match_reference():
    bad = [(c['code'], c['syntax'], parse(c['code']).ok) for c in GOLD['cases'] if parse(c['code']).ok != c['syntax']]
    assert not bad, bad[:5]
    assert sum(c['syntax'] for c in GOLD['cases']) > 100 and sum(not c['syntax'] for c in GOLD['cases']) > 50
    assert not parse('').ok
    assert parse('move').ok and parse('move').symbol == 'action'    # the stack machine's quirk


def test_canonical_programs_match_reference():
    n = 0
    for c in GOLD['cases']:
        if not c['syntax']:
            continue
        can = parse(c['code']).canonical()
        if c['canonical_len'] < 0:
            assert can is None, c['code']
            continue
        assert can is not None and len(can) == c['canonical_len'], c['code']
        assert hashlib.sha1(' '.join(can).encode()).hexdigest() == c['canonical_sha1'], c['code']
        if 'canonical' in c:
            assert can == c['canonical']
        n += 1
    assert n > 100
    a = parse('DEF run m( IFELSE c( frontIsClear c) i( move i) ELSE e( move e) m)').canonical()
    assert a == ['move']
    b = parse('DEF run m( IF c( not c( noMarkersPresent c) c) i( move i) m)').canonical()
    assert b == ['if', 'markersPresent', 'move']


def test_executions_match_reference():
    n_runs = n_fail = n_full = 0
    for c in GOLD['cases']:
        for r in c.get('runs', []):
            world = Karel_world(WORLDS[r['world']].copy(), make_error=r['make_error'])
            _, n, ok = parse(c['code']).run(world)
            tag = (c['code'], r['world'], r['make_error'])
            if r['raised']:
                assert not ok, tag         # the reference propagates an exception here
                continue
            assert ok == r['success'], tag
            assert n == r['n'], tag
            assert len(world.s_h) == r['len'], tag
            assert world.a_h == r['actions'], tag
            assert hashlib.sha1(pack(np.stack(world.s_h))).hexdigest() == r['sha1'], tag
            if 's_h_hex' in r:
                assert pack(np.stack(world.s_h)).hex() == r['s_h_hex']
                n_full += 1
            n_runs += 1
            n_fail += (not ok)
    assert n_runs > 1000 and n_fail > 100 and n_full > 20


def test_perception_vectors():
    for w, pv in zip(WORLDS, GOLD['percepts']):
        assert [bool(v) for v in Karel_world(w.copy()).get_perception_vector()] == pv


def test_world_error_modes():
    s = np.zeros((3, 3, 16), bool)
    s[:, :, 5] = True
    s[0, 0, 0] = True                                   # facing north on the top row
    w = Karel_world(s.copy(), make_error=True)
    with pytest.raises(RuntimeError):
        w.state_transition(0)
    assert len(w.s_h) == 1
    w = Karel_world(s.copy(), make_error=False)
    w.state_transition(0)                               # turns around instead
    assert w.get_location() == (0, 0, 2) and len(w.s_h) == 2
    with pytest.raises(RuntimeError):
        Karel_world(s.copy()).state_transition(3)       # nothing to pick up
    w = Karel_world(s.copy())
    for _ in range(9):
        w.state_transition(4)
    assert w.s[0, 0, 5 + 9] and w.marker_present()
    with pytest.raises(RuntimeError):
        w.state_transition(4)                           # a tenth marker


def test_state_generator_reproduces_reference_worlds():
    """KarelStateGenerator(seed=11): same RNG call order as karel_env/state_generator.py, so the
    first eight states equal the ones the reference generated for the fixture."""
    from demo2program_amd.karel_env.generator import KarelStateGenerator
    g = KarelStateGenerator(seed=11)
    mine = [g.generate_single_state(8, 8, 0.1)[0] for _ in range(6)]
    mine += [g.generate_single_state(8, 8, 0.35)[0] for _ in range(2)]
    for a, b in zip(mine, WORLDS):
        assert np.array_equal(a, b)


def test_sample_batch_layout():
    """dataset_karel.py:38-115 layout: one-hot program, zero padding past the lengths, <e> closing
    every action sequence, frames = the program's own execution."""
    from demo2program_amd.config import make_config
    from demo2program_amd.karel_env.generator import random_program, sample_batch
    rng = np.random.RandomState(3)
    for _ in range(100):
        p = parse(random_program(rng))
        assert p.ok and p.symbol == 'prog'
    cfg = make_config('karel_tiny')
    b = sample_batch(cfg, seed=9)
    B, k, T, A = cfg.batch_size, cfg.k, cfg.max_demo_len, cfg.action_space
    v = KarelVocab()
    for i in range(B):
        n = int(b['program_len'][i, 0])
        assert v.intseq2str(b['program_tokens'][i, :n]) == str(b['codes'][i])
        assert b['program'][i].sum() == n and not b['program'][i, :, n:].any()
        assert (b['program'][i].argmax(0)[:n] == b['program_tokens'][i, :n]).all()
        for pre, nd in (('', k), ('test_', cfg.test_k)):
            for d in range(nd):
                m = int(b[pre + 'demo_len'][i, d])
                assert 2 <= m <= T
                assert b[pre + 'a_h_tokens'][i, d, m - 1] == A - 1 and not b[pre + 'a_h_tokens'][i, d, m:].any()
                assert b[pre + 'a_h'][i, d].sum() == m and not b[pre + 's_h'][i, d, m:].any()
                w = Karel_world(b[pre + 's_h'][i, d, 0], make_error=True)
                _, _, ok = parse(str(b['codes'][i])).run(w)
                assert ok and len(w.s_h) == m
                assert np.array_equal(np.stack(w.s_h), b[pre + 's_h'][i, d, :m].astype(bool))
                assert np.array_equal(np.stack(w.p_v_h), b[pre + 'per'][i, d, :m].astype(bool))
