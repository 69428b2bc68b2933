#!/opt/conda/bin/python3.9
"""Generates tests/golden/karel_dsl.json by RUNNING THE REFERENCE's own Karel DSL code.

Only runs in the build container (needs /root/reference and /opt/conda/bin/python3.9 with ply
and numpy 1.26); the committed JSON is what travels.  The reference modules are Python 2:
  * implicit relative imports  -> sys.path gets karel_env/ and karel_env/dsl/;
  * `np.bool`                  -> aliased;
  * `zip(...)[0]` in dsl_parse.py / dsl_enum_program.py -> those two files are exec'd from
    where they lie with a list-returning `zip` in their namespace (nothing is copied or edited).
Recorded per case: the token string, whether the stack parser accepts it (check_correct_syntax,
models/model_full.py:602-616), the canonical program of dsl_enum_program.parse
(exact_program_compare_karel, :713-729) and, for accepted programs, executions on generated
worlds exactly as generate_program_output_karel does (:745-780): success flag, call counter,
number of states and a SHA-1 over the packed state history (full histories for a few).
"""
import builtins
import hashlib
import json
import os
import sys
import types

import numpy as np

np.bool = bool
REF = '/root/reference'
sys.path[:0] = [REF, os.path.join(REF, 'karel_env'), os.path.join(REF, 'karel_env', 'dsl')]

from dsl import get_KarelDSL                     # noqa: E402  (reference, py2-style import)
import karel                                      # noqa: E402
from state_generator import KarelStateGenerator  # noqa: E402


def load_py2(path, name):
    mod = types.ModuleType(name)
    mod.__dict__['zip'] = lambda *a: list(builtins.zip(*a))
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), mod.__dict__)
    return mod


dsl_parse = load_py2(os.path.join(REF, 'karel_env', 'dsl', 'dsl_parse.py'), 'ref_dsl_parse')
dsl_enum = load_py2(os.path.join(REF, 'karel_env', 'dsl', 'dsl_enum_program.py'), 'ref_dsl_enum')


def pack(s):
    return np.packbits(np.asarray(s, dtype=np.uint8).reshape(-1)).tobytes()


def main():
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'karel_dsl.json')
    dsl = get_KarelDSL(dsl_type='prob', seed=123)
    rs = np.random.RandomState(7)
    vocab = list(dsl.int2token)

    programs = []
    seen = set()
    while len(programs) < 90:
        code = dsl.random_code(max_depth=int(rs.randint(2, 7)), max_nesting_depth=int(rs.randint(1, 5)))
        if code in seen or len(code.split()) > 50:
            continue
        seen.add(code)
        programs.append(code)
    hand = [
        'DEF run m( move m)', 'move', 'DEF run m( m)', 'DEF run m( move', 'm) move m( run DEF',
        'DEF run m( move move turnLeft m)',
        'DEF run m( REPEAT R=3 r( move r) m)', 'DEF run m( REPEAT R=0 r( move r) m)',
        'DEF run m( REPEAT R=19 r( turnLeft r) m)',
        'DEF run m( WHILE c( frontIsClear c) w( move w) m)',
        'DEF run m( WHILE c( not c( frontIsClear c) c) w( turnLeft w) m)',
        'DEF run m( WHILE c( noMarkersPresent c) w( turnLeft w) m)',
        'DEF run m( WHILE c( markersPresent c) w( pickMarker w) m)',
        'DEF run m( IF c( markersPresent c) i( pickMarker i) m)',
        'DEF run m( IF c( not c( not c( markersPresent c) c) c) i( pickMarker i) m)',
        'DEF run m( IFELSE c( frontIsClear c) i( move i) ELSE e( turnLeft e) m)',
        'DEF run m( IFELSE c( frontIsClear c) i( move i) ELSE e( move e) m)',
        'DEF run m( IFELSE c( not c( leftIsClear c) c) i( turnRight i) ELSE e( turnLeft move e) m)',
        'DEF run m( IFELSE c( noMarkersPresent c) i( putMarker i) ELSE e( pickMarker e) m)',
        'DEF run m( putMarker putMarker putMarker putMarker putMarker putMarker putMarker putMarker putMarker putMarker putMarker m)',
        'DEF run m( pickMarker m)', 'DEF run m( move move move move move move move move m)',
        'DEF run m( WHILE c( leftIsClear c) w( turnLeft IF c( frontIsClear c) i( move i) w) m)',
        'DEF run m( REPEAT R=4 r( WHILE c( rightIsClear c) w( turnRight w) move r) m)',
        'REPEAT R=2 r( move r)', 'frontIsClear', 'R=5', 'not c( frontIsClear c)',
        'IF c( frontIsClear c) i( move i)', 'DEF run m( move m) move', 'DEF DEF run m( move m)',
        'DEF run m( IF c( frontIsClear c) i( move i) ELSE e( move e) m)',
        'DEF run m( REPEAT r( move r) m)', 'DEF run m( WHILE c( move c) w( move w) m)',
        'move move', 'move turnLeft putMarker', 'c( frontIsClear c)',
    ]
    cases = [('random', p) for p in programs] + [('hand', p) for p in hand]
    # corruptions of generated programs: drop / duplicate / swap / truncate / random tokens
    for p in programs[:60]:
        t = p.split()
        kind = rs.randint(0, 5)
        if kind == 0 and len(t) > 1:
            del t[rs.randint(0, len(t))]
        elif kind == 1:
            i = rs.randint(0, len(t)); t.insert(i, t[i])
        elif kind == 2 and len(t) > 2:
            i, j = rs.randint(0, len(t), size=2); t[i], t[j] = t[j], t[i]
        elif kind == 3 and len(t) > 3:
            t = t[:rs.randint(1, len(t))]
        else:
            t[rs.randint(0, len(t))] = vocab[rs.randint(0, len(vocab))]
        cases.append(('corrupt', ' '.join(t)))
    for _ in range(30):
        n = rs.randint(1, 12)
        cases.append(('noise', ' '.join(vocab[i] for i in rs.randint(0, len(vocab), size=n))))

    s_gen = KarelStateGenerator(seed=11)
    worlds = [s_gen.generate_single_state(8, 8, 0.1)[0] for _ in range(6)]
    worlds += [s_gen.generate_single_state(8, 8, 0.35)[0] for _ in range(2)]
    # a world with a stack of markers under Karel (exercises the marker limits)
    w = worlds[0].copy()
    y, x, _ = np.argwhere(w[:, :, :4])[0]
    w[y, x, 5:] = False
    w[y, x, 5 + 8] = True
    worlds.append(w)

    records = []
    full_budget = 40
    for kind, code in cases:
        exe, ok = dsl_parse.parse(code)
        rec = {'kind': kind, 'code': code, 'syntax': bool(ok)}
        if ok:
            prog, ok2 = dsl_enum.parse(code)
            assert ok2
            if isinstance(prog, list):
                # WHILE expands 100x: keep a digest, and the list itself only when short
                rec['canonical_len'] = len(prog)
                rec['canonical_sha1'] = hashlib.sha1(' '.join(prog).encode()).hexdigest()
                if len(prog) <= 40:
                    rec['canonical'] = prog
            else:
                rec['canonical_len'] = -1        # top-level symbol is not `prog`: a function object
            runs = []
            for wi in (0, 1, 2, 3, 4, 5, 6, 7, 8) if kind in ('hand',) else tuple(rs.choice(9, size=3, replace=False)):
                for make_error in (True, False):
                    world = karel.Karel_world(worlds[int(wi)].copy(), make_error=make_error)
                    try:
                        k, n, s_run = exe(world, 0)
                        raised = None
                    except Exception as e:           # the reference would propagate this
                        n, s_run, raised = -1, False, type(e).__name__
                    hist = np.stack(world.s_h, axis=0)
                    r = {'world': int(wi), 'make_error': make_error, 'success': bool(s_run), 'n': int(n),
                         'len': int(hist.shape[0]), 'raised': raised,
                         'sha1': hashlib.sha1(pack(hist)).hexdigest(),
                         'actions': [int(a) for a in world.a_h]}
                    if full_budget > 0 and hist.shape[0] <= 12 and kind != 'noise':
                        r['s_h_hex'] = pack(hist).hex()
                        full_budget -= 1
                    runs.append(r)
            rec['runs'] = runs
        records.append(rec)

    # perception vectors on the worlds (front/left/right clear, markers present / absent)
    percepts = []
    for w in worlds:
        kw = karel.Karel_world(w.copy())
        percepts.append([bool(v) for v in kw.get_perception_vector()])

    doc = {
        'generator': 'tests/golden/make_karel_dsl_golden.py (reference karel_env code run under python3.9)',
        'vocab': vocab,
        'worlds_hex': [pack(w).hex() for w in worlds],
        'world_shape': [8, 8, 16],
        'percepts': percepts,
        'cases': records,
    }
    with open(out_path, 'w') as f:
        json.dump(doc, f, separators=(',', ':'))
    n_ok = sum(r['syntax'] for r in records)
    print('wrote %s: %d cases (%d accepted), %d runs' %
          (out_path, len(records), n_ok, sum(len(r.get('runs', [])) for r in records)))


if __name__ == '__main__':
    main()
