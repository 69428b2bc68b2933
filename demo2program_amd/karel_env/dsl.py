"""Karel DSL: vocabulary, stack parser, executor and canonical form.

What the model's metrics call (models/model_full.py):
  * check_correct_syntax (:602-616)      -> parse(code).ok
  * exact_program_compare_karel (:713-729)-> parse(code).canonical() equality
  * generate_program_output_karel (:745-780) -> parse(code).run(world)

The acceptance set is that of the reference's *shift-reduce stack machine*
(karel_env/dsl/dsl_parse.py:3-13,254-265), not of the yacc grammar: after every shift or
reduction the rule list is scanned in a fixed order and the first rule whose right-hand side
equals the top of the stack fires; a string is accepted when the input is exhausted and ONE
symbol -- of any kind -- is left.  So `move` alone is "correct syntax" (it reduces to a `stmt`),
and predicted token soup is judged exactly as the reference judges it.  Here the machine
builds a small AST; execution and canonicalisation are separate passes over it.

Execution semantics (dsl_parse.py:24-251): every node receives the call counter n and fails
once n > 100; wrappers (`prog`, every reduction to `stmt`, the first half of a sequence, `IF` /
`IFELSE` conditions, `REPEAT`) add one, actions and `WHILE` iterations do not -- the statement
wrapper around each body is what bounds loops.  An action that raises in the world is a
failure, not an exception.

Canonical form (dsl_enum_program.py:22-222): a flat token list -- sequences concatenate, REPEAT
unrolls, WHILE unrolls 100 times as IFs, IFELSE with identical branches collapses to the
branch, double negation cancels, noMarkersPresent = not markersPresent.
"""
import numpy as np

MAX_FUNC_CALL = 100
MAX_WHILE = 100

_TOKENS = (
    ['DEF', 'run', 'm(', 'm)', 'move', 'turnRight', 'turnLeft', 'pickMarker', 'putMarker', 'r(', 'r)'] +
    ['R=%d' % i for i in range(20)] +
    ['REPEAT', 'c(', 'c)', 'i(', 'i)', 'e(', 'e)', 'IF', 'IFELSE', 'ELSE',
     'frontIsClear', 'leftIsClear', 'rightIsClear', 'markersPresent', 'noMarkersPresent', 'not',
     'w(', 'w)', 'WHILE'])


class KarelVocab(object):
    """int <-> token maps in the reference's order (karel_env/dsl/dsl_prob.py:13-28 tokens,
    dsl_base.py:49-60 construct_vocab): 50 entries, 'm)' = 3."""

    def __init__(self, seed=None):
        self.int2token = list(_TOKENS)
        self.token2int = {t: i for i, t in enumerate(self.int2token)}
        self.rng = np.random.RandomState(seed)

    def str2intseq(self, code):
        return [self.token2int[t] for t in code.split()]

    code2intseq = str2intseq

    def intseq2str(self, intseq):
        return ' '.join(self.int2token[int(i)] for i in intseq)


def get_KarelDSL(dsl_type='prob', seed=None):
    if dsl_type != 'prob':
        raise ValueError('Undefined dsl type')
    return KarelVocab(seed=seed)


# ---------------------------------------------------------------------------------------------
# stack machine.  A rule = (right-hand side symbols, left-hand symbol, builder(values) -> node).
# Nodes are tuples: ('prog', s) ('stmt', inner) ('seq', a, b) ('if', c, s) ('ifelse', c, a, b)
# ('while', c, s) ('repeat', count, s) ('cond', inner) ('not', c) ('percept', name)
# ('action', index) ('cste', value).
# ---------------------------------------------------------------------------------------------
_ACTION_INDEX = {'move': 0, 'turnLeft': 1, 'turnRight': 2, 'pickMarker': 3, 'putMarker': 4}
_PERCEPTS = ('frontIsClear', 'leftIsClear', 'rightIsClear', 'markersPresent', 'noMarkersPresent')


def _rules():
    r = [(('DEF', 'run', 'm(', 'stmt', 'm)'), 'prog', lambda v: ('prog', v[3]))]
    for sym in ('while_stmt', 'repeat_stmt', 'stmt_stmt', 'action', 'if_stmt', 'ifelse_stmt'):
        r.append(((sym,), 'stmt', lambda v: ('stmt', v[0])))
    r.append((('stmt', 'stmt'), 'stmt_stmt', lambda v: ('seq', v[0], v[1])))
    r.append((('IF', 'c(', 'cond', 'c)', 'i(', 'stmt', 'i)'), 'if_stmt', lambda v: ('if', v[2], v[5])))
    r.append((('IFELSE', 'c(', 'cond', 'c)', 'i(', 'stmt', 'i)', 'ELSE', 'e(', 'stmt', 'e)'), 'ifelse_stmt',
              lambda v: ('ifelse', v[2], v[5], v[9])))
    r.append((('WHILE', 'c(', 'cond', 'c)', 'w(', 'stmt', 'w)'), 'while_stmt', lambda v: ('while', v[2], v[5])))
    r.append((('REPEAT', 'cste', 'r(', 'stmt', 'r)'), 'repeat_stmt', lambda v: ('repeat', v[1][1], v[3])))
    r.append((('cond_without_not',), 'cond', lambda v: ('cond', v[0])))
    r.append((('not', 'c(', 'cond', 'c)'), 'cond', lambda v: ('not', v[2])))
    for name in _PERCEPTS:
        r.append(((name,), 'cond_without_not', lambda v, name=name: ('percept', name)))
    for name in ('move', 'turnLeft', 'turnRight', 'pickMarker', 'putMarker'):
        r.append(((name,), 'action', lambda v, name=name: ('action', _ACTION_INDEX[name])))
    for i in range(20):
        r.append((('R=%d' % i,), 'cste', lambda v, i=i: ('cste', i)))
    return r


_RULES = _rules()


def _reduce_once(symbols, values):
    for rhs, lhs, build in _RULES:
        n = len(rhs)
        if len(symbols) >= n and tuple(symbols[-n:]) == rhs:
            node = build(values[-n:])
            del symbols[-n:], values[-n:]
            symbols.append(lhs)
            values.append(node)
            return True
    return False


class _Fail(Exception):
    pass


class Program(object):
    """Result of parse(): `.ok`, and for accepted strings `.symbol` (what the stack reduced to),
    `.run(world)` and `.canonical()`."""

    def __init__(self, ok, symbol=None, node=None):
        self.ok, self.symbol, self.node = ok, symbol, node

    # ---------------------------------------------------------------- execution
    def run(self, world, n=0):
        """-> (world, n, success), as the reference's compiled closure `exe(world, 0)`.
        Top-level symbols that are not statements (a bare condition or constant; the reference
        raises a TypeError/ValueError out of the py_func there) report failure."""
        if not self.ok or self.symbol not in ('prog', 'stmt', 'stmt_stmt', 'action', 'if_stmt', 'ifelse_stmt',
                                              'while_stmt', 'repeat_stmt'):
            return world, n, False
        n, ok = _exec(self.node, world, n)
        return world, n, ok

    # ---------------------------------------------------------------- canonical form
    def canonical(self):
        """Flat token list for programs that reduced to `prog`, else None (the reference compares
        function objects in that case, i.e. never equal)."""
        if not self.ok or self.symbol != 'prog':
            return None
        return _canon(self.node[1])


def _exec_cond(node, k, n):
    """-> (n, success, value)."""
    kind = node[0]
    if n > MAX_FUNC_CALL:
        return n, False, False
    if kind == 'cond':
        return _exec_cond(node[1], k, n)
    if kind == 'not':
        n, ok, c = _exec_cond(node[1], k, n)
        return n, ok, not c
    name = node[1]
    if name == 'frontIsClear':
        c = k.front_is_clear()
    elif name == 'leftIsClear':
        c = k.left_is_clear()
    elif name == 'rightIsClear':
        c = k.right_is_clear()
    elif name == 'markersPresent':
        c = k.marker_present()
    else:
        c = k.no_marker_present()
    return n, True, bool(c)


def _exec(node, k, n):
    """-> (n, success)."""
    kind = node[0]
    if n > MAX_FUNC_CALL:
        return n, False
    if kind in ('prog', 'stmt'):
        return _exec(node[1], k, n + 1)
    if kind == 'seq':
        n, ok = _exec(node[1], k, n + 1)
        if not ok:
            return n, False
        if n > MAX_FUNC_CALL:
            return n, False
        return _exec(node[2], k, n)
    if kind == 'if':
        n, ok, c = _exec_cond(node[1], k, n + 1)
        if not ok:
            return n, False
        return _exec(node[2], k, n) if c else (n, True)
    if kind == 'ifelse':
        n, ok, c = _exec_cond(node[1], k, n + 1)
        if not ok:
            return n, False
        return _exec(node[2] if c else node[3], k, n)
    if kind == 'while':
        n, ok, c = _exec_cond(node[1], k, n)
        if not ok:
            return n, False
        while c:
            n, ok = _exec(node[2], k, n)
            if not ok:
                return n, False
            n, ok, c = _exec_cond(node[1], k, n)
            if not ok:
                return n, False
        return n, True
    if kind == 'repeat':
        n += 1
        for _ in range(node[1]):
            n, ok = _exec(node[2], k, n)
            if not ok:
                return n, False
        return n, True
    if kind == 'action':
        try:
            k.state_transition(node[1])
        except Exception:
            return n, False
        return n, True
    raise ValueError('not a statement node: %r' % (kind,))


def _canon_cond(node):
    kind = node[0]
    if kind == 'cond':
        return _canon_cond(node[1])
    if kind == 'not':
        inner = _canon_cond(node[1])
        return inner[1:] if inner[0] == 'not' else ['not'] + inner
    return ['not', 'markersPresent'] if node[1] == 'noMarkersPresent' else [node[1]]


_ACTION_NAMES = ('move', 'turnLeft', 'turnRight', 'pickMarker', 'putMarker')


def _canon(node):
    kind = node[0]
    if kind == 'stmt':
        return _canon(node[1])
    if kind == 'seq':
        return _canon(node[1]) + _canon(node[2])
    if kind == 'if':
        return ['if'] + _canon_cond(node[1]) + _canon(node[2])
    if kind == 'ifelse':
        a, b = _canon(node[2]), _canon(node[3])
        if a == b:
            return a
        c = _canon_cond(node[1])
        neg = c[1:] if c[0] == 'not' else ['not'] + c
        return ['if'] + c + a + ['if'] + neg + b
    if kind == 'while':
        unit = ['if'] + _canon_cond(node[1]) + _canon(node[2])
        return unit * MAX_WHILE
    if kind == 'repeat':
        return _canon(node[2]) * node[1]
    if kind == 'action':
        return [_ACTION_NAMES[node[1]]]
    raise ValueError('not a statement node: %r' % (kind,))


def parse(code):
    """Shift-reduce the token string.  Returns a Program; `.ok` is the reference's second return
    value.  An empty string (the reference would raise IndexError) is a syntax error."""
    pending = code.split()[::-1]
    symbols, values = [], []
    if not pending:
        return Program(False)
    reduced = False
    while pending or len(symbols) != 1:
        if reduced:
            reduced = False
        else:
            tok = pending.pop()
            symbols.append(tok)
            values.append(tok)
        reduced = _reduce_once(symbols, values)
        if not reduced and not pending:
            return Program(False)
    return Program(True, symbols[0], values[0])
