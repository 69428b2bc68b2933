"""ViZDoom DSL: vocabulary, stack parser, executor and canonical form.

What the model's metrics call for dataset_type='vizdoom' (models/model_full.py):
  * the token tables behind intseq2str / the 'm)' end token (:74-77)  -> VizDoomDSLVocab
  * check_correct_syntax (:602-616)                                   -> parse(code).ok
  * exact_program_compare_vizdoom (:730-745)                          -> parse(code).canonical()
  * generate_program_output_vizdoom (:789-848)                        -> parse(code).run(world)

Same shift-reduce machine as the Karel one (vizdoom_env/dsl/dsl_parse.py:42-53,291-304): after
every shift or reduction the rule list is scanned in its fixed order, the first rule whose
right-hand side equals the top of the stack fires, and a string is accepted when the input is
exhausted and ONE symbol of any kind is left.  The machine here builds a small AST; execution
(dsl_parse.py:66-288) and canonicalisation (dsl_enum_program.py:66-270) are passes over it.

The world is duck-typed: `run` only calls `state_transition(action_name)`,
`exist_actor_in_distance_horizontal(actor, dist, horz)`, `in_target(actor)`, `is_there(actor)`
-- the methods of vizdoom_env/vizdoom_env.py:115,286-320.  The game engine itself is not part of
this build; anything with those four methods executes.

Execution differences from Karel that are kept: an action does NOT fail on an exhausted call
counter (its `n > MAX_FUNC_CALL` test has no return, dsl_parse.py:267-268) -- the statement
wrapper around it does; percepts and both condition wrappers test the counter.  Canonical
form: WHILE unrolls 1000 times (dsl_enum_program.py:62), percepts flatten to
[method, actor(, distance, horizontal)].
"""

MAX_FUNC_CALL = 100
MAX_WHILE = 1000

MONSTER_LIST = ['Demon', 'HellKnight', 'Revenant']
ITEMS_IN_INTEREST = ['MyAmmo']
ACTION_LIST = ['MOVE_FORWARD', 'MOVE_BACKWARD', 'MOVE_LEFT', 'MOVE_RIGHT', 'TURN_LEFT', 'TURN_RIGHT', 'ATTACK',
               'SELECT_WEAPON1', 'SELECT_WEAPON2', 'SELECT_WEAPON3', 'SELECT_WEAPON4', 'SELECT_WEAPON5']
# distance / horizontal words any perception type knows (dsl_parse.py:13-40; the parser accepts
# the union)
DISTANCE_WORDS = ('doncare_dist', 'far', 'mid_far', 'mid', 'close', 'very_close')
HORIZONTAL_WORDS = ('doncare_horz', 'center', 'slight_left', 'slight_right', 'mid_left', 'mid_right',
                    'left', 'right')

_STRUCTURE = ['DEF', 'run', 'm(', 'm)', 'WHILE', 'c(', 'c)', 'w(', 'w)']
_REPEAT = ['REPEAT', 'r(', 'r)', 'R=2', 'R=3', 'R=4', 'R=5', 'R=6']
_BRANCH = ['IF', 'i(', 'i)', 'IFELSE', 'ELSE', 'e(', 'e)', 'not']
_ACTIONS_NO_WEAPON2 = [a for a in ACTION_LIST if a != 'SELECT_WEAPON2']


class VizDoomDSLVocab(object):
    """int <-> token maps of vizdoom_env/dsl/vocab.py:17-82 for the perception types whose table
    is fully determined by that file: 'simple' (what the dataset generator writes,
    vizdoom_env/generator.py:283), 'more_simple', and level='simple'.  For 'clear' and the
    default type the reference appends `dict.keys()` of its distance / horizontal tables, whose
    order is Python 2's hash order -- not reproducible here, so those raise."""

    def __init__(self, perception_type='clear', level='not_simple'):
        if perception_type not in ('simple', 'more_simple'):
            raise NotImplementedError(
                "VizDoomDSLVocab(perception_type=%r): the reference's token order for this type is "
                "Python 2's dict order of its distance/horizontal tables (vocab.py:19-27); only "
                "'simple' / 'more_simple' vocabularies are built" % (perception_type,))
        if level == 'simple':
            actions = ACTION_LIST[:7]
            tokens = _STRUCTURE + _BRANCH + ['EXIST', 'IN', 'INTARGET']
        elif perception_type == 'simple':
            actions = _ACTIONS_NO_WEAPON2
            tokens = _STRUCTURE + _REPEAT + _BRANCH + ['INTARGET', 'ISTHERE']
        else:
            actions = _ACTIONS_NO_WEAPON2
            tokens = _STRUCTURE + _REPEAT + _BRANCH + ['ISTHERE']
        self.int2token = tokens + actions + MONSTER_LIST + ITEMS_IN_INTEREST
        self.token2int = {t: i for i, t in enumerate(self.int2token)}
        self.action_int2token = list(actions)
        self.action_token2int = {t: i for i, t in enumerate(self.action_int2token)}

    def str2intseq(self, string):
        return [self.token2int[t] for t in string.split()]

    def strlist2intseq(self, strlist):
        return [self.token2int[t] for t in strlist]

    def intseq2str(self, intseq):
        return ' '.join(self.int2token[int(i)] for i in intseq)

    def token_dim(self):
        return len(self.int2token)

    def action_str2intseq(self, string):
        return [self.action_token2int[t] for t in string.split()]

    def action_strlist2intseq(self, strlist):
        return [self.action_token2int[t] for t in strlist]

    def action_intseq2str(self, intseq):
        return ' '.join(self.action_int2token[int(i)] for i in intseq)

    def action_token_dim(self):
        return len(self.action_int2token)


# ---------------------------------------------------------------------------------------------
# stack machine.  A rule = (right-hand side symbols, left-hand symbol, builder(values) -> node).
# Statement nodes: ('prog', s) ('stmt', inner) ('seq', a, b) ('if', c, s) ('ifelse', c, a, b)
# ('while', c, s) ('repeat', count, s) ('action', name).  Condition nodes: ('cond', inner)
# ('not', c) ('exist', actor, dist, horz) ('intarget', actor) ('isthere', actor).  Leaves keep
# their word: ('word', text) / ('cste', value) / ('slot', value).
# ---------------------------------------------------------------------------------------------
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
    r.append((('percept',), 'cond', lambda v: ('cond', v[0])))
    r.append((('not', 'c(', 'cond', 'c)'), 'cond', lambda v: ('not', v[2])))
    r.append((('EXIST', 'actor', 'IN', 'distance', 'horizontal'), 'percept',
              lambda v: ('exist', v[1][1], v[3][1], v[4][1])))
    r.append((('INTARGET', 'actor'), 'percept', lambda v: ('intarget', v[1][1])))
    r.append((('ISTHERE', 'actor'), 'percept', lambda v: ('isthere', v[1][1])))
    r.append((('monster',), 'actor', lambda v: v[0]))
    for name in MONSTER_LIST:
        r.append(((name,), 'monster', lambda v, name=name: ('word', name)))
    r.append((('items',), 'actor', lambda v: v[0]))
    for name in ITEMS_IN_INTEREST:
        r.append(((name,), 'items', lambda v, name=name: ('word', name)))
    for name in DISTANCE_WORDS:
        r.append(((name,), 'distance', lambda v, name=name: ('word', name)))
    for name in HORIZONTAL_WORDS:
        r.append(((name,), 'horizontal', lambda v, name=name: ('word', name)))
    for i in range(1, 7):
        r.append((('S=%d' % i,), 'slot', lambda v, i=i: ('slot', i)))
    for name in ACTION_LIST:
        r.append(((name,), 'action', lambda v, name=name: ('action', name)))
    for i in range(20):
        r.append((('R=%d' % i,), 'cste', lambda v, i=i: ('cste', i)))
    return r


_RULES = _rules()
_STATEMENT_SYMBOLS = ('prog', 'stmt', 'stmt_stmt', 'action', 'if_stmt', 'ifelse_stmt', 'while_stmt', 'repeat_stmt')


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


class Program(object):
    """Result of parse(): `.ok`, and for accepted strings `.symbol` (what the stack reduced to),
    `.run(world)` and `.canonical()`."""

    def __init__(self, ok, symbol=None, node=None):
        self.ok, self.symbol, self.node = ok, symbol, node

    def run(self, world, n=0):
        """-> (world, n, success), as the reference's compiled closure `exe(world, 0)`.  Roots
        that are not statements (a bare condition, actor, constant ...; the reference's caller
        fails to unpack their result there) report failure."""
        if not self.ok or self.symbol not in _STATEMENT_SYMBOLS:
            return world, n, False
        n, ok = _exec(self.node, world, n)
        return world, n, ok

    def canonical(self):
        """Flat token list for programs that reduced to `prog`, else None (the reference compares
        closures in that case, i.e. never equal)."""
        if not self.ok or self.symbol != 'prog':
            return None
        return _canon(self.node[1])


def _exec_cond(node, world, n):
    """-> (n, success, value)."""
    if n > MAX_FUNC_CALL:
        return n, False, False
    kind = node[0]
    if kind == 'cond':
        return _exec_cond(node[1], world, n)
    if kind == 'not':
        n, ok, c = _exec_cond(node[1], world, n)
        return n, ok, not c
    if kind == 'exist':
        return n, True, world.exist_actor_in_distance_horizontal(node[1], node[2], node[3])
    if kind == 'intarget':
        return n, True, world.in_target(node[1])
    if kind == 'isthere':
        return n, True, world.is_there(node[1])
    raise ValueError('not a condition node: %r' % (kind,))


def _exec(node, world, n):
    """-> (n, success)."""
    kind = node[0]
    if kind == 'action':                       # no counter test (see the module docstring)
        try:
            world.state_transition(node[1])
        except Exception:
            return n, False
        return n, True
    if n > MAX_FUNC_CALL:
        return n, False
    if kind in ('prog', 'stmt'):
        return _exec(node[1], world, n + 1)
    if kind == 'seq':
        n, ok = _exec(node[1], world, n + 1)
        if not ok:
            return n, False
        if n > MAX_FUNC_CALL:
            return n, False
        return _exec(node[2], world, n)
    if kind == 'if':
        n, ok, c = _exec_cond(node[1], world, n + 1)
        if not ok:
            return n, False
        return _exec(node[2], world, n) if c else (n, True)
    if kind == 'ifelse':
        n, ok, c = _exec_cond(node[1], world, n + 1)
        if not ok:
            return n, False
        return _exec(node[2] if c else node[3], world, n)
    if kind == 'while':
        n, ok, c = _exec_cond(node[1], world, n)
        if not ok:
            return n, False
        while c:
            n, ok = _exec(node[2], world, n)
            if not ok:
                return n, False
            n, ok, c = _exec_cond(node[1], world, n)
            if not ok:
                return n, False
        return n, True
    if kind == 'repeat':
        n += 1
        for _ in range(node[1]):
            n, ok = _exec(node[2], world, n)
            if not ok:
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
    if kind == 'exist':
        return ['exist_actor_in_distance_horizontal', node[1], node[2], node[3]]
    if kind == 'intarget':
        return ['in_target', node[1]]
    if kind == 'isthere':
        return ['is_there', node[1]]
    raise ValueError('not a condition node: %r' % (kind,))


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
        return (['if'] + _canon_cond(node[1]) + _canon(node[2])) * MAX_WHILE
    if kind == 'repeat':
        return _canon(node[2]) * node[1]
    if kind == 'action':
        return [node[1]]
    raise ValueError('not a statement node: %r' % (kind,))


def parse(code):
    """Shift-reduce the token string -> Program; `.ok` is the reference's second return value.
    An empty string (the reference raises IndexError) is a syntax error."""
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
