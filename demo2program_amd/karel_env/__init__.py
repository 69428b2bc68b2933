"""Karel domain pieces the full model's metrics need: vocabulary, the stack
parser with its executor and canonicaliser, and the grid world.  Host-side Python like the
reference's `karel_env/` (they run inside tf.py_func there); no GPU work here.
Pinned by tests/golden/karel_dsl.json, produced by running the reference's own code."""
from .dsl import KarelVocab, get_KarelDSL, parse, Program   # noqa: F401
from .karel import Karel_world                               # noqa: F401
