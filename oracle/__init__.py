"""CPU oracle for the demo2program full-model training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``demo2program_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the reported CPU
baseline -- never as the thing measured or shipped.

PARITY UNPINNED: the arithmetic of the reference path lives in
``tensorflow-gpu==1.3.0`` (``/root/reference/requirements.txt:1``), which is not
vendored under ``/root/reference`` and is not installable here, and the
reference ships no tests, golden vectors or fixtures.  The oracle therefore
restates the published TF-1.3 semantics of each call site (cited per function)
and is pinned by hand-derived known-answer tests (``tests/test_oracle_known_answers.py``,
SURVEY.md Appendix C D1-D14) plus independent ``torch.nn`` second opinions, not
by a live TF1 run.
"""
from .model_full import (  # noqa: F401
    OracleConfig,
    forward,
    loss_and_grads,
    lrelu,
    same_pad_s2k3,
    conv2d_lrelu_bn,
    batch_norm_train,
    batch_norm_infer,
    basic_lstm_cell,
    dynamic_rnn,
    training_decoder,
    sequence_loss,
    rn_pool,
    adam_clip_step,
    exponential_decay_staircase,
    polynomial_decay,
    PARAM_ORDER,
    param_shapes,
    greedy_decoder,
    sequence_stats,
    greedy_program_and_actions,
)
