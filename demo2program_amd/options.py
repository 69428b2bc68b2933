"""Every process-level switch of the product path, in ONE table.

Rule: a switch exists only if BOTH of its settings are exercised by a test, or if it is
plumbing (library path, build jobs).  A/B switches whose experiment ended negative are deleted together with their
code path -- the measurement stays in DESIGN.md's appendix, the knob does not stay in the product.  There is no
switch that leaves work out of a step: the timing-only ablation hook of tools/step_ablation.py is set on a Model
OBJECT by that tool (`Model.set_ablation`), and `Trainer.train` / `Evaler` refuse to run a model that carries one.

tests/test_host_logic.py::test_every_environment_switch_is_declared checks that no other D2P_* name is read
anywhere in the package.

Not process switches: three attributes of a Model OBJECT that exist for the A side of a test's A/B and are never read
from the environment -- `fused_encoder` (set from D2P_FUSED_ENCODER at construction), `fused_rn` (the relation networks'
four-launch form; tests/test_model_gpu.py::test_relation_networks_in_four_launches_equal_the_separate_launches) and
`decoder_skip_past_len` (a training step's decoders stop at a row's length;
::test_training_step_decoders_skip_the_steps_past_a_rows_length); likewise `fold_bn` (batch norm folded into the conv launches)
and `grouped_decoder_grads` (::test_decoder_small_gradient_products_grouped_equal_the_separate_launches), `paired_kernel_grads`
(an encoder's two kernel-gradient halves as one product; ::test_encoder_kernel_gradient_halves_as_one_product).
"""
import os

# name -> (default, what '1' / '0' mean, the test that runs the non-default side)
SWITCHES = {
    'D2P_SIDE_STREAM': ('1', "two-stream eager schedule; '0' = everything on one stream (profiling passes, the "
                        "instrumented roofline pass, graph capture)", 'tests/test_model_gpu.py::test_rccl_single_rank_self_test'),
    'D2P_GRAPH': ('0', "'1' = forward + backward captured as one hipGraph per (n_prog, n_demo)",
                  'tests/test_model_gpu.py::test_graph_replay_survives_scratch_growth_and_new_shapes'),
    'D2P_DP_OVERLAP': ('0', "'1' = the decoders' gradient slice all-reduced beside the encoder backward",
                       'tests/test_model_gpu.py::test_rccl_single_rank_self_test'),
    'D2P_FORCE_DIST': ('0', "'1' = create the process group even for one rank (the exchange step runs as an identity)",
                       'tests/test_model_gpu.py::test_bench_starts_its_own_ranks'),
    'D2P_FUSED_ENCODER': ('1', "Karel State_Encoder in one launch per direction; '0' = the separate conv / batch-norm "
                          "launches (the fallback every other geometry takes)",
                          'tests/test_model_gpu.py::test_one_launch_state_encoder_equals_the_separate_launches'),
    'D2P_FUSED_LOSS': ('1', "loss value from the loss-backward launch; '0' = the forward cross-entropy launches (the "
                       "form evaluation takes)", 'tests/test_model_gpu.py::test_loss_value_from_the_loss_backward_launch'),
    'D2P_PER_FACTORED': ('1', "perception decoder input in factored form; '0' = row-wise fc + batch norm (the form "
                         "evaluation takes)", 'tests/test_model_gpu.py::test_perception_decoder_factored_input_equals_row_wise_form'),
    'D2P_TOKEN_PROJECTION': ('1', "token-input decoders project the embedding table; '0' = project the gathered rows "
                             "(the form scheduled sampling takes)",
                             'tests/test_model_gpu.py::test_token_decoders_project_the_table_not_the_rows'),
    'D2P_PRIORITY_STREAM': ('1', "`with Trainer.step_stream():` (Trainer.train, bench.py) runs the steps on a priority -1 stream of the "
                            "trainer's own (eager schedule); '0' = on the caller's current stream", 'tests/test_model_gpu.py::test_step_on_the_priority_stream_equals_the_step_on_the_callers_stream'),
    # plumbing
    'D2P_LIB_PATH': (None, 'path of libd2p_hip.so (default: in-tree csrc/)', 'tests/test_abi.py'),
    'D2P_BUILD_JOBS': (None, 'parallel hipcc jobs of build.py', '-'),
    'D2P_BENCH_SELF_SPAWNED': (None, "set by bench.py on the ranks it starts itself", '-'),
    'D2P_COMMIT': (None, 'commit id written into profile summaries (.git does not travel to the GPU box)', '-'),
}


def flag(name):
    """True when switch `name` is on (its default, unless the environment says otherwise)."""
    default = SWITCHES[name][0]
    return os.environ.get(name, default) == '1'
