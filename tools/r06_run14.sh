set -x
mkdir -p gpurun_out
python tools/step_ab.py d2p_conv_set_direct 3 2 --extra 2 2 --preset vizdoom --rounds 3 --steps 40 > gpurun_out/r06n_ab_fwd2.log 2>&1; tail -8 gpurun_out/r06n_ab_fwd2.log
