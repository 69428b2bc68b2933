set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "projection_weight or small_pair" > gpurun_out/r06p_pytest_k.log 2>&1; tail -15 gpurun_out/r06p_pytest_k.log
python -m pytest tests/test_model_gpu.py -x -q -k "grouped or small or headline_config_matches or one_training_step" > gpurun_out/r06p_pytest_m.log 2>&1; tail -15 gpurun_out/r06p_pytest_m.log
python tools/step_ab_attr.py grouped_decoder_grads 0 1 --rounds 3 --steps 150 > gpurun_out/r06p_ab_grouped.log 2>&1; tail -5 gpurun_out/r06p_ab_grouped.log
