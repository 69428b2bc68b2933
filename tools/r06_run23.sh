mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "lstm" > gpurun_out/r06s_pytest_lstm.log 2>&1; tail -3 gpurun_out/r06s_pytest_lstm.log
python tools/step_ab.py d2p_lstm_persist_set_skip_zero_pass 0 1 --rounds 4 --steps 150 > gpurun_out/r06s_ab_skip_zero_pass.log 2>&1; tail -4 gpurun_out/r06s_ab_skip_zero_pass.log
python tools/lstm_persist_steps.py 2>&1 | tail -4
