set -x
mkdir -p gpurun_out
python tools/bench_conv2.py > gpurun_out/r06m_bench_conv2.log 2>&1; cat gpurun_out/r06m_bench_conv2.log
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or wide or batch_norm or bn" > gpurun_out/r06m_pytest_conv.log 2>&1; tail -3 gpurun_out/r06m_pytest_conv.log
python -m pytest tests/test_model_gpu.py -x -q -k "vizdoom or k25 or folded or small" > gpurun_out/r06m_pytest_model.log 2>&1; tail -3 gpurun_out/r06m_pytest_model.log
python bench.py --preset vizdoom --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('vizdoom', d['ms_per_step'], d['value'])"
