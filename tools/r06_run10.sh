set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or wide or batch_norm or bn" > gpurun_out/r06j_pytest_conv.log 2>&1; tail -5 gpurun_out/r06j_pytest_conv.log
python tools/bench_conv_wide.py > gpurun_out/r06j_bench_conv_wide.log 2>&1; grep -E "^conv|dgrad" gpurun_out/r06j_bench_conv_wide.log
python -m pytest tests/test_model_gpu.py -x -q -k "vizdoom or k25 or folded or small" > gpurun_out/r06j_pytest_model.log 2>&1; tail -3 gpurun_out/r06j_pytest_model.log
python bench.py --preset vizdoom --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/r06j_bench_vizdoom.json 2> gpurun_out/r06j_bench_vizdoom.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06j_bench_vizdoom.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'])
for r in d['kernel_table']: print(r['group'], r['launches_per_step'], r['ms_per_step'], r['rate'])
PY
