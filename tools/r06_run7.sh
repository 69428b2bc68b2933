set -x
mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -x -q -k "vizdoom or k25 or folded or small" > gpurun_out/r06g_pytest_model.log 2>&1; tail -3 gpurun_out/r06g_pytest_model.log
python bench.py --preset vizdoom --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/r06g_bench_vizdoom.json 2> gpurun_out/r06g_bench_vizdoom.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06g_bench_vizdoom.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'])
for r in d['kernel_table']: print(r['group'], r['launches_per_step'], r['ms_per_step'], r['rate'])
PY
python bench.py --preset vizdoom_k25 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-roofline > gpurun_out/r06g_bench_vizdoom_k25.json 2>> gpurun_out/r06g_bench_vizdoom.err; tail -c 300 gpurun_out/r06g_bench_vizdoom_k25.json
