set -x
mkdir -p gpurun_out
python tools/cu_budget_sweep.py --budgets 0 224 192 160 128 --rounds 3 --steps 100 > gpurun_out/r06b_cu_budget_karel.log 2>&1
python tools/cu_budget_sweep.py --preset vizdoom --budgets 0 224 192 --rounds 2 --steps 40 > gpurun_out/r06b_cu_budget_vizdoom.log 2>&1
tail -8 gpurun_out/r06b_cu_budget_karel.log gpurun_out/r06b_cu_budget_vizdoom.log
