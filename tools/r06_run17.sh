mkdir -p gpurun_out
python tools/bench_small_products.py > gpurun_out/r06q_bench_small_products.log 2>&1; cat gpurun_out/r06q_bench_small_products.log
