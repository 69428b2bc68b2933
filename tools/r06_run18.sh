mkdir -p gpurun_out
python tools/bench_small_products.py > gpurun_out/r06q_bench_small_products.log 2>&1; cat gpurun_out/r06q_bench_small_products.log
python -m pytest tests/test_kernels_gpu.py -x -q -k "projection_weight or small_pair" 2>&1 | tail -2
python tools/step_ab_attr.py grouped_decoder_grads 0 1 --rounds 3 --steps 150 > gpurun_out/r06q_ab_grouped.log 2>&1; tail -4 gpurun_out/r06q_ab_grouped.log
