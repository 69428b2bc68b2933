mkdir -p gpurun_out
python tools/bench_small_products.py 2>&1 | tail -2
python -m pytest tests/test_kernels_gpu.py -x -q -k "small_pair" 2>&1 | tail -2
python -m pytest tests/test_model_gpu.py -x -q -k "grouped" 2>&1 | tail -2
