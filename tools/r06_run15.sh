set -x
mkdir -p gpurun_out
python tools/step_ab_attr.py small_stream 0 1 --rounds 3 --steps 150 --marks > gpurun_out/r06o_ab_small_stream.log 2>&1; grep -v amdgpu.ids gpurun_out/r06o_ab_small_stream.log | tail -50
python tools/step_ab_attr.py small_stream 0 1 --rounds 2 --steps 40 --preset vizdoom > gpurun_out/r06o_ab_small_stream_vizdoom.log 2>&1; tail -4 gpurun_out/r06o_ab_small_stream_vizdoom.log
