set -x
mkdir -p gpurun_out
python tools/step_marks.py --steps 60 > gpurun_out/r06i_step_marks_karel.log 2>&1; cat gpurun_out/r06i_step_marks_karel.log
D2P_SIDE_STREAM=0 python tools/step_marks.py --steps 60 > gpurun_out/r06i_step_marks_karel_one_stream.log 2>&1; cat gpurun_out/r06i_step_marks_karel_one_stream.log
