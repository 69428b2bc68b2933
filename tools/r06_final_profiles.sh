# Run on the GPU box (through gpurun): the round-6 profile set (outputs under gpurun_out/, copied into profiles/ by hand)
TAG=${1:-r06final}
set -x
mkdir -p gpurun_out
export D2P_COMMIT=${D2P_COMMIT:-unknown}
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for P in vizdoom vizdoom_k25; do
  python bench.py --preset $P --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_$P.json 2>> gpurun_out/${TAG}_bench.err
done
# kernel trace: eager one-stream launches (per-kernel durations), then the default two-stream schedule
D2P_SIDE_STREAM=0 bash tools/profile_bench.sh ${TAG}s > /dev/null 2>&1
DB=$(find gpurun_out/prof_${TAG}s -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 13 > gpurun_out/${TAG}_kernel_stats_serial.md
bash tools/profile_bench.sh ${TAG}g > /dev/null 2>&1
DB=$(find gpurun_out/prof_${TAG}g -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 13 > gpurun_out/${TAG}_kernel_stats_default.md
python tools/rocpd_streams.py $DB 5 --seq > gpurun_out/${TAG}_streams_timeline.txt
# the main-queue timeline without a profiler
python tools/step_marks.py --steps 60 > gpurun_out/${TAG}_step_marks_karel.log 2>&1
python tools/step_marks.py --steps 30 --preset vizdoom > gpurun_out/${TAG}_step_marks_vizdoom.log 2>&1
# HBM traffic counters, one counter per pass
bash tools/profile_pmc.sh $TAG > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.md
# config 4: kernel table + traffic
bash tools/profile_vizdoom.sh $TAG pmc > /dev/null 2>&1
# MFMA-pipe busy cycles per kernel, own pass
bash tools/profile_mfma.sh ${TAG}m > /dev/null 2>&1
DB=$(find gpurun_out/pmc_${TAG}m -name "*.db" | head -1)
python tools/mfma_summary.py $DB 2.4 gpurun_out/${TAG}_mfma_util.json > gpurun_out/${TAG}_mfma_util.md
find gpurun_out -name "*.db" -size +1M -delete
ls -la gpurun_out/ | grep $TAG
