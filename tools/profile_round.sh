# Run on the GPU box (through gpurun): everything a round's profiles/ entry is made of.
# usage: bash tools/profile_round.sh <tag>     (tag e.g. r02a; outputs under gpurun_out/)
TAG=${1:-r02}
set -x
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
# HBM traffic counters, one counter per pass (D2P_COMMIT: the commit these sources are at -- .git does not travel)
bash tools/profile_pmc.sh $TAG > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.md
# MFMA-pipe busy cycles per kernel, own pass
bash tools/profile_mfma.sh ${TAG}m > /dev/null 2>&1
DB=$(find gpurun_out/pmc_${TAG}m -name "*.db" | head -1)
python tools/mfma_summary.py $DB 2.4 gpurun_out/${TAG}_mfma_util.json > gpurun_out/${TAG}_mfma_util.md
find gpurun_out -name "*.db" -size +1M -delete
ls -la gpurun_out/ | tail -12
