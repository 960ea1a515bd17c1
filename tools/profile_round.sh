#!/bin/bash
# Runs on the GPU box: everything behind profiles/<round>_*: counter passes, kernel trace (warm launches only), lane
# occupancy (analysis build), and the same for the 100 M-triangle scene (BASELINE config 5 on one GPU, where HBM binds).
# Usage: tools/profile_round.sh r03 [big]      (big: also the G=7072 passes, ~8 minutes)
set -u
R=$1
BIG=${2:-}
OUT=gpurun_out/${R}_prof
mkdir -p $OUT
export TMPDIR=/tmp
KEY=g708_1920x1080x64_d8
bash tools/pmc_passes.sh $OUT/pmc --frames 3 --primary 2 || exit 1
python3 tools/pmc_summary.py $OUT/pmc $KEY > $OUT/${R}_pmc_summary.json || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/run_frame.py --frames 8 --primary 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_kernel_stats_all_launches.csv \;
# per-kernel statistics of the WARM launches only (the first two of every kernel dropped): what a per-step time may be compared with
python3 tools/warm_kernel_stats.py $OUT/trace 2 > $OUT/${R}_kernel_stats.csv || exit 1
RT_AMD_LIB=rt_amd/librt_amd_stats.so timeout -k 10 300 python3 tools/trav_stats.py --spp 64 > $OUT/${R}_lane_occupancy.jsonl 2> $OUT/trav.err || { tail -5 $OUT/trav.err; exit 1; }
RT_AMD_LIB=rt_amd/librt_amd_stats.so timeout -k 10 300 python3 tools/trav_stats.py --spp 64 --wave-times >> $OUT/${R}_lane_occupancy.jsonl 2>> $OUT/trav.err || exit 1
if [ -n "$BIG" ]; then
  BKEY=g7072_1920x1080x16_d8
  bash tools/pmc_passes.sh $OUT/pmc_big --grid 7072 --spp 16 --frames 2 --primary 1 || exit 1
  python3 tools/pmc_summary.py $OUT/pmc_big $BKEY > $OUT/${R}_pmc_summary_${BKEY}.json || exit 1
  RT_AMD_LIB=rt_amd/librt_amd_stats.so timeout -k 10 600 python3 tools/trav_stats.py --grid 7072 --spp 16 >> $OUT/${R}_lane_occupancy.jsonl 2>> $OUT/trav.err || { tail -5 $OUT/trav.err; exit 1; }
fi
echo profile_round done
