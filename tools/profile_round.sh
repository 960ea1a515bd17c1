#!/bin/bash
# Runs on the GPU box: everything behind profiles/<round>_*: counter passes, kernel trace, lane occupancy (analysis
# build), bench line.  Usage: tools/profile_round.sh r02
set -u
R=$1
OUT=gpurun_out/${R}_prof
mkdir -p $OUT
export TMPDIR=/tmp
KEY=g708_1920x1080x64_d8
bash tools/pmc_passes.sh $OUT/pmc --frames 3 --primary 2 || exit 1
python3 tools/pmc_summary.py $OUT/pmc $KEY > $OUT/${R}_pmc_summary.json || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/run_frame.py --frames 5 --primary 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_kernel_stats.csv \;
RT_AMD_LIB=rt_amd/librt_amd_stats.so timeout -k 10 300 python3 tools/trav_stats.py --spp 64 > $OUT/${R}_lane_occupancy.jsonl 2> $OUT/trav.err || { tail -5 $OUT/trav.err; exit 1; }
RT_AMD_LIB=rt_amd/librt_amd_stats.so timeout -k 10 300 python3 tools/trav_stats.py --spp 64 --wave-times >> $OUT/${R}_lane_occupancy.jsonl 2>> $OUT/trav.err || exit 1
echo profile_round done
