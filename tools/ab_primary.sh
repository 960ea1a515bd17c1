#!/bin/bash
# A/B of two library builds on primary rays: tools/ab_primary.sh LIB_A LIB_B [rounds] [run_frame args]
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    RT_AMD_LIB=$L timeout -k 10 120 python3 tools/run_frame.py --spp 1 --frames 1 --primary 9 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', round(d['primary_ms'],4))" || exit 1
  done
done
