#!/bin/bash
# A/B of two builds of the library on one box: tools/ab.sh LIB_A LIB_B [rounds] -- prints ms per frame, alternating
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    RT_AMD_LIB=$L timeout -k 10 120 python3 tools/run_frame.py --frames 4 --primary 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', round(d['render_ms'],2), round(d['primary_ms'],3))" || exit 1
  done
done
