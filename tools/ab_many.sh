#!/bin/bash
# several library builds on one box, in turn: tools/ab_many.sh ROUNDS LIB...
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    RT_AMD_LIB=$L timeout -k 10 120 python3 tools/run_frame.py --frames 4 --primary 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$L', round(d['render_ms'],2), round(d['primary_ms'],3))" || exit 1
  done
done
