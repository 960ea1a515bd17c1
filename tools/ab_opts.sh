#!/bin/bash
# several (library, run_frame options) variants on one box, in turn: tools/ab_opts.sh ROUNDS "LIB [run_frame options]" ...
N=$1; shift
VARIANTS=("$@")
for i in $(seq $N); do
  for v in "${VARIANTS[@]}"; do
    read -r L OPTS <<< "$v"
    RT_AMD_LIB=$L timeout -k 10 180 python3 tools/run_frame.py --frames 4 --primary 3 $OPTS | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['render_ms'],2), round(d['primary_ms'],3))" || exit 1
  done
done
