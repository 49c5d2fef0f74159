#!/bin/bash
# same-box A/B/C of the bench step under up to four environment settings: tools/ab_env3.sh reps "A=0" "A=1" "A=2" ...
set -e
reps=$1; shift
B="python bench.py --no_cpu_baseline --no_extra --steps 20 --warmup 5"
for i in $(seq 1 $reps); do
  for v in "$@"; do
    echo -n "$v  "; env $v timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
