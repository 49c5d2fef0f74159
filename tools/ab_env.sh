#!/bin/bash
# same-box A/B of the bench step under two environment settings: tools/ab_env.sh "A=0" "A=1" [reps] [extra bench flags]
set -e
B="python bench.py --no_cpu_baseline --no_extra --steps 20 --warmup 5 $4"
for i in $(seq 1 ${3:-2}); do
  for v in "$1" "$2"; do
    echo -n "$v  "; env $v timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
