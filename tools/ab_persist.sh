set -e
B="python bench.py --no_cpu_baseline --no_extra --steps 20 --warmup 5"
for i in 1 2; do
  for v in "MT_IGEMM_PERSIST=0" "MT_IGEMM_PERSIST=1" "MT_IGEMM_PERSIST=1 MT_PK_STORE_AUX=2"; do
    echo "== $v"; env $v timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
