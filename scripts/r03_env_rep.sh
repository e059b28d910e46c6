# Repeated A/B of environment settings on one workload: [FIELD=value] bash scripts/r03_env_rep.sh OUT REPEATS STEPS WORKLOAD "VAR=a" "VAR=b" ...
# (FIELD: the JSON field compared, ms_per_step unless given)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; N=$1; shift; S=$1; shift; W=$1; shift; mkdir -p $R/gpurun_out/r03; rm -f $O
cd $R
for rep in $(seq $N); do
  for setting in "$@"; do
    env $setting timeout -k 10 600 python3 bench.py --workload $W --no-cpu-baseline --steps $S --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$setting', d['${FIELD:-ms_per_step}'])
" >> $O || exit 1
  done
done
python3 - $O $W <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    k, x = ln.split(); v[k].append(float(x))
for k, xs in v.items():
    print(f"{sys.argv[2]} {k:24s} mean {statistics.mean(xs):.4f} median {statistics.median(xs):.4f} min {min(xs):.4f} max {max(xs):.4f} n {len(xs)}")
PY
