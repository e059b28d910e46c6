# Repeated A/B of library variants on one workload: bash scripts/r03_ab_rep.sh OUT REPEATS STEPS LIBS...  (prints the per-variant mean and spread)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; N=$1; shift; S=$1; shift; mkdir -p $R/gpurun_out/r03; rm -f $O
cd $R
for rep in $(seq $N); do
  for lib in "$@"; do
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps $S --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$lib', d['ms_per_step'])
" >> $O || exit 1
  done
done
python3 - $O <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    k, x = ln.split(); v[k].append(float(x))
for k, xs in v.items():
    print(f"{k:16s} mean {statistics.mean(xs):.4f} median {statistics.median(xs):.4f} min {min(xs):.4f} max {max(xs):.4f} n {len(xs)}")
PY
