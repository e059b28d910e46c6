# Repeated A/B of library variants on a named workload: bash scripts/r03_ab_rep_w.sh OUT REPEATS STEPS WORKLOAD LIBS...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; N=$1; shift; S=$1; shift; W=$1; shift; mkdir -p $R/gpurun_out/r03; rm -f $O
cd $R
for rep in $(seq $N); do
  for lib in "$@"; do
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 600 python3 bench.py --workload $W --no-cpu-baseline --steps $S --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('$lib', d['ms_per_step'], d.get('prefill_tok_s', 0), d.get('decode_tok_s', 0))
" >> $O || exit 1
  done
done
python3 - $O $W <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    k, x, p, d = ln.split(); v[k].append((float(x), float(p), float(d)))
for k, xs in v.items():
    ms = [a for a, _, _ in xs]
    print(f"{sys.argv[2]} {k:12s} mean {statistics.mean(ms):.4f} median {statistics.median(ms):.4f} min {min(ms):.4f} max {max(ms):.4f} n {len(ms)} prefill {statistics.mean(p for _, p, _ in xs):.0f} decode {statistics.mean(d for _, _, d in xs):.0f}")
PY
