# A/B of library variants on the benchmark steps: bash scripts/r03_ab.sh OUTFILE WORKLOADS LIBS...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; W="$1"; shift; mkdir -p $R/gpurun_out/r03
cd $R
for w in $W; do
  for lib in "$@"; do
    echo "== $w $lib" >> $O
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 400 python3 bench.py --workload $w --no-cpu-baseline --steps 64 --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], 'ms/step', d['value'], 'tok/s', 'attn us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])
" >> $O || exit 1
  done
done
cat $O
