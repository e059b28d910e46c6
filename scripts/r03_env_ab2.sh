# A/B of environment switches on one workload: bash scripts/r03_env_ab2.sh OUT WORKLOAD "VAR=a" "VAR=b" ... (3 interleaved repeats)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; W=$1; shift; mkdir -p $R/gpurun_out/r03
cd $R
for rep in 1 2 3; do
  for setting in "$@"; do
    echo "== $W $setting" >> $O
    env $setting timeout -k 10 400 python3 bench.py --workload $W --no-cpu-baseline --steps 96 --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], 'ms/step', d['value'], 'tok/s', 'attn', d['roofline']['avg_launch_us'])
" >> $O || exit 1
  done
done
