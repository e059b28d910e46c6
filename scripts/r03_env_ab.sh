# A/B of environment switches on the default bench: bash scripts/r03_env_ab.sh OUT "VAR=a" "VAR=b" ... (each run twice, interleaved)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03/$1; shift; mkdir -p $R/gpurun_out/r03
cd $R
for rep in 1 2 3; do
  for setting in "$@"; do
    echo "== $setting" >> $O
    env $setting timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 96 --warmup 8 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], 'ms/step', d['value'], 'tok/s')
" >> $O || exit 1
  done
done
cat $O
