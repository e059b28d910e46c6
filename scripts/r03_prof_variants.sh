# per-kernel rocprofv3 tables of the default bench for several library variants: bash scripts/r03_prof_variants.sh LIB...
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export NV_HIP_LIB=$R/byox_vllm_amd/$lib.so
  rm -rf $O/k_$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k_$lib -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_$lib.json 2>/dev/null || exit 1
  rm -f $O/k_$lib/*kernel_trace.csv
  echo "== $lib: $(python3 -c "import json;d=json.load(open('$O/bench_$lib.json'));print(d['ms_per_step'])") ms/step under rocprof"
  python3 $R/scripts/kernel_table.py $O/k_$lib 74 | head -8
done
