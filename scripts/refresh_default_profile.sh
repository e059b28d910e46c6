set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kdef -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/kdef/*kernel_trace.csv
ls -R $O | head -20
