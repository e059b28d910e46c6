# Round-3 evidence: bench lines of every workload, rocprofv3 per-kernel summaries, PMC traffic of the decode kernel.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
for w in qwen2-7b-w4a16-b128-s1024 qwen2-7b-w4a16-b512-s1024 qwen2-0.5b-bf16-b32-s1024 qwen2-0.5b-w8a16-b64-s1024 qwen2-0.5b-fp16-b1-s128; do
  timeout -k 10 400 python3 $R/bench.py --workload $w > $O/bench_$w.json 2>/dev/null || exit 1
done
timeout -k 10 300 python3 $R/bench.py --workload ref-bench --steps 3 > $O/bench_ref.json 2>/dev/null || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kdef -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/kdef/*kernel_trace.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k7b -o b -- python3 $R/bench.py --no-cpu-baseline --workload qwen2-7b-w4a16-b128-s1024 > $O/bench_7b_b128_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/k7b/*kernel_trace.csv
for which in pmc pmc7b; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$which -o a -- python3 $R/scripts/bench_attn.py $which 1024 > $O/f_$which.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$which -o a -- python3 $R/scripts/bench_attn.py $which 1024 > $O/w_$which.log 2>&1 || exit 1
  echo "== $which" >> $O/pmc_attn.txt
  python3 $R/scripts/pmc_table.py $O/f_$which paged_decode >> $O/pmc_attn.txt
  python3 $R/scripts/pmc_table.py $O/w_$which paged_decode >> $O/pmc_attn.txt
  grep "us/launch" $O/f_$which.log >> $O/pmc_attn.txt
  rm -rf $O/f_$which $O/w_$which
done
timeout -k 10 100 python3 $R/scripts/bench_head.py > $O/head_sampler.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/host_breakdown.py > $O/host_breakdown.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/bench_gemm.py 0.5b 128 > $O/gemm_05b.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/bench_gemm.py 7b 128 > $O/gemm_7b.txt 2>/dev/null
cat $O/pmc_attn.txt; cat $O/host_breakdown.txt; python3 $R/scripts/kernel_table.py $O/kdef 74 > $O/kernel_table_default.txt; python3 $R/scripts/kernel_table.py $O/k7b 74 > $O/kernel_table_7b.txt; cat $O/kernel_table_default.txt
