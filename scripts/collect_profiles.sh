set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 300 python3 $R/bench.py --workload qwen2-7b-w4a16-b128-s1024 > $O/bench_7b_b128.json 2>/dev/null || exit 1
timeout -k 10 300 python3 $R/bench.py --workload qwen2-7b-w4a16-b512-s1024 > $O/bench_7b_b512.json 2>/dev/null || exit 1
timeout -k 10 300 python3 $R/bench.py --workload ref-bench --steps 3 > $O/bench_ref.json 2>/dev/null || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k7b -o b -- python3 $R/bench.py --no-cpu-baseline --workload qwen2-7b-w4a16-b128-s1024 > $O/bench_7b_b128_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/k7b/*kernel_trace.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kdef -o b -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/kdef/*kernel_trace.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o a -- python3 $R/scripts/bench_attn.py 0.5b 1024 > $O/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o a -- python3 $R/scripts/bench_attn.py 0.5b 1024 > $O/pmc_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_gemm1 -o a -- python3 $R/scripts/bench_gemm.py gateup7b > $O/pmc_gemm1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_gemm2 -o a -- python3 $R/scripts/bench_gemm.py gateup7b > $O/pmc_gemm2.log 2>&1 || exit 1
python3 $R/scripts/pmc_table.py $O/pmc_fetch paged_decode > $O/pmc_attn.txt
python3 $R/scripts/pmc_table.py $O/pmc_write paged_decode >> $O/pmc_attn.txt
python3 $R/scripts/pmc_table.py $O/pmc_gemm1 gemm_ > $O/pmc_gemm.txt
python3 $R/scripts/pmc_table.py $O/pmc_gemm2 gemm_ >> $O/pmc_gemm.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_gemm1 $O/pmc_gemm2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kref -o b -- python3 $R/bench.py --no-cpu-baseline --workload ref-bench --steps 2 --warmup 1 > $O/bench_ref_under_rocprof.json 2>/dev/null || exit 1
rm -f $O/kref/*kernel_trace.csv
for w in 1 2 4 8; do NV_ATTN_WAVES=$w timeout -k 10 100 python3 $R/scripts/bench_attn.py qwen3 180 2>/dev/null; NV_ATTN_WAVES=$w timeout -k 10 100 python3 $R/scripts/bench_attn.py 7b 2>/dev/null | grep "splits=1"; NV_ATTN_WAVES=$w timeout -k 10 100 python3 $R/scripts/bench_attn.py pmc 1024 2>/dev/null; done > $O/attn_waves.txt
timeout -k 10 100 python3 $R/scripts/bench_head.py > $O/head_sampler.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/bench_attn.py prefill > $O/prefill_attn.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/bench_gemm.py 7b 128 > $O/gemm_7b.txt 2>/dev/null
timeout -k 10 200 python3 $R/scripts/bench_gemm.py 0.5b 128 > $O/gemm_05b.txt 2>/dev/null
cat $O/pmc_attn.txt $O/pmc_gemm.txt
