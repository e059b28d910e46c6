# HBM traffic of the paged decode-attention launch from the L2's memory-side counters: separate --pmc passes
# (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), each a run of its own.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for which in pmc pmc7b; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$which -o a -- python3 $R/scripts/bench_attn.py $which 1024 > $O/f_$which.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$which -o a -- python3 $R/scripts/bench_attn.py $which 1024 > $O/w_$which.log 2>&1 || exit 1
  echo "== $which" >> $O/pmc_attn2.txt
  python3 $R/scripts/pmc_table.py $O/f_$which paged_decode >> $O/pmc_attn2.txt
  python3 $R/scripts/pmc_table.py $O/w_$which paged_decode >> $O/pmc_attn2.txt
  grep "us/launch" $O/f_$which.log >> $O/pmc_attn2.txt
  rm -rf $O/f_$which $O/w_$which
done
cat $O/pmc_attn2.txt
