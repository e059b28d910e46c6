# Decode-attention variants (scripts/build_variant.py) against the shipped build, same box, same process environment.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
for lib in libnv_hip abl_nt0 abl_d3 abl_d4 abl_d4nt0; do
  for w in auto 8; do
    echo "== $lib waves=$w" >> $O/attn_variants.txt
    if [ $w = auto ]; then unset NV_ATTN_WAVES; else export NV_ATTN_WAVES=$w; fi
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 120 python3 scripts/bench_attn.py pmc 1024 >> $O/attn_variants.txt 2>&1 || exit 1
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 120 python3 scripts/bench_attn.py pmc7b 1024 >> $O/attn_variants.txt 2>&1 || exit 1
  done
done
unset NV_ATTN_WAVES
for lib in libnv_hip abl_d4; do
  echo "== $lib long/short contexts" >> $O/attn_variants.txt
  for S in 256 4000; do
    NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 120 python3 scripts/bench_attn.py pmc $S >> $O/attn_variants.txt 2>&1 || exit 1
  done
  NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 120 python3 scripts/bench_attn.py 7b512 1024 >> $O/attn_variants.txt 2>&1 || exit 1
  NV_HIP_LIB=$R/byox_vllm_amd/$lib.so timeout -k 10 120 python3 scripts/bench_attn.py qwen3 180 >> $O/attn_variants.txt 2>&1 || exit 1
done
cat $O/attn_variants.txt
