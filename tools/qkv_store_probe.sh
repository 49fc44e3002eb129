#!/bin/bash
# What do the fused q / k / v epilogue's stores cost the PREFILL QKV GEMM?  Builds probe variants of the fp16 library HERE (gemm_pp.hip with -DRS_PROBE=256 / 512 / 1024 / 1792:
# no V^T / K / Q / any store - kernels.h qkv_rope_store_t; everything else from the regular objects), then times one batched prefill pass per variant on the GPU box:
#   bash tools/qkv_store_probe.sh build        (build container)
#   gpurun -- 'bash tools/qkv_store_probe.sh run'
C=revisionllm_amd/csrc
if [ "$1" = build ]; then
  for p in 256 512 1024 1792 2048; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function -Wno-pass-failed -DRV_OP_F16=1 -DRS_PROBE=$p -DRV_TU=gemm_pp -c $C/gemm_pp.hip -o $C/build/probe_pp$p.o &
  done
  wait
  for p in 256 512 1024 1792 2048; do
    objs=$(ls $C/build/f16/*.o | grep -v "/gemm_pp.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o revisionllm_amd/librevision_hip_pp$p.so $objs $C/build/probe_pp$p.o
  done
  ls -la revisionllm_amd/librevision_hip_pp*.so
else
  cd "$GRAFT_REPO_ROOT"
  bash tools/prefill_prof.sh 4 2>&1 | grep "gemm_pp_sk<0, 0, [12]" | sed 's/^/regular      /'
  for p in 256 512 1024 1792 2048; do
    REVISION_HIP_LIB=$PWD/revisionllm_amd/librevision_hip_pp$p.so bash tools/prefill_prof.sh 4 2>&1 | grep "gemm_pp_sk<0, 0, [12]" | sed "s/^/RS_PROBE=$p /"
  done
fi
