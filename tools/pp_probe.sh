#!/bin/bash
# Ablation probes of the 256 x 256 x 64 ping-pong main loop (gemm_pp.hip, -DPP_ABL=n: 1 no LDS-DMA in the steady state, 2 no fragment reads, 4 no barriers,
# 8 no MFMAs; sums combine).  Results of a probe build are garbage - only its time means anything.
#   bash tools/pp_probe.sh build 1 2 3 4 7 8            (build container; extra compiler flags through PP_EXTRA="-D...")
#   gpurun -- 'bash tools/pp_probe.sh run 1 2 3 4 7 8'  (KSWEEP / ROWS as in tools/blas_yardstick.py)
set -e
cd "$(dirname "$0")/.."
C=revisionllm_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  python -c "from revisionllm_amd import build; build.build_library(flavours=('f16',))"
  for p in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function -Wno-pass-failed -DRV_OP_F16=1 -DPP_ABL=$p $PP_EXTRA -DRV_TU=gemm_pp -c $C/gemm_pp.hip -o $C/build/probe_pp$p.o &
  done
  wait
  for p in "$@"; do
    objs=$(ls $C/build/f16/*.o | grep -v "/gemm_pp.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o revisionllm_amd/librevision_hip_pp$p.so $objs $C/build/probe_pp$p.o
  done
  ls -la revisionllm_amd/librevision_hip_pp*.so
else
  export PYTHONPATH=.
  echo "== regular"; python tools/blas_yardstick.py f16 2>&1 | grep "M="
  for p in "$@"; do
    echo "== PP_ABL=$p"; REVISION_HIP_LIB=$PWD/revisionllm_amd/librevision_hip_pp$p.so python tools/blas_yardstick.py f16 2>&1 | grep "M="
  done
fi
