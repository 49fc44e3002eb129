#!/bin/bash
# Build probe variants of the library (gemm_rows.hip with -DRS_PROBE=n, everything else from the regular objects) HERE, then time them on the GPU box:
#   bash tools/rows_probe.sh build 1 2 3 4 5 8      (build container)
#   gpurun -- 'bash tools/rows_probe.sh run 1 2 3 4 5 8'
set -e
cd "$(dirname "$0")/.."
C=revisionllm_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  python -c "from revisionllm_amd import build; build.build_library()"
  for p in "$@"; do
    # the bf16 parts carry the probe (parts 0 and 2: 4 / 5 and 8 / 9 row blocks); the FP8 parts come from the regular objects
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function -Wno-pass-failed -DRV_OP_F16=1 -DRS_PROBE=$p $RS_EXTRA -DRV_TU=gemm_rows -c $C/gemm_rows.hip -o $C/build/probe$p.rows0.o &
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-function -Wno-pass-failed -DRV_OP_F16=1 -DRS_PROBE=$p $RS_EXTRA -DRV_TU=gemm_rows_p2 -c $C/gemm_rows_p2.hip -o $C/build/probe$p.rows2.o &
  done
  wait
  for p in "$@"; do
    objs=$(ls $C/build/f16/*.o | grep -v "/gemm_rows.o" | grep -v "/gemm_rows_p2.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map -o revisionllm_amd/librevision_hip_p$p.so $objs $C/build/probe$p.rows0.o $C/build/probe$p.rows2.o
  done
  ls -la revisionllm_amd/librevision_hip_p*.so
else
  export PYTHONPATH=.
  M=${ROWS:-70}
  echo "== regular"; python tools/rows_time.py $M 2>&1 | grep -E "N=22016 K= 4096|N= 4096 K= 4096|N=12288 K= 4096"
  for p in "$@"; do
    echo "== RS_PROBE=$p"; REVISION_HIP_LIB=$PWD/revisionllm_amd/librevision_hip_p$p.so python tools/rows_time.py $M 2>&1 | grep -E "N=22016 K= 4096|N= 4096 K= 4096|N=12288 K= 4096"
  done
fi
