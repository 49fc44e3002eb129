"""Build the HIP library (gfx950) in-tree with hipcc, in its two operand flavours: ``librevision_hip.so`` (fp16 operands, the default) and
``librevision_hip_bf16.so`` (bf16 operands) - the same sources compiled with ``-DRV_OP_F16=1`` / ``0`` (csrc/common.h).  No JIT cache: the
``.so`` files ship with the tree.  Only the entry points ``include/revision_hip.h`` declares are exported (``-fvisibility=hidden``)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
FLAVOURS = {"f16": ("librevision_hip.so", "-DRV_OP_F16=1"), "bf16": ("librevision_hip_bf16.so", "-DRV_OP_F16=0")}
LIB = os.path.join(HERE, FLAVOURS["f16"][0])


def lib_path(flavour):
    return os.path.join(HERE, FLAVOURS[flavour][0])
SOURCES = ["error.hip", "init.hip", "gemm.hip", "gemm_pp.hip", "gemm_arows.hip", "gemm_rows.hip", "gemm_rows_p1.hip", "gemm_rows_p2.hip", "gemm_rows_p3.hip", "rowops.hip", "attention.hip", "sample.hip", "engine.hip"]
HEADERS = ["exports.map", "common.h", "kernels.h", "gemv_finish.h", "gemm_rows.hip", os.path.join("..", "..", "include", "revision_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, flavours=("f16", "bf16")):
    """Compile every HIP translation unit for gfx950 in each flavour and link the shared libraries. Returns the default library's path."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for fl in flavours:
        os.makedirs(os.path.join(OBJ, fl), exist_ok=True)
        for s in SOURCES:
            src = os.path.join(CSRC, s)
            obj = os.path.join(OBJ, fl, s.replace(".hip", ".o"))
            if force or _stale(obj, [src] + hdrs):
                jobs.append([hipcc] + FLAGS + [FLAVOURS[fl][1], "-DRV_TU=" + s.replace(".hip", ""), "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)

    with ThreadPoolExecutor(max_workers=int(os.environ.get("REVISION_BUILD_JOBS", "7"))) as ex:
        list(ex.map(run, jobs))
    for fl in flavours:
        objs = [os.path.join(OBJ, fl, s.replace(".hip", ".o")) for s in SOURCES]
        lib = lib_path(fl)
        if force or _stale(lib, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    fl = tuple(a for a in sys.argv[1:] if a in FLAVOURS) or tuple(FLAVOURS)
    print(build_library(force="--force" in sys.argv, verbose=True, flavours=fl))
