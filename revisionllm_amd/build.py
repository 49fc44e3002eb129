"""Build librevision_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so ships with the tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "librevision_hip.so")
SOURCES = ["error.hip", "init.hip", "gemm.hip", "gemm_pp.hip", "gemm_arows.hip", "gemm_rows.hip", "gemm_rows_p1.hip", "gemm_rows_p2.hip", "gemm_rows_p3.hip", "rowops.hip", "attention.hip", "sample.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", "gemv_finish.h", "gemm_rows.hip", os.path.join("..", "..", "include", "revision_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=default", "-Wall", "-Wno-unused-function",
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


def build_library(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
