"""Stamps of the persistent GEMMs inside batched prefill passes (a PP_ABL & 32 probe build named by REVISION_HIP_LIB): per kernel kind the mean cycles of
a work item's main loop, the k-split loop's prologue and the whole-panel epilogue.  python tools/pp_pass_stamps.py [G]"""
import ctypes
import runpy
import sys
import numpy as np
from revisionllm_amd import hip

raw = ctypes.CDLL(hip.LIB_PATHS[hip.flavour()])
raw.rv_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
sys.argv = ["prefill_prof.py"] + sys.argv[1:]
runpy.run_path("tools/prefill_prof.py", run_name="__main__")
import torch
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 9 + 3 * 256 * 4, dtype=np.uint64)
assert raw.rv_pp_stamps(buf.ctypes.data, 0) == 0
ext = buf[256 * 2 * 9:].reshape(3, 256, 4).astype(np.float64)
for k, name in enumerate(("gate/up", "o / down", "q/k/v")):
    n = max(ext[k, :, 3].sum(), 1)
    print(f"{name:9s} whole-panel items {int(n):7d}: main loop {ext[k, :, 1].sum() / n:9.0f} cycles (incl. prologue {ext[k, :, 0].sum() / n:7.0f}), epilogue {ext[k, :, 2].sum() / n:7.0f}")
