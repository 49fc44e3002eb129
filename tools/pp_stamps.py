"""Per-slot cycle stamps of the ping-pong main loop (a PP_ABL & 32 probe build named by REVISION_HIP_LIB: tools/pp_probe.sh build 32): mean cycles of each of
the 8 barrier-to-barrier slots of a k-tile, per M-group.  python tools/pp_stamps.py [M N K]"""
import ctypes
import sys
import numpy as np
import torch
from revisionllm_amd import hip, ops

dev = torch.device("cuda:0")
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8040, 4096, 4096)
DT = hip.op_dtype()
lib = hip.lib()
raw = ctypes.CDLL(hip.LIB_PATHS[hip.flavour()])
raw.rv_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
ws_ = [ops.pack_fragments((torch.randn(N, K, device=dev) * 0.05).to(DT)) for _ in range(3)]
out = torch.empty(M, N, dtype=DT, device=dev)
ws = ops.stream_k_workspace(dev)


def run(i):
    hip.check(lib.rv_gemm(None, hip.ptr(x), K, hip.ptr(ws_[i % 3]), K, 1, None, None, 0, hip.ptr(out), N, hip.dtype_code(out), 0, M, N, K,
                          hip.ptr(ws), ws.numel(), hip.stream()), "rv_gemm")


for i in range(300):
    run(i)
torch.cuda.synchronize()
assert raw.rv_pp_stamps(None, 1) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(300):
    run(i)
b.record()
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 9 + 256 * 4, dtype=np.uint64)
assert raw.rv_pp_stamps(buf.ctypes.data, 0) == 0
ext = buf[256 * 2 * 9:].reshape(256, 4).astype(np.float64)
buf = buf[:256 * 2 * 9].reshape(256, 2, 9).astype(np.float64)
print(f"M={M} N={N} K={K}: {a.elapsed_time(b) / 300 * 1e3:.1f} us per launch")
items = 300.0 * 256
calls = buf[:, 0, 8].sum() / K * 64       # main-loop calls (k-tiles / k-tiles per whole panel)
print(f"per main-loop call: prologue {ext[:, 0].sum() / max(calls, 1):8.0f} cycles, whole loop {ext[:, 1].sum() / max(calls, 1):9.0f};  per whole-panel epilogue {ext[:, 2].sum() / max(ext[:, 3].sum(), 1):8.0f} cycles ({int(ext[:, 3].sum())} epilogues, {int(calls)} calls)")
names = ["own memory part 0", "own compute part 0", "own memory part 1", "own compute part 1", "own memory part 2", "own compute part 2", "own memory part 3", "own compute part 3"]
for g in (0, 1):
    tiles = buf[:, g, 8].sum()
    per = buf[:, g, :8].sum(axis=0) / max(tiles, 1)
    # stamp i sits behind the barrier that ends the wave's own part i (plus the s_memtime round trip, which waits for the wave's outstanding LDS reads)
    print(f"group {g}: k-tiles {int(tiles)}  cycles per k-tile {per.sum():7.1f}")
    for i in range(8):
        print(f"   {names[i]:20s} {per[i]:7.1f}")
