"""Do two batched prefill passes of G / 2 prefills each, on two streams with half the CUs each, beat one pass of G prefills on all CUs?
(Their epilogues, prologues, attention and norms would overlap the other stream's main loops.)  python tools/two_stream_probe.py [G] [cus]"""
import sys
import time
import torch
from revisionllm_amd import engine
from revisionllm_amd.utils import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CUS = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
shape = synth.LlamaShape(layers=8, vocab=32000)
B, P0, S, Smax, D = 7, 32, 139, 192, 4096


def make(gate, cus):
    e = engine.Engine(shape, adapter_text=False, device=dev, gate=gate)
    e.init_synthetic(seed=3, llm=True, clip=False)
    if cus:
        e.set_option("gemm_cus", cus)
    return e


def run(e, g, pool, h):
    e.llm_prefill_pool_groups(h, g, B, P0, pool, g * B, [B * i for i in range(g)], Smax)


e0 = make(None, 0)
pool0, _ = e0.new_kv_pool(G * B, Smax)
h0 = torch.randn(G * (P0 + B * S), D, device=dev) * 0.02
for _ in range(3):
    run(e0, G, pool0, h0.clone())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run(e0, G, pool0, h0.clone())
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / 10 * 1e3
print(f"one pass of {G} prefills, all CUs: {one:.3f} ms", flush=True)

es = [make(engine.PersistGate(), CUS) for _ in range(2)]
pools = [e.new_kv_pool(G // 2 * B, Smax)[0] for e in es]
hs = [torch.randn(G // 2 * (P0 + B * S), D, device=dev) * 0.02 for _ in es]
streams = [torch.cuda.Stream(dev) for _ in es]
torch.cuda.synchronize()


def pair():
    for e, p, h, s in zip(es, pools, hs, streams):
        with torch.cuda.stream(s):
            run(e, G // 2, p, h.clone())


for _ in range(3):
    pair()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    pair()
torch.cuda.synchronize()
two = (time.perf_counter() - t0) / 10 * 1e3
print(f"two passes of {G // 2} prefills on two streams, {CUS} CUs each: {two:.3f} ms per pair  ({one / two:.3f} x)", flush=True)
# and the halves one after the other on all CUs (what the pair costs without overlap)
for e in es:
    e.set_option("gemm_cus", 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    for e, p, h in zip(es, pools, hs):
        run(e, G // 2, p, h.clone())
torch.cuda.synchronize()
print(f"the two halves one after the other, all CUs: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per pair", flush=True)
