"""Isolated timing of merged decode steps (rv_llm_decode_rows) at 7B shapes for several row counts:
    python tools/decode_rows_time.py [rows ...]
Prints ms per step and the implied weight-streaming rate (13.2 GB of bf16 weights per step)."""
import sys
import torch
from revisionllm_amd import engine
from revisionllm_amd.utils import synth

fp8 = "--fp8" in sys.argv
rows = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [7, 14, 16, 21, 28, 32]
eng = engine.Engine(synth.LlamaShape(), adapter_text=False, device="cuda:0")
eng.init_synthetic(seed=0, llm=True, clip=False, fp8_decode=fp8)
for a in sys.argv:
    if a.startswith("--fill="):
        eng.set_option("rows_fill", int(a.split("=")[1]))
    if a == "--one-item-per-workgroup":
        eng.set_option("rows_persistent", 0)
    if a == "--single":
        eng.set_option("rows_single", 1)
    if a.startswith("--spread="):
        eng.set_option("rows_spread", int(a.split("=")[1]))
D, V = eng.shape.hidden, eng.shape.vocab
Smax = 256
wbytes = sum(t.numel() * t.element_size() for t in eng._llm_tensors()) if hasattr(eng, "_llm_tensors") else 13.2e9
for R in rows:
    kv, sm = eng.new_kv_pool(R, Smax)
    pos = torch.full((R,), 180, dtype=torch.int32, device="cuda:0")
    # --share: rows in groups of 7 (the calls of one recursion) share a 32-position prompt prefix, read from the group's first row
    share = torch.tensor([(r // 7 * 7) | (32 << 16) for r in range(R)], dtype=torch.int32, device="cuda:0") if "--share" in sys.argv else None
    h0 = torch.randn(R, D, device="cuda:0") * 0.02
    logits = torch.empty(R, V, device="cuda:0")
    for _ in range(3):
        eng.llm_decode_rows(h0.clone(), pos, kv, sm, logits=logits, row_share=share)
    torch.cuda.synchronize()
    n = 20
    hs = [h0.clone() for _ in range(n)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        eng.llm_decode_rows(hs[i], pos, kv, sm, logits=logits, row_share=share)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{'fp8 ' if fp8 else ''}rows {R:2d}: {ms:.3f} ms/step  {wbytes / ms / 1e9:.2f} TB/s  {ms / R * 1e3:.1f} us/row", flush=True)
    if "--graph" in sys.argv:
        # the same step captured into ONE hipGraph (torch.cuda.graph on a side stream; the library launches on torch's current stream) and replayed:
        # what is left of a step when the ~165 launches cost the host nothing and the device no front-end gaps
        g = torch.cuda.CUDAGraph()
        hg = h0.clone()
        with torch.cuda.graph(g):
            eng.llm_decode_rows(hg, pos, kv, sm, logits=logits, row_share=share)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for i in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        mg = e0.elapsed_time(e1) / n
        print(f"rows {R:2d}: {mg:.3f} ms/step replayed as one hipGraph ({mg / ms:.3f} x the stream-launched step)", flush=True)
