"""Layer-0 K / V^T caches after a 7-call shared-prefix prefill: ring kernel (variant 6) vs the persistent ping-pong QKV
(variant 2, 192-column panels + fused RoPE epilogue).  The QKV sums are whole-panel in both, so the caches must be bit-equal."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import hip, engine
from revisionllm_amd.utils import synth
dev = torch.device("cuda:0")
shape = synth.LlamaShape(hidden=4096, inter=11008, layers=1, heads=32, vocab=32000)
eng = engine.Engine(shape, device="cuda:0")
eng.init_synthetic(seed=1)
B, S, P0 = 7, 171, 32
M = P0 + B * (S - P0)
torch.manual_seed(0)
h0 = torch.randn(M, 4096, device=dev) * 0.02
out = {}
for v in (6, 2):
    hip.lib().rv_set_gemm_tile_variant(v)
    kv, Smax = eng.new_kv(B, S + 8, reuse=False)
    logits = eng.llm_prefill_shared(h0.clone(), B, P0, kv, Smax)
    per = B * 32 * Smax * 128
    K = kv[:per].view(B, 32, Smax, 128)[:, :, :S].clone()
    Vt = kv[per:2 * per].view(B, 32, 128, Smax)[..., :S].clone()
    out[v] = (K, Vt, logits.clone())
hip.lib().rv_set_gemm_tile_variant(2)
print("K equal", torch.equal(out[6][0], out[2][0]), "V^T equal", torch.equal(out[6][1], out[2][1]),
      "max |dK|", (out[6][0].float() - out[2][0].float()).abs().max().item(), "max |dV|", (out[6][1].float() - out[2][1].float()).abs().max().item())
d = (out[6][2] - out[2][2]).abs().max().item()
print("logits max diff", d, "of", out[6][2].abs().max().item())
