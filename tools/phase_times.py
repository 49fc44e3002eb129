"""Wall-clock split of one stage-2 recursion on the GPU (HIP events): adapter / prefill / decode step / sampling."""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import ops  # noqa: E402
from revisionllm_amd.model import ReVisionLlamaForCausalLM  # noqa: E402
from revisionllm_amd.utils import synth  # noqa: E402


def ev(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
    m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True,
                                                            clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768,
                                                            pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
    m.engine.init_synthetic(seed=0)
    eng = m.engine
    B, S = 7, int(sys.argv[1]) if len(sys.argv) > 1 else 171
    feats = torch.randn(100, 256, 768, device=dev).to(torch.bfloat16)
    qf = torch.randn(1, 16, 768, device=dev).to(torch.bfloat16)
    print(f"adapter 100x256x768      : {ev(lambda: eng.clip_encoder(feats, qf, torch.ones(1, 16), 'cls')):8.3f} ms")
    kv, Smax = eng.new_kv(B, S + 64)
    h0 = torch.randn(B, S, 4096, device=dev) * 0.02
    print(f"prefill B={B} S={S}        : {ev(lambda: eng.llm_forward(h0.clone(), 0, kv, Smax)):8.3f} ms")
    P0 = 37
    hs = torch.randn(P0 + B * (S - P0), 4096, device=dev) * 0.02
    print(f"prefill shared P0={P0}       : {ev(lambda: eng.llm_prefill_shared(hs.clone(), B, P0, kv, Smax)):8.3f} ms")
    h1 = torch.randn(B, 1, 4096, device=dev) * 0.02
    print(f"decode step B={B}          : {ev(lambda: eng.llm_forward(h1.clone(), S, kv, Smax), n=20):8.3f} ms")
    logits = torch.randn(B, 32000, device=dev)
    u = torch.rand(B, device=dev)
    print(f"sample                    : {ev(lambda: ops.sample(logits, u, True, 0.05, 50, 1.0), n=20):8.3f} ms")
    ids = torch.randint(3, 30000, (B, 1), device=dev).int()
    print(f"splice_embed decode       : {ev(lambda: eng.splice_embed(ids, None), n=20):8.3f} ms")


if __name__ == "__main__":
    main()
