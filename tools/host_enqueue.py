"""Host time to ENQUEUE one stage-2 recursion (launch_queries_sharded returns without waiting) vs. its device time."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import ops, parallel
from revisionllm_amd.eval import stage2
from revisionllm_amd.model import ReVisionLlamaForCausalLM
from revisionllm_amd.utils import synth

dev = torch.device("cuda:0")
m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls",
                                                        hierarchy=True, adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
m.engine.init_synthetic(seed=0)
m.generation_config.eos_token_id = None
tok = synth.FakeTokenizer()
feats = ops.init_hash_(torch.empty(100, 256, 768, dtype=torch.bfloat16, device=dev), "f", 0, synth.SQRT3)
qf = ops.init_hash_(torch.empty(16, 768, dtype=torch.bfloat16, device=dev), "q", 0, synth.SQRT3)
qc = ops.init_hash_(torch.empty(768, dtype=torch.float32, device=dev), "c", 0, synth.SQRT3)
plan = stage2.plan_groups(100, 100)
perms = stage2.make_perms(plan, torch.Generator().manual_seed(0))
st = parallel.HipStages(m, tok)
sent = "a person opens the door and walks into the kitchen while another person is sitting at the table reading a newspaper and then both of them leave the room together"
kw = dict(batch=100, perms=[perms], max_new_tokens=8)
for _ in range(3):
    parallel.run_queries_sharded(st, tok, feats, 100, [(qf, qc, sent)], **kw)
torch.cuda.synchronize()
enq, col = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = parallel.launch_queries_sharded(st, tok, feats, 100, [(qf, qc, sent)], **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    parallel.collect_queries(p)
    t3 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); col.append((t3 - t2) * 1e3)
    dev_ms = (t2 - t0) * 1e3
print("host enqueue ms per recursion:", [round(x, 1) for x in enq], "| collect ms:", [round(x, 2) for x in col], "| enqueue+device ms (last):", round(dev_ms, 1))
