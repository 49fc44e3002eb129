import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace
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
L = lambda: parallel.launch_query_sharded(st, tok, feats, 100, qf, qc, sent, batch=100, perms=perms, max_new_tokens=8)
for _ in range(3): parallel.collect_query(L())
torch.cuda.synchronize()
for rep in range(2):
    tl, tc = [], []
    t00 = time.perf_counter(); pend = None
    for i in range(6):
        t0 = time.perf_counter(); nxt = L(); t1 = time.perf_counter()
        if pend is not None: parallel.collect_query(pend)
        t2 = time.perf_counter(); pend = nxt
        tl.append((t1 - t0) * 1e3); tc.append((t2 - t1) * 1e3)
    parallel.collect_query(pend); torch.cuda.synchronize()
    print("pipelined: total/6 %.2f ms; launch host ms" % ((time.perf_counter() - t00) / 6 * 1e3), [round(x, 1) for x in tl], "collect ms", [round(x, 1) for x in tc], flush=True)
    t00 = time.perf_counter()
    for i in range(6): parallel.collect_query(L())
    torch.cuda.synchronize()
    print("sequential: total/6 %.2f ms" % ((time.perf_counter() - t00) / 6 * 1e3), flush=True)
