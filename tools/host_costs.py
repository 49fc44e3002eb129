"""Host-side enqueue cost of the pieces of the default pipeline (no device waits inside the timed calls): one recursion launch up to
its prefill ticket, one batched prefill pass, one merged decode step.  If their sum per recursion approaches the ~20 ms the GPU needs,
the scheduler - not the kernels - sets the rate."""
import time
from types import SimpleNamespace
import torch
from revisionllm_amd import ops, parallel, sched, serve
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
server = serve.DecodeServer(m, rows=56, smax=192, gmax=16, pools=2, gang=True, prefill_batch=4)
st.server = server
sent = "a person opens the door and walks into the kitchen while another person is sitting at the table reading a newspaper"
streams = [torch.cuda.Stream(dev) for _ in range(16)]
inter = sched.Interleaver(servers=[server])


def launch(k):
    g = lambda task: parallel.launch_queries_sharded_steps(st, tok, feats, 100, [(qf, qc, sent)], turn=task, batch=100, perms=[perms], max_new_tokens=8)
    return sched.Task(g, streams[k % 16], m.engine, k % 16)


for rnd in range(3):
    t_launch, tasks = [], []
    for k in range(16):
        t0 = time.perf_counter()
        tasks.append(inter.add(launch(k)))
        t_launch.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    res = [inter.finish(t) for t in tasks]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0 + sum(t_launch)
    print(f"round {rnd}: launch+first pump per task {1e3 * sum(t_launch) / 16:.2f} ms (max {1e3 * max(t_launch):.2f}); 16 recursions wall {1e3 * wall:.1f} ms "
          f"= {1e3 * wall / 16:.2f} ms each", flush=True)
# isolated enqueue costs
import cProfile, pstats
pr = cProfile.Profile()
tasks = []
pr.enable()
for k in range(16):
    tasks.append(inter.add(launch(k)))
res = [inter.finish(t) for t in tasks]
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
