"""Scheduling probe: two recursions per pair on two streams with their DECODE phases aligned (decode A waits, on the device,
until prefill B is done: hipStreamWaitValue32 / hipStreamWriteValue32 on a flag), against the free-running two-stream pipeline
of bench.py.  python -u tools/lockstep_probe.py [pairs]"""
import ctypes, os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revisionllm_amd import ops, parallel
from revisionllm_amd.eval import stage2
from revisionllm_amd.model import ReVisionLlamaForCausalLM
from revisionllm_amd.utils import synth

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls",
                                                        hierarchy=True, adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
m.engine.init_synthetic(seed=0)
m.generation_config.eos_token_id = None
tok = synth.FakeTokenizer()
feats = ops.init_hash_(torch.empty(100, 256, 768, dtype=torch.bfloat16, device=dev), "bench.feat.r0", 0, synth.SQRT3)
qf = ops.init_hash_(torch.empty(16, 768, dtype=torch.bfloat16, device=dev), "bench.q", 0, synth.SQRT3)
qc = ops.init_hash_(torch.empty(768, dtype=torch.float32, device=dev), "bench.qcls", 0, synth.SQRT3)
plan = stage2.plan_groups(100, 100)
perms = stage2.make_perms(plan, torch.Generator().manual_seed(0))
sent = ("a person opens the door and walks into the kitchen while another person is sitting at the table "
        "reading a newspaper and then both of them leave the room together")
st = parallel.HipStages(m, tok)
qs = [(qf, qc, sent)]
kw = dict(batch=100, perms=[perms], max_new_tokens=8)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
hiprt = ctypes.CDLL("libamdhip64.so")
flag = torch.zeros(64, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
state = {"slot": 0, "pair": 0, "gate": False}


def hook():
    if not state["gate"]:
        return
    s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = ctypes.c_void_p(flag.data_ptr())
    if state["slot"] == 1:
        rc = hiprt.hipStreamWriteValue32(s, p, ctypes.c_uint32(state["pair"]), ctypes.c_uint(0))
    else:
        rc = hiprt.hipStreamWaitValue32(s, p, ctypes.c_uint32(state["pair"]), ctypes.c_uint(0), ctypes.c_uint32(0xffffffff))  # flags 0 = GEQ
    if rc != 0:
        raise RuntimeError(f"hipStream*Value32 failed: {rc}")


m.after_prefill = hook


def launch(slot):
    state["slot"] = slot
    m.engine.slot = slot
    streams[slot].wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(streams[slot]):
        return parallel.launch_queries_sharded(st, tok, feats, 100, qs, **kw)


def run(n_pairs, gate):
    state["gate"] = gate
    pend, rec = [], None
    for _ in range(n_pairs):
        state["pair"] += 1
        pend.append(launch(0))
        pend.append(launch(1))
        while len(pend) > 2:
            rec = parallel.collect_queries(pend.pop(0))[0]
    while pend:
        rec = parallel.collect_queries(pend.pop(0))[0]
    return rec


for gate in (False, True, False, True):
    torch.manual_seed(0)
    run(3, gate)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = run(pairs, gate)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"decode phases aligned = {gate}: {dt / (2 * pairs) * 1e3:.2f} ms per recursion, {200 * pairs / dt:.0f} segments/s  answers[0] {rec['answers'][0][:40]}", flush=True)
