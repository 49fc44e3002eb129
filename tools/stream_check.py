"""Race check for recursions in flight: the bench's step, N times one at a time on the default stream, then the same N steps
pipelined on alternating HIP streams (device-side torch.rand draws reseeded before each mode).  Every record must be equal.

    python -u tools/stream_check.py [steps] [streams] [rounds]
"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda:0")
    from revisionllm_amd import ops, parallel
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    model = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
    model.get_model().initialize_vision_modules(SimpleNamespace(
        clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls", hierarchy=True,
        adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
    model.engine.init_synthetic(seed=0, llm=True, clip=True, fp8_decode=bool(os.environ.get('REVISION_FP8')))
    model.engine.set_option('fp8_decode', int(os.environ.get('REVISION_FP8_ON', '0')))
    model.generation_config.eos_token_id = None
    tok = synth.FakeTokenizer()
    feats = ops.init_hash_(torch.empty(100, 256, 768, dtype=torch.bfloat16, device=dev), "bench.feat.r0", 0, synth.SQRT3)
    qf = ops.init_hash_(torch.empty(16, 768, dtype=torch.bfloat16, device=dev), "bench.q", 0, synth.SQRT3)
    qc = ops.init_hash_(torch.empty(768, dtype=torch.float32, device=dev), "bench.qcls", 0, synth.SQRT3)
    plan = stage2.plan_groups(100, 100)
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(0))
    sentence = ("a person opens the door and walks into the kitchen while another person is sitting at the table "
                "reading a newspaper and then both of them leave the room together")
    stages = parallel.HipStages(model, tok)
    qs = [(qf, qc, sentence)]
    kw = dict(batch=100, perms=[perms], max_new_tokens=8)
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]

    def key(r):
        return (r["answers"], r["max_entropy"], r["mean_entropy"], r["score_cos"])

    def sequential():
        torch.manual_seed(1)
        model.engine.slot = 0
        return [key(parallel.run_queries_sharded(stages, tok, feats, 100, qs, **kw)[0]) for _ in range(steps)]

    def pipelined():
        torch.manual_seed(1)
        out, pend = [], []
        for i in range(steps):
            model.engine.slot = i % nstreams
            with torch.cuda.stream(streams[i % nstreams]):
                pend.append(parallel.launch_queries_sharded(stages, tok, feats, 100, qs, **kw))
            if len(pend) > nstreams:
                out.append(key(parallel.collect_queries(pend.pop(0))[0]))
        while pend:
            out.append(key(parallel.collect_queries(pend.pop(0))[0]))
        model.engine.slot = 0
        return out

    ref = sequential()
    torch.cuda.synchronize()
    bad = 0
    for r in range(rounds):
        again = sequential()
        torch.cuda.synchronize()
        par = pipelined()
        torch.cuda.synchronize()
        ds = [i for i in range(steps) if again[i] != ref[i]]
        dp = [i for i in range(steps) if par[i] != ref[i]]
        print(f"round {r}: sequential rerun differs at steps {ds}; pipelined differs at steps {dp}", flush=True)
        for i in dp[:2]:
            a, b = ref[i], par[i]
            print("   answers equal:", a[0] == b[0], "| first differing answer:",
                  next(((j, x, y) for j, (x, y) in enumerate(zip(a[0], b[0])) if x != y), None), flush=True)
        bad += len(ds) + len(dp)
        print("   last step answers[:2]:", ref[-1][0][:2], "|", par[-1][0][:2], flush=True)
    print("RESULT", "OK" if bad == 0 else f"{bad} differing records")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
