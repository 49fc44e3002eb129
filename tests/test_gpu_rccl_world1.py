"""RCCL next to the persistent stream-K GEMMs on ONE GPU.

The segment-parallel recursion issues two all-gathers per recursion (CLS rows, proposals) bracketed by the device's persistent-launch
gate (``parallel._gated`` / ``engine.PersistGate``) so that an RCCL kernel never sits on CUs a one-workgroup-per-CU stream-K GEMM of
another stream is waiting for.  Everything else about that code has only ever met gloo on CPU stand-in stages.  Here a REAL ``nccl``
process group of world size 1 is initialised and ``REVISION_FORCE_COLLECTIVES=1`` makes the driver issue both exchanges (through RCCL:
``all_gather_into_tensor``) for every recursion while other recursions' batched prefill passes (4 x 1005-row persistent GEMMs) and merged
decode steps run on other streams.  Asserted: every record equals the one produced without any collective, and nothing stalls.
"""
import os
import socket
import time

import pytest
import torch

from helpers import SEED, fl, op, tol

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_all_gathers_next_to_persistent_prefill_gemms(monkeypatch):
    import torch.distributed as dist
    from types import SimpleNamespace
    from revisionllm_amd import ops, parallel, sched, serve
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    dev = torch.device("cuda:0")
    m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device=dev)
    m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, clip_adapter_text=True, clip_adapter_feature="cls",
                                                            hierarchy=True, adapter_input_dim=768, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None))
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True)
    m.generation_config.eos_token_id = None
    tok = synth.FakeTokenizer()
    W, Tn, Lq, G, copies = 100, 256, 16, 6, 8
    sets = []
    for k in range(copies):
        feat = ops.init_hash_(torch.empty(W, Tn, 768, dtype=op(), device=dev), f"rccl1.feat{k}", SEED, synth.SQRT3)
        qf = ops.init_hash_(torch.empty(Lq, 768, dtype=op(), device=dev), f"rccl1.q{k}", SEED, synth.SQRT3)
        qc = ops.init_hash_(torch.empty(768, dtype=torch.float32, device=dev), f"rccl1.qc{k}", SEED, synth.SQRT3)
        g = torch.Generator().manual_seed(1000 + k)
        sets.append((feat, qf, qc, stage2.make_perms(stage2.plan_groups(W, 100), g, W=W)))
    u = torch.rand(G, 7, generator=torch.Generator().manual_seed(5)).to(dev)
    sentence = "a person opens the door and walks into the kitchen"

    def pipeline(group):
        st = parallel.HipStages(m, tok)
        server = serve.DecodeServer(m, rows=56, smax=192, gmax=16, pools=2, gang=True, prefill_batch=4)
        st.server = server
        streams = [torch.cuda.Stream(dev) for _ in range(copies)]
        torch.cuda.synchronize()
        inter = sched.Interleaver(servers=[server])
        tasks = []
        for i, (feat, qf, qc, perms) in enumerate(sets):
            kw = dict(batch=100, perms=[perms], max_new_tokens=G, uniforms=u, group=group)
            tasks.append(inter.add(sched.Task(lambda t, feat=feat, qf=qf, qc=qc, kw=kw: parallel.launch_queries_sharded_steps(st, tok, feat, W, [(qf, qc, sentence)], turn=t, **kw),
                                              streams[i], m.engine, i)))
        recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
        m.engine.slot = 0
        torch.cuda.synchronize()
        assert server.pf_batches < copies            # the prefills really rode together: persistent GEMMs of several recursions' rows
        return recs

    want = pipeline(parallel.LOCAL)                  # no process group involved at all
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        monkeypatch.setenv("REVISION_FORCE_COLLECTIVES", "1")
        calls = {"n": 0}
        real = dist.all_gather_into_tensor

        def counted(*a, **kw):
            calls["n"] += 1
            return real(*a, **kw)
        monkeypatch.setattr(dist, "all_gather_into_tensor", counted)
        t0 = time.perf_counter()
        got = pipeline(None)                         # the default group: exchanges forced through RCCL, gated against the GEMMs
        dt = time.perf_counter() - t0
        assert calls["n"] == copies * 4              # exchange 1: CLS rows + cosine scores; exchange 2: tokens + entropies - per recursion
        for a, b in zip(want, got):
            for k in ("answers", "max_entropy", "mean_entropy", "score_cos", "starts", "hierarchy_zooms"):
                assert a[k] == b[k], k
        assert dt < 60.0, f"{copies} recursions with forced RCCL exchanges took {dt:.1f} s: a stall"
        print(f"\n[RCCL world 1] {copies} recursions in flight, {calls['n']} all_gather_into_tensor calls through RCCL next to batched stream-K prefills: "
              f"records equal the collective-free run; {dt * 1e3:.0f} ms")
    finally:
        monkeypatch.delenv("REVISION_FORCE_COLLECTIVES", raising=False)
        dist.destroy_process_group()
