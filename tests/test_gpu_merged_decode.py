"""Merged decode steps: several generates share one KV pool and their KV-cached decode steps run as ONE pass over the weights
(rows at different positions, up to 32 rows).  A row's results must not depend on what it is batched with."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import SEED, T, feats, fl, op, rel_err, tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,act", [(21, 4096, 4096, 0), (32, 22016, 4096, 2), (17, 4096, 11008, 0), (28, 32000, 4096, 0), (21, 512, 1408 // 128 * 128, 0)])
def test_weight_streaming_kernel_17_to_32_rows(M, N, K, act):
    """rv_gemm with 17 .. 32 rows: the weight-streaming kernel with two MFMA column blocks per weight fragment.  Rows 0 .. 15 are
    BIT-identical to the 16-row launch of the same rows, rows 16 .. to a 16-row launch of THOSE rows; vs float64; fp8 weights too."""
    from revisionllm_amd import hip, ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(M * 7 + N)
    a = (torch.randn(M, K, generator=g) * 0.5).to(op()).to(dev)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(op()).to(dev)
    wp = ops.pack_fragments(w)
    res = torch.randn(M, N // 2 if act == 2 else N, generator=g).to(dev) if act == 0 else None
    od = op() if act == 2 else torch.float32
    y = ops.gemm(a, wp, residual=res, out_dtype=od, act=act, w_packed=True)
    lo = ops.gemm(a[:16], wp, residual=None if res is None else res[:16], out_dtype=od, act=act, w_packed=True)
    hi = ops.gemm(a[16:], wp, residual=None if res is None else res[16:], out_dtype=od, act=act, w_packed=True)
    assert torch.equal(y[:16], lo) and torch.equal(y[16:], hi)
    z = a.double() @ w.double().t()
    if act == 2:
        z3 = z.view(M, N // 32, 2, 16)
        z = (torch.nn.functional.silu(z3[:, :, 0]) * z3[:, :, 1]).reshape(M, N // 2)
    else:
        z = z + res.double()
    assert rel_err(y.float().cpu(), z.cpu()) < (1.5e-2 if act == 2 else 1e-4)
    assert torch.equal(y, ops.gemm(a, w, residual=res, out_dtype=od, act=act))              # row-major weights: same sums
    w8, sc = ops.pack_fragments_fp8(w.float())
    y8 = ops.gemv_fp8(a, w8, sc, residual=res, out_dtype=od, act=act)
    lo8 = ops.gemv_fp8(a[:16], w8, sc, residual=None if res is None else res[:16], out_dtype=od, act=act)
    assert torch.equal(y8[:16], lo8)


def _engine(layers=2, vocab=2048):
    from revisionllm_amd import engine
    from revisionllm_amd.utils import synth
    eng = engine.Engine(synth.LlamaShape(layers=layers, vocab=vocab), adapter_text=False, device="cuda:0")
    eng.init_synthetic(seed=SEED, llm=True, clip=False)
    return eng


def test_merged_decode_rows_equal_separate_generates():
    """Three 'generates' (7, 7 and 3 rows; prompts of different lengths: 171, 171 and 140) prefilled into ONE pool and decoded by merged
    steps (17 then, with the third group inactive, 14 active rows) against the same three run on their own caches: logits of every
    step BIT-identical, caches bit-identical; inactive rows leave their cache untouched."""
    eng = _engine()
    D, V, H = 4096, 2048, 32
    groups = [(7, 171), (7, 171), (3, 140)]
    R, Smax = 32, 192
    g = torch.Generator().manual_seed(3)
    hs = [torch.randn(B, S, D, generator=g).mul(0.02).cuda() for B, S in groups]
    steps = 4
    toks = [[torch.randn(B, 1, D, generator=g).mul(0.02).cuda() for _ in range(steps)] for B, S in groups]
    # --- reference: every group on its own cache
    ref_logits, ref_kv = [], []
    for (B, S), h, tk in zip(groups, hs, toks):
        kv, sm = eng.new_kv(B, Smax, reuse=False)
        assert sm == Smax
        out = [eng.llm_forward(h.clone(), 0, kv, Smax)]
        for s_ in range(steps):
            out.append(eng.llm_forward(tk[s_].clone(), S + s_, kv, Smax))
        ref_logits.append(out)
        ref_kv.append(kv)
    # --- pool: rows 0-6, 7-13, 14-16 (+ 15 unused rows)
    pool, sm = eng.new_kv_pool(R, Smax)
    row0 = [0, 7, 14]
    pos = torch.full((R,), -1, dtype=torch.int32, device="cuda:0")
    first = []
    for (B, S), h, r0 in zip(groups, hs, row0):
        first.append(eng.llm_prefill_pool(h.clone().view(B * S, D), B, 0, pool, R, r0, Smax))
        pos[r0:r0 + B] = S
    for gi in range(3):
        assert torch.equal(first[gi], ref_logits[gi][0])
    for s_ in range(steps):
        active = [0, 1, 2] if s_ < 2 else [0, 1]          # the third group stops after two steps
        hrow = torch.zeros(R, D, device="cuda:0")
        p = torch.full((R,), -1, dtype=torch.int32, device="cuda:0")
        for gi in active:
            B, S = groups[gi]
            hrow[row0[gi]:row0[gi] + B] = toks[gi][s_][:, 0]
            p[row0[gi]:row0[gi] + B] = S + s_
        logits = eng.llm_decode_rows(hrow, p, pool, Smax)
        for gi in active:
            B = groups[gi][0]
            assert torch.equal(logits[row0[gi]:row0[gi] + B], ref_logits[gi][1 + s_]), (s_, gi)
    # caches: K rows / V^T columns of every group equal its own-cache run up to the positions it wrote
    L = 2
    halfp = pool.numel() // 2
    kp = pool[:halfp].view(L, R, H, Smax, 128)
    vp = eng.vt_logical(pool[halfp:], L, R, H, Smax=Smax)          # (blocked by 8 positions on the device)
    for gi, ((B, S), r0) in enumerate(zip(groups, row0)):
        n = S + (2 if gi == 2 else steps)
        half = ref_kv[gi].numel() // 2
        kr = ref_kv[gi][:half].view(L, B, H, Smax, 128)
        vr = eng.vt_logical(ref_kv[gi][half:], L, B, H, Smax=Smax)
        assert torch.equal(kp[:, r0:r0 + B, :, :n], kr[:, :, :, :n]) and torch.equal(vp[:, r0:r0 + B, :, :, :n], vr[..., :n])
    assert (kp[:, 14:17, :, 142:] == 0).all()                 # the stopped group's rows were not written after it went inactive
    assert (kp[:, 17:] == 0).all() and (vp[:, 17:] == 0).all()    # unused rows never touched


@pytest.mark.parametrize("n_groups,R", [(6, 48), (9, 64), (10, 70), (11, 80), (16, 112), (18, 128), (19, 133), (20, 140), (20, 144)])
def test_wide_pools_33_to_144_rows_equal_separate_generates(n_groups, R):
    """Merged decode steps with 33 .. 144 rows (the split-K kernel: activations shared through LDS, 1 - 8 workgroups per column group,
    partial planes folded as subtrees of the 8-way tree): every group's logits at every step BIT-identical to the same group decoded
    on its own cache by the <= 16-row kernel; groups at different positions, the last one leaving after two steps."""
    eng = _engine()
    D = 4096
    rng = np.random.RandomState(n_groups)
    groups = [(7, int(rng.randint(40, 120))) for _ in range(n_groups)]
    Smax = 160
    g = torch.Generator().manual_seed(5)
    hs = [torch.randn(B, S, D, generator=g).mul(0.02).cuda() for B, S in groups]
    steps = 3
    toks = [[torch.randn(B, 1, D, generator=g).mul(0.02).cuda() for _ in range(steps)] for B, S in groups]
    ref = []
    for (B, S), h, tk in zip(groups, hs, toks):
        kv, sm = eng.new_kv(B, Smax, reuse=False)
        out = [eng.llm_forward(h.clone(), 0, kv, Smax)]
        for s_ in range(steps):
            out.append(eng.llm_forward(tk[s_].clone(), S + s_, kv, Smax))
        ref.append(out)
    pool, sm = eng.new_kv_pool(R, Smax)
    row0 = [7 * i for i in range(n_groups)]
    for (B, S), h, r0, want in zip(groups, hs, row0, ref):
        assert torch.equal(eng.llm_prefill_pool(h.clone().view(B * S, D), B, 0, pool, R, r0, Smax), want[0])
    for s_ in range(steps):
        active = range(n_groups) if s_ < 2 else range(n_groups - 1)
        hrow = torch.zeros(R, D, device="cuda:0")
        p = torch.full((R,), -1, dtype=torch.int32, device="cuda:0")
        for gi in active:
            B, S = groups[gi]
            hrow[row0[gi]:row0[gi] + B] = toks[gi][s_][:, 0]
            p[row0[gi]:row0[gi] + B] = S + s_
        logits = eng.llm_decode_rows(hrow, p, pool, Smax)
        for gi in active:
            assert torch.equal(logits[row0[gi]:row0[gi] + 7], ref[gi][1 + s_]), (s_, gi)


@pytest.mark.parametrize("G,P0", [(3, 40), (2, 0), (4, 48), (6, 40)])       # (6 x 7 = 42 last rows: the last block's row form in two pieces of <= 32)
def test_batched_prefill_groups_match_separate_prefills(G, P0):
    """rv_llm_prefill_pool_groups: G prefills of identical geometry ([P0 shared prefix ; 7 x S'] each) in ONE pass against the same G
    prefills one at a time: logits and caches agree to GEMM summation order (the stream-K split points depend on the row count, so
    not bit for bit), and with tiny prefills (the 128-row tile kernel either way) exactly."""
    eng = _engine()
    D, H, L = 4096, 32, 2
    B, S, Smax, R = 7, 96, 160, 32 if G <= 4 else 64
    g = torch.Generator().manual_seed(11 + G)
    hs = [torch.randn(P0 + B * S, D, generator=g).mul(0.02).cuda() for _ in range(G)]
    row0 = [3 + 7 * i for i in range(G)]
    pool_a, _ = eng.new_kv_pool(R, Smax)
    pool_b, _ = eng.new_kv_pool(R, Smax)
    sep = [eng.llm_prefill_pool(h.clone(), B, P0, pool_a, R, r0, Smax).clone() for h, r0 in zip(hs, row0)]
    bat = eng.llm_prefill_pool_groups(torch.cat(hs).contiguous(), G, B, P0, pool_b, R, row0, Smax)
    for gi in range(G):
        assert rel_err(bat[gi * B:(gi + 1) * B].cpu(), sep[gi].cpu()) < 1e-2, gi       # (bf16 activations: a different f32 summation order moves roundings)
    half = pool_a.numel() // 2
    ka, kb = pool_a[:half].view(L, R, H, Smax, 128).float(), pool_b[:half].view(L, R, H, Smax, 128).float()
    va, vb = eng.vt_logical(pool_a[half:], L, R, H, Smax=Smax).float(), eng.vt_logical(pool_b[half:], L, R, H, Smax=Smax).float()
    n = P0 + S
    assert rel_err(kb[:, :, :, :n].cpu(), ka[:, :, :, :n].cpu()) < 1e-2 and rel_err(vb[..., :n].cpu(), va[..., :n].cpu()) < 1e-2
    assert (kb[:, :3] == 0).all() and (kb[:, 3 + 7 * G:] == 0).all() and (kb[:, :, :, n:] == 0).all()      # nothing outside the groups' rows / positions
    # one decode step from either pool: same logits to the same tolerance
    pos = torch.full((R,), -1, dtype=torch.int32, device="cuda:0")
    pos[3:3 + 7 * G] = n
    hrow = torch.randn(R, D, generator=g).mul(0.02).cuda()
    la = eng.llm_decode_rows(hrow.clone(), pos, pool_a, Smax)
    lb = eng.llm_decode_rows(hrow.clone(), pos, pool_b, Smax)
    assert rel_err(lb[3:3 + 7 * G].cpu(), la[3:3 + 7 * G].cpu()) < 1e-2


@pytest.mark.parametrize("G,B,P0,S", [(4, 7, 32, 139), (3, 7, 40, 96), (2, 7, 0, 96), (1, 5, 32, 171)])
def test_prefill_attention_with_lds_staged_keys_is_bit_identical(G, B, P0, S):
    """The prefill's causal attention with its key blocks staged in LDS once per 64-row workgroup (attention.hip attn_body_lds1, option ``attn_lds``) against the
    per-wave form of rounds 1 - 5: logits AND every byte of the KV pool equal - shared prefix + per-call rows (the pair launch), several prefills to a pass
    (groups), no shared prefix (the groups launch), a lone prefill."""
    eng = _engine()
    D, R, Smax = 4096, 64, 256
    g = torch.Generator().manual_seed(31 + G)
    hs = torch.randn(G * (P0 + B * S), D, generator=g).mul(0.02).cuda()
    row0 = [3 + B * i for i in range(G)]
    outs = []
    for v in (1, 0):
        eng.set_option("attn_lds", v)
        pool, _ = eng.new_kv_pool(R, Smax)
        lg = (eng.llm_prefill_pool_groups(hs.clone(), G, B, P0, pool, R, row0, Smax) if G > 1 else eng.llm_prefill_pool(hs.clone(), B, P0, pool, R, row0[0], Smax))
        outs.append((lg.clone(), pool.clone()))
    eng.set_option("attn_lds", 1)
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _tiny_model(parity=False):
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    m = ReVisionLlamaForCausalLM(synth.TINY, device="cuda:0")
    m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None,
                                                            pretrain_mm_mlp_adapter=None, clip_adapter_text=True, clip_adapter_feature="cls",
                                                            hierarchy=True, adapter_input_dim=768))
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True, parity=parity)
    if parity:
        m.engine.set_option("precision", 1)
    m.generation_config.eos_token_id = None
    return m


@pytest.mark.parametrize("G,B,P0,S", [(4, 7, 32, 139), (3, 7, 40, 96), (2, 7, 0, 96), (1, 5, 32, 171), (4, 1, 0, 327), (8, 7, 32, 139), (2, 3, 17, 61)])
def test_prefill_qkv_epilogue_through_lds_writes_the_same_bytes(G, B, P0, S):
    """The fused q / k / v epilogue of the persistent prefill GEMM staged through LDS (gemm_pp.hip pp_epilogue_rope_lds, option ``qkv_lds``: whole 128-byte row slabs
    of Q / K, 16-byte pieces of 8 positions of V^T, 2-byte stores only where a group of 8 positions is cut by a tile edge or a sequence end) against the per-lane
    stores of rounds 1 - 5: the logits and EVERY byte of the KV pool are equal - shared prefix, no prefix, one / several / eight prefills to a pass, one-row
    sequences of 327 positions, short odd geometry (sequence ends inside tiles, groups of positions cut everywhere)."""
    eng = _engine()
    D, R, Smax = 4096, 64, 352
    g = torch.Generator().manual_seed(41 + G + S)
    hs = torch.randn(G * (P0 + B * S), D, generator=g).mul(0.02).cuda()
    row0 = [1 + B * i for i in range(G)]
    outs = []
    for v in (1, 0):
        eng.set_option("qkv_lds", v)
        pool, _ = eng.new_kv_pool(R, Smax)
        lg = (eng.llm_prefill_pool_groups(hs.clone(), G, B, P0, pool, R, row0, Smax) if G > 1 else eng.llm_prefill_pool(hs.clone(), B, P0, pool, R, row0[0], Smax))
        outs.append((lg.clone(), pool.clone()))
    eng.set_option("qkv_lds", 1)
    assert torch.isfinite(outs[0][0]).all() and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].float().abs().max()) > 0


def test_batched_adapter_calls_equal_separate_ones():
    """``DecodeServer(encode_batch=4)``: the adapter calls of recursions in flight ride in ONE rv_clip_encoder call (a query per recursion, grouped fold
    GEMMs): every recursion's CLS rows equal its own call's up to the GEMM plans' summation order (at these few rows the K = 2048 FFN-2 takes a stream-K plan
    that depends on the total row count: one 16-bit rounding moves, measured 3.9e-4 of the largest element in fp16 - with or without the folded attention,
    i.e. a property of the engine at small row counts, not of the batching; at the bench's 100 x 256-frame recursions the rows are bit-identical) - whatever it
    was batched with (3 tickets: a full batch is not needed, ``flush`` launches what waits)."""
    from revisionllm_amd import serve
    m = _tiny_model()
    eng = m.engine
    srv = serve.DecodeServer(m, rows=32, smax=64, gmax=8, pools=2, gang=True, prefill_batch=1, encode_batch=4)
    feats_ = [feats(f"be.x{i}", (9, 32, 768), bf16=fl()).to(op()).cuda() for i in range(3)]
    qfs = [feats(f"be.q{i}", (6, 768), bf16=fl()).to(op()).cuda() for i in range(3)]
    want = [eng.clip_encoder(f, q[None], torch.ones(1, 6), "cls").clone() for f, q in zip(feats_, qfs)]
    tickets = [srv.submit_encode(f, q) for f, q in zip(feats_, qfs)]
    assert all(t.ready is None for t in tickets) and not srv.pump()          # a partial batch waits ...
    assert srv.flush() and all(t.ready is not None for t in tickets)          # ... until nothing else can move
    tickets[0].ready.synchronize()
    assert srv.enc_batches == 1 and srv.enc_tickets == 3
    for t, w in zip(tickets, want):
        assert t.cls.shape == w.shape and rel_err(t.cls.cpu(), w.cpu()) < tol(8e-3)


@pytest.mark.parametrize("n_passes,streams,pools,pbatch", [(3, 3, 1, 1), (5, 2, 1, 1), (7, 6, 2, 1), (9, 8, 3, 1), (7, 6, 2, 4)])
def test_recursions_through_the_decode_server_equal_sequential(n_passes, streams, pools, pbatch):
    """Several stage-2 recursions in flight on their own HIP streams, their generates decoding through ONE DecodeServer (shared KV
    pool, merged steps with rows at different positions, rows joining and leaving as prefills complete) against the same
    recursions run one after the other through the classic loop: records identical (answers, entropies, cosine scores).
    ``pools`` > 1: the gang policy (a pool is filled with four 8-row generates, sealed, and stepped with all 32 rows while the
    next recursions prefill into the other pool; the last, partly filled pool is run when nothing else is left).
    ``pbatch`` > 1: the waiting prefills of identical geometry ride in one pass (``rv_llm_prefill_pool_groups``); at these sizes every
    GEMM is on the 128-row tile kernel whatever the batch, so the records stay identical."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    tok = synth.FakeTokenizer(vocab=synth.TINY.vocab)
    st = parallel.HipStages(m, tok)
    W, batch = 13, 8
    feat = feats("s2.feat", (W, 16, 768), bf16=fl()).to(op()).cuda()
    qfs = [feats(f"ms.q{i}", (5 + i % 3, 768), bf16=fl()).to(op()).cuda() for i in range(n_passes)]
    qc = feats("s2.qc", (768,)).cuda()
    plan = stage2.plan_groups(W, batch)                      # 4 + 2 + 2 calls -> generates of 8 rows (same row count per level here)
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(1))
    unis = [torch.rand(6, len(plan), generator=torch.Generator().manual_seed(10 + i)) for i in range(n_passes)]
    kw = dict(batch=batch, perms=[perms], max_new_tokens=6)
    seq = [parallel.run_queries_sharded(st, tok, feat, W, [(qfs[i], qc, f"query {i}")], uniforms=unis[i], **kw)[0] for i in range(n_passes)]
    server = serve.DecodeServer(m, rows=32, smax=128, gmax=16, pools=pools, gang=pools > 1, prefill_batch=pbatch)
    st.server = server
    hs = [torch.cuda.Stream("cuda:0") for _ in range(streams)]
    torch.cuda.synchronize()
    inter = sched.Interleaver(servers=[server])
    pending, par = [], []
    for i in range(n_passes):
        pending.append(inter.add(sched.Task(lambda t, i=i: parallel.launch_queries_sharded_steps(st, tok, feat, W, [(qfs[i], qc, f"query {i}")],
                                                                                              uniforms=unis[i], turn=t, **kw),
                                            hs[i % streams], m.engine, i % streams)))
        if len(pending) > streams:
            par.append(parallel.collect_queries(inter.finish(pending.pop(0)))[0])
    while pending:
        par.append(parallel.collect_queries(inter.finish(pending.pop(0)))[0])
    m.engine.slot = 0
    assert server.steps_run > 0 and server.rows_served > server.steps_run * len(plan) * 0.99      # the merged steps really carried the rows
    if n_passes == 3:
        assert server.rows_served > server.steps_run * len(plan)                                   # ... of more than one recursion at a time
    if pbatch > 1:
        assert server.pf_tickets > server.pf_batches > 0                                          # some prefills really rode together
    if pools > 1:
        assert server.rows_served > server.steps_run * len(plan) * 1.5, (server.rows_served, server.steps_run)   # gangs (the three levels of a recursion are
                                                                                                   # sequential generates, so pools also run partly filled here)
    for a, b in zip(seq, par):
        assert a["answers"] == b["answers"] and a["max_entropy"] == b["max_entropy"] and a["mean_entropy"] == b["mean_entropy"]
        assert a["score_cos"] == b["score_cos"]
    assert not server.jobs and sum(n for _, n, _ in server.free) == 32 * pools                     # every row was given back


def test_ragged_generate_of_a_33_window_recursion_equals_the_two_generate_form():
    """A 33-window recursion at batch 33 has 9 calls: 8 present 32 video tokens, one 33.  Through a DecodeServer with batched prefills they now
    run as ONE generate of right-padded sequences (``rv_llm_prefill_pool_groups_ragged``: the head reads every sequence's last VALID row; every
    row then decodes from its own length) instead of two generates with two prefill passes.  Records of 4 recursions in flight: the ragged form
    against the two-generate form (``server.ragged = False``) and against the sequential classic loop - answers equal, entropies to 1e-5 (the
    prefill GEMMs see other row counts), and the ragged run really used one ticket per recursion."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    tok = synth.FakeTokenizer(vocab=synth.TINY.vocab)
    st = parallel.HipStages(m, tok)
    W, batch, n = 33, 33, 4
    plan = stage2.plan_groups(W, batch)
    assert len(plan) == 9
    feat = feats("rg.feat", (W, 16, 768), bf16=fl()).to(op()).cuda()
    qfs = [feats(f"rg.q{i}", (5 + i % 3, 768), bf16=fl()).to(op()).cuda() for i in range(n)]
    qc = feats("rg.qc", (768,)).cuda()
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(2), W=W)
    unis = [torch.rand(6, len(plan), generator=torch.Generator().manual_seed(20 + i)) for i in range(n)]
    kw = dict(batch=batch, perms=[perms], max_new_tokens=6)
    seq = [parallel.run_queries_sharded(st, tok, feat, W, [(qfs[i], qc, f"query {i}")], uniforms=unis[i], **kw)[0] for i in range(n)]

    def pipeline(ragged):
        server = serve.DecodeServer(m, rows=27, smax=160, gmax=16, pools=2, gang=True, prefill_batch=4)      # (<= 32 rows: the tiny model's K = 512 is below the wide kernel's)
        server.ragged = ragged
        st.server = server
        hs = [torch.cuda.Stream("cuda:0") for _ in range(n)]
        torch.cuda.synchronize()
        inter = sched.Interleaver(servers=[server])
        tasks = [inter.add(sched.Task(lambda t, i=i: parallel.launch_queries_sharded_steps(st, tok, feat, W, [(qfs[i], qc, f"query {i}")], uniforms=unis[i],
                                                                                           turn=t, **kw), hs[i], m.engine, i)) for i in range(n)]
        recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
        m.engine.slot = 0
        st.server = None
        assert not server.jobs and sum(k for _, k, _ in server.free) == 27 * 2
        return recs, server
    rag, sv_r = pipeline(True)
    two, sv_t = pipeline(False)
    assert sv_r.pf_tickets == n and sv_t.pf_tickets == 2 * n                     # one generate per recursion instead of two
    for a, b, c in zip(seq, rag, two):
        assert a["answers"] == b["answers"] == c["answers"]
        assert a["starts"] == b["starts"] and a["hierarchy_zooms"] == b["hierarchy_zooms"] and a["score_cos"] == b["score_cos"]
        for k in ("max_entropy", "mean_entropy"):
            assert rel_err(b[k], a[k]) < 1e-5 and rel_err(c[k], a[k]) < 1e-5, k


def test_ragged_generate_that_no_pool_can_take_falls_back_to_equal_geometry_groups():
    """whether a recursion completes must not depend on the DecodeServer's capacity.  The 9 calls of a 33-window recursion
    form a ragged batch (8 x 32 + 1 x 33 video tokens); with a server whose pools hold only 8 rows (``fits`` is false for the padded batch), with a
    server whose Smax is too short, and with no server at all, ``generate_steps`` runs them as equal-geometry sub-batches (the two-generate form
    of round 3) instead of raising - same answers, entropies to 1e-5 of the ragged run through a server that CAN take it."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    tok = synth.FakeTokenizer(vocab=synth.TINY.vocab)
    st = parallel.HipStages(m, tok)
    W = batch = 33
    plan = stage2.plan_groups(W, batch)
    feat = feats("rg.feat", (W, 16, 768), bf16=fl()).to(op()).cuda()
    qf, qc = feats("rg.q0", (5, 768), bf16=fl()).to(op()).cuda(), feats("rg.qc", (768,)).cuda()
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(2), W=W)
    uni = torch.rand(6, len(plan), generator=torch.Generator().manual_seed(20))
    kw = dict(batch=batch, perms=[perms], max_new_tokens=6)

    def through(server):
        st.server = server
        inter = sched.Interleaver(servers=[server] if server is not None else ())
        task = inter.add(sched.Task(lambda t: parallel.launch_queries_sharded_steps(st, tok, feat, W, [(qf, qc, "query 0")], uniforms=uni, turn=t, **kw),
                                    torch.cuda.Stream("cuda:0"), m.engine, 0))
        rec = parallel.collect_queries(inter.finish(task))[0]
        m.engine.slot = 0
        st.server = None
        return rec
    roomy = serve.DecodeServer(m, rows=27, smax=160, gmax=16, pools=2, gang=True, prefill_batch=4)
    want = through(roomy)
    assert roomy.pf_tickets == 1                                                   # (one ragged generate)
    small = serve.DecodeServer(m, rows=8, smax=160, gmax=16, pools=2, gang=True, prefill_batch=4)          # 9 rows do not fit 8
    short = serve.DecodeServer(m, rows=27, smax=64, gmax=16, pools=2, gang=True, prefill_batch=4)          # S + G > Smax
    for server in (small, short, None):
        got = through(server)
        assert got["answers"] == want["answers"] and got["starts"] == want["starts"] and got["score_cos"] == want["score_cos"]
        for k in ("max_entropy", "mean_entropy"):
            assert rel_err(got[k], want[k]) < 1e-5, k
    assert small.pf_tickets == 2 and short.pf_tickets == 0       # two sub-batches through the pools, one after the other / nothing fits: both groups decode alone


@pytest.mark.parametrize("pools", [1, 2])
def test_decode_server_with_eos_equals_classic_loop(pools):
    """EOS in the merged path (``pools`` = 2: under the gang policy, the partly filled pool sealed when the scheduler runs dry): generates whose rows emit EOS at different steps (teacher-forced) leave the pool early; sequences and
    entropies equal the classic loop's (pad after a row's EOS, cut at the step where all rows are done)."""
    from revisionllm_amd import sched, serve
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    m.generation_config.eos_token_id, m.generation_config.pad_token_id = 2, 0
    P = 40
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None]
    cases = []
    forced_sets = [torch.tensor([[7, 9, 11], [2, 12, 13], [5, 14, 15], [6, 2, 2], [8, 9, 10], [8, 9, 10]]),           # ends after step 3
                   torch.tensor([[9, 9], [9, 9], [9, 9], [9, 9], [9, 9], [9, 9]]),                                       # never ends: 6 steps
                   torch.tensor([[2], [9], [9], [9], [9], [9]])]                                                         # ends at step 0
    for i, forced in enumerate(forced_sets):
        B = forced.shape[1]
        feat = feats(f"eoss.feat{i}", (B, 6, 16, 768), bf16=fl())
        q = (feats(f"eoss.q{i}", (B, 5, 768), bf16=fl()), torch.ones(B, 5))
        kw = dict(images=feat, query_feats=q, do_sample=True, temperature=0.05, max_new_tokens=6, forced_tokens=forced, return_dict_in_generate=True,
                  uniforms=torch.full((6, B), 0.5))
        cases.append((ids.repeat(B, 1), kw, m.generate(ids.repeat(B, 1), **kw)))
    assert [c[2]["sequences"].shape[1] - P for c in cases] == [4, 6, 1]
    server = serve.DecodeServer(m, rows=16, smax=96, gmax=8, pools=pools, gang=pools > 1)
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream("cuda:0") for _ in range(3)]
    tasks = [inter.add(sched.Task(m.generate_steps(c[0], server=server, **c[1]), streams[i], m.engine, i)) for i, c in enumerate(cases)]
    outs = [inter.finish(t) for t in tasks]
    m.engine.slot = 0
    for (ids_, kw, want), got in zip(cases, outs):
        assert torch.equal(got["sequences"], want["sequences"]) and torch.equal(got["entropy"], want["entropy"])
        assert torch.equal(got["entropy_raw"], want["entropy_raw"])
    assert not server.jobs and not server.draining and sum(n for _, n, _ in server.free) == 16 * pools


@pytest.mark.parametrize("parity", [False, True])
def test_one_row_generates_in_flight_share_prefill_passes_and_decode_steps(parity):
    """The stage-1 shape of the pipeline (bench.py ``workload_stage1_*``): six ONE-row generates (own window features each, no shared
    prefix) in flight - prefills of identical geometry ride four / two to a pass with P0 = 0 (their attention in ONE launch for all groups,
    ``k_attention_groups``), decode steps merged in a gang-filled pool - against the classic loop: same sequences and entropies (tiny
    model: the tile GEMM serves every pass size).  ``parity``: the same in the parity precision (split Q, split attention output)."""
    from revisionllm_amd import sched, serve
    from revisionllm_amd.utils import synth
    m = _tiny_model(parity)
    P = 40
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None]
    cases = []
    for i in range(6):
        feat = feats(f"s1row.feat{i}", (1, 6, 16, 768), bf16=fl())
        q = (feats(f"s1row.q{i}", (1, 5, 768), bf16=fl()), torch.ones(1, 5))
        kw = dict(images=feat, query_feats=q, do_sample=True, temperature=0.05, max_new_tokens=5, return_dict_in_generate=True,
                  uniforms=torch.rand(5, 1, generator=torch.Generator().manual_seed(100 + i)))
        cases.append((kw, m.generate(ids, **kw)))
    server = serve.DecodeServer(m, rows=16, smax=96, gmax=8, pools=2, gang=True, prefill_batch=4)
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream("cuda:0") for _ in range(6)]
    tasks = [inter.add(sched.Task(m.generate_steps(ids, server=server, **kw), streams[i], m.engine, i)) for i, (kw, _) in enumerate(cases)]
    outs = [inter.finish(t) for t in tasks]
    m.engine.slot = 0
    assert server.pf_tickets == 6 and server.pf_batches <= 3 and max(server.pf_hist) >= 2          # passes of 4 + 2 (or 4 + 1 + 1)
    assert server.rows_served > server.steps_run                                                  # merged steps carried several windows
    for (kw, want), got in zip(cases, outs):
        assert torch.equal(got["sequences"], want["sequences"]) and torch.equal(got["entropy"], want["entropy"])


def test_eos_job_running_all_its_steps_keeps_its_last_column():
    """Regression: EOS configured, ``max_new_tokens == gmax``, greedy stepping: a generate that has produced all its
    tokens waits in ``draining`` for its stop flags while ANOTHER generate keeps stepping in the same pool; the later merged steps
    scatter all rows and must not touch the finished generate's last column (they used to write token 0 / a stale entropy there)."""
    from revisionllm_amd import sched, serve
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    m.generation_config.eos_token_id, m.generation_config.pad_token_id = 2, 0
    P, G = 40, 8
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None]
    cases = []
    for i, steps in enumerate((G, G, 3)):           # two full-length generates and a short one that joins later
        B = 2
        feat = feats(f"eosg.feat{i}", (B, 6, 16, 768), bf16=fl())
        q = (feats(f"eosg.q{i}", (B, 5, 768), bf16=fl()), torch.ones(B, 5))
        forced = torch.full((steps, B), 9 + i)
        kw = dict(images=feat, query_feats=q, do_sample=True, temperature=0.05, max_new_tokens=steps, forced_tokens=forced, return_dict_in_generate=True,
                  uniforms=torch.full((steps, B), 0.5))
        cases.append((ids.repeat(B, 1), kw, m.generate(ids.repeat(B, 1), **kw)))
    server = serve.DecodeServer(m, rows=16, smax=96, gmax=G, pools=1, gang=False)          # greedy: jobs step as soon as they have joined
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream("cuda:0") for _ in range(3)]
    tasks = [inter.add(sched.Task(m.generate_steps(cases[0][0], server=server, **cases[0][1]), streams[0], m.engine, 0))]
    for _ in range(3):                               # let the first generate get a few steps ahead of the others
        inter.pump()
        torch.cuda.synchronize()
    tasks += [inter.add(sched.Task(m.generate_steps(c[0], server=server, **c[1]), streams[i], m.engine, i)) for i, c in enumerate(cases) if i > 0]
    outs = [inter.finish(t) for t in tasks]
    m.engine.slot = 0
    for (ids_, kw, want), got in zip(cases, outs):
        assert torch.equal(got["sequences"], want["sequences"]) and torch.equal(got["entropy"], want["entropy"])
    assert not server.jobs and not server.draining and sum(n for _, n, _ in server.free) == 16


def test_a_failing_generate_fails_alone_and_gives_its_rows_back():
    """Regression: with several generates in flight under the gang policy, an exception in ONE task's generator is raised by
    ``finish`` of THAT task only; its rows go back to the pool, so the pool can still seal and step for the others, and the
    scheduler does not spin."""
    from revisionllm_amd import sched, serve
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    P = 40
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None]
    server = serve.DecodeServer(m, rows=8, smax=96, gmax=8, pools=2, gang=True, prefill_batch=1)
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream("cuda:0") for _ in range(4)]

    def make(i, sampling_t):
        feat = feats(f"fail.feat{i}", (2, 6, 16, 768), bf16=fl())
        q = (feats(f"fail.q{i}", (2, 5, 768), bf16=fl()), torch.ones(2, 5))
        return dict(images=feat, query_feats=q, do_sample=True, temperature=sampling_t, max_new_tokens=4, return_dict_in_generate=True,
                    uniforms=torch.full((4, 2), 0.5))
    kws = [make(0, 0.05), make(1, 0.7), make(2, 0.05), make(3, 0.05)]            # generate 1 asks for other sampling settings: join() raises
    want = [m.generate(ids.repeat(2, 1), **kw) for kw in kws]
    tasks = [inter.add(sched.Task(m.generate_steps(ids.repeat(2, 1), server=server, **kws[0]), streams[0], m.engine, 0))]
    while not server.pools[0].jobs:                 # generate 0 joins first: its sampling settings are the pool's
        inter.pump()
    tasks += [inter.add(sched.Task(m.generate_steps(ids.repeat(2, 1), server=server, **kw), streams[i], m.engine, i)) for i, kw in enumerate(kws) if i > 0]
    results = []
    for i, t in enumerate(tasks):
        try:
            results.append(inter.finish(t))
        except ValueError as e:
            assert i == 1 and "sampling settings" in str(e)
            results.append(None)
    m.engine.slot = 0
    assert [r is None for r in results] == [False, True, False, False]
    for i in (0, 2, 3):
        assert torch.equal(results[i]["sequences"], want[i]["sequences"]) and torch.equal(results[i]["entropy"], want[i]["entropy"])
    assert all(p.pending == 0 and p.live == 0 for p in server.pools) and sum(n for _, n, _ in server.free) == 16


@pytest.mark.parametrize("R", [70, 112, 140])
def test_fp8_decode_weights_in_wide_merged_steps_7b_layer(R):
    """The fp8 LLM path of BASELINE configs[4] through the MERGED pipeline's kernels: one Vicuna-7B-shaped block + lm_head, R rows
    (ten / sixteen 7-row generates) prefilled into one KV pool, then merged decode steps streaming the FP8 weight copies
    (rv_llm_decode_rows above 32 rows: the split-K kernel) - against the CPU oracle running its decode steps on the same fake-quantised
    weights (``oracle.llama.fp8_decode_weights``), and bit-identical per generate to the <= 16-row FP8 kernel."""
    from oracle import llama
    from revisionllm_amd import engine
    from revisionllm_amd.utils import synth
    shape = synth.LlamaShape(layers=1, vocab=2048)
    eng = engine.Engine(shape, adapter_text=False, device="cuda:0")
    eng.init_synthetic(seed=SEED, llm=True, clip=False, fp8_decode=True)
    D, S, Smax, steps = 4096, 40, 64, 3
    n = R // 7
    g = torch.Generator().manual_seed(R)
    hs = [torch.randn(7, S, D, generator=g).mul(0.02) for _ in range(n)]
    toks = [[torch.randn(7, 1, D, generator=g).mul(0.02) for _ in range(steps)] for _ in range(n)]
    pool, sm = eng.new_kv_pool(R, Smax)
    for i in range(n):
        eng.llm_prefill_pool(hs[i].cuda().view(7 * S, D).contiguous(), 7, 0, pool, R, 7 * i, Smax)
    got = []
    for s_ in range(steps):
        hrow = torch.zeros(R, D, device="cuda:0")
        p = torch.full((R,), -1, dtype=torch.int32, device="cuda:0")
        for i in range(n):
            hrow[7 * i:7 * i + 7] = toks[i][s_][:, 0].cuda()
            p[7 * i:7 * i + 7] = S + s_
        got.append(eng.llm_decode_rows(hrow, p, pool, Smax).clone())
    # (a) the <= 16-row FP8 kernel on each generate's own cache: same bits
    for i in (0, n - 1):
        kv, _ = eng.new_kv(7, Smax, reuse=False)
        eng.llm_forward(hs[i].cuda().clone(), 0, kv, Smax)
        for s_ in range(steps):
            want = eng.llm_forward(toks[i][s_].cuda().clone(), S + s_, kv, Smax)
            assert torch.equal(got[s_][7 * i:7 * i + 7], want), (i, s_)
    # (b) the oracle with the same fake-quantised decode weights (prefill on the bf16 weights), generates 0 and n - 1
    w = {k: T(v) for k, v in synth.build_numpy(synth.llama_spec(shape), SEED).items()}
    w = {k: (v.to(op()).float() if v.dim() == 2 else v) for k, v in w.items()}
    cfg = llama.LlamaCfg(shape.hidden, shape.inter, shape.layers, shape.heads, shape.vocab, shape.eps, shape.theta)
    w8 = llama.fp8_decode_weights(w, cfg)
    for i in (0, n - 1):
        cache = llama.KVCache(cfg.layers)
        llama.forward(hs[i], w, cfg, cache=cache, last_only=True)
        for s_ in range(steps):
            want = llama.forward(toks[i][s_], w8, cfg, cache=cache)[:, -1]
            want16 = None
            assert rel_err(got[s_][7 * i:7 * i + 7].cpu(), want) < 2e-2, (i, s_)
    # it really is the quantised weights that ran: switching the copies off changes the logits
    eng.set_option("fp8_decode", 0)
    hrow = torch.zeros(R, D, device="cuda:0")
    p = torch.full((R,), S + steps, dtype=torch.int32, device="cuda:0")
    a16 = eng.llm_decode_rows(hrow + 0.01, p, pool, Smax).clone()
    eng.set_option("fp8_decode", 1)
    p2 = torch.full((R,), S + steps, dtype=torch.int32, device="cuda:0")
    a8 = eng.llm_decode_rows(hrow + 0.01, p2, pool, Smax)
    assert not torch.equal(a16, a8)


def test_handoff_status_travels_as_one_snapshot_and_still_raises():
    """Round 4: the hand-off status words of all workspaces go to pinned host memory as ONE snapshot behind a recursion's results
    (``Engine.handoff_status_async``) instead of one ``.item()`` per workspace when the record is collected.  A clean run yields a clean
    snapshot; a status word that a kernel's bounded wait would have set is reported through the snapshot AND through the synchronous
    form, names its workspace, and is cleared so that the next check is clean again."""
    from revisionllm_amd import hip
    m = _tiny_model()
    eng = m.engine
    P = 40
    from revisionllm_amd.utils import synth
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None]
    feat = feats("status.feat", (1, 6, 16, 768), bf16=fl())
    q = (feats("status.q", (1, 5, 768), bf16=fl()), torch.ones(1, 5))
    m.generate(ids, images=feat, query_feats=q, do_sample=False, max_new_tokens=3)       # workspaces exist now
    snap = eng.handoff_status_async()
    assert snap is not None and len(snap[0]) == snap[1].numel() >= 1
    torch.cuda.synchronize()
    eng.check_handoff_status(snap)                                                         # clean
    eng.check_handoff_status()
    key = snap[0][0]
    eng._ws[key][8188:8192].view(torch.int32).fill_(1)                                     # what a timed-out in-kernel wait writes
    bad = eng.handoff_status_async()
    torch.cuda.synchronize()
    with pytest.raises(hip.HipLibraryError, match="hand-off wait timed out"):
        eng.check_handoff_status(bad)
    eng.check_handoff_status()                                                             # cleared by the failed check
    eng._ws[key][8188:8192].view(torch.int32).fill_(1)
    with pytest.raises(hip.HipLibraryError, match="hand-off wait timed out"):
        eng.check_handoff_status()
    eng.check_handoff_status()
