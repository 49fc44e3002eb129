"""Python owner of an ``rv_ctx``: packs / binds weights, owns workspaces and KV caches (torch device tensors),
and exposes the high-level entry points of the C ABI.

Weight packing (build-defined layout): q/k/v rows fused into ``wqkv`` [3D,D]; gate/up rows
interleaved in 16-row blocks into ``wgu`` [2F,D] (so SiLU(gate)*up is a lane-local GEMM epilogue); every matrix
is then stored FRAGMENT-PACKED (ops.pack_fragments: each 16x32 MFMA operand fragment one contiguous 1 KiB block),
which makes the decode weight stream perfectly coalesced and the prefill LDS image conflict-free; the embedding
table stays row-major (it is gathered by row); vectors are f32.  LoRA never exists here: the builder merges it before packing (builder.py:53-60).
"""
import ctypes as C

import torch

from . import hip, ops
from .utils import synth


def _dev_f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _dev16(t, device, dtype):
    """A tensor as 16-bit operands of ``dtype`` on the device.  fp16 from wider types: values beyond +-65504 saturate (like the kernels' own
    conversions) instead of becoming inf."""
    if dtype == torch.float16 and t.dtype != torch.float16:
        t = t.to(device=device, dtype=torch.float32).clamp(-65504.0, 65504.0)      # (no host sync: an unconditional pass, only for inputs that are not fp16 already)
    return t.to(device=device, dtype=dtype).contiguous()


class PersistGate:
    """Orders the launches that need EVERY CU of a device at once.

    The prefill GEMMs are PERSISTENT kernels: one workgroup per CU, and a workgroup waits in-kernel for the partial tiles of its
    panel's other workgroups, so all of a launch's workgroups must become resident.  Two such launches running at once (calls in
    flight on different streams, or two engines on one GPU) could each hold half the CUs and wait for the other half until the
    bounded wait gives up; an RCCL collective kernel resident on a few CUs delays them the same way.  Whoever enqueues such a
    launch brackets it with ``begin`` / ``end``: a device-side event wait only - decode steps and the adapter overlap freely.
    The ordering domain is the DEVICE (its CUs are the contended resource), so by default every Engine on a device shares that
    device's gate (``PersistGate.for_device``); an Engine can be given its own gate explicitly."""

    _by_device = {}

    def __init__(self):
        self._event = None

    @classmethod
    def for_device(cls, device):
        key = torch.device(device).index or 0
        if key not in cls._by_device:
            cls._by_device[key] = cls()
        return cls._by_device[key]

    def begin(self, stream=None):
        if self._event is not None:
            (stream or torch.cuda.current_stream()).wait_event(self._event)

    def end(self, stream=None):
        ev = torch.cuda.Event()
        ev.record(stream or torch.cuda.current_stream())
        self._event = ev


class Engine:
    def __init__(self, shape: synth.LlamaShape = synth.VICUNA_7B, adapter_text=True, device="cuda:0", adapter_dim=768,
                 adapter_heads=8, adapter_ff=2048, adapter_layers=2, gate=None, op_dtype=None):
        """``op_dtype``: "f16" / "bf16" (or the torch dtype): the 16-bit operand type of weights, activation copies and KV caches = which build of
        the library this engine calls (hip.py; default: the process default, fp16)."""
        if not torch.cuda.is_available():
            raise hip.HipLibraryError("no GPU visible: revisionllm_amd runs only on the HIP device path")
        self.flavour = hip.flavour_of(op_dtype)
        self.op_dtype = hip.op_dtype(self.flavour)
        self.lib = hip.lib(self.flavour)
        self.shape = shape
        self.device = torch.device(device)
        self.adapter_text = bool(adapter_text)
        self.adapter_dim, self.adapter_ff, self.adapter_layers = adapter_dim, adapter_ff, adapter_layers
        cfg = hip.RvConfig(shape.hidden, shape.inter, shape.layers, shape.heads, shape.vocab, shape.eps, shape.theta,
                           adapter_dim, adapter_heads, adapter_ff, adapter_layers, int(adapter_text))
        self._ctx = C.c_void_p()
        hip.check(self.lib.rv_ctx_create(C.byref(cfg), C.byref(self._ctx)), "rv_ctx_create")
        self._keep = {}      # name -> tensor (keeps device memory alive while bound)
        self._ws = {}        # workspace cache
        self.slot = 0        # workspace / KV-pool namespace: one per in-flight call stream (weights are shared, read-only)
        self.slots_in_flight = set()   # slots whose generate (decoding alone) has started and not finished: see model.generate_steps
        self.gate = gate if gate is not None else PersistGate.for_device(self.device)
        self.has_llm = self.has_clip = self.has_linear = False
        # fp16 stores that saturated (the reference's bf16 path cannot overflow; this build's default operand type can): the library counts them in
        # a device word that travels with the hand-off status snapshots.  on_saturation: "warn" (once per new count), "raise" or "ignore"
        self.numeric_status = hip.numeric_status(self.flavour, self.device)
        self.on_saturation = "warn"
        self.saturated_seen = 0

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                self.lib.rv_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass

    # ---- tunables (per context: two engines in one process can differ) ---------------------------
    def set_option(self, key, value):
        hip.check(self.lib.rv_ctx_set_option(self._ctx, key.encode(), int(value)), f"rv_ctx_set_option({key})")
        return self

    def get_option(self, key):
        v = C.c_int64()
        hip.check(self.lib.rv_ctx_get_option(self._ctx, key.encode(), C.byref(v)), f"rv_ctx_get_option({key})")
        return int(v.value)

    # ---- weights -------------------------------------------------------------------------------
    def _dev16(self, t):
        return _dev16(t, self.device, self.op_dtype)

    def _dev_packed(self, t):
        """16-bit operands on the device in the fragment-packed GEMM layout."""
        return ops.pack_fragments(self._dev16(t))

    def bind(self, name, t):
        assert t.is_cuda and t.is_contiguous()
        self._keep[name] = t
        hip.check(self.lib.rv_weights_bind(self._ctx, name.encode(), hip.ptr(t), hip.dtype_code(t), t.numel()),
                  f"rv_weights_bind({name})")

    def weight(self, name):
        return self._keep[name]

    def _loaded(self):
        """Weights are written by kernels on the loader's stream; calls may later run on other (non-blocking) HIP streams
        that do not wait for it implicitly, so loading ends with a device synchronise (not on the hot path)."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _bind_matrix(self, name, w, fp8, fp8_prefill=False, parity=False):
        """Bind a row-major bf16 matrix fragment-packed; with ``fp8`` also its FP8 (e4m3fn, per-row scale) copy for the
        decode kernels (``<name>.f8`` uint8, ``<name>.s8`` f32 [rows]); with ``fp8_prefill`` the same quantised bytes in the
        prefill GEMM's layout (``<name>.f8p``); with ``parity`` the K-duplicated copy ``<name>.p2`` = [W | W] the parity precision
        multiplies split operands [hi | lo] with."""
        self.bind(name, ops.pack_fragments(w))
        if parity:
            self.bind(name + ".p2", ops.pack_fragments(torch.cat([w, w], dim=1).contiguous()))
        if fp8 or fp8_prefill:
            q, sc = ops.quantize_rows_fp8(w)
            self.bind(name + ".s8", sc)
            if fp8:
                self.bind(name + ".f8", ops.pack_fp8_decode(q))
            if fp8_prefill:
                self.bind(name + ".f8p", ops.pack_fp8_prefill(q))

    def load_llm(self, get, fp8_decode=False, fp8_prefill=False, parity=False):
        """``get(hf_name) -> tensor`` (any device / float dtype), HF Llama names.  Packs layer by layer.
        ``fp8_decode``: additionally keep an FP8 copy of every projection (+ lm_head) and stream THAT in KV-cached decode steps
        (opt-in "fp8 LLM path": half the decode weight bytes).  ``fp8_prefill``: additionally keep the layer projections in the
        FP8 prefill layout; prefill passes then quantise their activations per row and run FP8 x FP8 MFMA GEMMs (lm_head stays bf16).
        ``parity``: additionally keep K-duplicated copies of every projection and the lm_head (+ 2 x the weight bytes) so that
        ``set_option("precision", 1)`` can switch the forward to split-bf16 operands (16 mantissa bits in every GEMM input: the
        reference's fp32 scores to 1e-3, tests/test_gpu_full_depth_conditioned.py)."""
        s, dev = self.shape, self.device
        self.parity = bool(parity)
        self.fp8_decode = bool(fp8_decode)
        self.fp8_prefill = bool(fp8_prefill)
        self.bind("llm.embed", self._dev16(get("model.embed_tokens.weight")))
        # the lm_head's K-duplicated copy is bound in EVERY precision (2 x V x D x 2 B = + 0.52 GB for 32000 x 4096): the lm_head input is always a split pair (option lm_head_split; 0 = single-operand lm_head input, the p2 copy then goes unused)
        self._bind_matrix("llm.lm_head", self._dev16(get("lm_head.weight")), fp8_decode, parity=True)
        self.bind("llm.norm", _dev_f32(get("model.norm.weight"), dev))
        for i in range(s.layers):
            p = f"model.layers.{i}."
            q, k, v = (self._dev16(get(p + f"self_attn.{n}_proj.weight")) for n in "qkv")
            q, k = pair_interleave_heads(q, s.heads), pair_interleave_heads(k, s.heads)
            self._bind_matrix(f"llm.L{i}.wqkv", torch.cat([q, k, v], dim=0).contiguous(), fp8_decode, fp8_prefill, parity)
            del q, k, v
            self._bind_matrix(f"llm.L{i}.wo", self._dev16(get(p + "self_attn.o_proj.weight")), fp8_decode, fp8_prefill, parity)
            g = self._dev16(get(p + "mlp.gate_proj.weight"))
            u = self._dev16(get(p + "mlp.up_proj.weight"))
            self._bind_matrix(f"llm.L{i}.wgu", pack_gate_up(g, u), fp8_decode, fp8_prefill, parity)
            del g, u
            self._bind_matrix(f"llm.L{i}.wdown", self._dev16(get(p + "mlp.down_proj.weight")), fp8_decode, fp8_prefill, parity)
            self.bind(f"llm.L{i}.norm1", _dev_f32(get(p + "input_layernorm.weight"), dev))
            self.bind(f"llm.L{i}.norm2", _dev_f32(get(p + "post_attention_layernorm.weight"), dev))
        self.has_llm = True
        self._loaded()

    def load_clip_adapter(self, get):
        """``get(name)`` with ClipEncoder state-dict names (transformer.py:60-92), e.g. 'encoder.layers.0.linear1.weight'."""
        dev = self.device
        self.bind("adp.cls_token", _dev_f32(get("global_rep_token"), dev))
        self.bind("adp.cls_pos", _dev_f32(get("global_rep_pos"), dev))
        if self.adapter_dim == self.shape.hidden and self.adapter_dim != 768:
            # the hidden-wide `cross_attn` ClipEncoder (transformer.py:65-67,86): no output projector (nn.Identity), a text projector in front
            self.txt_proj = (self._dev_packed(get("text_mm_projector.weight")), _dev_f32(get("text_mm_projector.bias"), dev))
        else:
            self.bind("adp.proj_w", self._dev_packed(get("mm_projector.weight")))
            self.bind("adp.proj_b", _dev_f32(get("mm_projector.bias"), dev))
        stacks = ([("t2v_encoder", "t2v")] if self.adapter_text else []) + [("encoder", "enc")]
        for ref, tag in stacks:
            for l in range(self.adapter_layers):
                r, o = f"{ref}.layers.{l}.", f"adp.{tag}.{l}."
                self.bind(o + "w_in", self._dev_packed(get(r + "self_attn.in_proj_weight")))
                self.bind(o + "b_in", _dev_f32(get(r + "self_attn.in_proj_bias"), dev))
                self.bind(o + "w_out", self._dev_packed(get(r + "self_attn.out_proj.weight")))
                self.bind(o + "b_out", _dev_f32(get(r + "self_attn.out_proj.bias"), dev))
                self.bind(o + "w1", self._dev_packed(get(r + "linear1.weight")))
                self.bind(o + "b1", _dev_f32(get(r + "linear1.bias"), dev))
                self.bind(o + "w2", self._dev_packed(get(r + "linear2.weight")))
                self.bind(o + "b2", _dev_f32(get(r + "linear2.bias"), dev))
                for n in ("1", "2"):
                    self.bind(o + f"ln{n}_w", _dev_f32(get(r + f"norm{n}.weight"), dev))
                    self.bind(o + f"ln{n}_b", _dev_f32(get(r + f"norm{n}.bias"), dev))
        self.has_clip = True
        self._loaded()

    def load_linear_projector(self, get):
        if self.adapter_dim != 768:
            # next to a hidden-wide cross_attn ClipEncoder the Linear(768 -> hidden) projector runs IN FRONT of the encoder
            # (vtimellm_arch.py:125 then :127-144); rv_project_dense is tied to the context's adapter width, so it goes through rv_gemm
            self.frame_proj = (self._dev_packed(get("weight")), _dev_f32(get("bias"), self.device))
        else:
            self.bind("proj.w", self._dev_packed(get("weight")))
            self.bind("proj.b", _dev_f32(get("bias"), self.device))
        self.has_linear = True
        self._loaded()

    # ---- synthetic weights generated on the device (bench / smoke / tests) -----------------------
    def _synth_get(self, spec, seed, prefix, grid=None):
        table = {n: (shp, a, base) for n, shp, a, base in spec}
        grid_dt = {None: None, "bf16": torch.bfloat16, "f16": torch.float16}[grid]

        def get(name):
            shp, a, base = table[name]
            t = torch.empty(shp, dtype=torch.float32, device=self.device)
            ops.init_hash_(t, prefix + name, seed, a, base)
            if grid_dt is not None and len(shp) > 1:      # matrices of a "checkpoint" stored in that type (vectors stay fp32, as in the goldens)
                t = t.to(grid_dt).float()
            return t
        return get

    def init_synthetic(self, seed=0, llm=True, clip=True, linear=False, llm_prefix="", clip_prefix="model.mm_projector.",
                       linear_prefix="model.mm_projector.", fp8_decode=False, fp8_prefill=False, cond=None, parity=False, grid=None):
        """Random-init weights of the reference's shapes, bit-identical to ``synth.build_numpy`` on the host.
        ``cond``: a ``synth.Conditioning`` (the well-conditioned LLM amplitudes of golden G8c); None = plain N(0, 0.02).
        ``grid``: None = the hash stream's fp32 values converted straight to this engine's operand type (``build_numpy(bf16=<flavour>)`` on
        the host); "bf16" / "f16" = the matrices of a CHECKPOINT stored in that type (``build_numpy(bf16=grid)``), which this engine then
        converts to its operand type: exact when the operand type is at least as fine (an fp16 engine holds a bf16-grid matrix exactly up
        to fp16's subnormal range), rounded otherwise (a bf16 engine rounds an fp16 checkpoint - golden G8d measures that term)."""
        if llm:
            self.load_llm(self._synth_get(synth.llama_spec(self.shape, cond=cond), seed, llm_prefix, grid), fp8_decode=fp8_decode, fp8_prefill=fp8_prefill, parity=parity)
        if clip:
            self.load_clip_adapter(self._synth_get(synth.clip_encoder_spec(hidden=self.shape.hidden, text=self.adapter_text,
                                                                           cross_attn=self.adapter_dim != 768), seed,
                                                   clip_prefix, grid))
        if linear:
            self.load_linear_projector(self._synth_get(synth.linear_projector_spec(hidden=self.shape.hidden), seed, linear_prefix, grid))
        return self

    # ---- workspaces ------------------------------------------------------------------------------
    def _workspace(self, key, nbytes):
        key = (self.slot, key)
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)  # zero: stream-K flags
            self._ws[key] = t
        return t

    _NUMERIC_KEY = ("numeric", "saturated")

    def _status_words(self):
        words = [(key, ws[8188:8192].view(torch.int32)) for key, ws in list(self._ws.items())
                 if isinstance(key, tuple) and key[0] != "kv" and key[1] in ("llm", "clip") and ws.numel() >= 8192]
        if words:       # the saturation count rides along (same copy, same event)
            words.append((self._NUMERIC_KEY, self.numeric_status[:1]))
        return words

    def saturated(self, reset=False):
        """Sticky count of f32 -> fp16 stores that met a value outside +-65504 (``rv_ctx_get_option("saturated")``; waits for the device).  Always 0
        for a bf16 engine.  ``reset``: zero it afterwards."""
        n = self.get_option("saturated")
        if reset:
            self.set_option("saturated", 0)
            self.saturated_seen = 0
        return n

    def _note_saturation(self, count):
        if count <= self.saturated_seen:
            return
        new, self.saturated_seen = count - self.saturated_seen, count
        msg = (f"{new} fp16 store(s) saturated at +-65504 since the last check ({count} in total): activations of this checkpoint leave the fp16 range; "
               "the affected elements lost accuracy (no inf was produced).  Use op_dtype='bf16' for the reference's exponent range.")
        if self.on_saturation == "raise":
            raise hip.HipLibraryError(msg)
        if self.on_saturation == "warn":
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)

    def handoff_status_async(self):
        """Snapshot of every workspace's hand-off status word, copied to pinned host memory on the CURRENT stream (no host wait): a pipeline
        that copies its results to the host anyway adds this behind them and hands it to ``check_handoff_status`` once its event has fired."""
        words = self._status_words()
        if not words:
            return None
        host = torch.empty(len(words), dtype=torch.int32, pin_memory=True)
        host.copy_(torch.cat([w for _, w in words]), non_blocking=True)
        return [k for k, _ in words], host

    def check_handoff_status(self, snapshot=None):
        """Raise if a bounded in-kernel wait of the stream-K GEMMs / fused decode kernel ever gave up (a workgroup that never
        arrived would otherwise show up as silently wrong numbers).  The status word sits at int 2047 of the hand-off header
        at the start of every workspace; reading it synchronises (ONE device -> host copy for all workspaces), so call it where the host
        waits for results anyway - or pass the ``handoff_status_async`` snapshot that travelled with the results (no wait at all:
        with twenty recursions in flight the per-workspace reads were 22 host round trips per collected recursion, round 4)."""
        if snapshot is None:
            words = self._status_words()
            if not words:
                return
            keys, vals = [k for k, _ in words], torch.cat([w for _, w in words]).cpu()
        else:
            keys, vals = snapshot
        bad = [(k, int(v)) for k, v in zip(keys, vals.tolist()) if v != 0 and k != self._NUMERIC_KEY]
        sat = [int(v) & 0xffffffff for k, v in zip(keys, vals.tolist()) if k == self._NUMERIC_KEY]
        if sat:
            self._note_saturation(sat[0])
        if bad:
            for k, _ in bad:
                ws = self._ws.get(k)
                if ws is not None:
                    ws[8188:8192].zero_()
            key, st = bad[0]
            raise hip.HipLibraryError(f"workspace '{key}': an in-kernel hand-off wait timed out (status {st}); results of the last calls are invalid")

    # ---- adapter ---------------------------------------------------------------------------------
    def project_dense(self, x, out_dtype=torch.float32):
        """nn.Linear(768, D) on [..., 768] bf16 -> [..., D]."""
        lead = x.shape[:-1]
        x2 = self._dev16(x).reshape(-1, x.shape[-1])
        y = torch.empty(x2.shape[0], self.shape.hidden, dtype=out_dtype, device=self.device)
        hip.check(self.lib.rv_project_dense(self._ctx, hip.ptr(x2), hip.ptr(y), hip.dtype_code(y), x2.shape[0], hip.stream()),
                  "rv_project_dense")
        return y.reshape(*lead, self.shape.hidden)

    def clip_encoder(self, x, txt=None, txt_mask=None, feature="cls"):
        """x [N,T,768]; txt [Nq,Lq,768], txt_mask [Nq,Lq] (1 = valid) -> f32 [N,D] ('cls') or [N,T+1,D] ('all')."""
        N, T, _ = x.shape
        x = self._dev16(x)
        wide = self.adapter_dim != 768           # the hidden-wide cross_attn ClipEncoder: frames and text are projected to its width first
        if wide:
            if getattr(self, "frame_proj", None) is None or getattr(self, "txt_proj", None) is None:
                raise hip.HipLibraryError("the cross_attn ClipEncoder needs the Linear projector (load_linear_projector) and its text projector bound")
            x = ops.gemm(x.reshape(N * T, -1), self.frame_proj[0], bias=self.frame_proj[1], out_dtype=self.op_dtype, w_packed=True,
                         ctx=self).view(N, T, self.adapter_dim)
        if self.adapter_text:
            txt = self._dev16(txt)
            Nq, Lq = txt.shape[0], txt.shape[1]
            if wide:
                txt = ops.gemm(txt.reshape(Nq * Lq, -1), self.txt_proj[0], bias=self.txt_proj[1], out_dtype=self.op_dtype, w_packed=True,
                               ctx=self).view(Nq, Lq, self.adapter_dim)
            m = ops.h2d((txt_mask != 0).to(torch.uint8), self.device).contiguous()
        else:
            txt, m, Nq, Lq = None, None, 0, 0
        feat = hip.RV_FEAT_CLS if feature == "cls" else hip.RV_FEAT_ALL
        D = self.shape.hidden
        out = torch.empty((N, D) if feat == hip.RV_FEAT_CLS else (N, T + 1, D), dtype=torch.float32, device=self.device)
        nbytes = self.lib.rv_clip_encoder_ws_bytes(self._ctx, N, T, Nq, Lq)
        ws = self._workspace("clip", nbytes)
        hip.check(self.lib.rv_clip_encoder(self._ctx, hip.ptr(x), hip.ptr(txt), hip.ptr(m), N, T, Nq, Lq, feat, hip.ptr(out),
                                           hip.ptr(ws), ws.numel(), hip.stream()), "rv_clip_encoder")
        return out

    # ---- LLM -------------------------------------------------------------------------------------
    def splice_embed(self, row_map, video_rows):
        """row_map i32 [B,S] (>=0 token id, <0 video row -(v+1)); video_rows f32 [R,D] -> h f32 [B,S,D]."""
        B, S = row_map.shape
        h = torch.empty(B, S, self.shape.hidden, dtype=torch.float32, device=self.device)
        rm = ops.h2d(row_map, self.device, torch.int32).contiguous()
        vr = _dev_f32(video_rows, self.device) if video_rows is not None else None
        hip.check(self.lib.rv_splice_embed(self._ctx, hip.ptr(rm), hip.ptr(vr), hip.ptr(h), B * S, hip.stream()), "rv_splice_embed")
        return h

    def new_kv(self, B, Smax, reuse=True):
        """KV cache for B rows x Smax positions (rounded up to 32).  Caches are pooled per (B, Smax): a recycled cache holds
        stale but finite bf16 values, which is all the kernels need beyond the current length (P = 0 there)."""
        Smax = (Smax + 31) // 32 * 32
        key = ("kv", self.slot, B, Smax)
        t = self._ws.get(key) if reuse else None
        if t is None:
            nbytes = self.lib.rv_kv_bytes(self._ctx, B, Smax)
            t = torch.zeros(nbytes // 2, dtype=self.op_dtype, device=self.device)
            if reuse:
                for k_ in [k_ for k_ in self._ws if isinstance(k_, tuple) and k_[0] == "kv" and k_[1] == self.slot][:-3]:
                    del self._ws[k_]   # keep the pool small
                self._ws[key] = t
        return t, Smax

    @staticmethod
    def vt_logical(vt, *lead, head_dim=128, Smax=None):
        """The V^T half of a KV cache as a LOGICAL [*lead, head_dim, Smax] tensor (a copy).  On the device V^T is blocked by 8 positions:
        element (d, pos) of one (row, head) lives at ((pos >> 3) * head_dim + d) * 8 + (pos & 7) (csrc/kernels.h rv_vt_index)."""
        return vt.view(*lead, Smax // 8, head_dim, 8).movedim(-3, -2).reshape(*lead, head_dim, Smax)

    def _persist_begin(self):
        self.gate.begin(torch.cuda.current_stream(self.device))

    def _persist_end(self):
        self.gate.end(torch.cuda.current_stream(self.device))

    def llm_forward(self, h, pos0, kv, Smax, logits=None):
        """h f32 [B,S,D] (clobbered) -> logits f32 [B,V] of the last position; appends K/V at pos0..pos0+S-1."""
        B, S, _ = h.shape
        assert h.dtype == torch.float32 and h.is_contiguous()
        if logits is None:
            logits = torch.empty(B, self.shape.vocab, dtype=torch.float32, device=self.device)
        nbytes = self.lib.rv_llm_ws_bytes(self._ctx, B, S)
        ws = self._workspace("llm", nbytes)
        if B * S > 16:
            self._persist_begin()
        hip.check(self.lib.rv_llm_forward(self._ctx, hip.ptr(h), B, S, pos0, hip.ptr(kv), Smax, hip.ptr(logits), hip.ptr(ws),
                                          ws.numel(), hip.stream()), "rv_llm_forward")
        if B * S > 16:
            self._persist_end()
        return logits


    def llm_layers(self, h, pos0, kv, Smax, layer_begin, layer_end):
        """Blocks [layer_begin, layer_end) over h f32 [B,S,D] IN PLACE (no final norm / lm_head): the residual stream behind the
        last of them.  S > 1: prefill at positions 0..S-1; S == 1: one KV-cached decode step at pos0."""
        B, S, _ = h.shape
        assert h.dtype == torch.float32 and h.is_contiguous()
        ws = self._workspace("llm", self.lib.rv_llm_ws_bytes(self._ctx, B, S))
        if B * S > 16:
            self._persist_begin()
        hip.check(self.lib.rv_llm_layers(self._ctx, hip.ptr(h), B, S, pos0, hip.ptr(kv), Smax, layer_begin, layer_end, hip.ptr(ws), ws.numel(),
                                         hip.stream()), "rv_llm_layers")
        if B * S > 16:
            self._persist_end()
        return h

    def llm_prefill_shared(self, h, B, P0, kv, Smax, logits=None):
        """h f32 [P0 + B*S, D] (shared prefix rows first, then S rows per sequence) -> logits f32 [B,V]."""
        assert h.dtype == torch.float32 and h.is_contiguous() and (h.shape[0] - P0) % B == 0
        S = (h.shape[0] - P0) // B
        if logits is None:
            logits = torch.empty(B, self.shape.vocab, dtype=torch.float32, device=self.device)
        ws = self._workspace("llm", self.lib.rv_llm_prefill_shared_ws_bytes(self._ctx, B, P0, S))
        self._persist_begin()
        hip.check(self.lib.rv_llm_prefill_shared(self._ctx, hip.ptr(h), B, P0, S, hip.ptr(kv), Smax, hip.ptr(logits), hip.ptr(ws),
                                                 ws.numel(), hip.stream()), "rv_llm_prefill_shared")
        self._persist_end()
        return logits


    # ---- several generates sharing one KV pool: merged decode steps ---------------------------------------------------------
    def new_kv_pool(self, rows, Smax):
        """One cache tensor for ``rows`` sequences ([L, rows, H, Smax, 128] + its V^T twin), zero-initialised."""
        Smax = (Smax + 31) // 32 * 32
        nbytes = self.lib.rv_kv_bytes(self._ctx, rows, Smax)
        return torch.zeros(nbytes // 2, dtype=self.op_dtype, device=self.device), Smax

    def llm_prefill_pool(self, h, B, P0, kv, kv_rows, kv_row0, Smax, logits=None):
        """Prefill of B sequences into rows kv_row0.. of a pool: h f32 [P0 + B*S, D] (shared prefix first; P0 = 0: [B*S, D])."""
        assert h.dtype == torch.float32 and h.is_contiguous() and (h.shape[0] - P0) % B == 0
        S = (h.shape[0] - P0) // B
        if logits is None:
            logits = torch.empty(B, self.shape.vocab, dtype=torch.float32, device=self.device)
        ws = self._workspace("llm", self.lib.rv_llm_prefill_shared_ws_bytes(self._ctx, B, P0, S))
        self._persist_begin()
        hip.check(self.lib.rv_llm_prefill_pool(self._ctx, hip.ptr(h), B, P0, S, hip.ptr(kv), kv_rows, kv_row0, Smax, hip.ptr(logits), hip.ptr(ws),
                                               ws.numel(), hip.stream()), "rv_llm_prefill_pool")
        self._persist_end()
        return logits

    def llm_prefill_pool_groups(self, h, G, B, P0, kv, kv_rows, row0s, Smax, logits=None, last_rows=None):
        """G prefills of identical geometry in one pass: h f32 [G * (P0 + B*S), D], block g's cache rows start at row0s[g].
        -> logits f32 [G * B, V] (group-major).  ``last_rows`` (device int32 [G * B]): the sequences are right-padded to S rows and the
        head reads row last_rows[i] of h for sequence i (rv_llm_prefill_pool_groups_ragged)."""
        import ctypes
        assert h.dtype == torch.float32 and h.is_contiguous() and h.shape[0] % G == 0 and (h.shape[0] // G - P0) % B == 0
        S = (h.shape[0] // G - P0) // B
        if logits is None:
            logits = torch.empty(G * B, self.shape.vocab, dtype=torch.float32, device=self.device)
        ws = self._workspace("llm", self.lib.rv_llm_ws_bytes(self._ctx, h.shape[0], 1))
        rows = (ctypes.c_int32 * G)(*[int(r) for r in row0s])
        self._persist_begin()
        if last_rows is not None:
            assert last_rows.dtype == torch.int32 and last_rows.is_cuda and last_rows.numel() == G * B
            hip.check(self.lib.rv_llm_prefill_pool_groups_ragged(self._ctx, hip.ptr(h), G, B, P0, S, hip.ptr(kv), kv_rows, rows, Smax, hip.ptr(last_rows),
                                                                 hip.ptr(logits), hip.ptr(ws), ws.numel(), hip.stream()), "rv_llm_prefill_pool_groups_ragged")
        else:
            hip.check(self.lib.rv_llm_prefill_pool_groups(self._ctx, hip.ptr(h), G, B, P0, S, hip.ptr(kv), kv_rows, rows, Smax, hip.ptr(logits), hip.ptr(ws),
                                                          ws.numel(), hip.stream()), "rv_llm_prefill_pool_groups")
        self._persist_end()
        return logits

    def llm_decode_rows(self, h, row_pos, kv, Smax, logits=None, row_share=None):
        """One merged decode step: h f32 [R, D] (clobbered), row_pos int32 [R] on the device (< 0: inactive) -> logits f32 [R, V].
        ``row_share`` int32 [R] on the device (optional): sibling | len << 16 per row - its first ``len`` cache positions are bit-identical to row
        ``sibling``'s (the shared prompt prefix of a generate's rows); the decode attention then reads them from the sibling (L2 hits)."""
        R = h.shape[0]
        assert h.dtype == torch.float32 and h.is_contiguous() and row_pos.dtype == torch.int32 and row_pos.is_cuda
        if logits is None:
            logits = torch.empty(R, self.shape.vocab, dtype=torch.float32, device=self.device)
        ws = self._workspace("llm", self.lib.rv_llm_ws_bytes(self._ctx, R, 1))
        if row_share is not None:
            assert row_share.dtype == torch.int32 and row_share.is_cuda and row_share.numel() == R
            hip.check(self.lib.rv_llm_decode_rows_shared(self._ctx, hip.ptr(h), R, hip.ptr(row_pos), hip.ptr(row_share), hip.ptr(kv), Smax, hip.ptr(logits),
                                                         hip.ptr(ws), ws.numel(), hip.stream()), "rv_llm_decode_rows_shared")
            return logits
        hip.check(self.lib.rv_llm_decode_rows(self._ctx, hip.ptr(h), R, hip.ptr(row_pos), hip.ptr(kv), Smax, hip.ptr(logits), hip.ptr(ws), ws.numel(),
                                              hip.stream()), "rv_llm_decode_rows")
        return logits


def pair_interleave_heads(w, heads):
    """Permute the rows of a q / k projection inside every head: new row 2j = dim j, new row 2j+1 = dim j + dh/2, so the
    rotate_half partners are adjacent (the fused QKV epilogue rotates them in one lane; q.k is permutation invariant)."""
    rows, D = w.shape
    dh = rows // heads
    return w.view(heads, 2, dh // 2, D).transpose(1, 2).reshape(rows, D).contiguous()


def pack_gate_up(gate, up):
    """[F,D] x2 -> [2F,D] with 16-row blocks alternating gate / up."""
    F, D = gate.shape
    assert F % 16 == 0
    return torch.stack([gate.view(F // 16, 16, D), up.view(F // 16, 16, D)], dim=1).reshape(2 * F, D).contiguous()
