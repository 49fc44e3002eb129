"""The model object of the kept API: what ``load_pretrained_model`` returns and ``inference()`` drives.

Surface follows revisionllm/model/vtimellm_llama.py:23-110 + vtimellm_arch.py:10-147 as far as the inference
path touches it: ``generate(input_ids, images=, query_feats=, ...)``, ``get_model()`` (-> ``mm_projector``,
``initialize_vision_modules``, ``config``, adapter flags), ``bfloat16()/cuda()/to()/eval()``, ``device``.
All arithmetic runs in the HIP engine (revisionllm_amd/engine.py); this file is host orchestration:
adapter dispatch, building the splice map, the token loop, EOS bookkeeping and output assembly.
"""
from types import SimpleNamespace

import numpy as np
import torch

from .. import hip, ops
from ..constants import IMAGE_TOKEN_INDEX, MEMORY_TOKEN_INDEX
from ..engine import Engine
from ..utils import synth


class GenerateOutput(dict):
    """dict with attribute access, standing in for HF's ``GenerateDecoderOnlyOutput`` (``out['sequences']``,
    ``out['scores']`` are what the drivers read: inference.py:62, eval_nlq_retrieval_e2e2.py:356)."""
    __getattr__ = dict.get


class GenerationConfig(SimpleNamespace):
    pass


class _Projector:
    """Stand-in for ``model.get_model().mm_projector``: callable like the torch module it replaces."""

    def __init__(self, owner):
        self._o = owner

    def __call__(self, images, query_feats=None, query_mask=None, iteration_step=None):
        o = self._o
        if not o.clip_adapter:
            return o.engine.project_dense(images)
        feature = "cls" if (o.hierarchy or o.clip_adapter_feature == "cls") else "all"
        y = o.engine.clip_encoder(images, query_feats, query_mask, feature)
        if feature == "cls":
            return y[:, None]
        return y[:, 1:] if o.clip_adapter_feature == "temporal" else y


class _Inner:
    """``model.get_model()``: adapter topology flags + ``initialize_vision_modules`` (vtimellm_arch.py:12-73)."""

    def __init__(self, owner):
        self._owner = owner
        self.config = owner.config
        self.clip_adapter = False
        self.clip_adapter_text = False
        self.clip_adapter_feature = "cls"
        self.hierarchy = False
        self.pretrain_clip_adapter = None
        self.mm_projector = None

    @property
    def engine(self):
        return self._owner.engine

    def initialize_vision_modules(self, model_args, state_dict=None):
        """Select the adapter topology from the duck-typed args namespace (same fields as the reference reads)
        and bind its weights.  ``state_dict`` (adapter-relative names) overrides the ``pretrain_*`` files."""
        from . import builder
        a = model_args
        assert not getattr(a, "clip_adapter", False) or not getattr(a, "cross_attn", False), \
            "both clip_adapter and cross_attn cannot be true"
        self.cross_attn_variant = bool(getattr(a, "cross_attn", False))
        # cross_attn=True WITHOUT pretrain_clip_adapter (vtimellm_arch.py:52-57 -> transformer.py:65-67,86): the separate ``cross_attn`` module
        # is a ClipEncoder as wide as the LLM (d_model = hidden_size, 8 heads of hidden / 8) with a ``text_mm_projector`` in front and no
        # output projector; ``mm_projector`` stays the Linear(768 -> hidden) and runs FIRST (vtimellm_arch.py:125, then :127-144)
        self.cross_attn_dense = self.cross_attn_variant and getattr(a, "pretrain_clip_adapter", None) is None
        if self.cross_attn_dense and self.config.hidden_size != 4096:
            raise NotImplementedError("the hidden-wide cross_attn ClipEncoder is built for hidden_size = 4096 (8 heads of 512)")
        # chapters variant (scripts/chapters/*.sh: --cross_attn True --pretrain_clip_adapter ...): mm_projector is a Linear
        # whose output is computed and then DISCARDED (vtimellm_arch.py:125,132-133); the separate ``cross_attn`` module is a
        # 768-d ClipEncoder applied to the RAW features.  That is exactly the clip_adapter data path with the weights
        # living under ``cross_attn.`` instead of ``mm_projector.``.
        self.clip_adapter = bool(getattr(a, "clip_adapter", False)) or self.cross_attn_variant
        self.clip_adapter_text = bool(getattr(a, "clip_adapter_text", False))
        self.clip_adapter_feature = getattr(a, "clip_adapter_feature", "cls")
        if not isinstance(self.clip_adapter_feature, str):  # e2e2.py:72 declares it type=bool
            self.clip_adapter_feature = "cls"
        self.hierarchy = bool(getattr(a, "hierarchy", False))
        self.pretrain_clip_adapter = getattr(a, "pretrain_clip_adapter", None)
        eng = self._owner._ensure_engine(adapter_text=self.clip_adapter_text if self.clip_adapter else None,
                                         adapter_dim=self.config.hidden_size if self.cross_attn_dense else 768)
        if self.cross_attn_dense and state_dict is not None:
            # adapter-relative names of BOTH modules: 'mm_projector.{weight,bias}' (the Linear) and the ClipEncoder's own names
            eng.load_linear_projector(lambda n: state_dict["mm_projector." + n])
            eng.load_clip_adapter(lambda n: state_dict[n])
            state_dict = None
        if state_dict is None and not self.cross_attn_dense:
            path = self.pretrain_clip_adapter if self.clip_adapter else getattr(a, "pretrain_mm_mlp_adapter", None)
            if path is not None:
                state_dict = builder.remap_projector_keys(torch.load(path, map_location="cpu"), clip=self.clip_adapter)
        if state_dict is not None:
            if self.clip_adapter:
                eng.load_clip_adapter(lambda n: state_dict[n])
            else:
                eng.load_linear_projector(lambda n: state_dict[n])
        self.mm_projector = _Projector(self)
        if self.cross_attn_variant:
            self.cross_attn = self.mm_projector
        if self.clip_adapter_feature == "alternate":
            # vtimellm_arch.py:72-73: nn.LayerNorm(hidden) applied to the adapter output (:146-147); a fresh LayerNorm is the
            # identity affine - ``load_alternate_layer_norm`` binds trained values ("model.alternate_layer_norm.{weight,bias}")
            self.load_alternate_layer_norm(torch.ones(self.config.hidden_size), torch.zeros(self.config.hidden_size))

    def load_alternate_layer_norm(self, weight, bias):
        dev = self.engine.device
        self.alternate_layer_norm = (weight.to(device=dev, dtype=torch.float32).contiguous(), bias.to(device=dev, dtype=torch.float32).contiguous())


class ReVisionLlamaForCausalLM:
    """Drop-in for ``VTimeLLMLlamaForCausalLM`` on the inference path."""

    def __init__(self, shape: synth.LlamaShape = synth.VICUNA_7B, device="cuda:0", max_sequence_length=None, engine=None, op_dtype=None):
        """``op_dtype``: "f16" (default) / "bf16": the 16-bit operand type of the engine this model creates (ignored with ``engine=``).
        ``engine``: an existing ``Engine`` to share (its LLM weights, workspaces and options): a second model object with
        another adapter topology - e.g. the dense Linear projector next to the hierarchy ClipEncoder - then costs no second copy
        of the 13.5 GB of LLM weights."""
        self.shape = shape
        self.config = SimpleNamespace(hidden_size=shape.hidden, intermediate_size=shape.inter, num_hidden_layers=shape.layers,
                                      num_attention_heads=shape.heads, vocab_size=shape.vocab, rms_norm_eps=shape.eps,
                                      rope_theta=shape.theta, model_type="VTimeLLM", pad_token_id=0, bos_token_id=1, eos_token_id=2)
        if max_sequence_length is not None:
            self.config.max_sequence_length = max_sequence_length
        self.generation_config = GenerationConfig(top_k=50, top_p=1.0, temperature=1.0, eos_token_id=2, pad_token_id=0)
        self._device = torch.device(device) if engine is None else engine.device
        self.engine = engine
        self.model = _Inner(self)
        self.scores_mode = "processed"  # what transformers>=4.39 returns; "raw" = the override's intent (vtimellm_llama.py:321)
        self.op_flavour = engine.flavour if engine is not None else hip.flavour_of(op_dtype)
        self.dtype = hip.op_dtype(self.op_flavour)
        self.uniform_fn = None  # optional (step, B) -> uniforms hook for reproducible sampling through inference()
        self.after_prefill = None  # optional () -> None hook called between the prefill and the decode loop of generate()

    # ---- plumbing -------------------------------------------------------------------------------
    def _ensure_engine(self, adapter_text=None, adapter_dim=768):
        if self.engine is None:
            self.engine = Engine(self.shape, adapter_text=bool(adapter_text), device=self._device, adapter_dim=adapter_dim, op_dtype=self.op_flavour)
        elif adapter_text is not None and (bool(adapter_text) != self.engine.adapter_text or adapter_dim != self.engine.adapter_dim):
            raise RuntimeError("the engine's ClipEncoder topology (text-conditioned layers on / off, width) differs from this model's")
        return self.engine

    def get_model(self):
        return self.model

    @property
    def device(self):
        return self._device

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    _dtype_note_done = False

    def _dtype_request(self, what, asked):
        """``model.bfloat16()`` / ``.half()`` / ``.to(dtype)`` keep the reference's call sites working (e2e2.py:181-185) but do NOT pick the arithmetic:
        the operand type belongs to the library build this model was created for (``args.op_dtype`` / ``--op_dtype``; fp16 by default).  A request for
        the OTHER 16-bit type is answered once per process with a note instead of silently."""
        if asked is not None and asked != self.dtype and asked in (torch.float16, torch.bfloat16) and not ReVisionLlamaForCausalLM._dtype_note_done:
            ReVisionLlamaForCausalLM._dtype_note_done = True
            print(f"[revisionllm_amd] {what}: ignored - this model computes with {str(self.dtype).replace('torch.', '')} operands "
                  f"(librevision_hip{'_bf16' if self.op_flavour == 'bf16' else ''}.so, f32 accumulation and residual stream); pass op_dtype="
                  f"'{'bf16' if asked == torch.bfloat16 else 'f16'}' to load_pretrained_model's args (--op_dtype) to change the operand type", flush=True)
        return self

    def bfloat16(self):
        return self._dtype_request(".bfloat16()", torch.bfloat16)

    def half(self):
        return self._dtype_request(".half()", torch.float16)

    def float(self):
        return self

    def to(self, *args, **kwargs):
        """Weights live in HBM as 16-bit operands (fp16 by default, ``op_dtype``) with fp32 accumulation / residual stream; dtype moves are accepted and ignored.
        NOTE (differs from the reference): ``scores`` / ``logits`` returned by ``generate`` are float32 whatever dtype was
        asked for here (the reference returns them in the model dtype, bf16 on the GPU)."""
        asked = kwargs.get("dtype", next((a for a in args if isinstance(a, torch.dtype)), None))
        return self._dtype_request(f".to({asked})", asked)

    # ---- adapter dispatch (vtimellm_arch.py:102-147) ---------------------------------------------
    def encode_images(self, images, query_feats, iteration_step=None):
        """-> (video_rows f32 [R,D], rows_per_sample).  hierarchy: [b,v,t,d] -> v rows per sample;
        ClipEncoder: [b,t,d] -> 1 (cls) / t (temporal) rows; Linear: [b,t,d] -> t rows.
        ``clip_adapter_feature='alternate'`` (transformer.py:134-138, vtimellm_arch.py:112-123,146-147): ``iteration_step`` even ->
        the CLS row(s), odd -> the T temporal rows of [b,t,d] features; then ``alternate_layer_norm`` over the hidden dim."""
        m = self.model
        eng = self.engine
        if m.clip_adapter and m.clip_adapter_feature == "alternate":
            if iteration_step is None:   # the reference evaluates ``iteration_step % 2`` (transformer.py:135)
                raise TypeError("unsupported operand type(s) for %: 'NoneType' and 'int' (clip_adapter_feature='alternate' needs iteration_step)")
            if isinstance(images, (list, tuple)):
                # the reference sends list inputs to ``mm_projector(concat_images)`` without query features (vtimellm_arch.py:102-108),
                # which a ClipEncoder's forward rejects: same outcome here, said in words
                raise TypeError("clip_adapter_feature='alternate': a list of feature tensors is not a valid input of the ClipEncoder adapter "
                                "(the reference's list branch calls mm_projector without query features, vtimellm_arch.py:102-108)")
            if query_feats is None:      # the reference subscripts it unconditionally (vtimellm_arch.py:113-122)
                raise TypeError("'NoneType' object is not subscriptable (clip_adapter_feature='alternate' needs query_feats=(feats, mask))")
            qf, qm = query_feats[0], query_feats[1]
            # hierarchy models at an ODD step: the reference forwards ``images`` as they are to the encoder (vtimellm_arch.py:112-113), which
            # only accepts [b,t,d] - a [b,v,t,d] tensor fails there in ``src.permute(1, 0, 2)`` and is refused below for the same reason
            if int(iteration_step) % 2 == 0:
                if m.hierarchy:
                    b, v, t, d = images.shape
                    y, rps = eng.clip_encoder(images.reshape(b * v, t, d), qf, qm, "cls"), v
                else:
                    y, rps = eng.clip_encoder(images, qf, qm, "cls"), 1
            else:
                if images.dim() != 3:
                    raise ValueError("clip_adapter_feature='alternate' at an odd iteration_step projects the temporal rows of [b,t,d] features "
                                     "(vtimellm_arch.py:112-113)")
                y = eng.clip_encoder(images, qf, qm, "all")[:, 1:].reshape(-1, self.shape.hidden)
                rps = images.shape[1]
            w, b_ = m.alternate_layer_norm
            return ops.layernorm(y.contiguous(), w, b_, want=("f32",))[0], rps
        if isinstance(images, (list, tuple)):
            images = torch.cat(list(images), dim=0)
        if not m.clip_adapter:
            y = eng.project_dense(images)
            return y.reshape(-1, self.shape.hidden), images.shape[1]
        qf, qm = (query_feats[0], query_feats[1]) if query_feats is not None else (None, None)
        if m.hierarchy:
            b, v, t, d = images.shape
            y = eng.clip_encoder(images.reshape(b * v, t, d), qf, qm, "cls")
            return y, v
        if m.clip_adapter_feature == "cls":
            return eng.clip_encoder(images, qf, qm, "cls"), 1
        y = eng.clip_encoder(images, qf, qm, "all")
        if m.clip_adapter_feature == "temporal":
            return y[:, 1:].reshape(-1, self.shape.hidden), images.shape[1]
        # any other feature value: the reference projects all T+1 rows, CLS first (transformer.py:143-144)
        return y.reshape(-1, self.shape.hidden), images.shape[1] + 1

    @staticmethod
    def build_row_map_ragged(input_ids, rows_per_sample, pad_id=0):
        """``build_row_map`` for samples with DIFFERENT numbers of video rows (``rows_per_sample``: one count per sample; the video rows of
        sample b follow those of sample b - 1): rows are right-padded with ``pad_id`` to the longest one.  -> (int32 [B, S], lengths)."""
        out, lens, off = [], [], 0
        for b, row in enumerate(input_ids.cpu().tolist()):
            if MEMORY_TOKEN_INDEX in row or row.count(IMAGE_TOKEN_INDEX) != 1:
                raise NotImplementedError("ragged generates take exactly one <video> per sample and no <memory>")
            n = int(rows_per_sample[b])
            i = row.index(IMAGE_TOKEN_INDEX)
            out.append(row[:i] + [-(off + k + 1) for k in range(n)] + row[i + 1:])
            off += n
            lens.append(len(out[-1]))
        S = max(lens)
        return torch.tensor([r + [pad_id] * (S - len(r)) for r in out], dtype=torch.int32), lens

    @staticmethod
    def build_row_map(input_ids, rows_per_sample, attention_mask=None, memory=None):
        """Splice plan (vtimellm_arch.py:149-238): int32 [B,S]; >= 0 token id, < 0 -> video row -(v+1).
        Sample b consumes video rows [b*rows_per_sample, (b+1)*rows_per_sample) at its -200 slot.
        ``memory`` = (prefix ids [B][Lp], M, base): the ``<memory>`` path (vtimellm_arch.py:179-232, round 5) - the -300 slot of sample b becomes its
        prefix-memory TOKENS (``embed_tokens(prefix_memory)`` is what any token of the row gets) followed by the M projected memory rows
        base + b*M .. of the same feature-row table; the slot must follow the video (the reference's chunk order: text, video, text, memory, text)."""
        ids = input_ids.cpu().tolist()
        mask = attention_mask.cpu().tolist() if attention_mask is not None else None
        out = []
        for b, row in enumerate(ids):
            if mask is not None:
                row = [t for t, m in zip(row, mask[b]) if m]
            if (MEMORY_TOKEN_INDEX in row) != (memory is not None):
                raise ValueError("a <memory> slot (-300) needs visual_memory / prefix_memory and the other way round (inference.py:29-30 appends the "
                                 "marker exactly when a memory is given)")
            r, used, used_mem = [], 0, 0
            for t in row:
                if t == IMAGE_TOKEN_INDEX:
                    if used:
                        raise NotImplementedError("more than one <video> per sample")
                    r.extend(-(b * rows_per_sample + i + 1) for i in range(rows_per_sample))
                    used = 1
                elif t == MEMORY_TOKEN_INDEX:
                    if used_mem or not used:
                        raise NotImplementedError("one <memory> per sample, behind its <video> (vtimellm_arch.py:207-232)")
                    prefix, M, base = memory
                    r.extend(int(x) for x in prefix[b])
                    r.extend(-(base + b * M + i + 1) for i in range(M))
                    used_mem = 1
                else:
                    r.append(t)
            out.append(r)
        if len({len(r) for r in out}) != 1:
            raise NotImplementedError("ragged batches: every row of a generate() call must have the same length "
                                      "(inference() repeats one prompt, inference.py:36)")
        return torch.tensor(out, dtype=torch.int32)

    # ---- generate ----------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids, images=None, query_feats=None, **kwargs):
        """Prefill + KV-cached sampling loop (inference.py:45-59 kwargs); see ``generate_steps`` for the build-defined kwargs.
        Drives the step generator to the end, waiting on the host wherever it asks to."""
        gen = self.generate_steps(input_ids, images=images, query_feats=query_feats, **kwargs)
        try:
            while True:
                ev = next(gen)
                if ev is not None:
                    ev.synchronize()
        except StopIteration as stop:
            return stop.value

    def generate_steps(self, input_ids, images=None, query_feats=None, do_sample=False, temperature=None, num_beams=1,
                       max_new_tokens=None, use_cache=True, visual_memory=None, prefix_memory=None, output_scores=False,
                       return_dict_in_generate=False, output_hidden_states=False, output_logits=False, top_k=None, top_p=None,
                       attention_mask=None, uniforms=None, forced_tokens=None, video_rows=None, rows_per_sample=None,
                       share_prefix=True, eos_lookahead=1, server=None, iteration_step=None, new_tokens_only=False, **kwargs):
        """``generate`` as a generator: enqueues device work and YIELDS a ``torch.cuda.Event`` whenever the host has to learn
        something from the device before it may enqueue more - which only happens with an EOS id configured: the "all rows
        finished" flag of step s is copied to pinned host memory asynchronously and looked at only after step s + ``eos_lookahead``
        has been enqueued, so the device always has work queued (no per-step drain of the launch queue) and a scheduler can
        interleave several generates on different HIP streams (``run_interleaved``): while one waits for its flag, the others'
        launches are enqueued.  At most ``eos_lookahead`` steps are computed past the EOS step; they are cut off on the host
        (finished rows emit the pad id from the EOS step on, exactly like HF's ``_sample`` bookkeeping).  Returns (via
        StopIteration) what ``generate`` returns.

        Extra, build-defined kwargs: ``uniforms`` [G,B] (host-supplied draws for reproducible sampling; default
        ``torch.rand`` on the device), ``forced_tokens`` [G,B] (teacher forcing for parity tests),
        ``video_rows`` / ``rows_per_sample`` (pre-encoded adapter output, used by the batched recursion),
        ``share_prefix`` (the text prefix common to all rows goes through the prefill once; bit-identical results),
        ``server`` (a ``serve.DecodeServer``: prefill into its KV pool, then let ITS merged steps decode this generate's rows
        together with the other generates in flight - same tokens / entropies, one pass over the weights per step for all of
        them; only under ``sched.Interleaver(servers=[server])``; falls back to the loop below when it does not apply).
        ``iteration_step``: the reference's ``forward`` kwarg for ``clip_adapter_feature='alternate'`` (vtimellm_llama.py:55).
        ``new_tokens_only`` (with ``return_dict_in_generate``, through a server): ``out['new_tokens']`` int32 [B, G] and no ``sequences``
        (the prompt is not copied in front of them: the recursion drivers read only the new tokens).
        ``output_hidden_states`` is accepted and ignored: nothing on the path reads it.
        """
        if num_beams != 1:
            raise NotImplementedError("beam search is not on the grounding path (num_beams=1, inference.py:51)")
        if (visual_memory is None) != (prefix_memory is None):
            raise ValueError("visual_memory and prefix_memory come together (vtimellm_arch.py:220-222 concatenates them)")
        eng = self._ensure_engine()
        gc = self.generation_config
        temperature = gc.temperature if temperature is None else temperature
        top_k = gc.top_k if top_k is None else top_k
        top_k = 0 if top_k is None else int(top_k)      # HF: None / 0 = no top-k filter (the sampling kernel's top_k = 0 path)
        top_p = gc.top_p if top_p is None else top_p
        max_new_tokens = 20 if max_new_tokens is None else max_new_tokens
        eos, pad = gc.eos_token_id, gc.pad_token_id
        dev = eng.device

        if video_rows is None:
            if images is None:
                video_rows, rows_per_sample = None, 0
            else:
                video_rows, rows_per_sample = self.encode_images(images, query_feats, iteration_step)
        # host inputs go up once, before the loop, through pinned non-blocking copies: a pageable upload inside the loop
        # would make the host wait for the whole queue at every step and the launch queue would run dry behind it
        if uniforms is not None:
            uniforms = ops.h2d(uniforms, dev, torch.float32)
        if forced_tokens is not None:
            forced_tokens = ops.h2d(forced_tokens, dev, torch.long)
        memory = None
        if visual_memory is not None:
            # <memory> prompts (inference.py:29-30; vtimellm_arch.py:179-232): [text, video, text, embed(prefix_memory) ++ mm_projector(vis_mem), text].
            # The reference calls ``mm_projector(vis_mem)`` with one argument (arch.py:222): that is the Linear projector - a ClipEncoder adapter
            # fails there (transformer.py:119, ``src_txt`` None), so it is refused here with the reason instead of an AttributeError.
            if self.model.clip_adapter:
                raise NotImplementedError("visual_memory with a ClipEncoder adapter: the reference's mm_projector(vis_mem) call (vtimellm_arch.py:222) passes no "
                                          "query features and fails in ClipEncoder.forward (transformer.py:119); only the Linear projector runs this path")
            if video_rows is None or isinstance(rows_per_sample, (list, tuple)):
                raise NotImplementedError("visual_memory needs the video features of an ordinary (non-ragged) generate")
            vis = visual_memory[:, None] if visual_memory.dim() == 2 else visual_memory            # arch.py:220
            pm = torch.as_tensor(prefix_memory).long().cpu()
            if pm.dim() != 2 or pm.shape[0] != vis.shape[0] or vis.shape[0] != input_ids.shape[0]:
                raise ValueError("visual_memory [B,768] / [B,M,768] and prefix_memory int [B,Lp]: one memory per row of input_ids")
            mem_rows = eng.project_dense(vis).reshape(-1, self.shape.hidden)
            memory = (pm.tolist(), int(vis.shape[1]), int(video_rows.reshape(-1, self.shape.hidden).shape[0]))
            video_rows = video_rows.reshape(-1, self.shape.hidden)
            video_rows = torch.cat([video_rows, mem_rows.to(video_rows.dtype)], 0)
        lens = None
        if isinstance(rows_per_sample, (list, tuple)) and len(set(rows_per_sample)) > 1:
            # Samples with different numbers of video rows (the 9 calls of a 33-window recursion: 8 x 32 and 1 x 33): one generate of
            # right-padded sequences - under causal attention a valid position never sees a later (pad) one, so every row's tokens are what
            # its own generate would produce.  Only through a DecodeServer (rows decode at their own positions there anyway).
            if attention_mask is not None:
                raise NotImplementedError("ragged generates (different numbers of video rows per sample) do not take an attention_mask")
            row_map, lens = self.build_row_map_ragged(input_ids, rows_per_sample, pad if pad is not None else 0)
            ragged_ok = (server is not None and getattr(server, "prefill_batch", 1) > 1 and not output_scores and not output_logits
                         and self.after_prefill is None and max_new_tokens >= 1 and server.fits(row_map.shape[1], max_new_tokens, row_map.shape[0]))
            if not ragged_ok:
                # no DecodeServer pass can take the padded batch (no server, pool rows / Smax / gmax too small, scores asked for): the samples run as
                # equal-geometry sub-batches through the ordinary path - what the drivers did before ragged generates existed
                return (yield from self._generate_split_by_rows(input_ids, video_rows, rows_per_sample, uniforms, forced_tokens, pad,
                                                                dict(do_sample=do_sample, temperature=temperature, max_new_tokens=max_new_tokens, output_scores=output_scores,
                                                                     return_dict_in_generate=return_dict_in_generate, output_logits=output_logits, top_k=top_k, top_p=top_p,
                                                                     share_prefix=share_prefix, eos_lookahead=eos_lookahead, server=server, new_tokens_only=new_tokens_only)))
        else:
            if isinstance(rows_per_sample, (list, tuple)):
                rows_per_sample = int(rows_per_sample[0]) if len(rows_per_sample) else 0
            row_map = self.build_row_map(input_ids, rows_per_sample, attention_mask, memory)
        B, S = row_map.shape
        P0 = self._common_text_prefix(row_map) if (share_prefix and B > 1) else 0
        job = None
        if (server is not None and not output_scores and not output_logits and self.after_prefill is None
                and server.fits(S, max_new_tokens, B) and max_new_tokens >= 1):
            job = server.reserve(B)
            while job is None and getattr(server, "blocking", False):      # gang policy: wait for a pool to fill rather than decode alone
                from .. import sched
                yield sched.RETRY
                job = server.reserve(B)
        if job is not None:
            try:
                return (yield from self._generate_in_pool(server, job, eng, dev, input_ids, row_map, video_rows, B, S, P0, do_sample, temperature, top_k,
                                                          top_p, max_new_tokens, uniforms, forced_tokens, return_dict_in_generate, lens,
                                                          new_tokens_only))
            finally:
                if not job.finished:        # an exception here or in a task this one was pumped from, or the task was cancelled
                    server.abandon(job)
        if lens is not None:
            # a non-blocking server whose pools are momentarily full: the same fallback - equal-geometry sub-batches (each decodes in the pool if
            # it finds room, alone otherwise), so whether a recursion completes never depends on pool occupancy
            return (yield from self._generate_split_by_rows(input_ids, video_rows, rows_per_sample, uniforms, forced_tokens, pad,
                                                            dict(do_sample=do_sample, temperature=temperature, max_new_tokens=max_new_tokens, output_scores=output_scores,
                                                                 return_dict_in_generate=return_dict_in_generate, output_logits=output_logits, top_k=top_k, top_p=top_p,
                                                                 share_prefix=share_prefix, eos_lookahead=eos_lookahead, server=server, new_tokens_only=new_tokens_only)))
        # A generate that decodes alone owns its engine slot's KV cache and workspace from its prefill to its last step.  Under a cooperative
        # scheduler (it yields at the EOS flag polls) a second generate started on the SAME slot meanwhile would be handed the same recycled
        # cache and overwrite it: refuse loudly - every call in flight needs its own slot (sched.Task(..., slot=i); engine.slot).
        slot = eng.slot
        if slot in eng.slots_in_flight:
            raise RuntimeError(f"engine slot {slot} already has a generate in flight: give every call in flight its own workspace slot "
                               "(sched.Task(..., slot=i) / engine.slot) - two generates on one slot share one recycled KV cache")
        eng.slots_in_flight.add(slot)
        try:
            return (yield from self._generate_alone(eng, dev, input_ids, row_map, video_rows, B, S, P0, do_sample, temperature, top_k, top_p,
                                                    max_new_tokens, uniforms, forced_tokens, return_dict_in_generate, output_scores, output_logits,
                                                    eos, pad, eos_lookahead))
        finally:
            eng.slots_in_flight.discard(slot)

    def _generate_split_by_rows(self, input_ids, video_rows, rows_per_sample, uniforms, forced_tokens, pad, kw):
        """A batch whose samples present different numbers of video rows, run as one ``generate_steps`` per row count (sample order kept inside a
        sub-batch) and merged back into batch order.  Sub-batches that stop at different steps (EOS) are right-padded with the pad id / zeros."""
        counts = [int(r) for r in rows_per_sample]
        starts = [0]
        for r in counts:
            starts.append(starts[-1] + r)
        if kw.get("output_scores") or kw.get("output_logits"):
            raise NotImplementedError("ragged generates return no per-step scores / logits: call generate per equal-geometry group for those")
        parts = []
        for r in sorted(set(counts)):
            sel = [i for i, c in enumerate(counts) if c == r]
            rows = torch.cat([video_rows[starts[i]:starts[i + 1]] for i in sel], 0) if video_rows is not None else None
            out = yield from self.generate_steps(input_ids[sel], video_rows=rows, rows_per_sample=r,
                                                 uniforms=None if uniforms is None else uniforms[:, sel],
                                                 forced_tokens=None if forced_tokens is None else forced_tokens[:, sel], **dict(kw, return_dict_in_generate=True))
            parts.append((sel, out))
        B = len(counts)
        merged = GenerateOutput()
        for key in ("sequences", "new_tokens", "entropy", "entropy_raw"):
            vals = [(sel, out.get(key)) for sel, out in parts]
            if any(v is None for _, v in vals):
                continue
            width = max(v.shape[1] for _, v in vals)
            fill = (pad if pad is not None else 0) if key in ("sequences", "new_tokens") else 0
            full = torch.full((B, width), fill, dtype=vals[0][1].dtype, device=vals[0][1].device)
            for sel, v in vals:
                full[torch.as_tensor(sel, device=full.device), :v.shape[1]] = v
            merged[key] = full
        if not kw.get("return_dict_in_generate"):
            return merged["sequences"]
        return merged

    def _generate_in_pool(self, server, job, eng, dev, input_ids, row_map, video_rows, B, S, P0, do_sample, temperature, top_k, top_p,
                          max_new_tokens, uniforms, forced_tokens, return_dict_in_generate, lens=None, new_tokens_only=False):
        """The merged-decode path of ``generate_steps``: prefill into the server's pool, its merged steps do the rest."""
        pool = job.pool
        for ev in job.free_events:
            torch.cuda.current_stream(dev).wait_event(ev)                   # the rows' previous owners are done with them
        batched = getattr(server, "prefill_batch", 1) > 1
        if not batched and getattr(server, "fifo_prefill", False) and server.prefill_tail is not None:
            # prefills in launch order: each one saturates the GPU anyway, and the pool that fills first then decodes (HBM-bound)
            # under the NEXT pool's prefills (MFMA-bound) instead of all prefills in flight finishing together
            torch.cuda.current_stream(dev).wait_event(server.prefill_tail)
        if 16 <= P0 < S:
            flat = torch.cat([row_map[0, :P0], row_map[:, P0:].reshape(-1)])[None]
            h, p0 = eng.splice_embed(flat, video_rows)[0], P0
        else:
            h, p0 = eng.splice_embed(row_map, video_rows).view(B * S, -1), 0
        if lens is not None and not batched:
            raise NotImplementedError("ragged generates need a DecodeServer with prefill_batch > 1 (its prefill passes carry the per-sequence last rows)")
        if do_sample and uniforms is None:
            # drawn BEFORE the prefill is submitted: the ticket / ``ready`` event chain recorded behind it then covers the rand kernel, and the
            # pool stream's copy of the whole [steps, B] block at join time (gated by ``ready`` only) can never run ahead of it
            uniforms = (torch.rand(max_new_tokens, B, device=dev) if self.uniform_fn is None
                        else torch.stack([self.uniform_fn(s_, B).to(dev).float() for s_ in range(max_new_tokens)]))
        if batched:      # the server batches the waiting prefills of identical geometry into one pass (its own stream, FIFO)
            ticket = server.submit_prefill(job, h, B, p0, lens)
            from .. import sched
            while ticket.ready is None:
                yield sched.RETRY
            first, ready = ticket.first, ticket.ready
            torch.cuda.current_stream(dev).wait_event(ready)
        else:
            first = eng.llm_prefill_pool(h, B, p0, pool.kv, pool.R, job.r0, pool.Smax)
            ready = None
        if ready is None:
            ready = torch.cuda.Event()
            ready.record()
            if getattr(server, "fifo_prefill", False):
                server.prefill_tail = ready
        yield ready                     # join only once the prefill has COMPLETED: the decode stream never waits for a prefill
        start = S if lens is None else [int(n) for n in lens]      # (ragged: every row decodes from its own length; the server uploads it on ITS stream)
        server.join(job, start, first, ready, max_new_tokens, (bool(do_sample), float(temperature), int(top_k), float(top_p if top_p is not None else 1.0)),
                    uniforms=uniforms if do_sample else None, forced=forced_tokens, shared_prefix=p0)
        from .. import sched
        while not job.finished:
            yield sched.RETRY           # the scheduler pumps the server's merged steps
        yield job.done_event
        if new_tokens_only and return_dict_in_generate:
            return GenerateOutput(sequences=None, new_tokens=job.tokens, entropy=job.entropy, entropy_raw=job.entropy_raw)
        seqs = torch.cat([ops.h2d(input_ids, dev, torch.long), job.tokens.long()], dim=1)
        if not return_dict_in_generate:
            return seqs
        return GenerateOutput(sequences=seqs, entropy=job.entropy, entropy_raw=job.entropy_raw)

    def _generate_alone(self, eng, dev, input_ids, row_map, video_rows, B, S, P0, do_sample, temperature, top_k, top_p, max_new_tokens,
                        uniforms, forced_tokens, return_dict_in_generate, output_scores, output_logits, eos, pad, eos_lookahead):
        """The classic loop of ``generate_steps``: this generate's own KV cache and decode passes."""
        cap = min(max_new_tokens, 64)
        kv, Smax = eng.new_kv(B, S + cap)
        if 16 <= P0 < S:     # (P0 == S: identical text-only rows - nothing per-row is left, take the plain prefill)
            # every row starts with the same P0 text tokens (inference() repeats one prompt): under causal attention their
            # hidden states and K/V are identical for all rows, so they ride through the prefill once (rows of the
            # GEMM batch: [P0 shared ; B x (S - P0)]) and their K/V are written into every row's cache
            flat = torch.cat([row_map[0, :P0], row_map[:, P0:].reshape(-1)])[None]
            h = eng.splice_embed(flat, video_rows)[0]
            logits = eng.llm_prefill_shared(h, B, P0, kv, Smax)
        else:
            h = eng.splice_embed(row_map, video_rows)
            logits = eng.llm_forward(h, 0, kv, Smax)
        if self.after_prefill is not None:   # scheduling hook (e.g. gate this call's decode steps on another stream's prefill)
            self.after_prefill()

        seqs = ops.h2d(input_ids, dev, torch.long)
        unfinished = torch.ones(B, dtype=torch.int32, device=dev)
        raw_steps, score_steps, ent_p, ent_r, new_tokens = [], [], [], [], []
        pos = S
        if do_sample and uniforms is None and self.uniform_fn is None:
            uniforms = torch.rand(max_new_tokens, B, device=dev)        # one launch instead of one per step
        flags = []      # per step: (pinned int32 [1] = max over rows of "unfinished" after the step, event of its copy)
        if eos is not None:
            flag_host = torch.empty(max_new_tokens, dtype=torch.int32).pin_memory()
        n_steps = 0
        for step in range(max_new_tokens):
            if do_sample:
                if uniforms is not None:
                    u = uniforms[step].contiguous()
                else:
                    u = self.uniform_fn(step, B).to(dev).float().contiguous()
                o = ops.sample(logits, u, True, temperature, top_k, top_p, ctx=eng)
            else:
                o = ops.sample(logits, None, False, ctx=eng)
            nxt = o["tokens"] if forced_tokens is None else forced_tokens[step].int()     # int32 [B]
            ent_p.append(o["entropy_proc"])
            ent_r.append(o["entropy_raw"])
            if output_logits or (output_scores and (not do_sample or self.scores_mode == "raw")):
                raw_steps.append(logits.clone())
            if output_scores and do_sample and self.scores_mode == "processed" and (not top_k or top_k > hip.TOPK_CAP):
                # no top-k filter (top_k = 0 / None) or one wider than the candidate list (top_k > 64): there is no list - the processed scores are
                # logits / T with everything below the kernel's threshold (the smallest score the top-k and top-p filters keep) at -inf
                # - computed EXACTLY as the kernel computes them (sample.hip: v * (1.0f / temperature), both in f32): logits / T differs from that by
                # one ulp for about one value in seven at T = 0.05, and a smallest kept score that lands one ulp BELOW the threshold would be masked
                # although it was kept (with one kept token the whole row would turn -inf)
                sc = logits * float(np.float32(1.0) / np.float32(temperature))
                score_steps.append(sc.masked_fill(sc < o["threshold"][:, None], float("-inf")))
            elif output_scores and do_sample and self.scores_mode == "processed":
                V = logits.shape[1]
                sc = torch.full((B, V + 1), float("-inf"), device=dev)      # column V swallows the dropped candidates
                keep = torch.arange(hip.TOPK_CAP, device=dev)[None] < o["n_keep"][:, None]
                idx = torch.where(keep, o["topk_idx"], torch.full_like(o["topk_idx"], V)).long()
                score_steps.append(sc.scatter(1, idx, o["topk_val"])[:, :V].contiguous())
            if eos is not None:   # rows that already emitted EOS keep producing the pad id (HF _sample bookkeeping)
                nxt = nxt * unfinished + pad * (1 - unfinished)
                unfinished = unfinished * (nxt != eos).int()
                flag_host[step:step + 1].copy_(unfinished.max().reshape(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                flags.append(ev)
            new_tokens.append(nxt)
            n_steps = step + 1
            if step == max_new_tokens - 1:
                break
            if eos is not None and step >= eos_lookahead:
                # the flag of step (step - lookahead): its copy was enqueued ``lookahead`` steps ago - usually complete already
                yield flags[step - eos_lookahead]
                if int(flag_host[step - eos_lookahead]) == 0:
                    n_steps = step - eos_lookahead + 1
                    break
            if pos + 1 > Smax:
                kv, Smax = self._grow_kv(kv, B, Smax, min(S + max_new_tokens, Smax * 2))
            h1 = eng.splice_embed(nxt[:, None], None)
            logits = eng.llm_forward(h1, pos, kv, Smax)
            pos += 1
        if eos is not None and n_steps == len(new_tokens) and n_steps > 0:
            # the steps not yet looked at: the generate may have finished inside the lookahead window
            for s_ in range(max(0, n_steps - eos_lookahead - 1), n_steps):
                yield flags[s_]
                if int(flag_host[s_]) == 0:
                    n_steps = s_ + 1
                    break

        def cut(lst):
            return lst[:n_steps]
        new_tokens, ent_p, ent_r, raw_steps, score_steps = cut(new_tokens), cut(ent_p), cut(ent_r), cut(raw_steps), cut(score_steps)
        seqs = torch.cat([seqs, torch.stack(new_tokens, dim=1).long()], dim=1)
        if not return_dict_in_generate:
            return seqs
        out = GenerateOutput(sequences=seqs, entropy=torch.stack(ent_p, 1), entropy_raw=torch.stack(ent_r, 1))
        if output_scores:
            out["scores"] = tuple(score_steps) if (do_sample and self.scores_mode == "processed") else tuple(raw_steps)
        if output_logits:
            out["logits"] = tuple(raw_steps)
        return out

    @staticmethod
    def _common_text_prefix(row_map):
        """Number of leading positions that hold the same token id (not a video row) in every row."""
        same = (row_map == row_map[:1]).all(dim=0) & (row_map[0] >= 0)
        bad = (~same).nonzero()
        return int(bad[0]) if bad.numel() else row_map.shape[1]

    def _grow_kv(self, kv, B, Smax, new_smax):
        s = self.shape
        new, new_smax = self.engine.new_kv(B, new_smax)
        half_old, half_new = kv.numel() // 2, new.numel() // 2
        k_old = kv[:half_old].view(s.layers, B, s.heads, Smax, s.head_dim)
        # V^T is blocked by 8 positions ([.., Smax / 8, head_dim, 8], csrc/kernels.h rv_vt_index): a longer cache has more blocks behind the old ones
        v_old = kv[half_old:].view(s.layers, B, s.heads, Smax // 8, s.head_dim, 8)
        new[:half_new].view(s.layers, B, s.heads, new_smax, s.head_dim)[:, :, :, :Smax] = k_old
        new[half_new:].view(s.layers, B, s.heads, new_smax // 8, s.head_dim, 8)[:, :, :, :Smax // 8] = v_old
        return new, new_smax


#: the reference's class name, so ``from revisionllm.model import VTimeLLMLlamaForCausalLM`` keeps working
VTimeLLMLlamaForCausalLM = ReVisionLlamaForCausalLM
