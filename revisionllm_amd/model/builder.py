"""Model-builder API of the reference (revisionllm/model/builder.py:9-67), rebuilt for the HIP engine.

``load_pretrained_model(args, stage2, stage3)`` reads the HF checkpoint files of ``args.model_base`` directly
(config.json, *.safetensors / pytorch_model*.bin), applies ``non_lora_trainables.bin`` and merges the LoRA
deltas on the host (peft is not needed: W += (alpha/r) * B @ A, builder.py:53-60 via peft's merge_and_unload),
then packs everything into HBM.  After the merge the model is a plain dense Llama + adapter.
"""
import glob
import json
import os
import re

import torch

from ..utils import synth
from .revision_llama import ReVisionLlamaForCausalLM

_LORA_RE = re.compile(r"^(?:base_model\.model\.)?(.*)\.lora_([AB])(?:\.default)?\.weight$")


def strip_trainable_prefixes(sd):
    """Prefix rule of ``load_lora`` (builder.py:13-15): drop a leading 'base_model.', then, if any key starts with
    'model.model.', drop one leading 'model.'."""
    sd = {(k[11:] if k.startswith("base_model.") else k): v for k, v in sd.items()}
    if any(k.startswith("model.model.") for k in sd):
        sd = {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}
    return sd


def remap_projector_keys(weights, clip):
    """Key remap of ``initialize_vision_modules`` (vtimellm_arch.py:30-37 ``get_wc`` / :46-47 ``get_w``):
    returns adapter-relative names.  For the ClipEncoder a key containing 'mm_projector.mm_projector' is the
    inner Linear and keeps one 'mm_projector.' prefix."""
    kw = "mm_projector"
    out = {}
    for k, v in weights.items():
        if kw not in k:
            if clip:    # get_wc's else-branch indexes ``k.split('mm_projector.')[2]`` (vtimellm_arch.py:36): IndexError in the reference (golden G13)
                raise IndexError(f"list index out of range (adapter checkpoint key '{k}' does not contain '{kw}.')")
            continue
        if not clip or f"{kw}.{kw}" not in k:
            out[k.split(kw + ".")[1]] = v
        else:
            out[kw + "." + k.split(kw + ".")[2]] = v
    return out


def merge_lora(base, lora_sd, alpha, r):
    """In-place ``W += (alpha / r) * B @ A`` for every LoRA pair in ``lora_sd`` (peft key naming)."""
    pairs = {}
    for k, v in lora_sd.items():
        m = _LORA_RE.match(k)
        if m:
            pairs.setdefault(m.group(1), {})[m.group(2)] = v
    scale = float(alpha) / float(r)
    for mod, ab in pairs.items():
        name = mod + ".weight"
        if name not in base:
            raise KeyError(f"LoRA target {name} not in the base model")
        w = base[name]
        base[name] = (w.float() + scale * (ab["B"].float() @ ab["A"].float())).to(w.dtype)
    return base


def _load_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def read_hf_checkpoint(model_base):
    sd = {}
    files = sorted(glob.glob(os.path.join(model_base, "*.safetensors"))) or sorted(glob.glob(os.path.join(model_base, "pytorch_model*.bin")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {model_base}")
    for f in files:
        sd.update(_load_file(f))
    return sd


def shape_from_config(cfg):
    return synth.LlamaShape(hidden=cfg["hidden_size"], inter=cfg["intermediate_size"], layers=cfg["num_hidden_layers"],
                            heads=cfg["num_attention_heads"], vocab=cfg["vocab_size"], eps=cfg.get("rms_norm_eps", 1e-5),
                            theta=cfg.get("rope_theta", 10000.0))


def apply_lora_dir(sd, lora_path):
    """``load_lora`` + ``merge_and_unload`` on a host state dict (builder.py:9-19,55)."""
    extra = {}
    nl = os.path.join(lora_path, "non_lora_trainables.bin")
    if os.path.exists(nl):
        extra = strip_trainable_prefixes(torch.load(nl, map_location="cpu"))
        for k, v in extra.items():
            sd[k] = v
    cfg_path = os.path.join(lora_path, "adapter_config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as f:
            lc = json.load(f)
        files = [p for p in (os.path.join(lora_path, "adapter_model.safetensors"), os.path.join(lora_path, "adapter_model.bin"))
                 if os.path.exists(p)]
        if files:
            merge_lora(sd, _load_file(files[0]), lc["lora_alpha"], lc["r"])
    return sd, extra


def load_lora(model, lora_path, is_trainable=False):
    """Kept for API compatibility: merges into ``model``'s pending host state dict (see load_pretrained_model)."""
    if getattr(model, "_host_sd", None) is None:
        raise RuntimeError("load_lora needs a model created by load_pretrained_model (host weights already released)")
    model._host_sd, _ = apply_lora_dir(model._host_sd, lora_path)
    return model


def load_pretrained_model(args, stage2=None, stage3=None, load_ckp=False):
    """-> (tokenizer, model, context_len) exactly like the reference (builder.py:21-67)."""
    model_base = args.model_base
    print("Loading VTimeLLM from base model...")
    if "chatglm" in model_base:
        raise NotImplementedError("the ChatGLM backbone is out of scope (broken at reference HEAD, SURVEY section 2 #20)")
    from transformers import AutoTokenizer
    tokenizer = AutoTokenizer.from_pretrained(model_base, use_fast=False)
    with open(os.path.join(model_base, "config.json")) as f:
        cfg = json.load(f)
    # build-defined: args.op_dtype = "f16" (default: the checkpoints' own storage type - builder.py:22 loads them with torch_dtype=float16 -
    # held exactly; segment scores within 1e-3 of the reference's fp32 CPU path) or "bf16" (the reference's GPU dtype, e2e2.py:181-185)
    model = ReVisionLlamaForCausalLM(shape_from_config(cfg), max_sequence_length=cfg.get("max_sequence_length"), op_dtype=getattr(args, "op_dtype", None))
    print(f"[revisionllm_amd] operand type: {str(model.dtype).replace('torch.', '')} (args.op_dtype = {getattr(args, 'op_dtype', None)!r}; f32 accumulation / residual stream)", flush=True)
    model.fp8_prefill = bool(getattr(args, "fp8_prefill", False))  # build-defined opt-in: FP8 x FP8 prefill GEMMs
    model.fp8_decode = bool(getattr(args, "fp8_decode", False))   # build-defined opt-in: FP8 copies of the LLM weights for decode steps
    model.parity = bool(getattr(args, "parity", False))           # build-defined opt-in: K-duplicated copies for the parity precision (engine option precision = 1)
    gpath = os.path.join(model_base, "generation_config.json")
    if os.path.exists(gpath):
        with open(gpath) as f:
            for k, v in json.load(f).items():
                if k in ("top_k", "top_p", "temperature", "eos_token_id", "pad_token_id"):
                    setattr(model.generation_config, k, v)
    sd = read_hf_checkpoint(model_base)
    # stage 1: adapter topology + pretrain_* files (vtimellm_arch.py:12-73)
    model.get_model().initialize_vision_modules(args)
    model._host_sd = sd
    for tag, path in (("stage2", stage2), ("stage3", stage3 if stage2 is not None else None)):
        if path is not None:
            print(f"Loading {tag} weights...")
            load_lora(model, path)
            print(f"Merging {tag} weights...")
    finalize(model)
    context_len = getattr(model.config, "max_sequence_length", 2048)
    return tokenizer, model, context_len


def finalize(model):
    """Pack the (merged) host state dict into HBM and drop the host copy."""
    sd = model._host_sd
    eng = model._ensure_engine()
    eng.load_llm(lambda n: sd[n], fp8_decode=getattr(model, "fp8_decode", False), fp8_prefill=getattr(model, "fp8_prefill", False), parity=getattr(model, "parity", False))
    if getattr(model, "parity", False):
        eng.set_option("precision", 1)
    inner = model.get_model()
    if getattr(inner, "cross_attn_dense", False):
        # cross_attn=True without pretrain_clip_adapter: BOTH modules carry weights - the Linear 'model.mm_projector.*' in front and the
        # hidden-wide ClipEncoder 'model.cross_attn.*' (vtimellm_arch.py:42,52-57)
        lin = {k[len("model.mm_projector."):]: v for k, v in sd.items() if k.startswith("model.mm_projector.")}
        ca = {k[len("model.cross_attn."):]: v for k, v in sd.items() if k.startswith("model.cross_attn.")}
        if lin:
            eng.load_linear_projector(lambda n: lin[n])
        if ca:
            eng.load_clip_adapter(lambda n: ca[n])
        model._host_sd = None
        return model
    root = "model.cross_attn." if getattr(inner, "cross_attn_variant", False) else "model.mm_projector."
    proj = {k[len(root):]: v for k, v in sd.items() if k.startswith(root)}
    if proj:
        if model.get_model().clip_adapter:
            eng.load_clip_adapter(lambda n: proj[n])
        else:
            eng.load_linear_projector(lambda n: proj[n])
    model._host_sd = None
    return model
