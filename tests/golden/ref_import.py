"""Import the upstream reference (read-only at /root/reference) for golden-vector generation.

Runs ONLY in the build container: /root/reference does not exist on the GPU box, and nothing
under tests/ that carries the ``gpu`` marker, ``smoke()`` or ``bench.py`` imports this module.
The reference pins transformers==4.41.2 / torch 1.13; the container has transformers 5.x and
torch 2.10, so four shims are needed:

  1. namespace stubs for ``revisionllm``, ``revisionllm.model`` ... so the eager ``__init__`` imports
     (which pull peft / chatglm) are skipped;
  2. ``transformers.generation.Sample*Output`` aliases removed upstream after 4.38;
  3. ``DynamicCache.__getitem__`` (the decode branch at model/vtimellm_arch.py:93 indexes the cache);
  4. empty stub modules for packages absent from the image (decord, easydict, clip, lmdb,
     torchvision, peft).

No reference source is copied; the modules are imported from where they lie.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("REVISION_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "revisionllm"))


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = None
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_done = False


def install():
    """Install the shims and return the dict of imported reference modules."""
    global _done
    import importlib
    import transformers  # must come first (is_torchvision_available probes sys.modules)
    import transformers.generation as tg
    import transformers.generation.utils as tgu
    from transformers.cache_utils import DynamicCache

    if not _done:
        if not available():
            raise RuntimeError(f"reference not found at {REF_ROOT}")
        # (2) aliases
        if not hasattr(tg, "SampleDecoderOnlyOutput"):
            tg.SampleDecoderOnlyOutput = tg.GenerateDecoderOnlyOutput
            tg.SampleEncoderDecoderOutput = tg.GenerateEncoderDecoderOutput
        if not hasattr(tg, "validate_stopping_criteria"):
            tg.validate_stopping_criteria = lambda crit, max_length: crit
        if not hasattr(tgu, "SampleOutput"):
            tgu.SampleOutput = object
        # (3) cache indexing
        if not hasattr(DynamicCache, "__getitem__"):
            DynamicCache.__getitem__ = lambda self, i: (self.layers[i].keys, self.layers[i].values)
        # (4) stubs for absent packages
        for name in ("decord", "easydict", "clip", "lmdb", "peft"):
            if name not in sys.modules:
                try:
                    importlib.import_module(name)
                except Exception:
                    _stub(name)
        sys.modules["decord"].gpu = getattr(sys.modules["decord"], "gpu", None)
        sys.modules["decord"].VideoReader = getattr(sys.modules["decord"], "VideoReader", object)
        sys.modules["easydict"].EasyDict = getattr(sys.modules["easydict"], "EasyDict", dict)
        sys.modules["peft"].PeftModel = getattr(sys.modules["peft"], "PeftModel", object)
        try:
            importlib.import_module("torchvision.transforms")
        except Exception:
            class _IM:
                BICUBIC = 3
            tv = _stub("torchvision")
            tvt = _stub("torchvision.transforms", InterpolationMode=_IM, Compose=object, Resize=object,
                        CenterCrop=object, Normalize=object)
            tv.transforms = tvt
        # (1) namespace stubs
        base = os.path.join(REF_ROOT, "revisionllm")
        _ns("revisionllm", base)
        _ns("revisionllm.model", os.path.join(base, "model"))
        _ns("revisionllm.model.adapter", os.path.join(base, "model", "adapter"))
        _ns("revisionllm.eval", os.path.join(base, "eval"))
        _ns("revisionllm.uncertainty", os.path.join(base, "uncertainty"))
        _done = True

    mods = {}
    for short, name in [
        ("constants", "revisionllm.constants"),
        ("conversation", "revisionllm.conversation"),
        ("transformer", "revisionllm.model.adapter.transformer"),
        ("tensor_utils", "revisionllm.model.adapter.tensor_utils"),
        ("arch", "revisionllm.model.vtimellm_arch"),
        ("llama", "revisionllm.model.vtimellm_llama"),
        ("entropy", "revisionllm.uncertainty.funs_get_feature_X"),
        ("similarity", "revisionllm.eval.similarity"),
    ]:
        mods[short] = importlib.import_module(name)
    # builder.py star-imports revisionllm.model
    sys.modules["revisionllm.model"].VTimeLLMLlamaForCausalLM = mods["llama"].VTimeLLMLlamaForCausalLM
    sys.modules["revisionllm.model"].__all__ = ["VTimeLLMLlamaForCausalLM"]
    for short, name in [
        ("mm_utils", "revisionllm.mm_utils"),
        ("inference", "revisionllm.inference"),
        ("e2e2", "revisionllm.eval.eval_nlq_retrieval_e2e2"),
        ("metric", "revisionllm.eval.metric_retrieval_forward"),
        ("builder", "revisionllm.model.builder"),
    ]:
        try:
            mods[short] = importlib.import_module(name)
        except Exception as e:  # pragma: no cover - reported by make_goldens
            mods[short] = e
    # eval_nlq_negative imports ``vtimellm.*``: alias every revisionllm module under that name
    for k in list(sys.modules):
        if k == "revisionllm" or k.startswith("revisionllm."):
            sys.modules.setdefault("vtimellm" + k[len("revisionllm"):], sys.modules[k])
    try:
        mods["negative"] = importlib.import_module("revisionllm.eval.eval_nlq_negative")
    except Exception as e:  # pragma: no cover
        mods["negative"] = e
    return mods
