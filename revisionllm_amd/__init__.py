"""revisionllm_amd - MI355X-native implementation of ReVisionLLM's recursive temporal-grounding inference path.

Keeps the reference's Python surface (``inference``, ``mm_utils``, ``model.builder``, ``constants``,
``conversation``); all device arithmetic is hand-written HIP behind the C ABI in include/revision_hip.h.
``install_as_revisionllm()`` aliases this package as ``revisionllm`` so existing eval scripts import it unchanged.
"""
import sys

__version__ = "0.1.0"


def install_as_revisionllm():
    """Make ``import revisionllm...`` resolve to this package (drop-in for the reference's eval scripts)."""
    import importlib
    pkg = sys.modules[__name__]
    sys.modules["revisionllm"] = pkg
    for sub in ("constants", "conversation", "mm_utils", "inference", "utils", "model", "model.builder"):
        sys.modules["revisionllm." + sub] = importlib.import_module(__name__ + "." + sub)
    return pkg
