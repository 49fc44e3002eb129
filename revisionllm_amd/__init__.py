"""revisionllm_amd - MI355X-native implementation of ReVisionLLM's recursive temporal-grounding inference path.

Keeps the reference's Python surface (``inference``, ``mm_utils``, ``model.builder``, ``constants``,
``conversation``, ``eval.similarity``, ``uncertainty.funs_get_feature_X``); all device arithmetic is hand-written HIP
behind the C ABI in include/revision_hip.h.  ``install_as_revisionllm()`` aliases the WHOLE package as ``revisionllm``
(and under the drivers' older name ``vtimellm``, eval_nlq_negative.py:17-22) so existing eval scripts import it unchanged.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

__version__ = "0.2.0"

#: names the reference's drivers import this package under (eval_nlq_retrieval_e2e2.py:18-23, eval_nlq_negative.py:17-22)
ALIASES = ("revisionllm", "vtimellm")


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """``import <alias>.x.y`` -> the module object of ``revisionllm_amd.x.y`` itself (one module, two names: no second copy
    of the ctypes handle or of any module-level state)."""

    def __init__(self, alias, target):
        self.alias, self.target = alias, target
        self._real_spec = {}

    def _real(self, fullname):
        return self.target + fullname[len(self.alias):]

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + "."):
            return None
        try:
            real = importlib.util.find_spec(self._real(fullname))
        except (ImportError, ValueError):
            return None
        if real is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, self, is_package=real.submodule_search_locations is not None)

    def create_module(self, spec):
        mod = importlib.import_module(self._real(spec.name))
        self._real_spec[spec.name] = mod.__spec__      # the machinery overwrites __spec__ with the alias spec: put it back below
        return mod

    def exec_module(self, module):
        for name, real in list(self._real_spec.items()):
            if real is not None and real.name == module.__name__:
                module.__spec__ = real
                del self._real_spec[name]


def install_as_revisionllm(aliases=ALIASES):
    """Make ``import revisionllm...`` (and ``import vtimellm...``) resolve to this package: every submodule, lazily, as the
    SAME module object (drop-in for the reference's eval scripts)."""
    pkg = sys.modules[__name__]
    for alias in aliases:
        if not any(isinstance(f, _AliasFinder) and f.alias == alias for f in sys.meta_path):
            sys.meta_path.insert(0, _AliasFinder(alias, __name__))
        for name, mod in list(sys.modules.items()):      # whatever is imported already
            if mod is not None and (name == __name__ or name.startswith(__name__ + ".")):
                sys.modules[alias + name[len(__name__):]] = mod
    return pkg
