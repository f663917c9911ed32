"""``PYTHONPATH=<repo>/leopard_amd/hf_shim:<repo> python evaluations/models/llava_multiimg_siglip_anyres.py ...``

Python imports ``sitecustomize`` from sys.path at interpreter start; this one binds the model names the reference's evaluation
scripts import from ``transformers`` to the HIP engine (leopard_amd.reference_shim).  The rebinding is deferred until
``transformers`` is actually imported, so interpreters that never touch it (eval_utils.py's scorers) pay nothing."""
import importlib.abc
import importlib.util
import sys


class _PatchTransformers(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name != "transformers":
            return None
        sys.meta_path.remove(self)
        spec = importlib.util.find_spec("transformers")
        if spec is None or spec.loader is None:
            return None
        inner = spec.loader

        class Loader(importlib.abc.Loader):
            def create_module(self, s):
                return inner.create_module(s)

            def exec_module(self, module):
                inner.exec_module(module)
                from leopard_amd import reference_shim
                reference_shim.install()
        spec.loader = Loader()
        return spec


def _chain_other_sitecustomize():
    """This directory is in front of sys.path, so this module shadows any ``sitecustomize`` the environment already had (venv /
    distribution hooks).  Run that one too: the first ``sitecustomize`` found on the REST of sys.path."""
    import importlib.machinery
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    rest = [p for p in sys.path if p and os.path.abspath(p) != here]
    spec = importlib.machinery.PathFinder.find_spec("sitecustomize", rest)
    if spec is None or spec.loader is None or (spec.origin and os.path.abspath(spec.origin) == os.path.abspath(__file__)):
        return
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as e:                           # noqa: BLE001  (same policy as site.py: report, do not abort the interpreter)
        sys.stderr.write(f"leopard_amd hf_shim: the shadowed sitecustomize ({spec.origin}) raised {e!r}\n")


try:
    from leopard_amd import reference_shim as _rs
    _rs._scorer_fallbacks()
    sys.meta_path.insert(0, _PatchTransformers())
except ImportError:                                  # leopard_amd not importable: leave the interpreter untouched
    pass
_chain_other_sitecustomize()
