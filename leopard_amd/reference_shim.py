"""Run the reference's evaluation scripts UNMODIFIED on the HIP engine.

The scripts (evaluations/models/llava_multiimg_siglip_anyres.py, "EVAL"; evaluations/models/idefics2_multiimg.py, "IDEF") import
their model classes from ``transformers``:

    EVAL:6-9     from transformers import CLIPImageProcessor, LlavaForConditionalGeneration, SiglipImageProcessor, AutoTokenizer, LlavaConfig
    EVAL:195-198 class myLlavaForConditionalGeneration(LlavaForConditionalGeneration): __init__ -> super().__init__(config)
    EVAL:373-376 myLlavaForConditionalGeneration.from_pretrained(ckpt, torch_dtype=torch.float32); .eval(); .to('cuda:0')
    EVAL:448-452 llava.generate(input_ids, pixel_values=, attention_mask=, pad_token_id=, eos_token_id=[...], max_new_tokens=128, use_cache=True)
    IDEF:6,23-29 from transformers import AutoProcessor, AutoModelForVision2Seq; .from_pretrained(...).to(device); .generate(**inputs, max_new_tokens=128)

``install()`` rebinds exactly those model names inside the REAL ``transformers`` package (tokenizers and image processors stay
the third-party ones) to engine-backed classes with the same construction / call surface:

    transformers.LlavaForConditionalGeneration  -> LlavaForConditionalGeneration below (subclassable: the script's subclass and its
                                                   projector attribute are built, its ``forward`` is never needed: ``generate`` drives the
                                                   HIP engine directly, the whole forward of EVAL:201-361 happens in libleopard_amd.so)
    transformers.AutoModelForVision2Seq         -> leopard_amd.idefics2_compat.Idefics2ForConditionalGeneration
    transformers.AutoProcessor                  -> leopard_amd.idefics2_compat.Idefics2Processor

Two ways to use it, neither edits ``evaluations/``:

    python -m leopard_amd.run_reference_eval evaluations/models/llava_multiimg_siglip_anyres.py -- -c CKPT -d slidevqa -s direct
    PYTHONPATH=<repo>/leopard_amd/hf_shim:<repo> bash run_eval_llava_siglip_multiimg.sh direct CKPT      # the reference's own launcher

(the second form works because ``hf_shim/sitecustomize.py`` calls ``install()`` at interpreter start).  ``rouge`` and
``editdistance`` (scorer dependencies, EVAL:13, eval_utils.py) get minimal fallbacks only when they are not installed.

Precision: EVAL:373 asks for ``torch_dtype=torch.float32``; the engine computes in a 16-bit MFMA operand type with fp32
accumulation and fp32 residual streams (``LEOPARD_AMD_COMPUTE_DTYPE`` = f16 (default) | bf16).  That narrowing is NOT silent: every
``from_pretrained`` emits a ``UserWarning`` when the requested and the compute type differ and records both in
``leopard_amd_run_info.json`` in the checkpoint directory — next to the shard files ``*_shard_details.jsonl`` the script writes there
(EVAL:496-497; ``LEOPARD_AMD_RUN_INFO_DIR`` overrides) — so that a result row can always be traced to the arithmetic that produced it.
A float32 request is served by the ``lo4`` schedule (fp16 operands + the fp4 correction phase: full-depth logits within 1e-3 of fp32, ~1.2x the
fast schedule's prefill time); ``LEOPARD_AMD_PRECISION`` = fast | lo4 | split overrides (split: hi + lo 16-bit operand pairs, ~1.9x).

Smoke-run knobs (GPU-less containers / CI only; the product path needs none of them):
    LEOPARD_AMD_LIB             alternative C-ABI library (the CPU kernel-logic emulator build); implies host tensors
    LEOPARD_AMD_FORCE_DEVICE    device the model is built on, whatever the script asks for ("cpu" with the emulator)
    LEOPARD_AMD_MAX_NEW_TOKENS  cap on max_new_tokens
"""
from __future__ import annotations

import os
import sys
import types

_INSTALLED = False


def _ops_from_env():
    lib = os.environ.get("LEOPARD_AMD_LIB")
    if not lib:
        return None                                   # product library, loaded lazily by Ops()
    from . import _lib
    from .ops import Ops
    return Ops(lib=_lib.bind(lib), emulated=True)


def _device(requested):
    return os.environ.get("LEOPARD_AMD_FORCE_DEVICE") or requested


def _compute_dtype(default):
    import torch
    name = os.environ.get("LEOPARD_AMD_COMPUTE_DTYPE", "").lower()
    if not name:
        return default
    if name not in ("f16", "fp16", "float16", "bf16", "bfloat16"):
        raise ValueError("LEOPARD_AMD_COMPUTE_DTYPE must be f16 or bf16")
    return torch.bfloat16 if name.startswith("b") else torch.float16


_PRECISION_TEXT = {"fast": "fast (one rounding per MFMA-operand hand-over)",
                   "lo4": "lo4 (16-bit operands + the fp4 image of every layer-linear operand's rounding residual, same accumulators)",
                   "split": "split operands (hi + lo 16-bit pairs, GEMMs at 2 K)"}


def _record_run_info(model_class: str, checkpoint: str, requested, compute, precision: str = "fast", warn: bool = True) -> None:
    """Side file of the result rows: what arithmetic the script's ``torch_dtype`` request was actually served with.  Written where the
    script writes its result shards — the checkpoint directory (EVAL:496-497) — or ``LEOPARD_AMD_RUN_INFO_DIR``; a location that cannot be
    written falls back to the working directory, then to the warning alone."""
    import json
    import warnings
    req, cmp_ = str(requested).replace("torch.", ""), str(compute).replace("torch.", "")
    if warn and requested is not None and requested != compute:
        warnings.warn(f"{model_class}.from_pretrained: torch_dtype={req} was requested; leopard_amd computes with {cmp_} MFMA operands "
                      f"(fp32 accumulation, fp32 residual streams), precision mode {precision} — recorded in leopard_amd_run_info.json",
                      UserWarning, stacklevel=3)
    info = {"model_class": model_class, "checkpoint": str(checkpoint), "requested_torch_dtype": req, "compute_dtype": cmp_,
            "accumulate_dtype": "float32", "residual_stream_dtype": "float32",
            "precision_mode": _PRECISION_TEXT.get(precision, precision), "library": os.environ.get("LEOPARD_AMD_LIB") or "libleopard_amd.so",
            "fallback_scorers": sorted(_FALLBACK_SCORERS)}
    for d in (os.environ.get("LEOPARD_AMD_RUN_INFO_DIR"), str(checkpoint) if os.path.isdir(str(checkpoint)) else None, "."):
        if not d:
            continue
        try:
            with open(os.path.join(d, "leopard_amd_run_info.json"), "w") as f:
                json.dump(info, f, indent=1)
            return
        except OSError:                               # read-only location: try the next one; the warning above went out either way
            continue


_FALLBACK_SCORERS = set()


def _warn_fallback_scorer(name: str) -> None:
    import warnings
    if name not in _FALLBACK_SCORERS:
        _FALLBACK_SCORERS.add(name)
        warnings.warn(f"the '{name}' package is not installed: leopard_amd's minimal stand-in is in use.  Scores computed with it are for "
                      "smoke runs only and are NOT comparable with the reference's (install the real package for reportable numbers)",
                      UserWarning, stacklevel=3)


def _cap_tokens(n):
    cap = os.environ.get("LEOPARD_AMD_MAX_NEW_TOKENS")
    return min(int(n), int(cap)) if cap else int(n)


def _llava_class():
    import torch
    from .checkpoint import CheckpointSource, load_config
    from .compat import LeopardForConditionalGeneration

    class LlavaForConditionalGeneration(LeopardForConditionalGeneration):
        """Stands where transformers.LlavaForConditionalGeneration stands in EVAL: ``cls(config)`` construction (so that the
        script's subclass can add its projector attribute), ``from_pretrained(path, torch_dtype=)``, ``eval()``, ``to(device)``,
        ``device``, ``generate(...)``, ``forward(...)`` / ``__call__`` with the reference's argument names."""
        _pending = None

        def __init__(self, config):
            path, compute_dtype, ops, req, precision = type(self)._pending or (None, torch.float16, None, torch.float32, None)
            if path is None:
                raise RuntimeError("construct through from_pretrained(checkpoint_dir) — the engine streams the checkpoint's tensors")
            super().__init__(config, lambda dev, dt: CheckpointSource(path, dev, dt), compute_dtype, ops, torch_dtype=req, precision=precision)

        @classmethod
        def from_pretrained(cls, path, torch_dtype=torch.float32, compute_dtype=None, ops=None, precision=None, **unused):
            from .compat import resolve_precision
            cfg = load_config(path)
            if compute_dtype is None:                     # a 16-bit request is honoured as is; fp32 (EVAL:373) is served in 16 bits, loudly
                compute_dtype = _compute_dtype(torch_dtype if torch_dtype in (torch.float16, torch.bfloat16) else torch.float16)
            recorded = resolve_precision(torch_dtype, compute_dtype, precision)
            _record_run_info("LlavaForConditionalGeneration", path, torch_dtype, compute_dtype, recorded)
            cls._pending = (path, compute_dtype, ops if ops is not None else _ops_from_env(), torch_dtype, precision)
            try:
                m = cls(cfg)
            finally:
                cls._pending = None
            m._run_info = (path, torch_dtype, compute_dtype, recorded)
            return m

        def to(self, device):
            m = super().to(_device(device))
            info = getattr(self, "_run_info", None)
            if info is not None and self.precision != info[3]:      # the engine fell back (a shape lo4 does not cover): the side file says what RAN
                _record_run_info("LlavaForConditionalGeneration", info[0], info[1], info[2], self.precision, warn=False)
                self._run_info = info[:3] + (self.precision,)
            return m

        def generate(self, *args, max_new_tokens=128, **kw):
            return super().generate(*args, max_new_tokens=_cap_tokens(max_new_tokens), **kw)

        # the reference's subclass overrides forward() with EVAL:201-361, which needs HF sub-modules; calls go to the engine
        def __call__(self, *args, **kw):
            return LeopardForConditionalGeneration.forward(self, *args, **kw)
    return LlavaForConditionalGeneration


def _idefics2_classes():
    from . import idefics2_compat as IC

    class AutoModelForVision2Seq(IC.Idefics2ForConditionalGeneration):
        @classmethod
        def from_pretrained(cls, path, **kw):
            import torch
            kw.setdefault("ops", _ops_from_env())
            req = kw.get("torch_dtype", torch.float16)
            m = super().from_pretrained(path, **kw)
            _record_run_info("AutoModelForVision2Seq", path, req, m.compute_dtype, getattr(m, "precision", "fast"))
            return m

        def to(self, device):
            return super().to(_device(device))

        def generate(self, *args, max_new_tokens=128, **kw):
            return super().generate(*args, max_new_tokens=_cap_tokens(max_new_tokens), **kw)
    return IC.Idefics2Processor, AutoModelForVision2Seq


def _scorer_fallbacks():
    """rouge / editdistance are imported at module level by the reference's scripts (EVAL:13, eval_utils.py).  Where they are
    not installed, stand in with small pure-Python equivalents of the two calls the scorers make."""
    try:
        import rouge  # noqa: F401
    except ImportError:
        m = types.ModuleType("rouge")

        class Rouge:
            def __init__(self, *args, **kwargs):
                pass

            @staticmethod
            def _lcs(a, b):
                prev = [0] * (len(b) + 1)
                for x in a:
                    cur = [0]
                    for j, y in enumerate(b):
                        cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
                    prev = cur
                return prev[-1]

            def get_scores(self, hyps, refs, avg=False):
                _warn_fallback_scorer("rouge")
                if isinstance(hyps, str):
                    hyps, refs = [hyps], [refs]
                out = []
                for h, r in zip(hyps, refs):
                    ht, rt = h.split(), r.split()
                    def f(overlap, nh, nr):
                        p, rc = (overlap / nh if nh else 0.0), (overlap / nr if nr else 0.0)
                        return {"r": rc, "p": p, "f": (2 * p * rc / (p + rc) if p + rc else 0.0)}
                    uni = len(set(ht) & set(rt))
                    hb, rb = set(zip(ht, ht[1:])), set(zip(rt, rt[1:]))          # unique bigrams, as the package counts n-grams
                    out.append({"rouge-1": f(uni, len(set(ht)), len(set(rt))), "rouge-2": f(len(hb & rb), len(hb), len(rb)),
                                "rouge-l": f(self._lcs(ht, rt), len(ht), len(rt))})
                if avg:
                    keys = out[0].keys() if out else []
                    return {k: {m_: sum(o[k][m_] for o in out) / len(out) for m_ in ("r", "p", "f")} for k in keys}
                return out
        m.Rouge = Rouge
        sys.modules["rouge"] = m
    try:
        import editdistance  # noqa: F401
    except ImportError:
        m = types.ModuleType("editdistance")

        def _eval(a, b):
            _warn_fallback_scorer("editdistance")             # (exact Levenshtein distance, as the package computes; warned all the same)
            prev = list(range(len(b) + 1))
            for i, x in enumerate(a, 1):
                cur = [i]
                for j, y in enumerate(b, 1):
                    cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
                prev = cur
            return prev[-1]
        m.eval = _eval
        m.distance = _eval
        sys.modules["editdistance"] = m


def install() -> None:
    """Idempotent.  Imports the real ``transformers`` and rebinds the three model-side names (see the module docstring)."""
    global _INSTALLED
    if _INSTALLED:
        return
    _INSTALLED = True
    _scorer_fallbacks()
    import transformers
    transformers.LlavaForConditionalGeneration = _llava_class()
    proc, model = _idefics2_classes()
    transformers.AutoProcessor = proc
    transformers.AutoModelForVision2Seq = model
