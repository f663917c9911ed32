"""ctypes loader of libleopard_amd.so (the C ABI declared in include/leopard_amd.h).

The product path has exactly one backend: the HIP library built for gfx950.  If it is missing this module
raises — there is no CPU or PyTorch fallback (build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LEOPARD_AMD_LIB: another build of the same library (A/B of a compile-time kernel variant on one box); default = the in-tree build
LIB_PATH = os.environ.get("LEOPARD_AMD_LIB") or os.path.join(_HERE, "libleopard_amd.so")

LMI_F16, LMI_BF16, LMI_F32, LMI_FP8 = 0, 1, 2, 3
EPI_STORE, EPI_RESIDUAL, EPI_STORE_F32, EPI_SWIGLU, EPI_QKV_ROPE, EPI_SWIGLU_F32 = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SWIGLU = 0, 1, 2, 3      # ACT_SWIGLU: lmi_gemm_bias_act only
A_PLAIN, A_PIXEL_SHUFFLE = 0, 1

_P, _I, _F = C.c_void_p, C.c_int, C.c_float


class Lo4Desc(C.Structure):
    """``lmi_lo4`` of include/leopard_amd.h: the fp4 images a GEMM with the low-bit correction phase consumes / produces."""
    _fields_ = [("a4", _P), ("a4_scale", _P), ("w4", _P), ("w4_scale", _P), ("lda4", _I), ("ldw4", _I), ("lds4", _I), ("k4", _I),
                ("out4", _P), ("out4_scale", _P), ("ld_out4", _I), ("ld_out4s", _I), ("row_sel", _P), ("unit_sel", _P),
                ("sel_ranges", _P), ("n_sel_ranges", _I)]


# name -> argtypes  (restype is int unless noted); mirrors include/leopard_amd.h one to one
SIGNATURES = {
    "lmi_abi_version": [],
    "lmi_set_option": [C.c_char_p, _I],
    "lmi_fill_synthetic": [_P, C.c_int64, C.c_uint32, _I, _I, _P],
    "lmi_preprocess_tiles": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "lmi_preprocess_images": [_P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "lmi_resample_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "lmi_layernorm": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "lmi_rmsnorm": [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "lmi_add_rmsnorm": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "lmi_gemm": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_gemm_ex": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P, _P, _P, _I, _I, _P],
    "lmi_rmsnorm_rope": [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_quantize_fp8": [_P, _I, _P, _I, _I, _I, _I, _F, _P],
    "lmi_gemm_fp8": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "lmi_norm_fp8": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P],
    "lmi_gemm_bias_act": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_patch_embed": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_kv_append": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lmi_attn_varlen_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P],
    "lmi_rope_qk": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P],
    "lmi_rope_qk_at": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P],
    "lmi_attn_decode_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, C.c_int64, _I, _P],
    "lmi_embed_merge": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lmi_gemv": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lmi_gemv_rmsnorm": [_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P],
    "lmi_gemv_rmsnorm_rope": [_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P],
    "lmi_attn_decode_pool": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, C.c_int64, _I, _P],
    "lmi_rope_qk_rows": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, C.c_int64, _P, _I, _P],
    "lmi_attn_prep_fp8": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P, _I, _P, _P, _I, _P],
    "lmi_attn_fp8_fwd": [_P, _I, _P, _P, _P, _I, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _I, _I, _P],
    "lmi_attn_varlen_fwd_fp8": [_P, _P, _P, _P, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "lmi_rope_qkv_fp8": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_split_hi_lo": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lmi_gemm_lo4": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P, _P, _P, _I, C.POINTER(Lo4Desc), _I, _P],
    "lmi_rmsnorm_rope_lo4": [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(Lo4Desc), _I, _P],
    "lmi_attn_varlen_fwd_lo4": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "lmi_attn_varlen_fwd_lo4_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P, _I, _P],
    "lmi_norm_lo4_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _P],
    "lmi_split_lo4": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_norm_lo4": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "lmi_add_rmsnorm_lo4": [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "lmi_quantize_w4": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lmi_attn_varlen_fwd_f32": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P],
    "lmi_rope_qkv_skinny": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _F, _P, _P, _P, _P, _I, C.c_int64, _P, _I, _P],
    "lmi_gemm_skinny_ex": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P, _I, _P, _P, _I, _P],
    "lmi_gemm_skinny_hl": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P, _I, _P, _P, _I, _P],
    "lmi_rope_qkv_skinny_hl": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _F, _P, _P, _P, _P, _I, C.c_int64, _P, _I, _P],
    "lmi_split_rows_hl": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lmi_attn_decode_fwd_hl": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, C.c_int64, _I, _P],
    "lmi_attn_decode_pool_hl": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, C.c_int64, _I, _P],
    "lmi_debug_copy": [_P, _P, C.c_int64, _I, _P],
    "lmi_gemm_skinny": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_decode_advance": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "lmi_lm_head_last": [_P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "lmi_comm_unique_id": [_P],
    "lmi_comm_init": [_I, _I, _P, C.POINTER(C.c_void_p)],
    "lmi_comm_destroy": [_P],
    "lmi_comm_size": [_P],
    "lmi_allgather": [_P, _P, _P, C.c_int64, _I, _P],
    "lmi_allreduce": [_P, _P, _P, C.c_int64, _I, _P],
    "lmi_reduce_scatter": [_P, _P, _P, C.c_int64, _I, _P],
    "lmi_broadcast": [_P, _P, _P, C.c_int64, _I, _I, _P],
}


def bind(path: str) -> C.CDLL:
    """dlopen ``path`` and attach the prototypes of every symbol the header declares."""
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = the library does not export the ABI
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.lmi_attn_decode_workspace_bytes.argtypes = [_I, _I, _I, _I]
    lib.lmi_attn_decode_workspace_bytes.restype = C.c_int64
    lib.lmi_llm_prefill_workspace_bytes.argtypes = [C.c_int64, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_int64)]
    lib.lmi_llm_prefill_workspace_bytes.restype = C.c_int64
    lib.lmi_vit_workspace_bytes.argtypes = [C.c_int64, _I, _I, _I, _I, C.POINTER(C.c_int64)]
    lib.lmi_vit_workspace_bytes.restype = C.c_int64
    lib.lmi_last_error.argtypes = []
    lib.lmi_last_error.restype = C.c_char_p
    return lib


_LIB = None


def load() -> C.CDLL:
    """The product library.  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built (run `make` or "
                "`__graft_entry__.build()`); leopard_amd has no CPU fallback")
        # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 and the library is linked against the system one; whichever is
        # loaded first serves both (same soname).  Loaded in the order "library, then torch" the process ends up with a runtime torch was not
        # built for and the first kernel launch reports "no ROCm-capable device" (seen with build() followed by smoke() in ONE process on a GPU
        # box, round 6).  Device memory, streams and torch.distributed come from torch anyway: import it first, always.
        import torch  # noqa: F401
        _LIB = bind(LIB_PATH)
    return _LIB
