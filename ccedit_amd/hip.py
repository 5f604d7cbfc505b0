"""ctypes binding of libccedit_hip.so (the C ABI declared in include/ccedit_hip.h).

There is NO fallback: if the shared library is missing or a kernel reports an error, the product path
raises.  Nothing here touches the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  -- must be imported BEFORE the library is loaded: torch bundles its own libamdhip64 and
#                              both sides have to share ONE HIP runtime (device state, streams, allocations).

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCEDIT_HIP_LIB") or os.path.join(_HERE, "libccedit_hip.so")

ABI_VERSION = 12
ATTN_Q_LOG2 = 1          # CcAttnDesc.flags: CCEDIT_ATTN_Q_LOG2

GEMM_LINEAR, GEMM_CONV2D, GEMM_TEMPORAL = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_QUICK_GELU = 0, 1, 2, 3


class CcGemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int32), ("Cin", C.c_int32), ("Cin1", C.c_int32), ("taps", C.c_int32),
        ("mode", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("ksize", C.c_int32), ("upsample", C.c_int32),
        ("T", C.c_int32), ("HW", C.c_int32), ("lda", C.c_int32), ("lda2", C.c_int32), ("ldc", C.c_int32),
        ("Kpad", C.c_int32), ("act", C.c_int32), ("out_f32", C.c_int32), ("group_rows", C.c_int32),
        ("ldr1", C.c_int32), ("ldr2", C.c_int32), ("tile", C.c_int32), ("korder", C.c_int32), ("gn_rows", C.c_int32),
        ("Tsrc", C.c_int32), ("tsrc_off", C.c_int32), ("t0", C.c_int32), ("Tglob", C.c_int32), ("cgroup", C.c_int32),
        ("ldgb", C.c_int32), ("ln_eps", C.c_float),
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
        ("group_bias", C.c_void_p), ("res1", C.c_void_p), ("res2", C.c_void_p), ("out", C.c_void_p),
        ("gn_stats", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("split_k", C.c_int32), ("subpix", C.c_int32), ("ln_colsum", C.c_void_p), ("ln_stats", C.c_void_p),
        ("ln_sums", C.c_void_p), ("row_sums", C.c_void_p), ("ln_sums_eps", C.c_float), ("vpad", C.c_int32),
        ("halo_top", C.c_void_p), ("halo_bot", C.c_void_p), ("Wfrag", C.c_void_p),
    ]


class CcAttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
        ("heads", C.c_int32), ("d", C.c_int32), ("batches", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
        ("q_inner", C.c_int32), ("q_outer_rows", C.c_int64), ("q_inner_rows", C.c_int64), ("q_seq_rows", C.c_int64),
        ("kv_div", C.c_int32), ("kv_inner", C.c_int32), ("kv_outer_rows", C.c_int64), ("kv_inner_rows", C.c_int64),
        ("kv_seq_rows", C.c_int64), ("scale", C.c_float),
        ("seg1_len", C.c_int32), ("seg1_div", C.c_int32), ("seg1_mul", C.c_int32), ("seg1_add", C.c_int32),
        ("causal", C.c_int32), ("flags", C.c_int32),
    ]


class CcFf320Desc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("dim", C.c_int32), ("inner", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32),
        ("eps", C.c_float), ("ln", C.c_int32),
        ("x", C.c_void_p), ("out", C.c_void_p), ("wstream", C.c_void_p), ("b2p", C.c_void_p), ("dbg", C.c_void_p),
        ("a", C.c_void_p), ("res", C.c_void_p), ("res2", C.c_void_p), ("bop", C.c_void_p), ("bpp", C.c_void_p),
        ("lda", C.c_int32), ("ldr", C.c_int32), ("ldr2", C.c_int32), ("pad_", C.c_int32),
    ]


class CcGemmF32Desc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int64), ("N", C.c_int32), ("Cin", C.c_int32), ("Cpad", C.c_int32), ("Kpad", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("mode", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32), ("stride", C.c_int32),
        ("pad", C.c_int32), ("upsample", C.c_int32),
    ]


class HipLibraryError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

_SIGS = {
    "ccedit_abi_version": (C.c_int, []),
    "ccedit_last_error": (C.c_char_p, []),
    "ccedit_last_kernel": (C.c_char_p, []),
    "ccedit_device_info": (C.c_int, [C.c_char_p, C.c_int]),
    "ccedit_policy_set": (C.c_int, [C.c_char_p, C.c_int32]),
    "ccedit_policy_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "ccedit_policy_names": (C.c_char_p, []),
    "ccedit_gemm": (C.c_int, [C.POINTER(CcGemmDesc), C.c_void_p]),
    "ccedit_gemm_workspace_bytes": (C.c_int64, [C.POINTER(CcGemmDesc)]),
    "ccedit_ff320": (C.c_int, [C.POINTER(CcFf320Desc), C.c_void_p]),
    "ccedit_groupnorm_spatial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ccedit_groupnorm_spatial_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ccedit_groupnorm_spatial_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ccedit_groupnorm_temporal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ccedit_groupnorm_temporal_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ccedit_groupnorm_temporal_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                                  C.c_int32, C.c_void_p]),
    "ccedit_row_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "ccedit_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "ccedit_attention": (C.c_int, [C.POINTER(CcAttnDesc), C.c_void_p]),
    "ccedit_ncthw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "ccedit_nhwc_to_ncthw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    "ccedit_copy_row_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "ccedit_copy_2d_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "ccedit_cat_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                 C.c_void_p]),
    "ccedit_cat_add_gn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_void_p]),
    "ccedit_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ccedit_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ccedit_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ccedit_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                      C.c_float, C.c_void_p]),
    "ccedit_gemm_f32": (C.c_int, [C.POINTER(CcGemmF32Desc), C.c_void_p]),
    "ccedit_groupnorm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_float, C.c_int32, C.c_void_p]),
    "ccedit_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "ccedit_embedding_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_void_p]),
    "ccedit_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_float, C.c_void_p]),
    "ccedit_mask_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ccedit_cfg_denoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "ccedit_axpby": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Load (once) and return the kernel library.  Raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing — build it with `python ccedit_amd/csrc/build.py` (or __graft_entry__.build()). "
            "ccedit_amd has no CPU fallback.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = l.ccedit_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = l
    from . import policy
    policy.push_to_library(l)          # the library never reads the environment: the one policy table is pushed here
    return l


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().ccedit_last_error()
        raise HipLibraryError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
