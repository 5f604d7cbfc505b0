"""The dispatch policy: ONE table of named switches for the whole product (host side and kernel library).

Every entry selects between two realisations of the same fp32 computation (a specialised kernel against the generic one, a fused
launch against separate launches, graph replay against eager launches): results differ by summation order only.  All default to
the fast path.  The table is read ONCE per process from

    CCEDIT_POLICY="name=value,name=value"        e.g.  CCEDIT_POLICY="ff320=0,g8=0,graph=0"

(the pre-round-5 spelling — one variable per switch, CCEDIT_<NAME>=v — is still honoured for the names below).  Library-side entries
(`LIB`) are pushed through ccedit_policy_set when the library is loaded; the library itself never reads the environment
(csrc/common.h: CcPolicy).  Tuning variants of one kernel that were A/B remnants of rounds 1-2 (block shapes T4 / T6, K rotation,
channel-tile groups, tile orders, narrow channel tile) are gone: the block shapes remain selectable through CcGemmDesc.tile.
"""
from __future__ import annotations

import os
from typing import Dict

# name -> (default, side, what the non-default value selects)
TABLE = {
    # ---- host side (ccedit_amd/ops.py, network.py) ----
    "graph": (1, "py", "0: eager launches instead of HIP-graph replay of a network evaluation"),
    "overlap_controlnet": (1, "py", "0: ControlNet on the main stream instead of a side stream"),
    "split_cfg": (0, "py", "1: the two CFG halves as B = 1 passes on two streams (round-2 execution)"),
    "hint_dedup": (1, "py", "0: hint stem evaluated for both CFG halves"),
    "text_kv_batched": (1, "py", "0: one text K/V projection launch per transformer block"),
    "share_cfg_prefix": (1, "py", "0: identical CFG halves evaluated in full (no shared prefix up to the first text attention)"),
    "fuse_gn_stats": (1, "py", "0: two-pass GroupNorm everywhere (no statistics from the producers' epilogues)"),
    "ff320": (1, "py", "0: LayerNorm + two GEMMs instead of the fused dim-320 feed-forward"),
    "ln320": (1, "py", "0: separate LayerNorm pass in front of the K = 320 projections"),
    "lnf": (1, "py", "0: LayerNorm passes in front of the 640 / 1280-channel projections"),
    "ln_sums": (1, "py", "0: LayerNorm statistics by ccedit_row_stats instead of the producer's epilogue"),
    "subpix": (1, "py", "0: upsample + 3x3 conv through the nine-tap gather instead of four parity convs"),
    "conv1x1_linear": (1, "py", "0: 1 x 1 convs through the convolution mode instead of the Linear dispatch"),
    "attn_q_log2": (1, "py", "0: softmax scale applied inside the attention kernels instead of folded into to_q"),
    "fuse_halo_stats": (1, "py", "0 (rows sharded): GroupNorm statistics all-reduced on their own instead of riding on the 3x3 conv's halo exchange"),
    "wfrag": (1, "py", "0: the register-resident-weight kernels (K = 320 / 640 Linears, 320-channel Conv1d k3) preload their weights from the row-major matrix instead of its fragment-ordered copy"),
    "block_tail": (1, "py", "0: to_out / proj_out of the dim-320 transformer tails as their own launches, not inside ff320"),
    # (a precision option, not an A/B arm of equal arithmetic: the reference runs its first-stage model with autocast disabled)
    "vae_fp32": (2, "py", "the KL-VAE's precision.  2 (default): the engine follows the yaml's disable_first_stage_autocast, the flag's meaning in "
                          "the reference (diffusion.py:151-156; the shipped yamls set it => fp32 on v_mfma_f32_32x32x2_f32, ccedit_amd/vae_f32.py); "
                          "1: always fp32, also for a first stage built outside an engine; 0: always the bf16-storage kernels"),
    # ---- kernel library (csrc/common.h: CcPolicy) ----
    "conv_halo": (1, "lib", "0: 3x3 stride-1 convs on the tap-gather kernel; 2: the LDS-halo kernel with the two-slot weight ring of rounds 2-5"),
    "g8": (1, "lib", "0: long Linears on the tap_gemm block shapes"),
    "g8_conv": (1, "lib", "0: 3x3 convs onto >= 1024 channels not on the persistent kernel"),
    "g8_temporal": (1, "lib", "0: Conv1d k3 at >= 640 channels not on the persistent kernel"),
    "g8_split": (-1, "lib", "split-K at the 8x12 level: -1 auto, 0 off, n fixed"),
    "lin320": (1, "lib", "0: K = 320 Linears on tap_gemm"),
    "lin320s": (1, "lib", "0: the K-split lin320 kernel instead of the streaming one"),
    "lin640": (1, "lib", "0: K = 640 Linears on the persistent kernel"),
    "temp320": (1, "lib", "0: 320-channel Conv1d k3 on tap_gemm"),
    "attn_short": (1, "lib", "0: temporal attention through the general flash kernel"),
    "attn_text": (1, "lib", "0: text cross-attention through the general flash kernel"),
    "attn_spatial": (1, "lib", "0: the 6144-key (d = 40) and 1536-key (d = 80) self-attention through the general flash kernel; 2: only d = 40 on the specialised kernel"),
    "attn_pv16": (1, "lib", "0: PV product of the spatial attention in 32x32x16 tiles"),
    "attn_opt": (1, "lib", "0: softmax reference of the spatial attention tracked on every key tile (no optimistic first-tile reference)"),
    "gn_flat": (1, "lib", "0: temporal GroupNorm through the per-pixel kernels at the two large levels"),
    "gn_apply_flat": (1, "lib", "0: spatial GroupNorm apply with a wave per pixel row instead of a granule column per thread"),
    "f32_split": (1, "lib", "0: the fp32 first stage's contractions on v_mfma_f32_32x32x2_f32 instead of six exact bf16 x bf16 products per fp32 product "
                            "(three-way operand split, the dropped terms below fp32's own accumulation rounding) on the bf16 matrix pipe; 2: the split form on the eight-wave kernel only"),
}

_values: Dict[str, int] = {}


def _parse(what: str, text: str, legacy_switch: bool = False) -> int:
    """Integer value of a switch.  The pre-round-5 one-variable-per-switch spelling treated any string other than "0" as on
    (CCEDIT_GRAPH="", "off", "true" ...): those keep that meaning instead of failing the import; CCEDIT_POLICY entries must be
    integers and a malformed one names itself."""
    t = text.strip()
    try:
        return int(t)
    except ValueError:
        if legacy_switch:
            return 0 if t.lower() in ("0", "off", "false", "no") else 1
        raise ValueError(f"{what}: '{text}' is not an integer") from None


def _load():
    vals = {k: v[0] for k, v in TABLE.items()}
    for name in TABLE:                                   # legacy spelling
        legacy = {"g8_split": "CCEDIT_G8_SPLIT"}.get(name, "CCEDIT_" + name.upper())
        if legacy in os.environ:
            vals[name] = _parse(legacy, os.environ[legacy], legacy_switch=True)
    spec = os.environ.get("CCEDIT_POLICY", "")
    for item in filter(None, (x.strip() for x in spec.split(","))):
        if "=" not in item:
            raise ValueError(f"CCEDIT_POLICY: '{item}' is not name=value")
        k, v = item.split("=", 1)
        if k.strip() not in TABLE:
            raise ValueError(f"CCEDIT_POLICY: unknown switch '{k.strip()}' (known: {', '.join(TABLE)})")
        vals[k.strip()] = _parse(f"CCEDIT_POLICY: {k.strip()}", v)
    return vals


def get(name: str) -> int:
    if not _values:
        _values.update(_load())
    return _values[name]


def on(name: str) -> bool:
    return get(name) != 0


def non_default() -> Dict[str, int]:
    """Entries that differ from the fast-path defaults (bench.py prints them on its line)."""
    get("graph")
    return {k: v for k, v in _values.items() if v != TABLE[k][0]}


def generic() -> str:
    """The CCEDIT_POLICY string that switches every specialised kernel / fusion / overlap off: the generic kernels the small-size
    tests pin against the oracle and the reference goldens (tests/test_fullsize_gpu.py)."""
    off = {k: 0 for k in TABLE if k not in ("split_cfg", "vae_fp32")}      # (vae_fp32 is a precision choice, not a kernel arm: left at its default)
    return ",".join(f"{k}={v}" for k, v in off.items())


def push_to_library(lib) -> None:
    get("graph")
    for name, (dflt, side, _) in TABLE.items():
        if side == "lib" and _values[name] != dflt:
            rc = lib.ccedit_policy_set(name.encode(), int(_values[name]))
            if rc != 0:
                raise RuntimeError(f"ccedit_policy_set({name}) failed: {lib.ccedit_last_error().decode()}")
