"""Parity at BASELINE.json's FULL size (17 x 512 x 768, CFG-doubled batch, shipped widths), where the fp32 CPU oracle
would need minutes per evaluation: size-independent properties instead.

  * the specialised kernels the full-size shapes dispatch to (LDS-halo 3x3 conv, 320-channel block shape, K rotation,
    register-resident-weight K = 320 linears, the fused dim-320 feed-forward,
    channel-tile groups, persistent temporal attention, 8-wave attention blocks, fused GroupNorm statistics, two-stream
    CFG halves, ControlNet on a side stream) must reproduce the GENERIC kernels (tap-gather GEMM, flash kernel, two-pass
    GroupNorm, one stream) — the ones the small-size tests pin against the oracle and the reference goldens;
  * two runs give identical bits; identical CFG halves give identical predictions; the clips of a batch do not interact.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GENERIC = dict(CCEDIT_T6="0", CCEDIT_CONV_HALO="0", CCEDIT_ATTN_SHORT="0", CCEDIT_SPLIT_CFG="0", CCEDIT_OVERLAP_CONTROLNET="0",
               CCEDIT_KROT="0", CCEDIT_CGROUP="0", CCEDIT_FUSE_GN_STATS="0", CCEDIT_TEMPORAL_ORDER="0", CCEDIT_LIN320="0",
               CCEDIT_FF320="0", CCEDIT_LN320="0", CCEDIT_T4="0", CCEDIT_BALANCED="0", CCEDIT_CONV_NARROW="0")


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))


def _run(tmp_path, name, extra_env, workload=None):
    out = os.path.join(str(tmp_path), name + ".npz")
    env = dict(os.environ)
    for k in GENERIC:
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_fullsize_eval.py"), out] + ([workload] if workload else []), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.timeout(1800)
def test_full_size_properties(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    fast = _run(tmp_path, "fast", {})
    again = _run(tmp_path, "again", {})
    gen = _run(tmp_path, "generic", GENERIC)
    assert fast["eps"].shape == (2, 4, 17, 64, 96) and fast["frames"].shape == (1, 3, 3, 512, 768)
    assert np.isfinite(fast["eps"]).all() and np.isfinite(gen["eps"]).all() and np.isfinite(fast["frames"]).all()
    # (1) reproducible: another process computes the identical bits (GroupNorm statistics meet in double atomics whose
    #     arrival order cannot move the fp32 mean / rstd; everything else has a fixed summation order)
    for k in ("eps", "eps_same", "eps_other", "frames"):
        assert np.array_equal(fast[k], again[k]), f"{k}: two runs of the same evaluation differ"
    # (2) identical CFG halves (each on its own stream) give identical predictions; (3) clips do not interact
    assert np.array_equal(fast["eps_same"][0], fast["eps_same"][1])
    assert np.array_equal(fast["eps_other"][0], fast["eps"][0])
    # (4) specialised kernels == generic kernels, up to the bf16 noise floor: both are bf16 realisations of the same fp32
    #     computation with different summation orders, and a single flipped bf16 rounding spreads to that floor within a
    #     few layers (measured 2.0e-2 on eps, 1.1e-2 on decoded frames — the distance between ANY two summation orders);
    #     the stated tolerance of one network evaluation against the fp32 oracle is 5e-2.
    e, ev = _rel(fast["eps"], gen["eps"]), _rel(fast["frames"], gen["frames"])
    print(f"full size: fast vs generic kernels: eps {e:.4f}, VAE frames {ev:.4f}")
    assert e < 3.5e-2, e
    assert ev < 2.5e-2, ev


@pytest.mark.timeout(1800)
def test_full_size_tvi2v_properties(tmp_path):
    """BASELINE.json config 3 (TVI2V: controlnet_img + SpatialTransformer3DCA anchor attention over 2 x 6144 keys) at
    17 x 512 x 768 — the same size-independent properties, plus: the reference frame of one CFG half reaches only that half."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    fast = _run(tmp_path, "tv_fast", {}, "tvi2v")
    again = _run(tmp_path, "tv_again", {}, "tvi2v")
    gen = _run(tmp_path, "tv_generic", GENERIC, "tvi2v")
    assert fast["eps"].shape == (2, 4, 17, 64, 96) and np.isfinite(fast["eps"]).all() and np.isfinite(gen["eps"]).all()
    for k in ("eps", "eps_same", "eps_ref"):
        assert np.array_equal(fast[k], again[k]), f"{k}: two runs of the same evaluation differ"
    assert np.array_equal(fast["eps_same"][0], fast["eps_same"][1])
    assert np.array_equal(fast["eps_ref"][0], fast["eps"][0])
    assert _rel(fast["eps_ref"][1], fast["eps"][1]) > 1e-2          # the reference latent does condition the prediction
    e = _rel(fast["eps"], gen["eps"])
    print(f"full size TVI2V: fast vs generic kernels: eps {e:.4f}")
    assert e < 3.5e-2, e
