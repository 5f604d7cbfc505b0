"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
that include/ccedit_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def libpath():
    from ccedit_amd.csrc.build import build
    return build(force=False, verbose=False)


def _declared():
    src = open(os.path.join(ROOT, "include", "ccedit_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccedit_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ccedit_hip.h but not exported"
    lib.ccedit_abi_version.restype = ctypes.c_int
    from ccedit_amd import hip
    assert lib.ccedit_abi_version() == 12 == hip.ABI_VERSION


def test_binding_matches_header(libpath):
    from ccedit_amd import hip
    assert sorted(hip.EXPORTS) == _declared()
    # descriptor layouts: sizes the C side was compiled with (kept in sync by hand; a mismatch shows up here)
    assert ctypes.sizeof(hip.CcGemmDesc) == 8 + 34 * 4 + 9 * 8 + (8 + 8 + 2 * 4) + 4 * 8 + 2 * 4 + 2 * 8 + 8      # ... + workspace, workspace_bytes, split_k, subpix; ln_colsum, ln_stats, ln_sums, row_sums, ln_sums_eps, vpad (ABI 9); halo_top, halo_bot (ABI 10); Wfrag (ABI 12)
    assert ctypes.sizeof(hip.CcFf320Desc) == 8 + 6 * 4 + 5 * 8 + 5 * 8 + 4 * 4                # ... + a, res, res2, bop, bpp, lda, ldr, ldr2, pad (ABI 10: block tail)
    assert ctypes.sizeof(hip.CcAttnDesc) % 8 == 0
    assert ctypes.sizeof(hip.CcGemmF32Desc) == 5 * 8 + 8 + 16 * 4                               # ABI 11: fp32 first-stage model


def test_invalid_arguments_are_reported_not_crashing(libpath):
    """Argument validation runs before any HIP call, so it can be exercised without a GPU."""
    from ccedit_amd import hip
    lib = hip.lib()
    rc = lib.ccedit_gemm(None, None)
    assert rc == -1 and b"null descriptor" in lib.ccedit_last_error()
    d = hip.CcGemmDesc()
    d.M, d.N, d.Cin, d.taps, d.Kpad = 16, 6, 8, 1, 64
    d.A = d.W = d.out = 1
    d.lda, d.ldc = 8, 8
    rc = lib.ccedit_gemm(ctypes.byref(d), None)
    assert rc == -2 and b"multiple of 4" in lib.ccedit_last_error()
    a = hip.CcAttnDesc()
    assert lib.ccedit_attention(ctypes.byref(a), None) == -1
    f = hip.CcGemmF32Desc()
    f.A = f.W = f.out = 16
    f.M, f.N, f.Cin, f.Cpad, f.Kpad, f.lda, f.ldw, f.ldc = 64, 8, 6, 16, 16, 8, 16, 8
    assert lib.ccedit_gemm_f32(ctypes.byref(f), None) == -1 and b"multiples of 4" in lib.ccedit_last_error()
    assert lib.ccedit_groupnorm_f32(16, 16, 16, 16, 16, 1, 64, 96, 1e-6, 1, None) == -1 and b"128, 256, 512" in lib.ccedit_last_error()
    assert lib.ccedit_softmax_rows_f32(16, 4, 9000, 8000, 1.0, None) == -1          # ld < cols (rows of any length are served since round 6)


def test_ln_eps_is_refused_where_no_kernel_normalises(libpath):
    """CcGemmDesc.ln_eps folds a LayerNorm into the GEMM; only the K = 320 register-resident kernel implements it.  Any other
    block shape or geometry must be refused — a kernel that ignored the field would multiply gamma / beta-folded weights with
    un-normalised rows and return a plausible, wrong tensor."""
    from ccedit_amd import hip
    lib = hip.lib()

    def desc(cin, n, tile):
        d = hip.CcGemmDesc()
        d.M, d.N, d.Cin, d.Cin1, d.taps, d.Kpad = 4096, n, cin, cin, 1, cin
        d.A = d.W = d.out = 1
        d.lda, d.ldc, d.tile, d.ln_eps = cin, n, tile, 1e-5
        return d
    for cin, n, tile in ((320, 320, 1), (320, 320, 2), (320, 320, 11), (640, 320, 0), (320, 256, 0), (320, 320, 6)):
        rc = lib.ccedit_gemm(ctypes.byref(desc(cin, n, tile)), None)
        assert rc == -2 and b"ln_eps" in lib.ccedit_last_error(), (cin, n, tile, rc, lib.ccedit_last_error())
    d = desc(320, 320, 0)
    d.res1, d.ldr1 = 1, 320                     # a residual epilogue is not available together with the normalisation either
    assert lib.ccedit_gemm(ctypes.byref(d), None) == -2


def test_ln_stats_is_refused_where_no_kernel_applies_it(libpath):
    """CcGemmDesc.ln_stats / ln_colsum (LayerNorm applied in the epilogue of a GEMM on the raw rows): only the persistent
    eight-phase kernel implements it; every other block shape / geometry must refuse rather than multiply folded weights with
    un-normalised rows."""
    from ccedit_amd import hip
    lib = hip.lib()

    def desc(cin, n, tile, **kw):
        d = hip.CcGemmDesc()
        d.M, d.N, d.Cin, d.Cin1, d.taps, d.Kpad = 8192, n, cin, cin, 1, cin
        d.A = d.W = d.out = d.ln_stats = d.ln_colsum = 1
        d.lda, d.ldc, d.tile = cin, n, tile
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    bad = [desc(640, 640, 1), desc(640, 640, 4), desc(640, 640, 6), desc(320, 320, 9), desc(640, 640, 0, res1=1, ldr1=640),
           desc(640, 640, 0, ln_colsum=None), desc(640, 640, 0, ln_stats=None), desc(640, 640, 0, act=1), desc(640, 648, 0)]
    for d in bad:
        assert lib.ccedit_gemm(ctypes.byref(d), None) == -2 and b"ln_stats" in lib.ccedit_last_error(), lib.ccedit_last_error()
    # the producer side (row_sums: the epilogue accumulates the LayerNorm statistics of what it writes) likewise
    for d in (desc(640, 640, 1, ln_stats=None, ln_colsum=None, row_sums=1), desc(640, 640, 0, ln_stats=None, ln_colsum=None, row_sums=1, act=1),
              desc(320, 320, 9, ln_stats=None, ln_colsum=None, row_sums=1), desc(640, 640, 0, row_sums=1)):
        assert lib.ccedit_gemm(ctypes.byref(d), None) == -2, lib.ccedit_last_error()


def test_split_k_workspace_size_query(libpath):
    """ccedit_gemm_workspace_bytes: which calls would split their K loop over several workgroups, and how much scratch the
    caller is asked to lend (4096 bytes of arrival counters + one fp32 256 x 256 slot per tile and split).  No GPU needed."""
    from ccedit_amd import hip
    lib = hip.lib()

    def desc(m, n, cin, taps, mode):
        d = hip.CcGemmDesc()
        d.M, d.N, d.Cin, d.Cin1, d.taps, d.mode, d.Kpad, d.korder = m, n, cin, cin, taps, mode, cin * taps, int(taps > 1)
        d.lda, d.ldc = cin, n
        if mode == 1:
            d.Hin = d.Hout = 8
            d.Win = d.Wout = 12
            d.stride, d.pad, d.ksize = 1, 1, 3
        if mode == 2:
            d.T, d.HW = 17, 96
        d.A = d.W = d.out = 1
        return d
    slot = 256 * 256 * 4
    # the 8x12 level: 3264 pixels x 1280 channels = 13 x 5 tiles -> 3 splits fill 195 of the 256 CUs
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(3264, 1280, 1280, 9, 1))) == 4096 + 65 * 3 * slot
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(3264, 1280, 1280, 3, 2))) == 4096 + 65 * 3 * slot
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(3264, 1280, 5120, 1, 0))) == 4096 + 65 * 3 * slot
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(480, 1280, 1280, 9, 1))) == 4096 + 10 * 8 * slot      # at most 8 splits
    # short K loops, outputs that fill the chip, GEGLU and the other block shapes are never split
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(3264, 1280, 1280, 1, 0))) == 0
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(desc(13056, 1280, 1280, 9, 1))) == 0
    d = desc(3264, 1280, 5120, 1, 0)
    d.tile = 1
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(d)) == 0
    d = desc(3264, 10240, 1280, 1, 0)
    d.act = 2
    assert lib.ccedit_gemm_workspace_bytes(ctypes.byref(d)) == 0
    assert lib.ccedit_gemm_workspace_bytes(None) == 0


def test_no_packed_fp32_low_lane_high_half_reads(libpath):
    """The one instruction form that was not safe beside another stream's GEMM (csrc/build.py EXTRA_FLAGS, DESIGN.md section 3
    "Streams"): no object of the library may contain it — the build refuses it, and this checks the objects that are shipped."""
    from ccedit_amd.csrc import build
    if not os.path.exists(build.OBJDUMP):
        pytest.skip("llvm-objdump not present")
    for s in build.SOURCES:
        o = os.path.join(build.HERE, s.rsplit(".", 1)[0] + ".o")
        if s.endswith(".hip") and os.path.exists(o):
            assert build.check_isa(o) == 0, s


def test_graft_entry_build_runs(libpath):
    """__graft_entry__.build() is the driver's "does it build" check: it must pass on this (GPU-less) machine — compile, load, agree
    on the ABI version with the binding, import the package and the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_graft_entry_under_test", os.path.join(ROOT, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()


def test_product_never_imports_oracle():
    """The product package must not reach the CPU oracle (or any CPU fallback)."""
    import subprocess, sys
    out = subprocess.run(["grep", "-rnE", r"^\s*(from|import)\s+oracle|ccedit_oracle", os.path.join(ROOT, "ccedit_amd"),
                          os.path.join(ROOT, "sgm"), "--include=*.py"], capture_output=True, text=True).stdout
    assert out.strip() == "", out
