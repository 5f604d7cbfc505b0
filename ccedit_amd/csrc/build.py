#!/usr/bin/env python3
"""Build libccedit_hip.so (the C-ABI kernel library) for gfx950 with hipcc.  In-tree output:
ccedit_amd/libccedit_hip.so — git-ignored, but it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, "..", "libccedit_hip.so"))
SOURCES = ["gemm.hip", "gemm8p.hip", "convhalo.hip", "smallconv.hip", "lin320.hip", "lin640.hip", "temp320.hip", "ff320.hip", "norm.hip", "attention.hip", "attnspatial.hip", "attnshort.hip", "attntext.hip", "elementwise.hip", "f32vae.hip", "core.cpp"]
ARCH = "gfx950"
# per-file flags: ff320's GEGLU must stay scalar fp32 (packed fp32 VALU is several times slower beside MFMAs, see the file).
# norm.hip / attnshort.hip: without the SLP vectoriser nothing there becomes a packed-fp32 op whose LOW lane reads the HIGH half of
# a register pair (`v_pk_add_f32 ... op_sel:[0,1]`).  That form returned 0 for the swizzled operand in lanes 48-63 about once per
# 10^7 waves when another stream's tap_gemm kernel (AGPR-resident accumulators) shared the SIMD: LayerNorm beside a GEMM on a
# second stream was not run-to-run reproducible (tools/exp/repro_e4.py; DESIGN.md section 3, streams).  check_isa() refuses it.
EXTRA_FLAGS = {"ff320.hip": ["-fno-slp-vectorize"], "norm.hip": ["-fno-slp-vectorize"], "attnshort.hip": ["-fno-slp-vectorize"]}
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_BAD_ISA = re.compile(r"v_pk_(add|mul|fma)_f32.*op_sel:\[[01,]*1")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = ([os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
            + [os.path.join(HERE, "..", "..", "include", "ccedit_hip.h")])
    return any(os.path.getmtime(d) > t for d in deps)


def check_isa(obj: str) -> int:
    """Disassemble the gfx950 code object inside one of our object files and count the packed-fp32 form described at EXTRA_FLAGS.
    The guard is what allows the ControlNet side stream to be on by default, so a check that could not run is a build failure,
    not a pass (CCEDIT_SKIP_ISA_CHECK=1 builds anyway — then run with CCEDIT_OVERLAP_CONTROLNET=0)."""
    if os.environ.get("CCEDIT_SKIP_ISA_CHECK", "0") == "1":
        sys.stderr.write(f"[build] WARNING: ISA check of {os.path.basename(obj)} skipped on request\n")
        return 0
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f"{OBJDUMP} not found: the packed-fp32 op_sel check cannot run (set CCEDIT_SKIP_ISA_CHECK=1 to build without it)")
    with tempfile.TemporaryDirectory() as tmp:
        o = shutil.copy(obj, tmp)
        subprocess.run([OBJDUMP, "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        n, lines = 0, 0
        for f in os.listdir(tmp):
            if "amdgcn" in f:
                dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=False).stdout
                body = dis.splitlines()
                lines += sum(1 for line in body if "v_" in line or "s_" in line)
                n += sum(1 for line in body if _BAD_ISA.search(line))
        if lines == 0:
            raise RuntimeError(f"{obj}: no gfx950 code object was disassembled — the packed-fp32 op_sel check did not run")
    return n


def _headers():
    return ([os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
            + [os.path.join(HERE, "..", "..", "include", "ccedit_hip.h"), os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    objs = []
    t0 = time.time()
    procs = []
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    for s in SOURCES:
        o = os.path.join(HERE, s.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        src = os.path.join(HERE, s)
        # per-file incremental: an object newer than its source and every header is kept (a forced build recompiles all)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *EXTRA_FLAGS.get(s, []), "-x", "hip", "-c", src, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    ok = True
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            ok = False
            sys.stderr.write(f"[build] {s} FAILED\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(out)
    if not ok:
        raise RuntimeError("hipcc failed")
    rebuilt = {s for s, _ in procs}
    for s, o in zip(SOURCES, objs):
        if s.endswith(".hip") and s in rebuilt:      # (kept objects passed the check when they were built)
            n = check_isa(o)
            if n:
                raise RuntimeError(f"{s}: {n} packed-fp32 instruction(s) with a low-lane read of a high half (op_sel:[..1..]) — "
                                   f"not safe beside another stream's GEMM, see EXTRA_FLAGS in {__file__}")
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", OUT]
    subprocess.check_call(link)
    if verbose:
        sys.stderr.write(f"[build] {OUT} built in {time.time() - t0:.1f}s\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
