#!/usr/bin/env python3
"""Build libccedit_hip.so (the C-ABI kernel library) for gfx950 with hipcc.  In-tree output:
ccedit_amd/libccedit_hip.so — git-ignored, but it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, "..", "libccedit_hip.so"))
SOURCES = ["gemm.hip", "gemm8p.hip", "convhalo.hip", "smallconv.hip", "lin320.hip", "ff320.hip", "norm.hip", "attention.hip", "attnshort.hip", "elementwise.hip", "core.cpp"]
ARCH = "gfx950"
# per-file flags: ff320's GEGLU must stay scalar fp32 (packed fp32 VALU is several times slower beside MFMAs, see the file)
EXTRA_FLAGS = {"ff320.hip": ["-fno-slp-vectorize"]}


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


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    objs = []
    t0 = time.time()
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.rsplit(".", 1)[0] + ".o")
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *EXTRA_FLAGS.get(s, []), "-x", "hip", "-c",
               os.path.join(HERE, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
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
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", OUT]
    subprocess.check_call(link)
    if verbose:
        sys.stderr.write(f"[build] {OUT} built in {time.time() - t0:.1f}s\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
