"""ccedit_gemm against the vendor GEMM (torch.matmul = hipBLASLt; CALIBRATION ONLY — nothing on the product path calls a library
GEMM) on the plain-GEMM shapes of DESIGN.md §3.1, hot (one A, one W, repeated launches: operands in the Infinity Cache) and cold
(every launch reads an activation that a producer pass just wrote, six rotating buffer sets: what a GEMM sees inside the network).

Columns: vendor, the best older block shape (t1 128x128 / t4 256x256 four-stage / t6 320x128), and tile 11 = the persistent
eight-phase kernel (gemm8p.hip; 256ch x 256pix, or 128ch x 512pix when Cout = 640).  The vendor's `cold` column uses a
preallocated output (`out=`), its `hot` column allocates per call like the product path does.

    python tools/exp/gemm_vs_vendor.py            # the table
"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import g8_check as g      # noqa: E402  (perf(): hot + cold timing of one shape for a list of block shapes)

if __name__ == "__main__":
    # the six plain-GEMM shapes of the table (the three conv-as-GEMM rows are run as LINEAR with K = 9 Cin, as in round 2)
    g.perf(8192, 8192, 8192, tiles=(1, 4, 11))
    g.perf(52224, 5120, 640, tiles=(1, 4, 11))
    g.perf(13056, 10240, 1280, tiles=(1, 6, 11))
    g.perf(13056, 1280, 1280, tiles=(1, 4, 11))
    g.perf(52224, 640, 5760, tiles=(1, 6, 11))
    g.perf(13056, 1280, 11520, tiles=(1, 4, 11))
    # with the network's epilogues
    g.perf(52224, 5120, 640, geglu=True, tiles=(2, 11))
    g.perf(13056, 10240, 1280, geglu=True, tiles=(6, 11))
    g.perf(52224, 640, 2560, res=True, tiles=(1, 6, 11))
    g.perf(13056, 1280, 5120, res=True, tiles=(4, 6, 11))
    g.perf(13056, 1280, 1280, res=True, tiles=(1, 4, 11))
