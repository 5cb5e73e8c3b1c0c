#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused chain kernels from the stamps of a -DMPNHIP_CHAIN_TS build.

    make clean && make EXTRA=-DMPNHIP_CHAIN_TS
    MPNHIP_CHAIN_TS=gpurun_out/stamps python bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/chain_stamps.py gpurun_out/stamps_fwd.txt fwd ;  python tools/chain_stamps.py gpurun_out/stamps_bwd.txt bwd

`ideal` = MFMA count of the phase x 64 cycles (v_mfma_f32_32x32x2_f32) for the 128-d dims (tiles 10/2/7/4)."""
import sys
import numpy as np

PH = {
    "fwd": [("start->chunk0", 0, 1, 0), ("phase1 loop", 1, 3, 640), ("phase2 (+Pc, Pf gathers)", 3, 4, 320),
            ("phase3", 4, 5, 32), ("phase4", 5, 6, 224), ("phase5 mfma", 6, 7, 448), ("epilogue", 7, 8, 0)],
    "bwd": [("start->chunk0", 0, 1, 0), ("B1 dZM", 1, 2, 0), ("B2 mfma", 2, 3, 448), ("HF mask/save", 3, 4, 0),
            ("B3", 4, 5, 224), ("B4+dZ2", 5, 6, 32), ("B5 mfma", 6, 7, 320), ("H1 mask/save", 7, 8, 0), ("B6", 8, 9, 640)],
}

def main():
    a = np.loadtxt(sys.argv[1], dtype=np.int64)
    kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    last = PH[kind][-1][2]
    a = a[(a[:, 0] > 0) & (a[:, last] > 0)]
    life = a[:, last] - a[:, 0]
    print("waves %d  life cycles: min %d median %d max %d;  first start -> last end %d" %
          (len(a), life.min(), np.median(life), life.max(), a[:, last].max() - a[:, 0].min()))
    order = np.argsort(life)
    fast, slow = order[: len(order) // 4], order[-(len(order) // 4):]
    print("%-26s %9s %9s %9s %9s   %s" % ("phase", "mean", "fast25%", "slow25%", "min", "ideal"))
    for name, i, j, mf in PH[kind]:
        d = a[:, j] - a[:, i]
        print("%-26s %9.0f %9.0f %9.0f %9d   %d" % (name, d.mean(), d[fast].mean(), d[slow].mean(), d.min(), mf * 64))
    placement(a, last)

def placement(a, last):
    """How the dispatcher spread the waves: waves per CU from the HW_ID / XCC_ID stamp (slot 15)."""
    w = a[:, 15]
    if not (w > 0).any():
        return
    hw, xcc = w & 0xffffffff, (w >> 32) & 0xf
    cu, sh, se, simd = (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7, (hw >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus, counts = np.unique(key, return_counts=True)
    print("placement: %d CUs used; waves per CU histogram:" % len(cus), dict(zip(*np.unique(counts, return_counts=True))))
    life = a[:, last] - a[:, 0]
    for c in sorted(set(counts)):
        sel = np.isin(key, cus[counts == c])
        print("  CUs with %2d waves: mean wave life %8.0f cycles, last end %d" % (c, life[sel].mean(), (a[sel, last] - a[:, 0].min()).max()))
    skey = key * 4 + simd
    _, sc = np.unique(skey, return_counts=True)
    print("  waves per SIMD histogram:", dict(zip(*np.unique(sc, return_counts=True))))


if __name__ == "__main__":
    main()
