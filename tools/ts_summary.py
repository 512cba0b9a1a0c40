"""Per-XCD, per-residency-round summary of the per-workgroup time stamps build/gemm_check writes with TS=1
(uint64[wg][16]: 0 entry, 1 / 3 k-loop entered (first / second item), 2 / 4 k-loop left, 5 done, 7 XCC id, 10 first
instruction; 100 MHz clock).   usage: python3 tools/ts_summary.py ts_nn.bin"""
import sys
import numpy as np

h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
ok = h[:, 1] > 0
t0 = h[ok, 10].min()
us = lambda a: (a - t0) / 100.0
print("%d workgroup records; first start to last end %.1f us" % (ok.sum(), (h[ok, :6].max() - t0) / 100.0))
for x in (0, 7):
    idx = np.arange(x, len(h), 8)
    print("XCD %d: round: start of first items [us] (spread), first-item loop mean/min/max, second-item loop mean/min/max, "
          "gap between the items mean" % x)
    for r in range(0, min(len(idx) // 64, 18)):
        w = idx[r * 64:(r + 1) * 64]
        w = w[h[w, 1] > 0]
        if len(w) == 0:
            continue
        s1 = us(h[w, 1])
        l1 = (h[w, 2] - h[w, 1]) / 100.0
        l2 = (h[w, 4] - h[w, 3]) / 100.0
        gap = (h[w, 3] - h[w, 2]) / 100.0
        print("  round %2d: %9.1f (%6.1f)   %6.1f / %6.1f / %6.1f   %6.1f / %6.1f / %6.1f   %5.1f" % (
            r, s1.min(), s1.max() - s1.min(), l1.mean(), l1.min(), l1.max(), l2.mean(), l2.min(), l2.max(), gap.mean()))
