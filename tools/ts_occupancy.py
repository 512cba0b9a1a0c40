"""Occupancy over time of one engine launch from the per-workgroup time stamps build/gemm_check writes with TS=1:
how many workgroups are inside a k-loop in each 20 us bin, and how the launch's span compares with (sum of k-loop time) / slots.
usage: python3 tools/ts_occupancy.py ts_nn.bin [slots=512]"""
import sys
import numpy as np

h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ok = h[:, 1] > 0
h = h[ok]
t0 = h[:, 10].min()
us = lambda a: (a - t0) / 100.0
end = np.where(h[:, 5] > 0, h[:, 5], np.where(h[:, 3] > 0, h[:, 3], h[:, 2]))
span = us(end).max()
loops = (h[:, 2] - h[:, 1]) + np.where(h[:, 4] > 0, h[:, 4] - h[:, 3], 0.0)
life = end - h[:, 10]
print("%d workgroups, span %.1f us; k-loop time per workgroup %.1f us (lifetime %.1f): loops fill %.1f %% of span x %d slots, lifetimes %.1f %%"
      % (len(h), span, loops.mean() / 100, life.mean() / 100, 100 * loops.sum() / 100 / (span * slots), slots,
         100 * life.sum() / 100 / (span * slots)))
bins = np.arange(0, span + 20, 20.0)
occ = np.zeros(len(bins))
for s, e in zip(us(h[:, 10]), us(end)):
    occ[int(s // 20):int(e // 20) + 1] += 1
step = max(1, len(bins) // 40)
print("resident workgroups over time (every %d us):" % (20 * step), " ".join("%d" % v for v in occ[::step]))
