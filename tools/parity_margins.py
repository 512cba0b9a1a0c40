"""Table of achieved parity errors against their asserted bounds, from the log tests/margins.py writes.
    usage (GPU box): GPR_MARGINS_LOG=gpurun_out/margins.jsonl python -m pytest tests -m gpu -q
                     python3 tools/parity_margins.py gpurun_out/margins.jsonl > profiles/r06_parity_margins.txt
One line per (test function, quantity, bound): number of checks, worst error, bound, bound / worst.  The summary at the
end lists, per (quantity, bound), the worst error over the whole run -- the figure the TOL_* constants of
tests/test_gpu_parity.py are set from (<= 10 x the worst observed, rounded up to one digit)."""
import collections
import json
import re
import sys

rows = collections.defaultdict(lambda: [0, 0.0, ""])
for path in sys.argv[1:]:
    for line in open(path):
        try:
            r = json.loads(line)
        except ValueError:
            continue
        fn = re.sub(r"\[.*\]$", "", r["test"].split("::")[-1])
        key = (fn, r["what"], r["tol"])
        e = rows[key]
        e[0] += 1
        if r["err"] >= e[1]:
            e[1] = r["err"]
            e[2] = r["test"].split("::")[-1]
print("%-78s %-24s %6s %10s %9s %9s  %s" % ("test", "quantity", "checks", "worst", "bound", "bound/w", "worst case"))
summary = collections.defaultdict(lambda: [0, 0.0, ""])
for (fn, what, tol), (cnt, worst, case) in sorted(rows.items()):
    print("%-78s %-24s %6d %10.2e %9.1e %9.1f  %s" % (fn, what, cnt, worst, tol, tol / max(worst, 1e-300) if worst > 0 else float("inf"),
                                                       case if case != fn else ""))
    s = summary[(what, tol)]
    s[0] += cnt
    if worst >= s[1]:
        s[1], s[2] = worst, case
print()
print("# per quantity and bound, over the whole run")
print("%-24s %9s %8s %10s %9s  %s" % ("quantity", "bound", "checks", "worst", "bound/w", "worst case"))
for (what, tol), (cnt, worst, case) in sorted(summary.items()):
    print("%-24s %9.1e %8d %10.2e %9.1f  %s" % (what, tol, cnt, worst, tol / max(worst, 1e-300) if worst > 0 else float("inf"), case))
