"""Dump the per-kernel summary of a rocprofv3 rocpd database (ROCm 7.2 default output) as text.
usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows:
        lines.append("%-110s %8d %14.1f %12.2f %6.2f%%" % (name[:110], calls, tot, avg, pct))
    try:  # counter passes only: the counters_collection view is what tools/pmc_summary.py reads too
        pm = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                              "group by kernel_name, counter_name"))
    except sqlite3.Error:
        pm = []  # a --stats-only database has no counter view
    if pm:
        lines.append("")
        lines.append("%-90s %-28s %8s %18s" % ("kernel", "counter", "n", "avg_per_dispatch"))
        for kn, pn, n, av in pm:
            lines.append("%-90s %-28s %8d %18.1f" % (kn[:90], pn, n, av))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
