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
    try:
        pm = list(cur.execute(
            "select k.name, p.name, count(*), avg(e.value), sum(e.value) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join rocpd_info_kernel_symbol k on d.kernel_id = k.id group by k.name, p.name"))
        if pm:
            lines.append("")
            lines.append("%-90s %-28s %8s %18s" % ("kernel", "counter", "n", "avg_per_dispatch"))
            for kn, pn, n, av, sm in pm:
                lines.append("%-90s %-28s %8d %18.1f" % (kn[:90], pn, n, av))
    except sqlite3.Error as e:
        lines.append("(no pmc data: %s)" % e)
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
