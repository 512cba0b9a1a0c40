"""Kernel timeline of a rocprofv3 rocpd database: start (us, relative), duration, gap to the previous kernel.
usage: python tools/rocpd_timeline.py <results.db> [first] [count]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    ncol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in ncol else ("display_name" if "display_name" in ncol else "name")
    q = "select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)
    rows = list(cur.execute(q))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    t0 = rows[first][1]
    prev_end = None
    for name, st, en in rows[first:first + count]:
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        print("%10.1f %9.1f %7.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, name[:90]))
        prev_end = en
    print("# columns: start_us dur_us gap_us name;", len(rows), "dispatches; cols:", cols)


if __name__ == "__main__":
    main()
