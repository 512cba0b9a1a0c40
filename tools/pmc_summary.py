"""Per-kernel PMC summary from rocprofv3 rocpd databases (separate passes, as the HBM section of
MI355X_MICROARCH.md prescribes): FETCH_SIZE (x2 gfx950 correction for wide coalesced reads, KiB units),
WRITE_SIZE, MFMA busy.  usage: python tools/pmc_summary.py <fetch.db> <write.db> <mfma.db> [out.json]"""
import collections
import json
import sqlite3
import sys


def load(path):
    cur = sqlite3.connect(path).cursor()
    out = collections.OrderedDict()
    q = ("select dispatch_id, kernel_name, grid_size_x, counter_name, value, duration "
         "from counters_collection order by dispatch_id")
    for did, kn, gx, cn, val, dur in cur.execute(q):
        out.setdefault(did, dict(name=kn, gx=gx, dur=dur))[cn] = val
    return out


def group(rows, key_counter):
    agg = collections.OrderedDict()
    for r in rows.values():
        k = (r["name"].split("(")[0].replace("void ", ""), r["gx"])
        a = agg.setdefault(k, dict(n=0, dur=0.0, val=collections.defaultdict(float)))
        a["n"] += 1
        a["dur"] += r["dur"]
        for c in key_counter:
            a["val"][c] += r.get(c, 0.0)
    return agg


def main():
    f, w, m = (load(p) for p in sys.argv[1:4])
    fa, wa, ma = group(f, ["FETCH_SIZE"]), group(w, ["WRITE_SIZE"]), group(
        m, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_MFMA_F64"])
    report = []
    for k, a in fa.items():
        if a["dur"] / a["n"] < 2e5:  # skip kernels shorter than 0.2 ms per launch
            continue
        fetch = 2.0 * a["val"]["FETCH_SIZE"] / a["n"] * 1024.0          # bytes per launch, x2 correction
        write = wa.get(k, dict(val={"WRITE_SIZE": 0}, n=1))
        wbytes = write["val"]["WRITE_SIZE"] / max(write["n"], 1) * 1024.0
        mm = ma.get(k)
        util = clk = None
        if mm and mm["val"]["GRBM_GUI_ACTIVE"] > 0:
            gui = mm["val"]["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
            util = mm["val"]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui   # 1024 SIMDs
            clk = gui / mm["dur"]                                         # cycles per ns = GHz
        report.append(dict(kernel=k[0], grid_x=k[1], launches=a["n"], avg_ms=a["dur"] / a["n"] / 1e6,
                           hbm_fetch_bytes_per_launch=fetch, hbm_write_bytes_per_launch=wbytes,
                           mfma_util=util, clock_ghz=clk))
    for r in report:
        print("%-44s grid %8d  n=%3d  %8.3f ms  fetch %8.2f GB  write %7.2f GB  mfma %s  clk %s" % (
            r["kernel"][:44], r["grid_x"], r["launches"], r["avg_ms"], r["hbm_fetch_bytes_per_launch"] / 1e9,
            r["hbm_write_bytes_per_launch"] / 1e9,
            "%.1f%%" % (100 * r["mfma_util"]) if r["mfma_util"] else "-",
            "%.2f" % r["clock_ghz"] if r["clock_ghz"] else "-"))
    if len(sys.argv) > 4:
        json.dump(report, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
