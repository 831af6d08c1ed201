#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the text/JSON kept under profiles/.

  python tools/rocprof_summary.py --trace a.db [--pmc FETCH_SIZE=b.db --pmc WRITE_SIZE=c.db] \
         --out profiles/r01_kernels

Kernel durations come from the kernel-trace run; PMC counters from their own runs (never combined
with tracing domains).  FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B? no: rocprofv3 reports
them in kilobytes (1 unit = 1024 B); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM), so `hbm_read_bytes_corrected` = 2 * FETCH_SIZE * 1024.
"""
import argparse
import json
import sqlite3


def kernel_stats(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    return [dict(name=r[0], calls=r[1], total_us=r[2] / 1e3, avg_us=r[3] / 1e3, min_us=r[4] / 1e3,
                 max_us=r[5] / 1e3, pct=100.0 * r[2] / tot) for r in rows]


def pmc_stats(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    cname = "counter_name" if "counter_name" in cols else "name"
    val = "value" if "value" in cols else "counter_value"
    q = ("select %s, count(*), avg(%s), sum(%s) from counters_collection where %s = ? group by %s"
         % (name_col, val, val, cname, name_col))
    return {r[0]: dict(dispatches=r[1], avg=r[2], total=r[3]) for r in cur.execute(q, (counter,))}


def short(n):
    n = n.split("(")[0]
    return n.replace("esvio::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--pmc", action="append", default=[], help="COUNTER=path.db")
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    ks = kernel_stats(a.trace)
    pmc = {}
    for item in a.pmc:
        c, path = item.split("=", 1)
        try:
            pmc[c] = pmc_stats(path, c)
        except Exception as e:  # pragma: no cover
            pmc[c] = {"error": repr(e)}
    lines = ["# rocprofv3 --kernel-trace --stats summary" + (" — " + a.note if a.note else ""), "",
             "| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k in ks:
        lines.append("| %s | %d | %.1f | %.3f | %.3f | %.3f | %.2f |" % (
            short(k["name"]), k["calls"], k["total_us"], k["avg_us"], k["min_us"], k["max_us"], k["pct"]))
    if pmc:
        lines += ["", "## PMC (separate passes; per-dispatch averages)", "",
                  "| kernel | " + " | ".join(pmc) + " |", "|---|" + "---|" * len(pmc)]
        names = sorted({n for c in pmc.values() for n in c if n != "error"})
        for n in names:
            lines.append("| %s | " % short(n) + " | ".join(
                "%.1f" % pmc[c][n]["avg"] if n in pmc[c] else "-" for c in pmc) + " |")
    open(a.out + ".md", "w").write("\n".join(lines) + "\n")
    json.dump(dict(kernels=ks, pmc=pmc, note=a.note), open(a.out + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
