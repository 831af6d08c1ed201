#!/usr/bin/env python3
"""raw_cycle.py <rocprofv3 rocpd .db> [--cycle K] [--n 1]: every dispatch between the K-th k_select start and the
(K+n)-th, not averaged (tools/timeline.py averages; this shows one cycle as it ran)."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--cycle", type=int, default=15)
    ap.add_argument("--n", type=int, default=1)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = con.execute("select name, start, end, %s from kernels order by start" % sid).fetchall()

    def short(n):
        n = n.split("(")[0].split("<")[0]
        return n.split("::")[-1].replace("void ", "").strip()
    rows = [(short(n), s, e, q) for n, s, e, q in rows]
    sel = [s for n, s, e, q in rows if n.startswith("k_select")]
    t0, t1 = sel[a.cycle], sel[a.cycle + a.n]
    for n, s, e, q in rows:
        if t0 <= s < t1:
            print("%8.1f %8.1f (%6.1f us) q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))


if __name__ == "__main__":
    main()
