"""timeline.py — per-frame GPU timeline from a rocprofv3 --kernel-trace database (rocpd sqlite):
start/end of every dispatch relative to the frame's k_select start, averaged over the published
frames.  Usage: python tools/timeline.py <trace.db> [--skip N]"""
import argparse
import sqlite3
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=int, default=10, help="leading published frames to ignore")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    q = "select name, start, end, %s from kernels order by start" % sid
    rows = con.execute(q).fetchall()
    if not rows:
        sys.exit("no dispatches")

    def short(n):
        n = n.split("(")[0].split("<")[0]
        return n.split("::")[-1].replace("void ", "").strip()
    rows = [(short(n), s, e, qid) for n, s, e, qid in rows]
    # how much of the device time is overlapped: sum of the kernel durations against the time at
    # least one kernel is running (union of the intervals), whole trace and per HIP stream / queue
    def union(iv):
        iv = sorted(iv)
        tot, cs, ce = 0, None, None
        for s0, e0 in iv:
            if cs is None or s0 > ce:
                if cs is not None:
                    tot += ce - cs
                cs, ce = s0, e0
            else:
                ce = max(ce, e0)
        return tot + (ce - cs if cs is not None else 0)
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e6
    ssum = sum(r[2] - r[1] for r in rows) / 1e6
    ubusy = union([(r[1], r[2]) for r in rows]) / 1e6
    print("whole trace: span %.3f ms, sum of kernel durations %.3f ms, union busy %.3f ms "
          "(concurrency %.2fx, device idle %.1f%% of the span)" % (span, ssum, ubusy, ssum / max(ubusy, 1e-9),
                                                                   100.0 * (1 - ubusy / span)))
    by_q = defaultdict(list)
    for r in rows:
        by_q[r[3]].append((r[1], r[2]))
    for qid, iv in sorted(by_q.items(), key=lambda kv: -sum(e0 - s0 for s0, e0 in kv[1])):
        print("  stream/queue %-4s %6d dispatches, busy %.3f ms" % (qid, len(iv), union(iv) / 1e6))
    sel = [i for i, r in enumerate(rows) if "k_select" in r[0]]
    if not sel:
        print(sorted(set(r[0] for r in rows)))
    print("dispatches", len(rows), "k_select launches", len(sel), "columns", cols[:12])
    # one cycle = from a k_select start to the next k_select start
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    cyc = []
    for a_i, b_i in zip(sel[a.skip:-1], sel[a.skip + 1:]):
        t0 = rows[a_i][1]
        cyc.append((rows[b_i][1] - t0) / 1e3)
        seen = defaultdict(int)
        for n, s, e, qid in rows[a_i:b_i]:
            seen[(n, qid)] += 1
            k = (n, qid, seen[(n, qid)])
            acc[k][0] += (s - t0) / 1e3
            acc[k][1] += (e - t0) / 1e3
            acc[k][2] += 1
    print("cycle (select start -> next select start): mean %.1f us over %d cycles" % (sum(cyc) / len(cyc), len(cyc)))
    out = sorted(acc.items(), key=lambda kv: kv[1][0] / kv[1][2])
    for (n, qid, occ), (s, e, cnt) in out:
        if cnt < len(cyc) * 0.5:
            continue
        print("%8.1f %8.1f  (%6.1f us)  q%-3s %s #%d  [%d]" % (s / cnt, e / cnt, (e - s) / cnt, qid, n[:48], occ, cnt))


if __name__ == "__main__":
    main()
