#!/usr/bin/env python3
"""Which hardware queues did the concurrent callers of a process get?  Reads a rocprofv3 --kernel-trace CSV, finds the windows in
which kernels of several host threads overlap (the replay's three arguments: main thread + two std::threads) and prints, per
window, the (thread, stream, queue) triples with their kernel counts and busy time.  Streams that share a queue run in turn.
  python tools/queue_usage.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
by_thread = collections.defaultdict(list)
for r in rows:
    by_thread[r["Thread_Id"]].append(r)
main = max(by_thread, key=lambda t: len(by_thread[t]))
print(f"{len(rows)} kernels, {len(by_thread)} host threads (main: {main}), queues seen: {sorted({r['Queue_Id'] for r in rows}, key=int)}, "
      f"streams seen: {len({r['Stream_Id'] for r in rows})}")
for t, rs in by_thread.items():
    if t == main or len(rs) < 20:
        continue
    lo, hi = min(r["s"] for r in rs), max(r["e"] for r in rs)
    inside = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
    use = collections.defaultdict(lambda: [0, 0])
    for r in inside:
        k = (r["Thread_Id"], r["Stream_Id"], r["Queue_Id"])
        use[k][0] += 1
        use[k][1] += r["e"] - r["s"]
    print(f"window of thread {t}: {(hi - lo) / 1e6:.2f} ms")
    for (th, st, q), (cnt, busy) in sorted(use.items(), key=lambda kv: -kv[1][0]):
        print(f"   thread {th} stream {st:>3s} queue {q:>2s}: {cnt:4d} kernels, {busy / 1e6:.2f} ms busy")
    qs = collections.defaultdict(set)
    for (th, st, q) in use:
        qs[q].add(st)
    shared = {q: sorted(s) for q, s in qs.items() if len(s) > 1}
    print(f"   queues shared by several busy streams: {shared if shared else 'none'}")
