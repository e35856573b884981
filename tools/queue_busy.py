"""Per hardware queue of a rocprofv3 kernel trace (tools/prof.sh run of bench.py): over the timed iterations, the time each
queue has a kernel running, the time exactly k queues have one running at once, and the longest kernels per queue.
usage: tools/queue_busy.py <tag> [first_fraction last_fraction]   (window of the trace, default the middle 0.35-0.6)"""
import collections
import csv
import glob
import re
import sys

tag = sys.argv[1]
lo_f, hi_f = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.6)
rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/{tag}/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
lo, hi = t0 + (t1 - t0) * lo_f, t0 + (t1 - t0) * hi_f
sel = [r for r in rows if int(r["Start_Timestamp"]) >= lo and int(r["End_Timestamp"]) <= hi]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
print(f"window: {len(sel)} dispatches over {span:.2f} ms")
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ks) / 1e6
    per = collections.Counter()
    for s, e, n in ks:
        n = re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "")).replace("void ", "")
        per[re.sub(r"<.*$", "", n)] += (e - s) / 1e6
    top = ", ".join(f"{k} {v / span * 100:.0f}%" for k, v in per.most_common(6))
    print(f"queue {q}: {len(ks):5d} kernels, running {busy / span * 100:5.1f}% of the window | {top}")
# how many queues have a kernel running at once
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("kernels running at once: " + ", ".join(f"{k}: {v / tot * 100:.1f}%" for k, v in sorted(hist.items())))
