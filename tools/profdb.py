"""Per-iteration kernel table from a rocprofv3 --kernel-trace database (rocpd .db): iterations are delimited by the
generator's Adam launch (the largest Adam grid), kernels are summed over a window of whole iterations.

    python tools/profdb.py gpurun_out/prof/x_results.db [first_iter last_iter] [--csv out.csv]
"""
import collections
import re
import sqlite3
import sys

import numpy as np


def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(?:\(anonymous namespace\)::)?([A-Za-z0-9_:]+)(<[^(]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    if csv:
        args.remove(csv)
    c = sqlite3.connect(args[0])
    rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
    ad = [(r[1], r[4]) for r in rows if "adam" in r[0]]
    gmax = max(g for _, g in ad)
    marks = [t for t, g in ad if g == gmax]
    lo, hi = (int(args[1]), int(args[2])) if len(args) >= 3 else (len(marks) // 3, len(marks) // 3 + 20)
    t0, t1, iters = marks[lo], marks[hi], hi - lo
    win = [r for r in rows if t0 <= r[1] < t1]
    tot, cnt = collections.Counter(), collections.Counter()
    for n, s, e, *_ in win:
        tot[short(n)] += e - s
        cnt[short(n)] += 1
    S = sum(tot.values())
    # time with at least one kernel running
    ev = sorted([(r[1], 1) for r in win] + [(r[2], -1) for r in win])
    busy, depth, last = 0, 0, None
    for t, d in ev:
        if depth > 0:
            busy += t - last
        depth += d
        last = t
    print(f"window: {iters} iterations, {(t1 - t0) / iters / 1e6:.3f} ms each; kernel time summed {S / iters / 1e6:.3f} ms, "
          f"some kernel running {busy / iters / 1e6:.3f} ms, launches {len(win) / iters:.0f} per iteration")
    lines = ["name,calls_per_iter,avg_us,ms_per_iter,percent"]
    for n, v in tot.most_common():
        lines.append(f"{n},{cnt[n] / iters:.1f},{v / cnt[n] / 1e3:.2f},{v / iters / 1e6:.4f},{100 * v / S:.2f}")
    print("\n".join(lines[:60]))
    if csv:
        open(csv, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
