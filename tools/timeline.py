"""Timeline summary of a rocprofv3 --kernel-trace run of bench.py: for the last graph replays, how the wall time of one
iteration splits into "a convolution kernel is running", "only other kernels are running" and "nothing is running",
and which non-convolution kernels run while no convolution does.   usage: python tools/timeline.py <tag> [iterations]"""
import csv, glob, sys, collections
tag = sys.argv[1]
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(f"gpurun_out/{tag}/*/*_kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
# iteration boundary: adam on the generator store (largest adam launch) -- take the last nit+1 of them
adam = [i for i, r in enumerate(rows) if "::adam_kernel" in r[2]]
# adam launches alternate 5 x discriminator, 1 x generator per iteration: the generator's is every 6th
marks = adam[5::6]
# the nit consecutive iterations that took the least time: the run's fastest route in steady state (calibration passes of
# the slower routes, recording and warm-up iterations are longer)
ends = [rows[i][1] for i in marks]
best = min(range(len(ends) - nit), key=lambda j: ends[j + nit] - ends[j])
marks = marks[best:best + nit + 1]
lo, hi = rows[marks[0]][1], rows[marks[-1]][1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
isconv = lambda n: ("conv_" in n and "c1" not in n)
ev = []
for s, e, n, q in sel:
    ev.append((s, 1, isconv(n), n)); ev.append((e, -1, isconv(n), n))
ev.sort(key=lambda x: (x[0], x[1]))
t_prev, nconv, nother = lo, 0, 0
acc = collections.Counter(); alone = collections.Counter(); running = collections.Counter()
for t, d, c, n in ev:
    dt = t - t_prev
    if dt > 0:
        key = "conv" if nconv else ("other" if nother else "idle")
        acc[key] += dt
        if key == "other":
            for k, v in running.items():
                if v > 0: alone[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]] += dt
    t_prev = t
    if c: nconv += d
    else:
        nother += d; running[n] += d
tot = hi - lo
print(f"{nit} iterations, {tot / nit / 1e6:.3f} ms each")
for k in ("conv", "other", "idle"):
    print(f"  {k:6s} {acc[k] / nit / 1e6:7.3f} ms  {100 * acc[k] / tot:5.1f}%")
print("kernels running while no convolution kernel runs (ms per iteration):")
for k, v in alone.most_common(16):
    print(f"  {v / nit / 1e6:7.3f}  {k}")
