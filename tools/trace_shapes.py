"""Per launch shape summary of a rocprofv3 --kernel-trace run: groups the dispatches of *_kernel_trace.csv by
(kernel, grid size, workgroup size).  usage: python tools/trace_shapes.py gpurun_out/<tag> [min_share_percent]"""
import collections
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/*/*_kernel_trace.csv")[0]
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    grid = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
    agg[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot / 1e3:.2f} ms over {sum(len(v) for v in agg.values())} dispatches")
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    share = 100 * sum(v) / tot
    if share < cut:
        continue
    v.sort()
    print(f"{share:5.2f}% n={len(v):5d} avg={sum(v) / len(v):8.2f} med={v[len(v) // 2]:8.2f} min={v[0]:8.2f} us  "
          f"{name[:48]:48s} blocks={grid[0]}x{grid[1]} wg={grid[2]}")
