"""Per-launch-shape kernel times from a rocprofv3 kernel trace: usage  tools/trace_shapes.py <tag> [bench.json]

rocprofv3's --stats table aggregates per kernel symbol, i.e. per template instantiation, which mixes launch shapes (batch 64
and 128 chains, different channel counts).  This reduces the per-dispatch trace of the same run (gpurun_out/<tag>/*/
*_kernel_trace.csv) to one line per (kernel, grid, workgroup): calls, average / minimum / maximum duration.  A launch shape of
bench.py's roofline.per_shape is one such group (the grid is a function of the shape), so its avg_us can be recomputed here;
with the bench line given, its per_shape entries are printed beside the groups of the same kernel family for comparison."""
import collections
import csv
import glob
import json
import re
import sys

tag = sys.argv[1]
files = glob.glob(f"gpurun_out/{tag}/*/*_kernel_trace.csv")
if not files:
    sys.exit(f"no kernel trace under gpurun_out/{tag}")
groups = collections.defaultdict(list)
for row in csv.DictReader(open(files[0])):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)                      # drop the argument list, keep template arguments
    name = re.sub(r"^void ", "", name)
    grid = tuple(int(row[f"Grid_Size_{a}"]) for a in "XYZ")
    wg = tuple(int(row[f"Workgroup_Size_{a}"]) for a in "XYZ")
    blocks = tuple(g // max(w, 1) for g, w in zip(grid, wg))
    groups[(name, blocks, wg[0])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
# regions between the marker pairs bench.py puts around every timed launch shape (a kernel of the loader that nothing else
# in the run uses), in the order of roofline.per_shape / hbm_kernels of the same run's JSON line
rows = sorted(csv.DictReader(open(files[0])), key=lambda r: int(r["Start_Timestamp"]))
regions, cur = [], None
for row in rows:
    if "pack_sites_kernel" in row["Kernel_Name"]:
        if cur is None:
            cur = []
        else:
            regions.append(cur)
            cur = None
    elif cur is not None:
        cur.append(row)
shape_names = []
if len(sys.argv) > 2:
    line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1]
    shape_names = json.loads(line).get("timed_shapes", [])
if regions:
    print(f"{len(regions)} timed regions (marker pairs); per region: dispatches, span / launches of the leading kernel, kernels")
    for i, reg in enumerate(regions):
        if not reg:
            continue
        span = (max(int(r["End_Timestamp"]) for r in reg) - min(int(r["Start_Timestamp"]) for r in reg)) / 1e3
        per = collections.defaultdict(list)
        for r in reg:
            nm = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
            per[nm].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        launches = max(len(v) for v in per.values())
        label = shape_names[i] if i < len(shape_names) else ""
        kern = "; ".join(f"{k[:44]} x{len(v)} avg {sum(v) / len(v):.2f}" for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])))
        print(f"  {i:3d} {label:44s} span/launch {span / launches:8.2f} us  sum of kernels/launch {sum(sum(v) for v in per.values()) / launches:8.2f} us | {kern}")
    print()
print(f"{len(groups)} (kernel, blocks, threads) groups in {files[0]}")
print(f"{'kernel':58s} {'blocks':>16s} {'thr':>4s} {'calls':>6s} {'avg us':>9s} {'min':>8s} {'max':>8s} {'total ms':>9s}")
for (name, blocks, thr), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    b = "x".join(str(v) for v in blocks if v > 1) or "1"
    print(f"{name[:58]:58s} {b:>16s} {thr:4d} {len(d):6d} {sum(d) / len(d):9.2f} {min(d):8.2f} {max(d):8.2f} {sum(d) / 1e3:9.3f}")
if len(sys.argv) > 2:
    line = [l for l in open(sys.argv[2]) if l.startswith('{"metric"')][-1]
    per = json.loads(line)["roofline"].get("per_shape", {})
    print("\nroofline.per_shape of the bench line (HIP events around back-to-back launches):")
    for k, v in per.items():
        print(f"  {k:48s} {json.dumps(v)}")
