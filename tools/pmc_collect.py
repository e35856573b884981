"""Collect the rocprofv3 --pmc passes left by tools/pmc_table.sh under gpurun_out/ into one JSON table (stdout): per launch
shape the averages of FETCH_SIZE and WRITE_SIZE (KB per launch, raw counters) of the convolution kernel and of the
helpers launched with it (memset of split-K outputs, slab sum), next to the algorithmic bytes."""
import collections
import csv
import glob
import json
import os
import re

LAYER = {16: (128, 64), 8: (256, 128), 4: (512, 256)}
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_*_f")):
    m = re.match(r"gpurun_out/pmc_(down|up|wgrad)_(\d+)_(\d+)_f", d)
    if not m:
        continue
    kind, N, hs = m.group(1), int(m.group(2)), int(m.group(3))
    Cs, Cl = LAYER[hs]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for part, counter in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
        fs = glob.glob(d[:-1] + part + "/*/*counter_collection.csv")
        if not fs:
            continue
        for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):      # the latest pass of that shape
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            vals[k][counter].append(float(r["Counter_Value"]))
    small, large, planes = N * Cs * hs * hs * 4 / 1024, N * Cl * 4 * hs * hs * 4 / 1024, Cs * Cl * 16 * 4 / 1024
    alg = {"down": dict(inp=large + planes, out=small), "up": dict(inp=small + planes, out=large),
           "wgrad": dict(inp=small + large, out=Cs * Cl * 16 * 4 / 1024)}[kind]
    ent = dict(algorithmic_in_KB=round(alg["inp"]), algorithmic_out_KB=round(alg["out"]), kernels={})
    fetch = write = 0.0
    for k, v in vals.items():
        if "conv_" not in k and "slab_sum" not in k and "fillBuffer" not in k:
            continue
        f_ = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
        w_ = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        ent["kernels"][k[:60]] = dict(FETCH_SIZE_KB=round(f_), WRITE_SIZE_KB=round(w_), launches=len(v["FETCH_SIZE"]))
        if "conv_" in k:
            fetch, write = f_, w_
    ent["FETCH_SIZE_KB"], ent["WRITE_SIZE_KB"] = round(fetch), round(write)
    ent["fetch_over_algorithmic"] = round(fetch / alg["inp"], 2)
    ent["fetch_x2_over_algorithmic"] = round(2 * fetch / alg["inp"], 2)
    ent["write_over_algorithmic"] = round(write / alg["out"], 2)
    out[f"conv_{kind}[N={N},Cs={Cs},Cl={Cl},hs={hs}]"] = ent
print(json.dumps(out, indent=1))
