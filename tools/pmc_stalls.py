"""Summarise the counter passes left by tools/pmc_stalls.sh: per launch shape, how a wave's cycles split into issuing,
issue-stalled and parked, how busy the matrix pipe was, and what the LDS cost."""
import collections
import csv
import glob
import os
import re

rows = []
for d in sorted(glob.glob("gpurun_out/st_*_a")):
    m = re.match(r"gpurun_out/st_(up|down|wgrad)_(\d+)_(\d+)_a", d)
    if not m:
        continue
    vals = collections.defaultdict(list)
    for part in "ab":
        fs = glob.glob(d[:-1] + part + "/*/*counter_collection.csv")
        for f in ([max(fs, key=os.path.getmtime)] if fs else []):      # the latest pass of that shape
            for r in csv.DictReader(open(f)):
                if "_x3_kernel" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = {k: sum(x) / len(x) for k, x in vals.items() if x}
    if "SQ_WAVE_CYCLES" not in v:
        continue
    wc = v["SQ_WAVE_CYCLES"]
    pct = lambda k: 100.0 * v.get(k, 0.0) / wc
    rows.append((m.group(1), int(m.group(2)), int(m.group(3)), pct("SQ_ACTIVE_INST_ANY"), pct("SQ_WAIT_INST_ANY"),
                 pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_LDS"),
                 100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                 v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1), v.get("SQ_INSTS_LDS", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1),
                 v.get("SQ_INSTS_VMEM", 0) / max(v.get("SQ_INSTS_MFMA", 1), 1)))
print("kernel N hs | % of wave cycles: issuing / issue-stalled / parked (of which LDS issue stall) | "
      "LDS bank-conflict % of LDS cycles | per MFMA: VALU, LDS, VMEM instructions")
for r in rows:
    print(f"{r[0]:5s} {r[1]:4d} {r[2]:2d} | {r[3]:5.1f} / {r[4]:5.1f} / {r[5]:5.1f} ({r[6]:4.1f}) | {r[7]:5.1f} | "
          f"{r[8]:5.2f} {r[9]:5.2f} {r[10]:5.2f}")
