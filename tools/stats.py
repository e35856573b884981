import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/{tag}/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total ms", round(tot / 1e6, 2))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
    print(f"{n:60s} calls {int(r['Calls']):5d} tot {float(r['TotalDurationNs'])/1e6:8.3f} ms avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
