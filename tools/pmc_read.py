import csv, glob, sys, collections
tag = sys.argv[1]
for part in "abcfw":
    fs = glob.glob(f'gpurun_out/{tag}_{part}/*/*counter_collection.csv')
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:50]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'conv' not in k: continue
        print(k)
        for c, v in d.items():
            print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
