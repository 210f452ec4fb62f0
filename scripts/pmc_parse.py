import csv, glob, collections, re, sys
pat = sys.argv[1]; kfilter = sys.argv[2] if len(sys.argv) > 2 else "spmv_w<1"
def short(k):
    m = re.search(r"k_[a-z_0-9]+(<[^>(]*>)?", k)
    return m.group(0) if m else k[:30]
for f in sorted(glob.glob(pat)):
    agg = collections.defaultdict(list); dur = []
    for row in csv.DictReader(open(f)):
        if kfilter in short(row["Kernel_Name"]):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for c, vals in agg.items():
        vals = sorted(vals)
        print(f"{c:36s} n={len(vals):3d} median={vals[len(vals)//2]:16.1f}")
    if dur:
        dur.sort(); print(f"   (kernel duration under PMC: median {dur[len(dur)//2]/1e3:.1f} us)")
