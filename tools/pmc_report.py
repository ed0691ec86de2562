"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "gemm_f32<%s>" % ("".join(c for c in k if c in "01")[-2:]) if "gemm_f32_kernel" in k else k[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    if not any(s in k for s in ("gemm", "Cijk", "MT")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
