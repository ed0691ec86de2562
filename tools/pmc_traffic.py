"""HBM traffic of the GEMM kernel from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slots), both in KiB; on gfx950 FETCH_SIZE reports half the
bytes of 16-B/lane streaming reads, so it is doubled.  WRITE_SIZE is uncalibrated (used as is).
Usage: python tools/pmc_traffic.py <dir with *counter_collection.csv> <out.json>"""
import collections
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
B16 = "bf16-storage GEMMs (gemm_b16r_kernel, gemm_b16w_kernel, gemm_x3p_kernel<.., 1>)"      # the label bench.py looks up for mixed_precision configs
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = ("gemm_f32_kernel" if "gemm_f32_kernel" in k else "gemm_bf16_kernel" if "gemm_bf16_kernel" in k
               else "gemm_x3_kernel" if ("gemm_x3_kernel" in k or "gemm_x3w_kernel" in k)          # both tilings of the x3 fp32 GEMM (gemm_f32.hip / gemm_x3w.hip)
               else B16 if ("gemm_b16r_kernel" in k or "gemm_b16w_kernel" in k)
               else (B16 if k.rstrip().endswith("1>(pulse::XpArgs)") or ", 1>" in k else "gemm_x3p_kernel") if "gemm_x3p_kernel" in k
               else ("other_pulse" if "pulse" in k else "torch"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # [r6] the two tilings of the x3 GEMM also get rows of their own (they are different kernels with different traffic per launch)
        if "gemm_x3w_kernel" in k:
            agg["gemm_x3w_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "gemm_x3_kernel" in k:
            agg["gemm_x3_kernel (128 x 128 / 64 x 128 tile only)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for key, cs in agg.items():
    f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
    if not f or not w:
        continue
    fetch = 2.0 * 1024.0 * sum(f) / len(f)          # KiB -> B, x2 gfx950 correction
    write = 1024.0 * sum(w) / len(w)
    res[key] = {"launches_fetch_pass": len(f), "launches_write_pass": len(w), "fetch_bytes_per_launch": fetch,
                "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    hit, miss = cs.get("TCC_HIT_sum", []), cs.get("TCC_MISS_sum", [])
    if hit and miss:          # L2 hit rate (MI355X_MICROARCH.md, L2 section): the fabric-side FETCH_SIZE above counts L2 MISSES, Infinity-Cache hits included
        res[key]["l2_hit_rate"] = sum(hit) / max(1.0, sum(hit) + sum(miss))
        res[key]["l2_requests_per_launch"] = (sum(hit) + sum(miss)) / len(hit)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
