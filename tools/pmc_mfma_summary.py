"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA):
    python tools/pmc_mfma_summary.py <pmc_dir> <out.json>
MfmaUtil follows rocprofiler's multi-XCC formula 100 * XCC_NUM * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CU_NUM * 4)
(GRBM_GUI_ACTIVE comes back summed over the 8 XCCs: 1.49 M for a 90 us launch; SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA for
v_mfma_f32_16x16x32_bf16, summed over all SIMDs).  The raw per-launch averages are kept next to it (ROCm 7.2 ships no gfx950
derived-counter section, MI355X_MICROARCH.md).  It is a fraction of the cycles the chip actually ran (~1.8-1.9 GHz under this
load), not of the 2.4 GHz the 2.5 PFLOP/s peak is quoted at."""
import csv, glob, json, sys, collections
CU_NUM, XCC_NUM = 256, 8
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"][0] == 0:
        continue
    avg = {n: v[0] / v[1] for n, v in c.items()}
    e = {"launches": c["SQ_VALU_MFMA_BUSY_CYCLES"][1], "avg": {n: round(v, 1) for n, v in avg.items()}}
    if avg.get("GRBM_GUI_ACTIVE"):
        e["MfmaUtil_percent"] = round(100 * XCC_NUM * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] * CU_NUM * 4), 2)
    if avg.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_over_cu_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CU_CYCLES"], 4)
    out[k] = e
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -- python bench.py --steps 3 --warmup 1 "
                     "--no-cpu-baseline --no-decode --no-prof --no-aux (28 frames/step); per-launch averages",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1]["avg"]["SQ_VALU_MFMA_BUSY_CYCLES"] * kv[1]["launches"])[:8]:
    print(k[:60], e.get("MfmaUtil_percent"), e.get("mfma_busy_over_cu_busy"), e["launches"])
