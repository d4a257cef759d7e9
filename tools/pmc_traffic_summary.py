"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) into profiles/-sized JSON.
    python tools/pmc_traffic_summary.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys


def load(d, cname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != cname:
                continue
            n = r["Kernel_Name"]
            for key in ("gemm256p_kernel", "gemm256_kernel", "gemm_kernel", "attn_kernel", "skinny_lds_kernel", "skinny_kernel", "splitk_reduce", "norm_wave_fixed"):
                if key in n:
                    agg[key][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
                    break
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, over "
                 "`python bench.py --batch 28 --no-pipeline --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-prof --no-aux` (28 frames/step). "
                 "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies "
                 "128-B requests at 64 B for wide coalesced reads; cross-check in this very run: the weight-streaming dual skinny "
                 "kernel reads 235 MB algorithmic and 2*FETCH_SIZE agrees within 2 %). FETCH counts L2 misses to the fabric "
                 "(Infinity-Cache hits included); WRITE_SIZE is taken as is.", "kernels": {}}
for k in f:
    tf = sum(sum(v) for v in f[k].values()); nf = sum(len(v) for v in f[k].values())
    tw = sum(sum(v) for v in w[k].values()); nw = max(1, sum(len(v) for v in w[k].values()))
    out["kernels"][k] = {"launches": nf, "fetch_size_kb_avg": tf / nf, "write_size_kb_avg": tw / nw,
                         "hbm_bytes_per_launch": (2 * tf / nf + tw / nw) * 1024,
                         "by_grid": {str(g): {"n": len(v), "fetch_size_kb": sum(v) / len(v),
                                              "write_size_kb": (sum(w[k][g]) / len(w[k][g])) if w[k].get(g) else None}
                                     for g, v in sorted(f[k].items())}}
if "gemm256_kernel" in out["kernels"]:
    # the dominant kernel = the 256x256 tiled GEMM in both forms (one tile per block; persistent tile walk): launch-weighted average
    ks = [out["kernels"][k] for k in ("gemm256_kernel", "gemm256p_kernel") if k in out["kernels"]]
    n = sum(k["launches"] for k in ks)
    out["hbm_bytes_per_launch"] = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / n
    out["kernel"] = "gemm256_kernel + gemm256p_kernel (launch-weighted)"
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out["kernels"].items()}))
