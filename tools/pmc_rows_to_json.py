"""FETCH_SIZE / WRITE_SIZE passes of tools/prof_rows.py -> profiles/<name>.json.
    python tools/pmc_rows_to_json.py gpurun_out/prof_X profiles/r02_rows_pmc.json"""
import csv, glob, json, sys

def per_launch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if "gemm_rows_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
    return out

root, dst = sys.argv[1], sys.argv[2]
fe, wr = per_launch(root + "/pmc_rows_fetch", "FETCH_SIZE"), per_launch(root + "/pmc_rows_write", "WRITE_SIZE")
V, D = 151936, 4096
res = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/prof_rows.py (separate passes)",
       "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KiB", "kernels": {}}
for k in fe:
    M = 32 if "<2," in k else 128
    f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
    alg = V * D * 2 + M * D * 2 + M * V * 2
    x_reread = (V // 128 + 1) * M * D * 2
    res["kernels"][k] = {"rows": M, "launches": len(fe[k]), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                         "traffic_bytes_per_launch": (2 * f + w) * 1024, "algorithmic_bytes_per_launch": alg,
                         "traffic_over_algorithmic": (2 * f + w) * 1024 / alg,
                         "x_bytes_reread_by_the_column_tiles (L2 side, not HBM)": x_reread}
json.dump(res, open(dst, "w"), indent=1)
print(open(dst).read())
