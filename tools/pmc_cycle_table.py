"""HBM traffic per launch of every kernel of the batch-32 cycle, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/batched_only.py: per (kernel, grid) the average over the steady-state launches.
    python tools/pmc_cycle_table.py gpurun_out/prof_X/pmc_b32_fetch gpurun_out/prof_X/pmc_b32_write profiles/r02_batch32_pmc.json"""
import collections, csv, glob, json, sys

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"][:64], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))].append(float(r["Counter_Value"]))
    return agg

fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fe:
    v = fe[k][len(fe[k]) // 3:]                      # drop the warm-up third
    w = wr.get(k, [0.0]); w = w[len(w) // 3:]
    f_kb, w_kb = sum(v) / len(v), sum(w) / max(1, len(w))
    rows.append({"kernel": k[0], "workgroups": k[1], "launches": len(v), "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
                 "hbm_read_MB (FETCH x2, gfx950 correction)": 2 * f_kb * 1024 / 1e6, "hbm_write_MB": w_kb * 1024 / 1e6})
rows.sort(key=lambda r: -r["hbm_read_MB (FETCH x2, gfx950 correction)"] * r["launches"])
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/batched_only.py (separate passes)",
           "kernels": rows[:24]}, open(sys.argv[3], "w"), indent=1)
for r in rows[:16]:
    print(f"{r['hbm_read_MB (FETCH x2, gfx950 correction)']:9.1f} MB read {r['hbm_write_MB']:8.1f} MB written  x{r['launches']:4d}  wg={r['workgroups']:6d}  {r['kernel'][:60]}")
