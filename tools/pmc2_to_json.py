"""Turn the PMC passes of tools/gpu_profile2.sh into profiles/<round>_loss_pmc.json (HBM traffic of the fused loss
kernels, the file bench.py's train.loss_roofline.traffic reads) and profiles/<round>_gemm_mfma_pmc.json (MFMA pipe
occupancy of the training lm_head GEMM).
    python tools/pmc2_to_json.py gpurun_out/prof_r01v9 profiles/r01
"""
import csv
import glob
import json
import sys


def counter(d, kernel, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)


root, out = sys.argv[1], sys.argv[2]
N, V, D = 2048, 151936, 4096
# gfx950: FETCH_SIZE counts a 128-B request as 64 B for wide coalesced reads -> x2; WRITE_SIZE exact; unit KiB
loss = {"shape": {"N": N, "V": V, "D": D},
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/prof_loss.py (separate passes)",
        "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; unit KiB"}
tot_t = tot_a = 0.0
for k, alg in (("ce_partial_kernel", 2.0 * N * V * 2 + 2.0 * N * D * 2), ("ce_combine_kernel", 0.0),
               ("ce_bwd_kernel", 3.0 * N * V * 2 + 3.0 * N * D * 2)):
    fe, n = counter(root + "/pmc_loss_fetch", k, "FETCH_SIZE")
    wr, _ = counter(root + "/pmc_loss_write", k, "WRITE_SIZE")
    t = (fe * 2 + wr) * 1024
    loss[k] = {"launches": n, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "traffic_bytes_per_launch": t,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (t / alg) if alg else None}
    tot_t += t
    tot_a += alg
loss["traffic_bytes_fwd_plus_bwd"] = tot_t
loss["algorithmic_bytes_fwd_plus_bwd"] = tot_a
loss["traffic_over_algorithmic"] = tot_t / tot_a
json.dump(loss, open(out + "_loss_pmc.json", "w"), indent=1)

M, Ng, K = 4096, 151936, 4096
busy, n = counter(root + "/pmc_gemm", "gemm_nt8_kernel", "SQ_VALU_MFMA_BUSY_CYCLES")
gui, _ = counter(root + "/pmc_gemm", "gemm_nt8_kernel", "GRBM_GUI_ACTIVE")
f = glob.glob(root + "/trace_gemm/**/*kernel_trace.csv", recursive=True)[0]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9 for r in csv.DictReader(open(f)) if "gemm_nt8_kernel" in r["Kernel_Name"]]
sec = sum(dur) / len(dur)
cyc = gui / 8                                  # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs
flops = 2.0 * M * Ng * K
json.dump({"kernel": "gemm_nt8_kernel<pingpong> (training lm_head, %d x %d x %d bf16)" % (M, Ng, K),
           "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/prof_gemm.py; durations from a separate --kernel-trace run",
           "launches": n, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_xcds": gui,
           "mfma_instructions_16x16x32": flops / 16384, "busy_cycles_per_mfma": busy / (flops / 16384),
           "kernel_cycles_per_xcd": cyc, "avg_duration_s_traced": sec, "effective_clock_GHz": cyc / sec / 1e9,
           "mfma_pipe_occupancy": busy / (1024 * cyc),
           "note": "occupancy = busy cycles / (1024 SIMDs x kernel cycles); TFLOP/s = occupancy x 1024 SIMDs x 1024 flop/cycle x clock",
           "TFLOPs_traced": flops / sec / 1e12}, open(out + "_gemm_mfma_pmc.json", "w"), indent=1)
print(open(out + "_loss_pmc.json").read())
print(open(out + "_gemm_mfma_pmc.json").read())
