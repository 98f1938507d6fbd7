"""Copy the artifacts of tools/gpu_profile_r02.sh from gpurun_out/prof_<tag>/ into profiles/r02_*.
    python tools/collect_profiles.py <tag>"""
import os, shutil, sys
src = os.path.join("gpurun_out", "prof_" + sys.argv[1])
for a, b in (("bench.json", "r02_bench.json"), ("bench_kernel_stats.csv", "r02_bench_kernel_stats.csv"),
             ("decode_cycle_timeline.txt", "r02_decode_cycle_timeline.txt"), ("batch32_cycle_timeline.txt", "r02_batch32_cycle_timeline.txt"),
             ("train_step_timeline.txt", "r02_train_step_timeline.txt"), ("lm_head_trace.json", "r02_lm_head_trace.json"),
             ("lm_head_pmc.json", "r02_lm_head_pmc.json")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", b))
        print("profiles/" + b)
    else:
        print("missing", p)
