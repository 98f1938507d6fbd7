"""bench.py's decode leg under a weight-stream kernel variant: python tools/bench_variant.py <variant int> [bench args]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eagle_amd import _hip
v = int(sys.argv[1], 0)
_hip.lib().eagle_debug_set_skinny_variant(v, 0, 0)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import json, io, contextlib
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"variant {v:#x}: value {d['value']:.1f} steps/s, ms/cycle {d['ms_per_step']:.4f}, lm_head us {d['roofline']['us_per_launch']:.2f} frac {d['roofline']['frac']:.3f}", flush=True)
