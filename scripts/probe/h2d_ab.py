"""Diagnostic (GPU box): the h2d_per_launch workload of bench.py alone, for A/B of runtime settings (HSA_ENABLE_SDMA ...)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
args = bench.parse_args()
b = bench.Bench(args)
spl = int(os.environ.get("SPL", 16))
out = bench.h2d_point(b, 64, 6, 1600, spl)
tr = b.make_trainer(64, 6, 64 * 7, spl, None, None)
b.fill_slots(tr, 64, 6)
sec, reps = b.timed(tr, 1600, 160)
print(json.dumps({"h2d_steps_per_sec": out["steps_per_sec"], "h2d_us": out["ms_per_step"] * 1e3,
                  "resident_steps_per_sec": round(1600 / sec, 1), "sdma": os.environ.get("HSA_ENABLE_SDMA", "default"), "spl": spl}))
