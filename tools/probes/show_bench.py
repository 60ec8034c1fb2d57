"""Print the headline numbers of a bench.py JSON line (file argument): value, ms/step, per-kernel ms and TFLOP/s."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'{d["config"]["workload"]}: {d["value"]:.3f} {d["unit"]}, {d["ms_per_step"]:.1f} ms/step; dominant {r["kernel"]} '
      f'{r["achieved"]:.1f} {r["unit"]} frac {r["frac"]:.3f}')
for k, v in sorted(r.get("kernels", {}).items(), key=lambda kv: -kv[1]["ms"]):
    print(f'  {v["ms"]:8.2f} ms  {v["tflops"]:7.1f} TF  x{v["launches"]:<4d} {k}')
