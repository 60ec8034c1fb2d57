"""BUILD-CONTAINER ONLY (reads /root/reference): the ``model=dict(type=..., ...)`` section of every model config the
reference ships under configs/patchrefiner*/ and configs/baseline*/ (``_base_`` inheritance resolved), dumped as data to
tests/golden/reference_model_configs.json.  tests/test_host_logic.py builds each of them through the product's registry and
reports how many construct -- the drop-in check of SURVEY.md 8(b) "Registry" for the whole config tree."""
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from patchrefinerv2_amd.registry import Config  # noqa: E402

REF = "/root/reference/configs"
out = {}
for path in sorted(glob.glob(os.path.join(REF, "*", "*.py"))):
    rel = os.path.relpath(path, REF)
    if rel.startswith("_base_"):
        continue
    try:
        cfg = Config.fromfile(path)
    except Exception as e:  # noqa: BLE001
        print("skip (does not load):", rel, type(e).__name__, e)
        continue
    if "model" not in cfg:
        continue
    out[rel] = cfg["model"].to_dict()
dst = os.path.join(REPO, "tests", "golden", "reference_model_configs.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
types = {}
for v in out.values():
    types[v.get("type")] = types.get(v.get("type"), 0) + 1
print(f"wrote {dst}: {len(out)} model configs", types, f"{os.path.getsize(dst) >> 10} KiB")
