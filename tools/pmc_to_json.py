"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter) into the per-kernel JSON that
bench.py's roofline.traffic reads:  python tools/pmc_to_json.py out.json <dir-or-csv> [<dir-or-csv> ...]

Per kernel: launches and the per-launch average of every counter found (FETCH_SIZE / WRITE_SIZE are in KB; the
gfx950 x2 correction for FETCH_SIZE is applied by the reader, bench.py:pmc_traffic, not here)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    files = []
    for s in srcs:
        files += [s] if os.path.isfile(s) else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0]
            c = row["Counter_Name"]
            tot[k][c] += float(row["Counter_Value"])
            disp[k][c].add((f, row["Dispatch_Id"]))
    res = {}
    for k in tot:
        e = {"launches": max(len(v) for v in disp[k].values())}
        for c, v in tot[k].items():
            e[f"{c}_KB_per_launch" if c.endswith("_SIZE") else f"{c}_per_launch"] = v / len(disp[k][c])
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(files)} csv -> {out}: {len(res)} kernels")


if __name__ == "__main__":
    main()
