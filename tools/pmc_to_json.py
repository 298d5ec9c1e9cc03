#!/usr/bin/env python3
"""Turn the rocprofv3 PMC CSVs of tools/prof_pmc2.sh into profiles/<name>.json: per-launch medians of every counter, per kernel
of the library (asdr_*), and for the dominant kernel (most wave cycles / launches) the HBM traffic figure bench.py reports as
roofline.traffic:   traffic = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024   [bytes per launch]
(MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream; WRITE_SIZE is exact
for 16-B-per-lane stores; both are in KiB.)      python3 tools/pmc_to_json.py <dir> <out.json> [workload label]"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else "bench.py C2, 65536 channels x 1 block per launch"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if k.startswith("asdr_") and "reset" not in k and "spin" not in k:
                per[(k.split("(")[0], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, d, name), v in per.items():
        acc[k][name].append(v)
kern = {}
for k, cs in acc.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in cs.items()}
    med["launches_seen"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        med["hbm_read_bytes"] = 2 * med["FETCH_SIZE"] * 1024
        med["hbm_write_bytes"] = med["WRITE_SIZE"] * 1024
        med["traffic_bytes_per_launch"] = med["hbm_read_bytes"] + med["hbm_write_bytes"]
    kern[k] = med
_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiosdr_amd", "libasdr_hip.so")
_sha = hashlib.sha256(open(_lib, "rb").read()).hexdigest() if os.path.exists(_lib) else None
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audiosdr_amd import build as _build
main = max(kern, key=lambda k: kern[k].get("SQ_WAVE_CYCLES", 0) * kern[k]["launches_seen"]) if kern else None
res = {"library_sha256": _sha, "source_sha256": _build.source_sha256(), "workload": label, "kernel": main, "kernels": kern}
if main:
    res["counters_median_per_launch"] = {k: v for k, v in kern[main].items()}
    for k in ("hbm_read_bytes", "hbm_write_bytes", "traffic_bytes_per_launch"):
        if k in kern[main]:
            res[k] = kern[main][k]
    res["traffic_bytes_per_call_all_kernels"] = sum(v.get("traffic_bytes_per_launch", 0) for v in kern.values())
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, sort_keys=True)[:600])
