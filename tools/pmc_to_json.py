#!/usr/bin/env python3
"""Turn the rocprofv3 PMC CSVs of tools/prof_pmc.sh into profiles/<name>.json (per-launch medians for
asdr_update_kernel) including the HBM traffic figure bench.py reports as roofline.traffic:
  traffic = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024   [bytes per launch]
(MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream; WRITE_SIZE is exact
for 16-B-per-lane stores; both are in KiB.)"""
import csv, glob, json, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Kernel_Name", "").startswith("asdr_update_kernel"):
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, name), v in per.items():
        acc[name].append(v)
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
import hashlib, os
_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiosdr_amd", "libasdr_hip.so")
_sha = hashlib.sha256(open(_lib, "rb").read()).hexdigest() if os.path.exists(_lib) else None
import sys as _sys
_sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from audiosdr_amd import build as _build
res = {"library_sha256": _sha, "source_sha256": _build.source_sha256(), "kernel": "asdr_update_kernel", "workload": "bench.py C2, 65536 channels x 1 block per launch", "counters_median_per_launch": med}
if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
    res["hbm_read_bytes"] = 2 * med["FETCH_SIZE"] * 1024
    res["hbm_write_bytes"] = med["WRITE_SIZE"] * 1024
    res["traffic_bytes_per_launch"] = res["hbm_read_bytes"] + res["hbm_write_bytes"]
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, sort_keys=True)[:600])
