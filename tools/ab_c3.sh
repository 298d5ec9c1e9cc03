#!/bin/bash
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    lib=${v%%@*}
    if [ "$lib" = intree ]; then unset ASDR_TOOLS_LIB; else export ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_$lib.so; fi
    python3 tools/bench_configs.py c3 > "$out/${v}_cfg_$r.jsonl" 2> "$out/${v}_cfg_$r.err"
  done
done
python3 - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.jsonl"))):
    for l in open(f):
        l = l.strip()
        if not l.startswith("{"): continue
        d = json.loads(l)
        print(os.path.basename(f), str(d.get("config"))[:20], d.get("steady_ms_per_launch"), d.get("steady_ms_per_launch_batch_stream"), d.get("parity"))
PY
