#!/bin/bash
# GPU box: interleaved A/B of library variants (audiosdr_amd/variants/libasdr_<name>.so; "intree" = the in-tree build), one process per run
# (tools/_variant.py): bench.py's driver command (--steps 20 --warmup 5: a fresh bank) and its default steady-state run.
#   tools/r5_ab.sh <outdir> <rounds> <name> [<name> ...]
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    # <name>[@VAR=VALUE]: an environment setting for the run (e.g. intree@ASDR_MW=0)
    lib=${v%%@*}; envset=""; if [ "$lib" != "$v" ]; then envset=${v#*@}; fi
    if [ "$lib" = intree ]; then unset ASDR_TOOLS_LIB; else export ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_$lib.so; fi
    if [ -n "$envset" ]; then export "$envset"; fi
    python3 tools/bench_variant.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-robustness --no-configs > "$out/${v}_drv_$r.json" 2> "$out/${v}_drv_$r.err"
    python3 tools/bench_variant.py --no-cpu-baseline --no-host-path --no-configs ${AB_EXTRA} > "$out/${v}_steady_$r.json" 2> "$out/${v}_steady_$r.err"
    if [ -n "$envset" ]; then unset "${envset%%=*}"; fi
  done
done
python3 - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print("unreadable", f, e); continue
    name = os.path.basename(f)[:-5]
    v, kind, r = name.rsplit("_", 2)
    rf = d.get("roofline", {})
    rb = d.get("robustness", {}) or {}
    rows.setdefault((v, kind), []).append((rf.get("kernel_ms"), (rf.get("steady_state") or {}).get("ms_per_step"), (rf.get("caller_stream_ordered") or {}).get("kernel_ms"),
                                           [x.get("kernel_ms") for x in rb.values()] if kind == "steady" else None))
for k in sorted(rows):
    print(k, rows[k])
PY
