#!/bin/bash
# GPU box: interleaved A/B of library variants on the OTHER workloads (one process per run): tools/bench_configs.py c3 / c4 and
# bench.py --single-process (two shards on one GPU).   tools/r5_ab_cfg.sh <outdir> <rounds> <name>[@VAR=VALUE] ...
out=$1; rounds=$2; shift 2
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    lib=${v%%@*}; envset=""; if [ "$lib" != "$v" ]; then envset=${v#*@}; fi
    if [ "$lib" = intree ]; then unset ASDR_TOOLS_LIB; else export ASDR_TOOLS_LIB=audiosdr_amd/variants/libasdr_$lib.so; fi
    if [ -n "$envset" ]; then export "$envset"; fi
    python3 tools/bench_configs.py c3 c4 > "$out/${v}_cfg_$r.jsonl" 2> "$out/${v}_cfg_$r.err"
    python3 tools/bench_variant.py --single-process --gpus 2 --devices 0,0 --channels 32768 --steps 500 > "$out/${v}_sp_$r.json" 2> "$out/${v}_sp_$r.err"
    python3 tools/bench_variant.py --config c4 --no-cpu-baseline --steps 200 > "$out/${v}_c4big_$r.json" 2> "$out/${v}_c4big_$r.err"
    if [ -n "$envset" ]; then unset "${envset%%=*}"; fi
  done
done
python3 - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*"))):
    if f.endswith(".err"): continue
    for l in open(f):
        l = l.strip()
        if not l.startswith("{"): continue
        d = json.loads(l)
        if "config" in d and isinstance(d["config"], str):
            print(os.path.basename(f), d["config"][:28], d.get("steady_ms_per_launch"), d.get("steady_ms_per_launch_batch_stream"), d.get("parity"))
        else:
            print(os.path.basename(f), d.get("ms_per_step"), d.get("value"))
PY
