#!/bin/bash
# One GPU-box call that produces everything profiles/ needs for the current build (copy the results from gpurun_out/<tag>/):
#   bench.json            python bench.py (default flags: the driver's N=1 command)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (default flags, CPU baseline leg skipped; kernel rows only)
#   pmc.json              rocprofv3 --pmc passes of the same command (tools/prof_pmc2.sh; HBM traffic = 2*FETCH_SIZE + WRITE_SIZE)
#   configs_1gpu.jsonl    tools/bench_configs.py: BASELINE configs 1-5 on one GPU with their parity checks
#   c3_kernel_stats.csv, c4_kernel_stats.csv   rocprofv3 --kernel-trace --stats of tools/bench_configs.py c3 / c4 (which launches, how long)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/stats.log" 2>&1 )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
bash tools/prof_pmc2.sh "gpurun_out/$TAG/pmc" > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc.json" "$OUT/pmc_final.json" 2>/dev/null
python3 tools/bench_configs.py c1 c2 c2s c3 c4 c5 > "$OUT/configs_1gpu.jsonl" 2> "$OUT/configs.err"
for c in c3 c4; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$c" -- python3 "$ROOT/tools/bench_configs.py" $c > "$OUT/stats_$c.log" 2>&1 )
  f=$(find "$OUT/stats_$c" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${c}_kernel_stats.csv"
done
cat "$OUT/bench.json"; head -5 "$OUT/kernel_stats.csv"; cat "$OUT/configs_1gpu.jsonl"
