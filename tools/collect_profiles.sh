#!/bin/bash
# One GPU-box call that produces everything profiles/ needs for the current build (copy the results from gpurun_out/<tag>/):
#   bench.json              python bench.py (default flags: the driver's N=1 command; headline + robustness + cpu_baseline)
#   bench_driver.json       python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's exact command line)
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-robustness --no-host-path` (the headline: on
#                           ASDR_STREAM_BATCH every step is TWO half-size launches of asdr_update_kernel, one per lane, in flight together)
#   kernel_stats_caller_stream.csv   the same with --caller-stream (strict stream order: one launch per step)
#   pmc.json                rocprofv3 --pmc passes of the C2 command (tools/prof_pmc2.sh; HBM traffic = 2*FETCH_SIZE + WRITE_SIZE)
#   pmc_c3.json, pmc_c4.json   the same passes of tools/bench_configs.py c3 / c4, per kernel
#   configs_1gpu.jsonl      tools/bench_configs.py: BASELINE configs 1-5 on one GPU with their parity checks
#   bench_c4.json, bench_c5.json   python bench.py --config c4 / c5 (N = 1: the whole job on one GPU)
#   c3_kernel_stats.csv, c4_kernel_stats.csv   rocprofv3 --kernel-trace --stats of tools/bench_configs.py c3 / c4
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-robustness --no-host-path --no-configs > "$OUT/stats.log" 2>&1 )
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_cs" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-robustness --no-host-path --no-configs --caller-stream > "$OUT/stats_cs.log" 2>&1 )
f=$(find "$OUT/stats_cs" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_caller_stream.csv"
bash tools/prof_pmc2.sh "gpurun_out/$TAG/pmc" > "$OUT/pmc.log" 2>&1
bash tools/prof_pmc3.sh "gpurun_out/$TAG/pmc_mix" > "$OUT/pmc_mix.log" 2>&1
export BENCH_CONFIGS_NO_LANES=1   # the counter passes and kernel traces of C3 / C4: strict stream order (one launch form per kernel name)
bash tools/prof_pmc2.sh "gpurun_out/$TAG/pmc_c3" "tools/bench_configs.py c3: SAM, 262144 channels x 1 block per launch" python3 "$ROOT/tools/bench_configs.py" c3 > "$OUT/pmc_c3.log" 2>&1
bash tools/prof_pmc2.sh "gpurun_out/$TAG/pmc_c4" "tools/bench_configs.py c4: mixed modes + ALS, 131072 channels x 1 block per launch" python3 "$ROOT/tools/bench_configs.py" c4 > "$OUT/pmc_c4.log" 2>&1
unset BENCH_CONFIGS_NO_LANES
python3 tools/bench_configs.py c1 c2 c2s c3 c4 c5 > "$OUT/configs_1gpu.jsonl" 2> "$OUT/configs.err"
python3 bench.py --config c4 --no-cpu-baseline --steps 300 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config c5 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
python3 bench.py --config c5 --channels 512 --no-cpu-baseline --steps 64 > "$OUT/bench_c5_share.json" 2> "$OUT/bench_c5_share.err"
python3 tools/bench_front.py 65536 1 > "$OUT/front_bench_1blk.jsonl" 2> "$OUT/front.err"
python3 tools/bench_front.py 65536 16 > "$OUT/front_bench_16blk.jsonl" 2>> "$OUT/front.err"
python3 bench.py --single-process --gpus 2 --devices 0,0 --channels 32768 --steps 500 > "$OUT/bench_single_process_2shards_1gpu.json" 2> "$OUT/bench_sp.err"
export BENCH_CONFIGS_NO_LANES=1
for c in c3 c4; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$c" -- python3 "$ROOT/tools/bench_configs.py" $c > "$OUT/stats_$c.log" 2>&1 )
  f=$(find "$OUT/stats_$c" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${c}_kernel_stats.csv"
done
# round 6: the four-wave kernel's phase timeline (profiling build: audiosdr_amd/variants/libasdr_timeline.so, `python tools/latency_model.py build` here),
# inside the full launch (steady and fresh bank) and as a lone workgroup, the per-mix issue intervals at 1..4 waves per SIMD, and the latency model made of them
python3 tools/timeline.py mw > "$OUT/mw_timeline.txt" 2> "$OUT/mw_timeline.err"
ASDR_MW_MIN_WAVES=1 python3 tools/timeline.py mw 300 32 >> "$OUT/mw_timeline.txt" 2>> "$OUT/mw_timeline.err"
python3 tools/latency_model.py run "$OUT/roofline_latency.json" > /dev/null 2> "$OUT/roofline_latency.err"
python3 tools/small_batch_stream.py 128 > "$OUT/small_batch_stream.jsonl" 2> "$OUT/small_batch_stream.err"
rm -rf "$OUT/stats" "$OUT/stats_cs" "$OUT/stats_c3" "$OUT/stats_c4" "$OUT"/pmc/*/ "$OUT"/pmc_mix/*/ "$OUT"/pmc_c3/*/ "$OUT"/pmc_c4/*/
cat "$OUT/bench.json"; head -5 "$OUT/kernel_stats.csv"; cat "$OUT/configs_1gpu.jsonl"; cat "$OUT/bench_c4.json" "$OUT/bench_c5.json" "$OUT/bench_c5_share.json"
