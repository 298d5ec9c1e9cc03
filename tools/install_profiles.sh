#!/bin/bash
# After `bash tools/collect_profiles.sh <tag>` on the GPU box: copy gpurun_out/<tag>/ into profiles/ under the committed names and rebuild
# pmc_latest.json / issue_latest.json (the bench line attaches them only while their source sha matches the tree).   bash tools/install_profiles.sh r05
set -e
T=${1:-r05}; S=gpurun_out/$T
cp $S/bench.json profiles/${T}_c2_bench.json
cp $S/bench_driver.json profiles/${T}_c2_bench_driver_cmd.json
cp $S/kernel_stats.csv profiles/${T}_c2_kernel_stats.csv
cp $S/kernel_stats_caller_stream.csv profiles/${T}_c2_kernel_stats_caller_stream.csv
cp $S/pmc.json profiles/${T}_c2_pmc.json; cp $S/pmc.json profiles/pmc_latest.json
cp $S/pmc_mix.json profiles/${T}_c2_pmc_mix.json
cp $S/c3_kernel_stats.csv profiles/${T}_c3_kernel_stats.csv; cp $S/c4_kernel_stats.csv profiles/${T}_c4_kernel_stats.csv
cp $S/configs_1gpu.jsonl profiles/${T}_configs_1gpu.jsonl
cp $S/front_bench_16blk.jsonl profiles/${T}_front_bench_16blk.jsonl; cp $S/front_bench_1blk.jsonl profiles/${T}_front_bench_1blk.jsonl
cp $S/pmc_c3.json profiles/${T}_pmc_c3.json; cp $S/pmc_c4.json profiles/${T}_pmc_c4.json
cp $S/bench_c4.json profiles/${T}_bench_c4.json; cp $S/bench_c5.json profiles/${T}_bench_c5.json; cp $S/bench_c5_share.json profiles/${T}_bench_c5_share.json
cp $S/bench_single_process_2shards_1gpu.json profiles/${T}_bench_single_process_2shards_1gpu.json
[ -f $S/mw_timeline.txt ] && grep -v "^/opt" $S/mw_timeline.txt > profiles/${T}_mw_timeline.txt
[ -f $S/roofline_latency.json ] && cp $S/roofline_latency.json profiles/${T}_roofline_latency.json && cp $S/roofline_latency.json profiles/roofline_latency_latest.json
[ -f $S/small_batch_stream.jsonl ] && cp $S/small_batch_stream.jsonl profiles/${T}_small_batch_stream.jsonl
python3 tools/issue_model.py profiles/${T}_c2_pmc.json profiles/${T}_c2_pmc_mix.json profiles/${T}_c2_kernel_stats_caller_stream.csv profiles/issue_latest.json > /dev/null
python3 - "$T" <<'PY'
import json, sys
T = sys.argv[1]
for f in ["profiles/%s_c2_bench.json" % T, "profiles/%s_c2_bench_driver_cmd.json" % T]:
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "kernel", r.get("kernel_ms"), "steady", (r.get("steady_state") or {}).get("ms_per_step"), "caller", (r.get("caller_stream_ordered") or {}).get("kernel_ms"))
    print("   robustness", {k: (v.get("kernel_ms"), v.get("frac")) for k, v in (d.get("robustness") or {}).items()})
    print("   configs", {k: (v.get("ms_per_step"), v.get("frac")) for k, v in (d.get("configs") or {}).items()})
    print("   h2d", (d.get("h2d_d2h_inclusive") or {}).get("ms_per_call"), (d.get("h2d_d2h_inclusive") or {}).get("frac_of_63GBps"))
i = json.load(open("profiles/issue_latest.json"))
print("issue", {k: i[k] for k in ("valu_instructions_per_wave", "valu_cycles_per_wave", "valu_floor_ms", "valu_busy_frac", "wave_lifetime_cycles_in_the_full_launch", "wave_cycles_issuing_frac", "wave_cycles_issue_stalled_frac", "wave_cycles_parked_at_a_wait_frac", "kernel_avg_ms_rocprofv3", "source_sha256")})
p = json.load(open("profiles/pmc_latest.json")); print("traffic", p.get("traffic_bytes_per_launch"), p.get("hbm_read_bytes"), p.get("hbm_write_bytes"))
PY
