#!/bin/bash
# Instruction-cache / fetch counters of asdr_update_kernel on the C2 workload (GPU box), one rocprofv3 --pmc pass.
set -u
OUT=${1:-gpurun_out/pmc_icache}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/ic" -- $CMD > "$ROOT/$OUT/ic.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d "$ROOT/$OUT/ic2" -- $CMD > "$ROOT/$OUT/ic2.log" 2>&1
find "$ROOT/$OUT" -name "*counter_collection.csv" | head
