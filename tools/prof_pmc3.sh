#!/bin/bash
# VALU instruction mix of asdr_update_kernel on the C2 workload (GPU box).
set -u
OUT=${1:-gpurun_out/pmc3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
# (caller's stream: ONE launch of the update kernel per step, so that per-launch medians are per-step figures)
CMD="python3 $ROOT/bench.py --steps 40 --warmup 10 --settle 0 --no-cpu-baseline --no-robustness --no-host-path --no-configs --caller-stream"
p() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$n" -- $CMD > "$ROOT/$OUT/$n.log" 2>&1; }
p a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64
p b SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
python3 $ROOT/tools/pmc_to_json.py "$ROOT/$OUT" "$ROOT/$OUT.json" > /dev/null
cat "$ROOT/$OUT.json"
