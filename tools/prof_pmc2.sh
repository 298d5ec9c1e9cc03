#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass: MI355X_MICROARCH.md, PMC slots) of one command, summarised per kernel.  (GPU box.)
#   bash tools/prof_pmc2.sh <out dir under the repo> [workload label] [command ...]
# Default command: bench.py's C2 on a caller's stream (ONE launch of asdr_update_kernel per step, so that the per-launch medians are per-step
# figures; on ASDR_STREAM_BATCH a step is two half-size launches in flight together); the C3 / C4 passes use tools/bench_configs.py c3 / c4.
# Never combined with a trace domain other than --kernel-trace (gpurun refuses that), program named directly after `--`.
set -u
OUT=${1:-gpurun_out/pmc2}
LABEL=${2:-"bench.py C2, 65536 channels x 1 block per launch"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
# the command stays an array, so a repository path with spaces survives; the program itself follows `--` (python3 <script>)
if [ $# -gt 0 ]; then CMD=("$@"); else CMD=(python3 "$ROOT/bench.py" --steps 40 --warmup 10 --settle 0 --no-cpu-baseline --no-robustness --no-host-path --no-configs --caller-stream); fi
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
p() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$n" -- "${CMD[@]}" > "$ROOT/$OUT/$n.log" 2>&1; }
p a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
p b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
p c SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM
p e GRBM_GUI_ACTIVE
p f FETCH_SIZE
p g WRITE_SIZE
python3 "$ROOT/tools/pmc_to_json.py" "$ROOT/$OUT" "$ROOT/$OUT.json" "$LABEL" > /dev/null
cat "$ROOT/$OUT.json"
