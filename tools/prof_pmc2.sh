#!/bin/bash
# Wider SQ counter passes for asdr_update_kernel on the C2 workload (GPU box): where a wave's cycles go.
set -u
OUT=${1:-gpurun_out/pmc2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
p() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/$n" -- $CMD > "$ROOT/$OUT/$n.log" 2>&1; }
p a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
p b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
p c SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM
p d SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES
p e GRBM_GUI_ACTIVE
p f FETCH_SIZE
p g WRITE_SIZE
python3 $ROOT/tools/pmc_to_json.py "$ROOT/$OUT" "$ROOT/$OUT.json" > /dev/null
cat "$ROOT/$OUT.json"
