#!/bin/bash
# PMC passes for asdr_update_kernel on the C2 workload (GPU box).  Separate passes per counter group
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8 slots; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$ROOT/$OUT/sq1" -- $CMD > "$ROOT/$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$ROOT/$OUT/sq2" -- $CMD > "$ROOT/$OUT/sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/fetch" -- $CMD > "$ROOT/$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/write" -- $CMD > "$ROOT/$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/grbm" -- $CMD > "$ROOT/$OUT/grbm.log" 2>&1
find "$ROOT/$OUT" -name "*counter_collection.csv" | head
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$ROOT/$OUT/tcc" -- $CMD > "$ROOT/$OUT/tcc.log" 2>&1
