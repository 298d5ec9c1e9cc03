#!/bin/bash
# GPU box: the LDS counters of asdr_update_kernel for the phase-ablation builds (tools/ablate.py build: audiosdr_amd/variants/libasdr_no_*.so,
# full, io_only): which phase the bank conflicts belong to (full minus no_<phase>).  One rocprofv3 --pmc pass per build, kernel trace only.
#   bash tools/pmc_lds_by_phase.sh [out dir under the repo] [workload]
set -u
OUT=${1:-gpurun_out/pmc_lds}
WL=${2:-c2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for f in "$ROOT"/audiosdr_amd/variants/libasdr_*.so; do
  n=$(basename "$f" .so); n=${n#libasdr_}
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d "$ROOT/$OUT/$n" -- python3 "$ROOT/tools/c2_loop.py" "$f" 40 "$WL" > "$ROOT/$OUT/$n.log" 2>&1
  echo "== $n"
  python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT/$n" | grep -E "LDS|WAVE_CYCLES"
done
