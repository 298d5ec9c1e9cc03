#!/bin/bash
# GPU box: the detector's phases by ablation builds (audiosdr_amd/variants/libasdr_pre<mask>.so: -DASDR_PRE_ABLATE=<mask>): kernel time
# of the one-block detector call per variant (results of the ablated builds are garbage: timing only).   bash tools/front_ablate.sh [blocks]
cd "$(dirname "$0")/.."
T=${1:-1}
echo "tree: $(python3 tools/bench_front.py 65536 $T 2>/dev/null | grep 'detector on' | sed 's/.*kernel_ms_median": \([0-9.]*\).*/\1/')"
for f in audiosdr_amd/variants/libasdr_pre*.so; do
  echo "$(basename $f): $(ASDR_TOOLS_LIB=$PWD/$f python3 tools/bench_front.py 65536 $T 2>/dev/null | grep 'detector on' | sed 's/.*kernel_ms_median": \([0-9.]*\).*/\1/')"
done
