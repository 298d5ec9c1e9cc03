#!/bin/bash
# GPU box: the GPU suite N times in a row (default 10), one process per run -- a single green run says nothing about the aborts that depend on the
# allocation history of a long-lived process (round 6: every second run died silently until the auto-pinning tests got processes of their own).
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/loop_gpu_suite.sh 10'
n=${1:-10}; ok=0
for i in $(seq 1 "$n"); do
  r=$(timeout 900 python3 -X faulthandler -m pytest tests -q -m gpu -x 2>&1 | tail -1)
  case "$r" in *" passed"*) case "$r" in *failed*) echo "run $i: $r";; *) ok=$((ok + 1));; esac;; *) echo "run $i: $r";; esac
done
echo "green runs: $ok of $n"
[ "$ok" -eq "$n" ]
