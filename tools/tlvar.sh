#!/bin/bash
# GPU box: the four-wave kernel's timeline for several profiling builds (audiosdr_amd/variants/libasdr_<name>.so, each built with -DASDR_TIMELINE plus the
# knob under test: `b.build(force=True, extra_flags=["-DASDR_TIMELINE", ...], out=...)`), the lines that show balance: per-duty lifetimes, the IF / FIR phases,
# the wait at the audio barrier.   tools/tlvar.sh timeline tl_noprio ...
for v in "$@"; do
  echo "=== $v"
  TIMELINE_LIB=audiosdr_amd/variants/libasdr_$v.so python3 tools/timeline.py mw 300 2>/dev/null | grep -E "wave lifetime|IF pipeline|Hilbert: FIR|barrier 3|NB chain duty|SUM|parked at"
done
