#!/bin/bash
# record-stream depth (RRT_STREAM_DEPTH) variants built on the GPU box; config 5 at 256 and 16 queries, config 2 x 256 on one CU each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
cd $R/rrtplanner_amd/csrc
pids=""
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  ( make exp NAME=$name EXP="$flags" > $O/build_$name.log 2>&1 ) &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
cd /tmp
for v in "$@"; do
  name=${v%%:*}
  lib=$R/rrtplanner_amd/librrt_hip_exp_$name.so
  [ -f $lib ] || { echo "$name: build failed"; tail -n 5 $O/build_$name.log; continue; }
  for shape in "--config 5 --steps 2 --warmup 1" "--config 5 --queries 16 --steps 2 --warmup 1" "--queries 256 --no-batched"; do
    RRT_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py $shape --no-cpu-baseline > $O/var.json 2>/dev/null
    python3 - "$name" "$shape" <<PY
import json,sys
try:
    d=json.load(open("$O/var.json")); print("%-10s %-46s kernel %8.3f ms  %s" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms"], d["roofline"].get("kernel")))
except Exception as e: print(sys.argv[1], sys.argv[2], "ERR", e)
PY
  done
done
