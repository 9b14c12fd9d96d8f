#!/bin/bash
# builds variants of the library on the GPU box (make exp) and runs the one-CU bench shapes with each
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
  for shape in "--team 1 --no-batched" "--queries 256 --no-batched" "--config 4 --team 1"; do
    RRT_HIP_LIB=$lib timeout -k 10 120 python3 $R/bench.py $shape --no-cpu-baseline > $O/var.json 2>/dev/null
    python3 - "$name" "$shape" <<PY
import json,sys
try:
    d=json.load(open("$O/var.json")); print("%-10s %-28s kernel %8.3f ms  %s" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms"], d["roofline"].get("kernel")))
except Exception as e: print(sys.argv[1], sys.argv[2], "ERR", e)
PY
  done
done
