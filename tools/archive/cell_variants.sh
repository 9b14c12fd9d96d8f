#!/bin/bash
# cell-size variants (RRT_CELL_DIV) built on the GPU box; the default bench (config 2 + config 4 batched), config 3, and the one-CU shapes
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
  for shape in "--no-cpu-baseline" "--config 3 --no-batched" "--team 1 --no-batched" "--queries 256 --no-batched" "--config 4 --team 1" "--team 4 --no-batched"; do
    RRT_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py $shape --no-cpu-baseline > $O/var.json 2>/dev/null
    python3 - "$name" "$shape" <<PY
import json,sys
try:
    d=json.load(open("$O/var.json")); b=d.get("batched")
    print("%-8s %-30s kernel %8.3f ms  %s%s" % (sys.argv[1], sys.argv[2], d["roofline"]["kernel_ms"], d["roofline"].get("kernel"), ("   batched leg %.3f ms" % b["roofline"]["kernel_ms"]) if b else ""))
except Exception as e: print(sys.argv[1], sys.argv[2], "ERR", e)
PY
  done
done
