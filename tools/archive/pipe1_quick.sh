#!/bin/bash
# quick look: the one-CU shapes (config 2 on one CU, 256 queries, config 4 on one CU each), config 5 at 16 queries; stamps of both pipelines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 python3 $R/bench.py --team 1 --no-cpu-baseline --no-batched > $O/pipe1_c2_team1.json 2>$O/pipe1_c2_team1.err && \
timeout -k 10 120 python3 $R/bench.py --queries 256 --no-cpu-baseline --no-batched > $O/pipe1_c2_q256.json 2>/dev/null && \
timeout -k 10 120 python3 $R/bench.py --config 4 --team 1 --no-cpu-baseline > $O/pipe1_c4_team1.json 2>/dev/null && \
timeout -k 10 300 python3 $R/bench.py --config 5 --queries 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/pipe1_c5_q16.json 2>/dev/null
python3 - <<PY
import json
for f in ("pipe1_c2_team1","pipe1_c2_q256","pipe1_c4_team1","pipe1_c5_q16"):
    try:
        d=json.load(open("$O/"+f+".json")); r=d["roofline"]
        print("%-18s ms/step %8.3f kernel %8.3f value %.4g  %s" % (f, d["ms_per_step"], r["kernel_ms"], d["value"], r.get("kernel")))
    except Exception as e:
        print(f, "ERR", e)
PY
if [ "$1" = "stamps" ]; then
  cd $R
  make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
  RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/pipe_stamps.py > $O/pipe_stamps.txt 2>&1
  RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/pipe_stamps.py --n 20000 --queries 64 >> $O/pipe_stamps.txt 2>&1
  RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/dubins_stamps.py > $O/dubins_stamps.txt 2>&1
  cat $O/pipe_stamps.txt $O/dubins_stamps.txt
fi
