#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_dubins.py -m gpu -x -q -k "block or batch or many or dubins or config5" > $O/pipe1_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 $O/pipe1_tests.log
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 400 python3 $R/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/tiny_c5.json 2>/dev/null
timeout -k 10 120 python3 $R/bench.py --queries 256 --no-cpu-baseline --no-batched > $O/tiny_c2_q256.json 2>/dev/null
python3 - <<PY
import json
for f in ("tiny_c5","tiny_c2_q256"):
    d=json.load(open("$O/"+f+".json")); r=d["roofline"]
    print("%-18s ms/step %8.3f kernel %8.3f value %.4g  %s" % (f, d["ms_per_step"], r["kernel_ms"], d["value"], r.get("kernel")))
PY
cd $R
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 400 python3 tools/tail_probe.py --config 5 --queries 256 > $O/tail_config5.txt 2>&1; cat $O/tail_config5.txt
