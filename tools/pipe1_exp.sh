#!/bin/bash
# The barrier-free one-CU kernel (rrt_pipe.h) against the block kernel on one CU: parity tests of both, then the bench shapes that run one CU per query.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block or batch or many" > $O/pipe1_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 $O/pipe1_tests.log
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 120 python3 $R/bench.py --team 1 --no-cpu-baseline --no-batched > $O/pipe1_c2_team1.json 2>$O/pipe1_c2_team1.err && \
timeout -k 10 120 python3 $R/bench.py --queries 256 --no-cpu-baseline --no-batched > $O/pipe1_c2_q256.json 2>/dev/null && \
timeout -k 10 120 python3 $R/bench.py --config 4 --team 1 --no-cpu-baseline > $O/pipe1_c4_team1.json 2>/dev/null
python3 - <<PY
import json
for f in ("pipe1_c2_team1","pipe1_c2_q256","pipe1_c4_team1"):
    try:
        d=json.load(open("$O/"+f+".json")); r=d["roofline"]
        print("%-18s ms/step %8.3f kernel %8.3f value %.4g  %s" % (f, d["ms_per_step"], r["kernel_ms"], d["value"], r.get("kernel")))
    except Exception as e:
        print(f, "ERR", e)
PY
