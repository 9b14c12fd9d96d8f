#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team2 or team3 or config4 or every_team_size or loses_a_member or two_contexts" > $O/wide_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 6 $O/wide_tests.log
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline > $O/wide_default.json 2>/dev/null
timeout -k 10 200 python3 $R/bench.py --config 4 --team 2 --no-cpu-baseline > $O/wide_c4_team2.json 2>/dev/null
timeout -k 10 200 python3 $R/bench.py --team 2 --no-cpu-baseline --no-batched > $O/wide_c2_team2.json 2>/dev/null
python3 - <<PY
import json
for f in ("wide_default","wide_c4_team2","wide_c2_team2"):
    try:
        d=json.load(open("$O/"+f+".json")); r=d["roofline"]; b=d.get("batched")
        print("%-16s kernel %8.3f ms  %s%s" % (f, r["kernel_ms"], r.get("kernel"), ("   batched leg %.3f ms %s" % (b["roofline"]["kernel_ms"], b["roofline"].get("kernel"))) if b else ""))
    except Exception as e: print(f, "ERR", e)
PY
