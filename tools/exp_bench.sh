#!/bin/bash
# Compare experiment builds of the library (make -C rrtplanner_amd/csrc exp EXP=... NAME=...) on the bench workloads.
#   gpurun -- bash tools/exp_bench.sh base prio walk8 ...      ("base" = the product library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/exp
mkdir -p $O
for name in "$@"; do
    lib=$R/rrtplanner_amd/librrt_hip_exp_$name.so
    [ "$name" = base ] && lib=$R/rrtplanner_amd/librrt_hip.so
    for cfg in "4" "2" ${EXP_CONFIGS}; do
        RRT_HIP_LIB=$lib python3 $R/bench.py --config $cfg --no-cpu-baseline --no-batched > $O/${name}_c$cfg.json 2> $O/${name}_c$cfg.err || echo "$name c$cfg FAILED: $(tail -n 1 $O/${name}_c$cfg.err)"
    done
    python3 - <<PY
import json
out=[]
for cfg in "4 2 ${EXP_CONFIGS}".split():
    try:
        d=json.load(open("$O/${name}_c%s.json"%cfg)); out.append("c%s %.3f ms"%(cfg,d["roofline"]["kernel_ms"]))
    except Exception as e:
        out.append("c%s ERR"%cfg)
print("%-10s"%"$name", "  ".join(out))
PY
done
