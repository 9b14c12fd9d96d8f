#!/bin/bash
# Diagnostic: the wall-clock time line (TSMARK) and the workers' group phases for a window of 32 blocks from block $1 on, four-stamp build.
#   gpurun -- bash tools/ts_window.sh 700
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
make -C rrtplanner_amd/csrc -j16 exp EXP="-DRRT_STAMPS -DRRT_STAMPS_LIGHT=0x8016 -DRRT_TS_BASE=$1" NAME=ts$1 > $O/ts_build.log 2>&1; echo "build rc=$?"
RRT_STAMPS_DUMP=1 RRT_STAMPS_RAW=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_exp_ts$1.so timeout -k 10 200 python3 tools/stamps.py > $O/ts_$1.txt 2>&1
grep -E "kernel|raw" $O/ts_$1.txt; grep "ts block" $O/ts_$1.txt | tail -32 | sed -n 3,14p
