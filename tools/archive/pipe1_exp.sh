#!/bin/bash
# The barrier-free one-CU kernel (rrt_pipe.h) against the block kernel on one CU: parity tests of both, then the bench shapes that run one CU per query.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_dubins.py -m gpu -x -q -k "block or batch or many or dubins or config5" > $O/pipe1_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 $O/pipe1_tests.log
[ $rc -eq 0 ] || exit $rc
bash $R/tools/pipe1_quick.sh $1
