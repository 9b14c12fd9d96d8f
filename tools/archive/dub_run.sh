set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_dubins.py -m gpu -x -q > gpurun_out/r03/dubins_tests.log 2>&1; echo "tests rc=$?"; tail -n 15 gpurun_out/r03/dubins_tests.log
