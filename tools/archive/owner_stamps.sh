cd $GRAFT_REPO_ROOT/rrtplanner_amd/csrc
make exp NAME=owner EXP="-DRRT_STAMPS -DRRT_STAMPS_OWNER" > /tmp/o.log 2>&1; echo "build rc=$?"
cd $GRAFT_REPO_ROOT
RRT_STAMPS_PIPE=1 RRT_STAMPS_RAW=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_exp_owner.so timeout -k 10 300 python3 tools/stamps.py --n 20000 --queries 64 2>&1 | grep "kernel\|raw wcyc\[16"
