cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
export RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so RRT_STAMPS_PIPE=1
(echo "## config 2 default (64+1), one kernel"; timeout -k 10 120 python3 tools/stamps.py; echo "## config 2, RRT_FLAG_SPLIT_COMMIT: committer as its own 8-wave kernel"; timeout -k 10 120 python3 tools/stamps.py --split) > $O/stamps_split.txt 2>&1
cat $O/stamps_split.txt
