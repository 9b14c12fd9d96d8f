cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
RRT_STAMPS_PIPE=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/stamps.py --n 20000 --queries 64 > $O/stamps_c4.txt 2>&1; cat $O/stamps_c4.txt
