cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/dubins_stamps.py block 1 > $O/dubins_stamps.txt 2>&1
RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/dubins_stamps.py block 256 > $O/dubins_stamps_q256.txt 2>&1
cat $O/dubins_stamps.txt $O/dubins_stamps_q256.txt
