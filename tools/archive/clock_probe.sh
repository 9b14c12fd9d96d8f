cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
for q in 1 64 256; do
RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/pipe_stamps.py --queries $q 2>&1 | grep "kernel\|GHz"
done
(rocm-smi --showclocks 2>&1 | head -20) || true
