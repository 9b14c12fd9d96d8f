cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
for q in 256 64 16; do
RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 400 python3 tools/tail_probe.py --config 5 --queries $q > $O/tail_config5_q$q.txt 2>&1; cat $O/tail_config5_q$q.txt
done
