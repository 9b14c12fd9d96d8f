# config 5 on the GPU box: bench line at 256 queries (new kernel), the serial kernel at 16 for comparison, stamps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python3 bench.py --config 5 --steps 3 --warmup 1 > $O/bench_config5.json 2> $O/bench_config5.err; echo "c5 rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/bench_config5.json"))
print("config5 x256: ms_per_step %.1f kernel_ms %.1f value %.4g nodes/s, words made/model %d/%d"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["value"], d["roofline"]["valu_f64_model"]["dubins_word_evaluations_made"], d["roofline"]["valu_f64_model"]["dubins_word_evaluations_model"]))
print("cpu", d.get("cpu_baseline"))
PY
if [ "$1" = "stamps" ]; then
  make -C rrtplanner_amd/csrc ../librrt_hip_stamps.so > /dev/null 2>&1; echo "stamps build rc=$?"
  RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 300 python3 tools/dubins_stamps.py > $O/dubins_stamps.txt 2>&1; cat $O/dubins_stamps.txt
fi
