cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 200 python3 bench.py --split --steps 20 --warmup 5 --no-batched > $O/bench_split.json 2> $O/bench_split.err; echo "split rc=$?"; tail -n 3 $O/bench_split.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-batched --no-cpu-baseline > $O/bench_nosplit.json 2>/dev/null
python3 - <<PY
import json
for f in ("bench_split","bench_nosplit"):
    d=json.load(open("$O/"+f+".json")); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"], d["config"]["team_fallbacks"], d.get("cpu_baseline",{}).get("device_result_equals_oracle"))
PY
