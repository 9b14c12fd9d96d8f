#!/bin/bash
# One GPU-box session: the -m gpu suite, the bench lines and the rocprofv3 passes whose summaries go to profiles/.
#   gpurun --timeout 1200 -- bash tools/gpu_round.sh [tests|bench|prof|pmc ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
# every summary that goes to profiles/ names the binary it measured (bench.py reports a counter pass only on the same build)
BUILD=$(python3 -c "import hashlib;print(hashlib.sha256(open('$R/rrtplanner_amd/librrt_hip.so','rb').read()).hexdigest()[:16])")
echo "librrt_hip.so sha256[:16] = $BUILD" > $O/BUILD_ID.txt
for what in "$@"; do
case $what in
smoke)
    (cd $R && python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1); echo "smoke rc=$?"; tail -n 4 $O/smoke.log ;;
tests)
    (cd $R && python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1); echo "tests rc=$?"; tail -n 3 $O/gpu_tests.log ;;
driver)
    # the driver's exact command line (BENCH_rNN.json)
    (cd $R && python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err); echo "driver-cmd bench rc=$?"
    python3 - <<PY
import json
d=json.load(open("$O/bench_driver_cmd.json"))
print("ms_per_step %.3f kernel_ms %.3f host_gap_ms %.3f value %.4g"%(d["ms_per_step"], d["roofline"]["kernel_ms"], d["host_gap_ms"], d["value"]))
PY
    ;;
total512)
    # BASELINE configs[3] as the job it names -- 512 queries -- on ONE GPU (the denominator of the 8-GPU strong-scaling claim)
    python3 $R/bench.py --config 4 --total-queries 512 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config4_total512_n1.json 2> $O/bench_config4_total512_n1.err; echo "total512 rc=$?"; tail -n 2 $O/bench_config4_total512_n1.err
    python3 -c "
import json; d=json.load(open('$O/bench_config4_total512_n1.json')); print('512 queries on one GPU: ms_per_step %.2f kernel_ms %.2f value %.4g cus/query %s kernel %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['cus_per_query'], d['roofline']['kernel']))" ;;
gap)
    python3 $R/tools/step_gap.py --steps 20 > $O/step_gap.txt 2>&1; echo "gap rc=$?"; cat $O/step_gap.txt
    python3 $R/tools/step_gap.py --steps 20 --config 4 >> $O/step_gap.txt 2>&1; tail -n 9 $O/step_gap.txt ;;
bench)
    python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -n 2 $O/bench_default.err
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_PORT=29999 python3 $R/bench.py --no-cpu-baseline --no-batched > $O/bench_one_rank_comm.json 2> $O/bench_one_rank_comm.err; echo "bench(comm) rc=$?"
    python3 $R/bench.py --config 3 --no-batched > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench c3 rc=$?"
    python3 $R/bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench c4 rc=$?"
    python3 $R/bench.py --queries 8 --no-cpu-baseline --no-batched > $O/bench_config2_q8.json 2>/dev/null
    python3 $R/bench.py --queries 256 --no-cpu-baseline --no-batched > $O/bench_config2_q256.json 2>/dev/null
    for t in 1 2 4 8 16 32; do python3 $R/bench.py --team $t --no-cpu-baseline --no-batched > $O/bench_config2_team$t.json 2>/dev/null; done
    for t in 1 2 4 8; do python3 $R/bench.py --config 4 --team $t --no-cpu-baseline > $O/bench_config4_team$t.json 2>/dev/null; done
    python3 $R/bench.py --config 5 --steps 3 --warmup 1 > $O/bench_config5.json 2> $O/bench_config5.err; echo "bench c5 rc=$?"
    python3 $R/bench.py --config 5 --serial --queries 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5_serial_q16.json 2>/dev/null
    python3 $R/bench.py --config 5 --queries 16 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_config5_q16.json 2>/dev/null
    python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]; m = r.get("measured") or {}
        print("%-32s ms/step %8.3f kernel %8.3f gap %6.3f value %.4g cus %s frac %s measured_frac %s" % (os.path.basename(f), d["ms_per_step"], r["kernel_ms"], d.get("host_gap_ms", 0), d["value"], d["config"]["cus_per_query"], r["frac"], m.get("hbm_frac")))
        b = d.get("batched")
        if b: print("%-32s ms/step %8.3f kernel %8.3f value %.4g cus %s" % ("  batched leg", b["ms_per_step"], b["roofline"]["kernel_ms"], b["value"], b["cus_per_query"]))
    except Exception as e:
        print(f, "ERR", e)
PY
    ;;
quick)
    python3 $R/bench.py --no-cpu-baseline > $O/q_default.json 2>$O/q_default.err; echo "default rc=$?"
    python3 $R/bench.py --config 4 --team 1 --no-cpu-baseline > $O/q_c4_team1.json 2>/dev/null
    python3 $R/bench.py --config 3 --no-cpu-baseline --no-batched > $O/q_c3.json 2>/dev/null
    python3 $R/bench.py --team 1 --no-cpu-baseline --no-batched > $O/q_c2_team1.json 2>/dev/null
    python3 - <<PY
import json
for f in ("q_default","q_c4_team1","q_c3","q_c2_team1"):
    try:
        d=json.load(open("$O/"+f+".json"))
        b=d.get("batched")
        print(f, "kernel_ms=%.3f value=%.4g"%(d["roofline"]["kernel_ms"], d["value"]), ("batched kernel_ms=%.3f value=%.4g"%(b["roofline"]["kernel_ms"], b["value"])) if b else "")
    except Exception as e:
        print(f, "ERR", e)
PY
    ;;
prof)
    for c in 2 3 4 5; do
        rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -- python3 $R/bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-batched > $O/prof_c$c.json 2> $O/prof_c$c.err
        echo "prof c$c rc=$?"
        f=$(ls -t $O/prof_c$c/*/*kernel_stats.csv 2>/dev/null | head -n 1)
        [ -n "$f" ] && (echo "# build $BUILD (sha256[:16] of librrt_hip.so); rocprofv3 --kernel-trace --stats -- bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-batched"; cat $f) > $O/config${c}_kernel_stats.csv
    done
    # the one-CU-per-query pipeline (rrt_pipe_kernel): 256 x config 2
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2_q256 -- python3 $R/bench.py --queries 256 --steps 5 --warmup 1 --no-cpu-baseline --no-batched > $O/prof_c2_q256.json 2> $O/prof_c2_q256.err
    echo "prof c2 q256 rc=$?"
    f=$(ls -t $O/prof_c2_q256/*/*kernel_stats.csv 2>/dev/null | head -n 1)
    [ -n "$f" ] && (echo "# build $BUILD (sha256[:16] of librrt_hip.so); rocprofv3 --kernel-trace --stats -- bench.py --queries 256 --steps 5 --warmup 1 --no-cpu-baseline --no-batched"; cat $f) > $O/config2_q256_kernel_stats.csv ;;
stamps)
    # the diagnostic build's cycle stamps (never the product's run times): all sixteen per block, and the four-stamp form whose
    # balance of the ring is close to the product build's
    (cd $R && make -C rrtplanner_amd/csrc -j16 stamps > $O/stamps_build.log 2>&1 && make -C rrtplanner_amd/csrc -j16 exp EXP="-DRRT_STAMPS -DRRT_STAMPS_LIGHT=0x8016" NAME=light >> $O/stamps_build.log 2>&1); echo "stamps builds rc=$?"
    (echo "# tools/stamps.py, diagnostic builds of the sources whose product build is $BUILD (run times of these builds are not the product's)"
     echo "## config 2 default (64+1), all stamps"; cd $R && RRT_STAMPS_PIPE=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 200 python3 tools/stamps.py
     echo "## config 4 share: 64 queries n=20000 default team (3+1 pipelined), all stamps"; RRT_STAMPS_PIPE=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 200 python3 tools/stamps.py --n 20000 --queries 64
     echo "## config 2 default (64+1), LIGHT stamps (mask 0x8016): committer wcyc[1] = block start .. part A done, [2] = .. lists + rounds done, [4] = .. published, [15] = .. past the end-of-block barrier; [14] low half = fetch of the next block's records, high half = its wait for the workers' flags; worker 1 wcyc[16+1] = take + resolve + hand over, [16+2] = go wait (cycles over the whole run: divide by 782 blocks)"
     RRT_STAMPS_RAW=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_exp_light.so timeout -k 10 200 python3 tools/stamps.py | grep -E "kernel|raw") > $O/stamps.txt 2>&1
    grep -E "^##|kernel|committer" $O/stamps.txt | cut -c1-200 ;;
pmc)
    SPECS=("2 1" "3 1" "4 64" "2 8" "2 256" "5 256")
    for spec in "${SPECS[@]}"; do
        set -- $spec; c=$1; q=$2
        for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
            name=${ctr%% *}
            rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c${c}_q${q}_$name -- python3 $R/bench.py --config $c --queries $q --steps 2 --warmup 1 --no-cpu-baseline --no-batched > $O/pmc_c${c}_q${q}_$name.json 2> $O/pmc_c${c}_q${q}_$name.err
            echo "pmc c$c q$q $name rc=$?"
        done
    done
    (cd $R && python3 tools/pmc_to_json.py $O "${SPECS[@]}" && cp profiles/r04_traffic.json profiles/r04_sq_counters.json $O/) ;;
esac
done
