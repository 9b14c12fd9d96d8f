#!/bin/bash
# Development loop on the GPU box: the team kernels against the oracle, the default bench line, the committer's stamps.
#   gpurun --timeout 900 -- bash tools/dev_check.sh [tests] [bench] [stamps] [c4]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
cd $R
for what in "$@"; do
case $what in
tests)
    timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_vs_oracle or fuzz or every_team_size or full_size_properties or config4" > $O/dev_tests.log 2>&1; rc=$?
    echo "tests rc=$rc"; tail -n 6 $O/dev_tests.log
    [ $rc -eq 0 ] || exit $rc ;;
bench)
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batched > $O/dev_bench.json 2> $O/dev_bench.err; echo "bench rc=$?"; tail -n 3 $O/dev_bench.err
    python3 -c "
import json; d=json.load(open('$O/dev_bench.json')); print('ms_per_step %.3f kernel_ms %.3f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" ;;
c4)
    timeout -k 10 300 python3 bench.py --config 4 --no-cpu-baseline > $O/dev_c4.json 2> $O/dev_c4.err; echo "c4 rc=$?"
    python3 -c "
import json; d=json.load(open('$O/dev_c4.json')); print('c4 ms_per_step %.3f kernel_ms %.3f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" ;;
stamps)
    make -C rrtplanner_amd/csrc -j16 stamps > $O/stamps_build.log 2>&1; echo "stamps build rc=$?"
    (echo "## config 2 default (64+1)"; RRT_STAMPS_DUMP=1 RRT_STAMPS_PIPE=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 200 python3 tools/stamps.py) > $O/dev_stamps.txt 2>&1; cat $O/dev_stamps.txt ;;
light)
    # four stamps per block instead of sixteen (mask 0x8016 = stamps 1, 2, 4, 15): the balance of the ring close to the product build's
    make -C rrtplanner_amd/csrc -j16 exp EXP="-DRRT_STAMPS -DRRT_STAMPS_LIGHT=0x8016" NAME=light > $O/light_build.log 2>&1; echo "light build rc=$?"
    (echo "## config 2 default (64+1), light stamps: committer [1] = block start .. part A done, [2] = .. rounds done, [4] = .. published, [15] = end-of-block barrier; worker [1] = resolve + hand over (+ take), [2] = go wait"
     RRT_STAMPS_DUMP=1 RRT_STAMPS_RAW=1 RRT_HIP_LIB=rrtplanner_amd/librrt_hip_exp_light.so timeout -k 10 200 python3 tools/stamps.py) > $O/dev_light.txt 2>&1; grep -E "kernel|raw" $O/dev_light.txt; grep "dbg2 groups\|dbg2 resolve>32" $O/dev_light.txt | tail -3 ;;
pstamps)
    # the barrier-free pipeline on config 4's share, one CU per query
    make -C rrtplanner_amd/csrc -j16 stamps > $O/stamps_build.log 2>&1; echo "stamps build rc=$?"
    (RRT_HIP_LIB=rrtplanner_amd/librrt_hip_stamps.so timeout -k 10 200 python3 tools/pipe_stamps.py --n 20000 --queries 64 --team 1) > $O/dev_pstamps.txt 2>&1; cat $O/dev_pstamps.txt ;;
c5)
    timeout -k 10 400 python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/dev_c5.json 2> $O/dev_c5.err; echo "c5 rc=$?"
    python3 -c "
import json; d=json.load(open('$O/dev_c5.json')); print('c5 ms_per_step %.1f kernel_ms %.1f value %.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))" ;;
dubtests)
    timeout -k 10 800 python -m pytest tests/test_dubins.py -m gpu -x -q > $O/dev_dubtests.log 2>&1; rc=$?
    echo "dubins tests rc=$rc"; tail -n 5 $O/dev_dubtests.log
    [ $rc -eq 0 ] || exit $rc ;;
stress)
    timeout -k 10 600 python3 tools/stress_team.py --reps 6 > $O/dev_stress.txt 2>&1; echo "stress rc=$?"; tail -n 15 $O/dev_stress.txt ;;
esac
done
