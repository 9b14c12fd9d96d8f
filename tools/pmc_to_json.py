#!/usr/bin/env python3
"""rocprofv3 PMC passes of bench.py (tools/gpu_round.sh pmc) -> profiles/r04_traffic.json and profiles/r04_sq_counters.json,
which bench.py reports as roofline.measured / roofline.bound_observed when workload and kernel variant match.

    python3 tools/pmc_to_json.py gpurun_out/r03 "2 1" "3 1" "4 64" ...      (specs: "CONFIG QUERIES")

Traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (each with --kernel-trace only), mean per dispatch of the
expansion kernel; HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- both counters are in KB and gfx950 counts a
128-byte read request as 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section).
SQ: one more pass with SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_ACTIVE_INST_VALU: the share of its
cycles a wave of the kernel spends waiting (s_waitcnt, barriers), issuing any instruction, issuing VALU instructions."""
import csv
import glob
import json
import os
import sys

KERNELS = ("rrt_expand_block_kernel", "rrt_dubins_block_kernel", "rrt_pipe_kernel", "rrt_expand_kernel")
SQ = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")


def mean_counters(dirname, counters):
    vals = {c: [] for c in counters}
    paths = sorted(glob.glob(dirname + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:  # the latest pass only (gpurun merges every call's output into the same directory)
        for row in csv.DictReader(open(path)):
            if any(k in row["Kernel_Name"] for k in KERNELS) and row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {c: (sum(v) / len(v) if v else None) for c, v in vals.items()}


def main(argv):
    out_dir, specs = argv[0], argv[1:]
    traffic, sq = [], []
    for spec in specs:
        c, q = spec.split()
        base = f"{out_dir}/pmc_c{c}_q{q}_"
        try:
            line = json.load(open(base + "FETCH_SIZE.json"))
        except (OSError, ValueError) as e:
            print("skip", spec, e)
            continue
        cfg = line["config"]
        team = cfg["cus_per_query"] - (1 if cfg["pipelined"] else 0)
        key = dict(config=int(c), queries_per_gpu=int(q), n=cfg["n"], team=team, pipelined=bool(cfg["pipelined"]), kernel=line["roofline"]["kernel"],
                   build_sha256=(line.get("build") or {}).get("lib_sha256"))  # the binary the pass measured (bench.py reports a pass only on the same build)
        f = mean_counters(base + "FETCH_SIZE", ("FETCH_SIZE",))["FETCH_SIZE"]
        w = mean_counters(base + "WRITE_SIZE", ("WRITE_SIZE",))["WRITE_SIZE"]
        if f is not None and w is not None:
            traffic.append(dict(key, fetch_size_kb=f, write_size_kb=w, hbm_bytes_per_launch=int((2 * f + w) * 1024),
                                note="separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (--kernel-trace only), mean per dispatch; FETCH_SIZE doubled per "
                                     "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"))
        s = mean_counters(base + "SQ_WAVE_CYCLES", SQ)
        if s["SQ_WAVE_CYCLES"]:
            wc = s["SQ_WAVE_CYCLES"]
            wait, issue, valu = s["SQ_WAIT_ANY"] / wc, s["SQ_ACTIVE_INST_ANY"] / wc, s["SQ_ACTIVE_INST_VALU"] / wc
            kind = "latency (waves wait on s_waitcnt / barriers / hand-offs most of their cycles)" if wait > 0.6 and issue < 0.3 else "instruction issue"
            sq.append(dict(key, kind=kind, waves_waiting_frac=wait, waves_issuing_frac=issue, valu_busy_frac=valu, counters=s,
                           source="profiles/r04_sq_counters.json (rocprofv3 --pmc " + " ".join(SQ) + ", an earlier run of this workload)"))
    json.dump({"entries": traffic}, open("profiles/r04_traffic.json", "w"), indent=1)
    json.dump({"entries": sq}, open("profiles/r04_sq_counters.json", "w"), indent=1)
    for e in traffic:
        print("traffic", e["config"], e["queries_per_gpu"], e["team"], e["pipelined"], e["hbm_bytes_per_launch"])
    for e in sq:
        print("sq", e["config"], e["queries_per_gpu"], "wait %.2f issue %.2f valu %.2f" % (e["waves_waiting_frac"], e["waves_issuing_frac"], e["valu_busy_frac"]))


if __name__ == "__main__":
    main(sys.argv[1:])
