"""Turn rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/r02_traffic.json, which bench.py reports as roofline.traffic when config, queries, n, team size and pipeline match.

    python tools/traffic_from_pmc.py CONFIG QUERIES N TEAM PIPELINED FETCH_DIR WRITE_DIR [... 7 more ...]

TEAM = workers per query (bench line: cus_per_query minus the committer), PIPELINED = 0 / 1.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB and gfx950 counts a
128-byte read request as 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import sys

KERNEL = "rrt_expand_block_kernel"
NOTE = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with --kernel-trace only), per dispatch of "
        + KERNEL + "; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)")


def mean_counter(dirname, counter):
    vals = []
    paths = sorted(glob.glob(dirname + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:  # the latest pass only (gpurun merges every call's output into the same directory)
        for row in csv.DictReader(open(path)):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, KERNEL, dirname))
    return sum(vals) / len(vals)


def main(argv):
    entries = []
    for k in range(0, len(argv), 7):
        cfg, q, n, team, pipe, fdir, wdir = argv[k:k + 7]
        f, w = mean_counter(fdir, "FETCH_SIZE"), mean_counter(wdir, "WRITE_SIZE")
        entries.append(dict(config=int(cfg), queries_per_gpu=int(q), n=int(n), team=int(team), pipelined=bool(int(pipe)),
                            fetch_size_kb=f, write_size_kb=w, hbm_bytes_per_launch=int((2 * f + w) * 1024), note=NOTE))
    json.dump({"entries": entries}, open("profiles/r02_traffic.json", "w"), indent=1)
    for e in entries:
        print(e["config"], e["queries_per_gpu"], e["team"], e["pipelined"], e["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main(sys.argv[1:])
