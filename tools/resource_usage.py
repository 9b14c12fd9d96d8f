#!/usr/bin/env python3
"""Table of the kernels' register / scratch / LDS use from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/resource_usage.py [ROLE]        (ROLE 1 / 2: only that half of the pipelined kernels, like `make asm-role`)

Compiles rrt_engine.hip for gfx950 (device only, no GPU needed) and prints one line per kernel."""
import re
import subprocess
import sys

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
       f"-I{ROOT}/include", f"-I{ROOT}/rrtplanner_amd/csrc", "--cuda-device-only", "-S", "-o", "/dev/null", f"{ROOT}/rrtplanner_amd/csrc/rrt_engine.hip",
       "-Rpass-analysis=kernel-resource-usage"]
if len(sys.argv) > 1:
    cmd.insert(1, f"-DRRT_ONLY_ROLE={sys.argv[1]}")
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch B/lane':>14s} {'LDS B':>7s}")
for r in rows:
    name = re.sub(r"rrtdev::|\(rrtdev::BatchView\)|void ", "", r["name"])
    print(f"{name:78s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('SGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} {r.get('SGPRs Spill', 0):6d} "
          f"{r.get('ScratchSize', 0):14d} {r.get('LDS Size', 0):7d}")
