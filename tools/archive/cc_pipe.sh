#!/bin/bash
# compile rrt_pipe.h alone for gfx950 (seconds) and print the kernel's resource usage; $1 = extra flags (e.g. -S -o /tmp/pipe_only.s)
cd /root/repo/rrtplanner_amd/csrc
printf '#include <hip/hip_runtime.h>\n#include "rrt_hip.h"\n#include "rrt_kernels.h"\n#include "rrt_block.h"\n#include "rrt_pipe.h"\n#include "rrt_dubins_block.h"\n' > /tmp/pipe_only.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I. -I../../include --cuda-device-only ${@:--c -o /tmp/pipe_only.o} /tmp/pipe_only.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "Name: _ZN6rrtdev15rrt_pipe\|Name: _ZN6rrtdev23rrt_dubins_block\|error\|warning" | grep -v "^--\|      |\|^ *[0-9]* |" | head -60
