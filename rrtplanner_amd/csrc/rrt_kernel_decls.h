// rrt_kernel_decls.h -- the expansion kernels that are plain (non-template) functions, declared for the translation unit that
// launches them (rrt_engine.hip).  Definitions: rrt_pipe.h and rrt_dubins_block.h, compiled by kernels_tu.hip.
#pragma once

#include "rrt_kernels.h"

namespace rrtdev {

__global__ __launch_bounds__(TPB) void rrt_pipe_kernel(BatchView bv);
__global__ __launch_bounds__(TPB) void rrt_dubins_block_kernel(BatchView bv);

}  // namespace rrtdev
