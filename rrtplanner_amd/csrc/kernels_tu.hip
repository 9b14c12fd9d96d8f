// kernels_tu.hip -- the definitions of the expansion kernels, dealt to translation units (make passes -DRRT_TU=k) so that they
// compile side by side; rrt_engine.hip only launches them.  No unit calls device code of another, so the objects link without
// relocatable device code.
#include <hip/hip_runtime.h>

#if RRT_TU == 1  // one CU per query, no barrier in the loop (RRTStandard / RRTStar)
#define RRT_BLOCK_DECL_ONLY
#define RRT_SERIAL_DECL_ONLY
#include "rrt_pipe.h"

#elif RRT_TU == 2  // Dubins planners: the pipeline, and the one-sample-per-iteration kernel kept as its cross-check
#include "rrt_dubins_block.h"
namespace rrtdev {
template __global__ void rrt_expand_kernel<false, true>(BatchView);
}

#elif RRT_TU == 3  // one sample per iteration: cross-check of the block kernel, and the opt-in true rewire
#include "rrt_kernels.h"
namespace rrtdev {
template __global__ void rrt_expand_kernel<false, false>(BatchView);
template __global__ void rrt_expand_kernel<true, false>(BatchView);
}

#else  // teams of compute units: rrt_expand_block_kernel<G, BSM, PIPE, INF>
#define RRT_SERIAL_DECL_ONLY
#include "rrt_block.h"
namespace rrtdev {
#define K(G, BSM, PIPE, INF) template __global__ void rrt_expand_block_kernel<G, BSM, PIPE, INF>(BatchView);
#if RRT_TU == 10
K(64, 1, true, false)
#elif RRT_TU == 11
K(64, 1, true, true)
#elif RRT_TU == 12
K(32, 2, true, false) K(32, 2, true, true)
#elif RRT_TU == 13
K(16, 4, true, false) K(16, 4, true, true)
#elif RRT_TU == 14
K(8, 8, true, false) K(8, 8, true, true)
#elif RRT_TU == 15
K(4, 16, true, false) K(4, 16, true, true)
#elif RRT_TU == 16
K(3, 16, true, false) K(3, 16, true, true)
#elif RRT_TU == 17
K(2, 16, true, false) K(2, 16, true, true) K(2, 32, true, false)
#elif RRT_TU == 18
K(64, 1, false, false) K(64, 1, false, true) K(32, 2, false, false) K(32, 2, false, true)
#elif RRT_TU == 19
K(16, 4, false, false) K(16, 4, false, true) K(8, 8, false, false) K(8, 8, false, true)
#elif RRT_TU == 20
K(4, 16, false, false) K(4, 16, false, true) K(2, 16, false, false) K(2, 16, false, true)
#elif RRT_TU == 21
K(1, 16, false, false) K(1, 16, false, true)
#else
#error "RRT_TU: unknown translation unit"
#endif
#undef K
}
#endif
