"""rrtplanner_amd -- MI355X-native drop-in for the tree-expansion hot path of rland93/rrtplanner.

Same planner classes as ``rrtplanner.rrt`` (RRTStandard / RRTStar / RRTStarInformed); the
sample -> nearest -> line-of-sight -> choose-parent -> insert loop runs as hand-written HIP
kernels for gfx950 behind the C ABI of include/rrt_hip.h.
"""
from .rrt import RRT, RRTStandard, RRTStar, RRTStarInformed, r2norm, random_point_og
from .oggen import perlin_occupancygrid
from .dubins import RRTDubins, RRTStarDubins  # no reference counterpart (README only): see include/rrt_dubins.h

__version__ = "0.1.0"
__all__ = ["RRT", "RRTStandard", "RRTStar", "RRTStarInformed", "r2norm", "random_point_og", "perlin_occupancygrid",
           "RRTDubins", "RRTStarDubins"]
