"""Cost functions for the custom-costfn tests (reference rrt.py:55, :70-80: ``costfn(vcosts, points, v, x) -> float``).
Shared by tests/golden/make_golden_costfn.py (which runs the real reference with them) and the tests (which run this build)."""
from math import sqrt


def discount(vcosts, points, v, x):
    """Half the cost so far plus the edge: the rewire predicate cost(vn -> xnew) < vcosts[vn] (rrt.py:536) is true whenever the
    edge is shorter than half of vn's cost, so the reference's rewire block fires all the time -- stale costs, child lists that
    lose a vertex without the new parent gaining it, and second rewires of one vertex (ValueError, rrt.py:541-546 / :740)."""
    d = points[v] - x
    return 0.5 * vcosts[v] + sqrt(d[0] * d[0] + d[1] * d[1])


def manhattan(vcosts, points, v, x):
    """Cost so far plus the L1 edge length: never rewires (the edge term is non-negative), but chooses other parents than the
    Euclidean default."""
    d = points[v] - x
    return vcosts[v] + float(abs(d[0]) + abs(d[1]))


def downhill(vcosts, points, v, x):
    """A cost that depends on where the NEW point lies more than on the path: rewires towards samples with a small x + y."""
    d = points[v] - x
    return 0.25 * vcosts[v] + 0.5 * sqrt(d[0] * d[0] + d[1] * d[1]) + 0.05 * float(x[0] + x[1])


COSTFNS = {"discount": discount, "manhattan": manhattan, "downhill": downhill}
