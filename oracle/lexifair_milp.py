"""TEST INFRASTRUCTURE ONLY -- the reference's own fair-assignment PROCEDURE, with HiGHS in Gurobi's place.

``oracle/lexifair.py`` restates what ``solve_fair_assignment`` (reference ``marl_fair_assign.py:16-55``) computes
(the lexicographically minimal descending cost vector).  This file restates HOW the reference computes it, statement
by statement, so that the definition is checked against the procedure and not only against itself:

  marl_fair_assign.py:5-14   binary x[i, j], every task covered once, every agent assigned once
  :24-27                     continuous z, cost_helper[i, j] * x[i, j] <= z, minimise z
  :32-52                     n rounds: solve; (r, c) = argmin |costs - z*|; cost_helper[r, c] = 0; fix row r to its
                             current assignment

The reference drives Gurobi 10.0.2 through pyomo (neither installed nor vendored here); the MILPs are handed to
``scipy.optimize.milp`` (HiGHS) instead.  Each round's MILP has many optimal assignments -- only the bottleneck
entry is forced -- and which one a solver returns is solver-dependent, but the row that gets fixed is the
bottleneck row, whose column is the same in every optimal assignment when the costs are distinct.  So for distinct
costs the result does not depend on the solver, which is the sense in which the assignment is "parity unpinned" at
Gurobi yet well defined.
"""
import numpy as np
from scipy.optimize import Bounds, LinearConstraint, milp


def solve_fair_assignment_milp(costs):
    """-> (x (n, n) int one-hot, objs descending sorted assigned costs) like the reference's return value."""
    costs = np.asarray(costs, dtype=np.float64)
    n, nj = costs.shape
    assert n == nj
    cost_helper = costs.copy()
    nx = n * nj                                   # variables: x[i, j] at i * nj + j, then z
    cover = np.zeros((nj, nx + 1))
    assign = np.zeros((n, nx + 1))
    for i in range(n):
        for j in range(nj):
            cover[j, i * nj + j] = 1.0            # :12 each task performed by exactly one agent
            assign[i, i * nj + j] = 1.0           # :13 each agent performs exactly one task
    fixed_lo, fixed_hi = np.zeros(nx + 1), np.ones(nx + 1)
    fixed_lo[nx], fixed_hi[nx] = -np.inf, np.inf
    objective = np.zeros(nx + 1)
    objective[nx] = 1.0                           # :27 minimise z
    integrality = np.ones(nx + 1)
    integrality[nx] = 0
    x = None
    for _ in range(n):                            # :32
        aux = np.zeros((nx, nx + 1))              # :25 cost_helper[i, j] * x[i, j] - z <= 0
        aux[np.arange(nx), np.arange(nx)] = cost_helper.ravel()
        aux[:, nx] = -1.0
        res = milp(objective, integrality=integrality, bounds=Bounds(fixed_lo, fixed_hi),
                   constraints=[LinearConstraint(cover, 1.0, 1.0), LinearConstraint(assign, 1.0, 1.0),
                                LinearConstraint(aux, -np.inf, 0.0)])
        assert res.success, res.message
        x = np.rint(res.x[:nx]).reshape(n, nj).astype(int)       # :34
        obj = float(res.x[nx])                                    # :35
        r, c = np.unravel_index(np.argmin(np.abs(costs - obj)), (n, nj))   # :38
        cost_helper[r, c] = 0.0                                   # :41
        for j in range(nj):                                       # :49-51 m.x[r, j] == x[r, j]
            fixed_lo[r * nj + j] = fixed_hi[r * nj + j] = float(x[r, j])
    objs = np.sort(np.sum(costs * x, axis=1))[::-1]               # :53
    return x, objs


def lexifair_milp(costs):
    """perm[i] = task of agent i (navigation_graph.py:559: ``np.where(x == 1)[1]``)."""
    x, _ = solve_fair_assignment_milp(costs)
    return np.where(x == 1)[1]
