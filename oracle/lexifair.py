"""TEST INFRASTRUCTURE ONLY — oracle for the lexicographic-fair goal assignment.

Restates ``solve_fair_assignment`` (reference ``marl_fair_assign.py:16-55``).  The
reference hands an iterated min-max MILP to Gurobi through pyomo (``gurobipy==10.0.2``
in ``requirements.txt``; pyomo unpinned).  Neither is vendored under ``/root/reference``
nor installed in this image, and the reference ships no test that records a solver
output, so this file is **parity unpinned** at the solver: it restates the published
mathematical definition instead.

Definition (reference ``marl_fair_assign.py:32-52``): repeat N times { minimise z subject
to ``cost_helper[i,j] * x[i,j] <= z`` over perfect assignments that keep the rows fixed in
earlier rounds; locate the entry whose cost equals z* (``np.argmin(|costs - z*|)``); zero
that entry of ``cost_helper``; fix its row }.  For distinct costs that is the unique
assignment whose descending-sorted cost vector is lexicographically minimal.

Two independent implementations are kept so they can check each other:
``lexifair_bruteforce`` (all N! permutations, N <= 8) and ``lexifair`` (polynomial:
bottleneck threshold by binary search over the sorted costs + augmenting-path matching,
then fix the bottleneck edge and recurse on the remaining rows/columns).
"""
from itertools import permutations

import numpy as np


def lexifair_bruteforce(costs):
    """Return perm (perm[i] = task of agent i) minimising the descending-sorted cost vector."""
    costs = np.asarray(costs, dtype=np.float64)
    n = costs.shape[0]
    assert costs.shape == (n, n) and n <= 8
    best, best_key = None, None
    rows = np.arange(n)
    for p in permutations(range(n)):
        key = tuple(sorted(costs[rows, list(p)], reverse=True))
        if best_key is None or key < best_key:
            best, best_key = p, key
    return np.array(best, dtype=np.int64)


def _perfect_matching(allowed, rows, cols):
    """Kuhn augmenting paths on the bipartite graph ``allowed[r, c]`` restricted to rows/cols.

    Returns dict row->col if a perfect matching of ``rows`` exists, else None.
    """
    match_col = {}

    def try_row(r, seen):
        for c in cols:
            if allowed[r, c] and c not in seen:
                seen.add(c)
                if c not in match_col or try_row(match_col[c], seen):
                    match_col[c] = r
                    return True
        return False

    for r in rows:
        if not try_row(r, set()):
            return None
    return {r: c for c, r in match_col.items()}


def lexifair(costs):
    """Polynomial lexicographic-bottleneck assignment.  Returns perm (int64, shape (N,)).

    Ties are broken by the total order (cost, row, col): the costs are replaced by their ranks
    in that order, which keeps every strict comparison and makes all entries distinct (so the
    "bottleneck edge is in every optimal matching" step is exact); the device kernel uses the
    same order.  The sorted cost vector of the result is lexicographically minimal either way.
    """
    costs = np.asarray(costs, dtype=np.float64)
    n = costs.shape[0]
    assert costs.shape == (n, n)
    order = np.lexsort((np.arange(n * n), costs.ravel()))
    ranks = np.empty(n * n, dtype=np.float64)
    ranks[order] = np.arange(n * n)
    costs = ranks.reshape(n, n)
    rows = list(range(n))
    cols = list(range(n))
    perm = np.full(n, -1, dtype=np.int64)
    while rows:
        sub = costs[np.ix_(rows, cols)]
        vals = np.unique(sub)  # sorted ascending
        lo, hi = 0, len(vals) - 1
        while lo < hi:  # smallest threshold admitting a perfect matching
            mid = (lo + hi) // 2
            if _perfect_matching(costs <= vals[mid], rows, cols) is not None:
                hi = mid
            else:
                lo = mid + 1
        t = vals[lo]
        # the entry with cost t is in every perfect matching at threshold t (distinct costs);
        # with ties pick the first (row-major) tied entry that some perfect matching contains.
        allowed = costs <= t
        fixed = None
        for r in rows:
            for c in cols:
                if costs[r, c] == t:
                    rr = [x for x in rows if x != r]
                    cc = [x for x in cols if x != c]
                    if not rr or _perfect_matching(allowed, rr, cc) is not None:
                        fixed = (r, c)
                        break
            if fixed is not None:
                break
        assert fixed is not None
        r, c = fixed
        perm[r] = c
        rows.remove(r)
        cols.remove(c)
    return perm


def solve_fair_assignment(costs):
    """Drop-in for the reference signature: returns (x one-hot (N,N) int, objs sorted desc)."""
    costs = np.asarray(costs, dtype=np.float64)
    n = costs.shape[0]
    perm = lexifair(costs)
    x = np.zeros((n, n), dtype=int)
    x[np.arange(n), perm] = 1
    objs = np.sort(np.sum(costs * x, axis=1))[::-1]
    return x, objs
