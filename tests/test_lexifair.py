"""Oracle lexifair: polynomial solver vs brute force, KAT-7, reference call-site shapes.  CPU only."""
import numpy as np
import pytest

from oracle.lexifair import lexifair, lexifair_bruteforce, solve_fair_assignment
from oracle.nav_oracle import cost_matrix


def test_kat7_reference_main_example():
    """marl_fair_assign.py:62-63 example (the only concrete input in the reference)."""
    goals = np.array([[0., -0.5], [0.45, -0.5], [0.9, -0.5]])
    agents = np.array([[-0.9, -0.9], [-0.9, 0.], [-0.9, 0.9]])
    costs = cost_matrix(agents, goals)
    assert np.allclose(costs[0], [0.98488578, 1.40801278, 1.84390889])
    x, objs = solve_fair_assignment(costs)
    assert np.array_equal(np.where(x == 1)[1], [2, 1, 0])  # navigation_graph.py:559
    assert np.allclose(objs, [1.8439088915, 1.6643316977, 1.4396180049])


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 6, 7])
def test_polynomial_vs_bruteforce(n):
    rs = np.random.RandomState(n)
    for _ in range(40 if n < 7 else 8):
        a, g = rs.uniform(-1, 1, (n, 2)), rs.uniform(-0.8, 0.8, (n, 2))
        c = cost_matrix(a, g)
        assert np.array_equal(lexifair(c), lexifair_bruteforce(c))


def test_lexicographic_property_n32():
    """No single swap of two agents' goals may improve the sorted-descending cost vector."""
    rs = np.random.RandomState(0)
    c = cost_matrix(rs.uniform(-1, 1, (32, 2)), rs.uniform(-0.8, 0.8, (32, 2)))
    p = lexifair(c)
    assert sorted(p) == list(range(32))
    base = np.sort(c[np.arange(32), p])[::-1]
    for i in range(32):
        for j in range(i + 1, 32):
            q = p.copy(); q[i], q[j] = p[j], p[i]
            alt = np.sort(c[np.arange(32), q])[::-1]
            assert tuple(base) <= tuple(alt)


def test_ties_total_order():
    """With tied costs the reference's result is solver-dependent (Gurobi picks any optimum of each
    round, marl_fair_assign.py:33-39); the oracle and the device define ties by the total order
    (cost, row, col).  The bottleneck (largest assigned cost) is still the brute-force minimum."""
    rs = np.random.RandomState(5)
    for n in (3, 4, 5, 6):
        for _ in range(30):
            c = rs.randint(0, 4, size=(n, n)).astype(np.float64)
            p = lexifair(c)
            assert sorted(p) == list(range(n))
            assert c[np.arange(n), p].max() == c[np.arange(n), lexifair_bruteforce(c)].max()
            eps = c + 1e-9 * np.arange(n * n).reshape(n, n)   # the same order made explicit
            assert np.array_equal(p, lexifair_bruteforce(eps))


@pytest.mark.parametrize('n', [1, 2, 3, 5, 7, 10, 16])
def test_reference_procedure_with_highs_gives_the_same_assignment(n):
    """oracle/lexifair_milp.py restates the reference's PROCEDURE (marl_fair_assign.py:16-55: n rounds of a min-max
    MILP, fix the bottleneck row) and solves the MILPs with HiGHS instead of Gurobi; for distinct costs the result
    must be the assignment the definition-based solvers return -- on the reference's own call-site inputs
    (cdist of agent and goal positions, navigation_graph.py:555) and on its __main__ example."""
    from oracle.lexifair_milp import lexifair_milp, solve_fair_assignment_milp
    rs = np.random.RandomState(100 + n)
    for _ in range(6 if n <= 10 else 2):
        c = cost_matrix(rs.uniform(-1, 1, (n, 2)), rs.uniform(-0.8, 0.8, (n, 2)))
        p = lexifair_milp(c)
        assert np.array_equal(p, lexifair(c))
        if n <= 7:
            assert np.array_equal(p, lexifair_bruteforce(c))
    if n == 3:
        goals = np.array([[0., -0.5], [0.45, -0.5], [0.9, -0.5]])
        agents = np.array([[-0.9, -0.9], [-0.9, 0.], [-0.9, 0.9]])
        x, objs = solve_fair_assignment_milp(cost_matrix(agents, goals))
        assert np.array_equal(np.where(x == 1)[1], [2, 1, 0])
        assert np.allclose(objs, [1.8439088915, 1.6643316977, 1.4396180049])
