"""A level of the exact-order path solved as STRIPS of solver bands on several streams (csrc/api.hip: smooth_flow_strips;
opt-in, PAPOF_STRIPS = 2 | 3 | 4 -- measured slower than the one-stream order, DESIGN.md §5.1, and therefore off by
default): the same kernels on row ranges, so the results must be the BITS of the default order.

* whole calls with 2, 3 and 4 strips against the default handle (and through it against the oracle / goldens, which the
  default order is tested against in test_gpu_parity.py), at sizes that use the plain and the two-sweeps-per-wave kernel;
* the strip schedule itself (papof_strip_plan): every pair of kernels that touch the same rows of the same buffer from
  different streams, one of them writing, must be ordered by stream order + the one event per iteration and boundary --
  checked on a model of each kernel's reads and writes (the stencil reaches are restated here, not imported);
* the bounded waits: a raised abort word still ends a strip solve with PAPOF_ETIMEOUT.
"""
import itertools

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from papteam_opticalflow_amd import Papof
    g = Papof(0)
    yield g
    g.close()


def _run(g, a, b, levels, sched):
    return g.coarse2fine_flow_sched(a, b, levels, 0.012, 0.75, *sched)[:3]


@pytest.mark.parametrize("strips", ["2", "3", "4"])
def test_strips_give_the_bits_of_the_default_order(gpu, strips, monkeypatch):
    from papteam_opticalflow_amd import Papof
    monkeypatch.setenv("PAPOF_STRIPS", strips)
    g = Papof(0)
    try:
        cut = []
        for res, levels, sched in (("1920", 5, (3, 0, 1, 30, 0)),    # config 4: the two-sweeps-per-wave kernel on level 0
                                   ("960", 3, (4, 1, 1, 30, 3)),     # plain kernel, sweep counts 30 / 33 / 36, 4-6 solves
                                   ("1920", 2, (2, 0, 1, 7, 0))):    # odd sweep count
            a, b = cases.load_pair(res)
            got = _run(g, a, b, levels, sched)
            cut.append(g.last_sor_stats()[0])
            want = _run(gpu, a, b, levels, sched)
            solves = sum(sched[0] + k * sched[1] for k in range(levels))
            assert gpu.last_sor_stats()[0] == solves  # the default order: one launch per solve
            assert cut[-1] > solves                   # ... and this handle really cut levels into strips
            for name, x, y in zip(("vx", "vy", "warpI2"), got, want):
                assert np.array_equal(x, y), "%s strips, %s L%d: %s differs from the default order" % (strips, res, levels, name)
        print("strips=%s: solver launches per call %s" % (strips, cut))
    finally:
        g.close()


def test_strips_sequence_and_graph_replay(monkeypatch):
    """strips inside a captured hipGraph (forked streams joined again) and in sequence mode"""
    from papteam_opticalflow_amd import Papof
    monkeypatch.setenv("PAPOF_STRIPS", "2")
    frames = [cases.load_frame_u8("960", i) for i in (1, 2, 1)]  # the 960 fixture holds two frames
    g = Papof(0)
    monkeypatch.delenv("PAPOF_STRIPS")
    ref = Papof(0)
    try:
        want = [ref.coarse2fine_flow_u8(frames[i], frames[i + 1], 2)[:3] for i in range(2)]
        g.set_graph_mode(True)
        for call in range(4):  # eager, capture, replay, replay
            got = g.coarse2fine_flow_u8(frames[0], frames[1], 2)
            assert all(np.array_equal(x, y) for x, y in zip(got[:3], want[0])), call
        g.set_graph_mode(False)
        g.seq_reset()
        assert g.seq_push(frames[0], 2) is None
        for i in (1, 2):
            out = g.seq_push(frames[i], 2)
            assert all(np.array_equal(x, y) for x, y in zip(out[:3], want[i - 1])), i
    finally:
        g.close()
        ref.close()


@pytest.mark.parametrize("h,w,n_sor,split,delay", [(1080, 1920, 7, 9, 0), (1080, 1920, 9, 9, 100), (1080, 1920, 11, 5, 300),
                                                   (1080, 1920, 3, 9, 0), (1080, 1920, 33, 12, 100), (1080, 1920, 30, 9, 50),
                                                   (810, 1440, 9, 7, 0), (810, 1440, 30, 3, 200), (607, 1080, 12, 5, 0)])
def test_one_solve_in_two_launches_equals_the_whole_solve(gpu, h, w, n_sor, split, delay):
    """the solver alone: bands < split on one stream, the others `delay` microseconds later on another -- every cell of
    the (du, dv) planes, intermediate sweeps included, as the one-launch solve leaves it (plain and two-sweeps-per-wave
    kernel, even and ODD sweep counts: the odd ones used to go wrong here, see sor.hip f_step)"""
    mm, nb = gpu.test_sor_strips(h, w, n_sor, split, reps=6, delay_us=delay)
    print("%dx%d sweeps %d: %d bands cut at %d, %d mismatching cells" % (w, h, n_sor, nb, split, mm))
    assert mm == 0


def test_strips_raised_abort_word_reports_timeout(monkeypatch):
    from papteam_opticalflow_amd import Papof
    from papteam_opticalflow_amd.capi import PapofError
    monkeypatch.setenv("PAPOF_STRIPS", "2")
    g = Papof(0)
    try:
        a, b = cases.load_pair("960")
        want = _run(g, a, b, 1, (2, 0, 1, 10, 0))
        monkeypatch.setenv("PAPOF_SOR_INJECT_ABORT", "1")
        with pytest.raises(PapofError) as err:
            _run(g, a, b, 1, (2, 0, 1, 10, 0))
        assert err.value.code == -5
        monkeypatch.delenv("PAPOF_SOR_INJECT_ABORT")
        got = _run(g, a, b, 1, (2, 0, 1, 10, 0))
        assert all(np.array_equal(x, y) for x, y in zip(got, want))
    finally:
        g.close()


# ---- the schedule: a model of what each kernel reads and writes -------------------------------------------------------
def _accesses(plan, n, s, H, BR, koff, n_outer):
    """[(stage, [(buffer, lo, hi, 'r' | 'w'), ...]), ...] of strip s before / in solve n, in stream order.
    Rows are image rows; 'du' is indexed by solver band.  u planes alternate: iteration n >= 1 reads u[(n-1) % 2] and writes u[n % 2]."""
    band, rU, rP, rS, rA = plan[n][s]
    band1, rU1, rP1, rS1, rA1 = plan[n][s + 1]
    clip = lambda lo, hi: (max(0, lo), min(H, hi))
    ops = []
    if n == 0:
        unew = "u0"
        ops.append(("init", [("u0", rU, rU1, "w"), ("warp", rU, rU1, "w")]))              # resize + warp: pointwise in the level
        ops.append(("phi", [("u0",) + clip(rP, rP1 + 1) + ("r",), ("phi", rP, rP1, "w")]))  # forward differences: row i + 1
    else:
        uold, unew = "u%d" % ((n - 1) % 2), "u%d" % (n % 2)
        b_lo, b_hi = (rU + koff) // BR, (min(H - 1, rU1) + koff) // BR + 1                 # bands that hold rows rU .. rU1 (phi: i + 1)
        ops.append(("update", [("du", b_lo, b_hi, "r"), (uold,) + clip(rU, rU1 + 1) + ("r",), (unew, rU, rU1, "w"),
                               ("warp", rU, rU1, "w"), ("phi", rU, rU1, "w")]))
    if n < n_outer:
        ops.append(("smooth", [("warp",) + clip(rS - 2, rS1 + 2) + ("r",), ("blend", rS, rS1, "w")]))       # 5 x 5
        ops.append(("assemble", [("blend",) + clip(rA - 2, rA1 + 2) + ("r",), ("phi",) + clip(rA - 1, rA1) + ("r",),
                                 (unew,) + clip(rA - 1, rA1 + 1) + ("r",), ("coef", rA, rA1, "w")]))          # 5-point, Laplacian
        # the strip's bands: every coefficient row a band touches in any sweep (it climbs koff + 1 rows; one ghost row
        # above), its own (du, dv) blocks, and the block above (the strip above writes it in the same solve: concurrent by
        # design, ordered by the progress counters)
        ops.append(("solve", [("coef",) + clip(BR * band - koff - 2, BR * band1) + ("r",), ("du", band, band1, "w")]))
    return ops


@pytest.mark.parametrize("H,W,n_sor,n_outer,want", [(1080, 1920, 30, 3, 2), (1080, 1920, 30, 7, 3), (810, 1440, 33, 8, 4),
                                                    (607, 1080, 36, 9, 2), (1080, 1920, 57, 3, 2), (1080, 1920, 64, 3, 2),
                                                    (540, 960, 30, 3, 2)])
def test_strip_schedule_orders_every_conflicting_pair(gpu, H, W, n_sor, n_outer, want):
    p = gpu.strip_plan(H, W, n_sor, n_outer, want)
    S, BR, koff, nb = p["S"], p["band_rows"], p["koff"], p["bands"]
    if S == 1:
        pytest.skip("%dx%d with %d sweeps x %d solves is not cut (bands %d)" % (W, H, n_sor, n_outer, nb))
    plan = p["plan"]
    assert len(plan) == n_outer + 1 and all(len(r) == S + 1 for r in plan)
    for n in range(n_outer + 1):
        assert plan[n][0] == (0, 0, 0, 0, 0) and plan[n][S][1:] == (H, H, H, H) and plan[n][S][0] == nb
        for s in range(1, S + 1):
            assert all(x > y for x, y in zip(plan[n][s], plan[n][s - 1])), "strips must be non-empty in every stage"
    # nodes in enqueue order: (n, s, k-th kernel); edges: stream order, and strip s waits at the START of iteration n for
    # the event strip s-1 recorded BEHIND its non-solver kernels of iteration n (in front of its solve)
    nodes, order, prev_in_stream, edges = [], {}, {}, set()
    for n in range(n_outer + 1):
        for s in range(S):
            ops = _accesses(plan, n, s, H, BR, koff, n_outer)
            for k, (stage, acc) in enumerate(ops):
                node = (n, s, stage)
                order[node] = len(nodes)
                nodes.append((node, acc))
                if s in prev_in_stream:
                    edges.add((prev_in_stream[s], node))
                prev_in_stream[s] = node
                if k == 0 and s > 0:  # the wait
                    last_ns = [st for st, _ in _accesses(plan, n, s - 1, H, BR, koff, n_outer) if st != "solve"][-1]
                    edges.add(((n, s - 1, last_ns), node))
    idx = {node: i for i, (node, _) in enumerate(nodes)}
    N = len(nodes)
    reach = np.zeros((N, N), dtype=bool)
    for a, b in edges:
        reach[idx[a], idx[b]] = True
    for k in range(N):  # enqueue order is a topological order: one forward pass per node closes the relation
        reach[:, :] |= np.outer(reach[:, k], reach[k, :])
    bad = []
    for (i, (na, accs_a)), (j, (nb_, accs_b)) in itertools.combinations(enumerate(nodes), 2):
        if na[1] == nb_[1]:
            continue  # same stream
        if na[2] == "solve" and nb_[2] == "solve" and na[0] == nb_[0]:
            continue  # strips of one solve: the progress counters
        for (ba, la, ha, ma), (bb, lb, hb, mb) in itertools.product(accs_a, accs_b):
            if ba == bb and "w" in (ma, mb) and la < hb and lb < ha and not (reach[i, j] or reach[j, i]):
                bad.append((na, nb_, ba, (la, ha, ma), (lb, hb, mb)))
    assert not bad, "unordered conflicting accesses: %s" % bad[:5]
    print("%dx%d sweeps %d solves %d: S = %d, %d kernels, every conflicting pair ordered" % (W, H, n_sor, n_outer, S, N))
