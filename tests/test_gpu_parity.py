"""Parity of the HIP path (through the C ABI, libsphhip.so) against the CPU oracle and the golden
vectors generated from the reference's own serial step.

Bars:
  * integer work -- cell keys, sorted order, cell start/end, neighbour counts: bit-exact;
  * floating point vs the oracle's cell-list functions (same per-particle summation order,
    snapshot velocity semantics): bit-exact -- the kernels evaluate the reference's expressions
    in IEEE binary32 without contraction;
  * floating point vs the reference's serial path (all-pairs j order, Gauss-Seidel velocity
    update, SPH/sph.cpp:96-97): stated tolerances below (summation order + snapshot semantics).
"""
import os

import numpy as np
import pytest

import orc
from sph_amd import scenes
from sph_amd.params import PARTICLE_DTYPE, SphParams
from sph_amd.solver import Solver

pytestmark = pytest.mark.gpu

O = orc.Oracle()
P = O.default_params()


def G(name):
    return np.load(os.path.join(orc.GOLDEN, name + ".npz"), allow_pickle=False)


def random_state(n, lo, hi, seed, vel=1.0, invalid=0.0):
    rng = np.random.default_rng(seed)
    a = np.zeros(n, PARTICLE_DTYPE)
    a["pos"] = rng.uniform(lo, hi, (n, 2)).astype(np.float32)
    a["vel"] = rng.normal(0, vel, (n, 2)).astype(np.float32)
    a["density"] = 1000.0
    a["valid"] = rng.random(n) >= invalid
    return a


@pytest.fixture(scope="module")
def solver():
    s = Solver(P, n_max=1 << 16)
    yield s
    s.close()


def test_roundtrip(solver):
    a = random_state(5000, 0.05, 1.95, 1, invalid=0.1)
    a["density"] = np.random.default_rng(2).uniform(1, 2000, len(a)).astype(np.float32)
    a["pressure"] = np.random.default_rng(3).uniform(0, 1e5, len(a)).astype(np.float32)
    solver.upload(a)
    assert solver.n == len(a)
    assert orc.same_bits(solver.download(), a)
    solver.buildNeighbors()  # reorders internally; must stay invisible
    assert orc.same_bits(solver.download(), a)


def test_integrate_walls_golden(solver):
    g = G("f7_walls")
    solver.upload(g["state"])
    solver.integrate()
    assert orc.same_bits(solver.download(), g["after_integrate"])


@pytest.mark.parametrize("n,lo,hi,seed", [(1, 0.5, 0.6, 0), (2, 0.5, 0.6, 1), (777, 0.05, 1.95, 2), (4096, 0.05, 1.95, 3),
                                          (3000, 0.9, 1.1, 4), (50000, 0.05, 1.95, 5)])
def test_cells_bit_exact(solver, n, lo, hi, seed):
    a = random_state(n, lo, hi, seed)
    solver.upload(a)
    solver.buildNeighbors()
    keys, order, cs, ce = solver.cells()
    okeys, oorder, oskeys, ocs = O.build_cells(P, a)
    assert np.array_equal(keys, oskeys)
    assert np.array_equal(order, oorder)
    assert np.array_equal(cs, ocs[:-1])
    assert np.array_equal(ce, ocs[1:])


def test_cells_after_shuffled_history(solver):
    """The sorted order must be canonical (key, id) whatever order the state was in before."""
    a = scenes.lattice(4096)
    solver.upload(a)
    solver.stepSimulation(nsteps=3)  # internal order is now the order of an earlier sort
    b = solver.download()
    solver.buildNeighbors()
    keys, order, cs, ce = solver.cells()
    okeys, oorder, oskeys, ocs = O.build_cells(P, b)
    assert np.array_equal(keys, oskeys) and np.array_equal(order, oorder)
    assert np.array_equal(cs, ocs[:-1]) and np.array_equal(ce, ocs[1:])


@pytest.mark.parametrize("name,key", [("f2_serial_traj_n1024", "s1"), ("f2_serial_traj_n1024", "s100"),
                                      ("f5_dense_block", "state")])
def test_density_and_counts(solver, name, key):
    a = G(name)[key].copy()
    solver.upload(a)
    solver.computeDensityPressure()
    got = solver.download()
    want, cnt = O.density_cells(P, a.copy())
    assert orc.same_bits(got, want)  # bit-exact vs oracle (same summation order)
    counts = solver.neighbor_counts()
    assert np.array_equal(counts, cnt)
    gk = {"s1": "n1024_s1", "s100": "n1024_s100", "state": "dense"}[key]
    assert np.array_equal(counts, G("f8_neighbor_counts")[gk])  # reference predicate, all pairs
    # vs the reference's all-pairs sum: summation order only.  rel tol 4e-6 (fp32, <= ~500 terms)
    ref = O.density_allpairs(P, a.copy())
    assert np.allclose(got["density"], ref["density"], rtol=4e-6, atol=0)


@pytest.mark.parametrize("mouse,strength", [((0.0, 0.0), 0.0), ((1.0, 1.0), 50000.0), ((0.7, 1.3), -50000.0)])
def test_forces_bit_exact_vs_oracle(solver, mouse, strength):
    for a in (G("f4_phases_n1024")["after_density"], G("f5_dense_block")["after_density"],
              G("f9_invalid")["state"]):
        solver.upload(a)
        solver.computeForces(mouse, strength)
        got = solver.download()
        want = O.forces_cells(P, a.copy(), mouse, strength)
        assert orc.same_bits(got, want)


def test_forces_vs_reference_serial_golden(solver):
    """One force pass against the reference's serial output.  Differences: j-summation order and
    Gauss-Seidel vs snapshot neighbour velocities.  Acceleration = dv/DT; tolerance relative to the
    largest acceleration in the scene (the pair terms nearly cancel on a lattice)."""
    g = G("f4_phases_n1024")
    solver.upload(g["after_density"])
    solver.computeForces()
    got = solver.download()
    dt = np.float32(0.0005)
    acc_got = (got["vel"] - g["after_density"]["vel"]) / dt
    acc_ref = (g["after_forces"]["vel"] - g["after_density"]["vel"]) / dt
    scale = np.abs(acc_ref).max()
    # vs serial (Gauss-Seidel velocities, sph.cpp:96-97): the viscosity term sees updated v_j for j < i
    assert np.abs(acc_got - acc_ref).max() <= 1e-3 * scale
    # vs the same all-pairs sum with snapshot velocities (MPI_sph.cpp semantics): summation order only
    snap = O.forces_allpairs(P, g["after_density"].copy(), flags=orc.SNAPSHOT)
    acc_snap = (snap["vel"] - g["after_density"]["vel"]) / dt
    assert np.abs(acc_got - acc_snap).max() <= 2e-5 * scale


@pytest.mark.parametrize("n,steps", [(1024, 10), (1000, 3), (2704, 2), (4096, 3)])
def test_step_bit_exact_vs_oracle(solver, n, steps):
    a = scenes.lattice(n)
    solver.upload(a)
    solver.stepSimulation(nsteps=steps)
    got = solver.download()
    want = O.step_cells(P, a.copy(), steps)
    assert orc.same_bits(got, want)


def test_phases_equal_fused_step(solver):
    a = G("f2_serial_traj_n1024")["s50"]
    solver.upload(a)
    solver.stepSimulation((0.8, 0.9), 50000.0, 4)
    fused = solver.download()
    solver.upload(a)
    for _ in range(4):
        solver.buildNeighbors()
        solver.computeDensityPressure()
        solver.computeForces((0.8, 0.9), 50000.0)
        solver.integrate()
    assert orc.same_bits(solver.download(), fused)


def test_trajectory_vs_reference_serial_golden(solver):
    """N-step parity with the reference's serial path on its default scene (N=1024, main.cpp:68).
    Stated tolerance (fp32; snapshot-vs-Gauss-Seidel gap dominates):
      10 steps : |dx| <= 2.4e-7 (2 ulp at x~2), |dv| <= 5e-5, density rel 5e-6
      100 steps: |dx| <= 2e-5,  |dv| <= 1e-3 (v_max 0.55),  density rel 1e-4
    Measured with the oracle (bit-identical to the HIP path): 1.2e-7 / 2.2e-5 / 1.3e-6 at 10 steps,
    5.8e-6 / 2.2e-4 / 4.4e-5 at 100 steps.
    """
    g = G("f2_serial_traj_n1024")
    solver.upload(g["s0"])
    solver.stepSimulation(nsteps=10)
    s10 = solver.download()
    assert np.abs(s10["pos"] - g["s10"]["pos"]).max() <= 2.4e-7
    assert np.abs(s10["vel"] - g["s10"]["vel"]).max() <= 5e-5
    assert np.allclose(s10["density"], g["s10"]["density"], rtol=5e-6)
    solver.stepSimulation(nsteps=90)
    s100 = solver.download()
    assert np.abs(s100["pos"] - g["s100"]["pos"]).max() <= 2e-5
    assert np.abs(s100["vel"] - g["s100"]["vel"]).max() <= 1e-3
    assert np.allclose(s100["density"], g["s100"]["density"], rtol=1e-4)
    # and bit-exact against the snapshot-semantics oracle in the GPU's summation order
    want = O.step_cells(P, g["s0"].copy(), 100)
    assert orc.same_bits(s100, want)


def test_reference_order_mode_equals_reference_serial(solver):
    """The sequential reference-order mode (all pairs in ascending j, in-place velocity update) against the
    golden vectors the compiled reference produced: bit-exact, including the mouse force."""
    g = G("f2_serial_traj_n1024")
    solver.upload(g["s0"])
    done = 0
    for k in (1, 10, 50, 100, 200):
        solver.stepSimulationReferenceOrder(nsteps=k - done)
        done = k
        assert orc.same_bits(solver.download(), g["s%d" % k]), k
    a = G("f9_invalid")["state"]
    solver.upload(a)
    solver.stepSimulationReferenceOrder(nsteps=1)
    assert orc.same_bits(solver.download(), G("f9_invalid")["after_1step"])
    for mouse, strength in (((0.8, 0.9), 50000.0), ((1.2, 0.4), -50000.0)):
        b = g["s50"]
        solver.upload(b)
        solver.stepSimulationReferenceOrder(mouse, strength, 3)
        assert orc.same_bits(solver.download(), O.step_allpairs(P, b.copy(), 3, mouse, strength, flags=orc.GAUSS_SEIDEL))
    with pytest.raises(Exception):
        big = scenes.lattice(5000)
        solver.upload(big)
        solver.stepSimulationReferenceOrder(nsteps=1)


def test_invalid_particles_golden(solver):
    g = G("f9_invalid")
    solver.upload(g["state"])
    solver.stepSimulation(nsteps=1)
    got = solver.download()
    inv = g["state"]["valid"] == 0
    # invalid particles keep position and velocity (sph.cpp:59,103) -- bit-exact
    assert np.array_equal(got["pos"][inv], g["state"]["pos"][inv])
    assert np.array_equal(got["vel"][inv], g["state"]["vel"][inv])
    assert np.array_equal(got["valid"], g["state"]["valid"])
    assert np.abs(got["pos"] - g["after_1step"]["pos"]).max() <= 1.2e-7
    assert np.abs(got["vel"] - g["after_1step"]["vel"]).max() <= 1e-5


@pytest.mark.parametrize("tile_cells,cap", [(1, 64), (3, 200), (16, 1152), (64, 4608), (5, 64)])
def test_tuning_does_not_change_results(solver, tile_cells, cap):
    """Tile height and LDS budget (including the global-memory fallback when a tile's candidates
    exceed the budget) must not change a single bit."""
    a = G("f5_dense_block")["state"]
    want = O.step_cells(P, a.copy(), 2)
    solver.set_tuning(tile_cells, cap)
    try:
        solver.upload(a)
        solver.stepSimulation(nsteps=2)
        assert orc.same_bits(solver.download(), want)
    finally:
        solver.set_tuning(12, 1024)


def test_close_and_coincident_pairs(solver):
    """Pairs at distance 0, 1e-9 .. 2e-6 (the reference's `r_len < 1e-6` guards, sph.cpp:16,30; the
    force kernel clamps such distances before its rsq-based root): bit-exact through both neighbour
    paths of the force pass -- masks from the density pass (step) and its own scan (forces alone)."""
    a = random_state(3000, 0.3, 1.5, 77, vel=0.3)
    twins = a[:600].copy()
    off = np.array([0.0, 1e-9, 1e-8, 1e-7, 5e-7, 9.9e-7, 1e-6, 1.1e-6, 2e-6, 1e-5], np.float32)
    twins["pos"][:, 0] += off[np.arange(600) % len(off)]
    twins["pos"][300:, 1] += off[(np.arange(300) // 3) % len(off)]
    both = np.zeros(len(a) + len(twins), PARTICLE_DTYPE)  # (np.concatenate drops the padded dtype)
    both[: len(a)] = a
    both[len(a):] = twins
    a = both
    solver.upload(a)
    solver.stepSimulation(nsteps=1)
    want = O.step_cells(P, a.copy(), 1)
    assert not np.isnan(want["pos"][600:3000]).any()  # particles without a twin stay finite
    got = solver.download()
    # Pairs a little farther apart than 1e-6 overflow the reference's Lennard-Jones term (sr^12 = inf,
    # then inf * 0 in the force cap, sph.cpp:32-37,86-90): NaN on both sides, but x86 and gfx950 differ in
    # the sign of the default NaN.
    assert np.isnan(want["vel"]).any()
    assert orc.same_bits_nan(got, want)
    solver.stepSimulation(nsteps=1)  # NaN positions from here on (r2 = NaN passes `r2 >= H*H`, sph.cpp:66)
    assert orc.same_bits_nan(solver.download(), O.step_cells(P, want.copy(), 1))
    b, _ = O.density_cells(P, a.copy())
    solver.upload(b)
    solver.computeForces((0.0, 0.0), 0.0)
    assert orc.same_bits_nan(solver.download(), O.forces_cells(P, b.copy(), (0.0, 0.0), 0.0))


def test_empty_and_tiny(solver):
    solver.upload(np.zeros(0, PARTICLE_DTYPE))
    solver.stepSimulation(nsteps=2)
    assert solver.n == 0
    for n in (1, 2, 3):
        a = scenes.lattice(n)
        solver.upload(a)
        solver.stepSimulation(nsteps=3)
        assert orc.same_bits(solver.download(), O.step_cells(P, a.copy(), 3))


def test_scaled_scene_miniatures():
    """Miniatures of the bench scenes (same spacing H/4, scaled walls): bit-exact vs the oracle,
    neighbour counts vs the all-pairs reference predicate."""
    for params, a in (scenes.dam_break(48), scenes.uniform_box(56)):
        with Solver(params, n_max=len(a)) as s:
            s.upload(a)
            s.computeDensityPressure()
            assert np.array_equal(s.neighbor_counts(), O.neighbor_counts_allpairs(params, a))
            s.upload(a)
            s.stepSimulation(nsteps=5)
            got = s.download()
        want = O.step_cells(params, a.copy(), 5)
        assert orc.same_bits(got, want)
        ref = O.step_allpairs(params, a.copy(), 5, flags=orc.GAUSS_SEIDEL)  # reference serial semantics
        assert np.abs(got["pos"] - ref["pos"]).max() <= 1e-6


@pytest.mark.parametrize("tp", ["1", "0"])
@pytest.mark.parametrize("n,cap,steps", [(8192, 0, 4), (8192, 200, 3), (16384, 0, 3), (16384, 448, 2)])
def test_dense_reference_box_bit_exact(monkeypatch, n, cap, steps, tp):
    """The reference's own scene at a large N (main.cpp's fixed 1.9 x 1.9 box: 50-100 particles per cell,
    stencil ranges of 400-900 candidates): windows of long ranges, candidate sequences staged chunk by
    chunk, and the tile height the host adapts after the first step must all keep every bit.  Neighbour
    counts against the all-pairs predicate of the reference.  tp: the term-parallel dense force kernel
    (a wave per particle, ordered sum over the lanes) or the lane-per-particle one."""
    monkeypatch.setenv("SPHHIP_DENSE_TP", tp)
    a = scenes.lattice(n)
    with Solver(P, n_max=n) as s:
        if cap:
            s.set_tuning(0, cap)
        s.upload(a)
        s.computeDensityPressure()
        assert np.array_equal(s.neighbor_counts(), O.neighbor_counts_allpairs(P, a))
        s.upload(a)
        for _ in range(steps):  # (one call per step: the host sees the tile count of the step before)
            s.stepSimulation(nsteps=1)
            s.synchronize()
        got = s.download()
    want = O.step_cells(P, a.copy(), steps)
    assert orc.same_bits(got, want)


def test_graph_replay_equals_plain_steps(monkeypatch):
    """Small scenes replay 8 captured steps as one hipGraph (ctx.hip: graph_steps).  Same bits as the
    plain launches (SPHHIP_GRAPH=0) and as the oracle -- across a change of the mouse arguments (new
    capture), a new upload with another particle count, and step counts that are not multiples of 8."""
    runs = {}
    for env in ("0", "1"):
        monkeypatch.setenv("SPHHIP_GRAPH", env)
        got = []
        with Solver(P, n_max=4096) as s:
            s.upload(scenes.lattice(1024))
            s.stepSimulation(nsteps=61)
            got.append(s.download())
            s.stepSimulation((1.0, 1.0), 50000.0, 45)
            got.append(s.download())
            s.upload(scenes.lattice(2704))
            s.stepSimulation(nsteps=40)
            s.stepSimulation(nsteps=3)
            got.append(s.download())
        runs[env] = got
    for x, y in zip(runs["0"], runs["1"]):
        assert orc.same_bits(x, y)
    want = O.step_cells(P, scenes.lattice(1024), 61)
    assert orc.same_bits(runs["1"][0], want)
    want = O.step_cells(P, want, 45, mouse=(1.0, 1.0), strength=50000.0)
    assert orc.same_bits(runs["1"][1], want)


@pytest.mark.parametrize("seed", range(30))
def test_random_scenes_fuzz(seed):
    """Random scenes against the oracle, bit for bit: random particle counts, clustered positions (dense
    cells, long stencil ranges: the dense kernels and their chunking), particles outside the walls (clamped
    cell coordinates), random velocities, a few invalid particles, optional mouse force, random LDS budget."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2500, 5000, 9000]))
    if seed in (20, 25):
        n = 50000  # enough dense particles for the lane-per-particle dense force kernel
    a = np.zeros(n, scenes.PARTICLE_DTYPE)
    centres = rng.uniform(0.1, 1.9, size=(int(rng.integers(1, 6)), 2))
    spread = float(rng.choice([0.002, 0.02, 0.1, 0.5])) if n < 50000 else 0.15
    pos = centres[rng.integers(0, len(centres), n)] + rng.normal(0.0, spread, size=(n, 2))
    if seed % 3 == 0:
        pos[rng.integers(0, n, max(1, n // 50))] += rng.uniform(-3.0, 3.0, size=(max(1, n // 50), 2))  # outside the walls
    a["pos"] = pos.astype(np.float32)
    a["vel"] = rng.normal(0.0, 1.0, size=(n, 2)).astype(np.float32)
    a["density"] = 1000.0
    a["valid"] = 1
    if seed % 4 == 1:
        a["valid"][rng.integers(0, n, max(1, n // 20))] = 0
    mouse, strength = ((1.0, 1.0), 50000.0) if seed % 2 else ((0.0, 0.0), 0.0)
    steps = 3
    want = O.step_cells(P, a.copy(), steps, mouse=mouse, strength=strength)
    with Solver(P, n_max=n) as s:
        cap = int(rng.choice([0, 0, 64, 150, 448]))
        if cap:
            s.set_tuning(0, cap)
        s.upload(a)
        for _ in range(steps):
            s.stepSimulation(mouse, strength, 1)
            s.synchronize()
        got = s.download()
    assert orc.same_bits_nan(got, want)


@pytest.mark.parametrize("seed", range(10))
def test_scaled_scenes_fuzz(seed):
    """Dam-break / box miniatures with jitter and random velocities, 1-40 steps in ONE call (graph replay and
    the one-workgroup grid build for the small ones), against the oracle."""
    rng = np.random.default_rng(5000 + seed)
    cols = int(rng.choice([8, 20, 33, 64, 100, 150]))
    params, a = scenes.dam_break(cols, int(rng.integers(4, cols + 1))) if seed % 2 else scenes.uniform_box(cols)
    a["pos"] += rng.uniform(-0.01, 0.01, size=a["pos"].shape).astype(np.float32)
    a["vel"] = rng.normal(0.0, float(rng.choice([0.0, 0.5, 3.0])), size=a["vel"].shape).astype(np.float32)
    steps = int(rng.choice([1, 7, 18, 27, 40]))
    want = O.step_cells(params, a.copy(), steps)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=steps)
        got = s.download()
    assert orc.same_bits_nan(got, want)


def test_step_in_two_halves(solver):
    """sphhip_step_begin + sphhip_step_finish = sphhip_step(1), whatever the edge widths; nothing but the
    halo pack may be called in between."""
    from sph_amd.lib import SphHipError

    a = G("f2_serial_traj_n1024")["s50"]
    solver.upload(a)
    solver.stepSimulation((0.8, 0.9), 50000.0, 3)
    want = solver.download()
    for edges in ((3, 3), (0, 5), (1, 0), (0, 0), (40, 40)):
        solver.upload(a)
        for _ in range(3):
            solver.stepBegin((0.8, 0.9), 50000.0, *edges)
            if 0 < sum(edges) < 20:  # (no interior, or no edge: begin did it all)
                with pytest.raises(SphHipError):
                    solver.stepSimulation(nsteps=1)
                with pytest.raises(SphHipError):
                    solver.download()
            solver.stepFinish()
        assert orc.same_bits(solver.download(), want), edges
    solver.stepFinish()  # without a begin: nothing to do


def test_full_size_dam_break_bit_exact():
    """BASELINE configs[1] at full size (1M-particle dam break): the cell-list oracle finishes a step in a
    fraction of a second, so the whole state is compared bit for bit after 3 steps, the neighbour counts
    after the first density pass, and -- size-independent -- the ids must stay a permutation and every
    particle inside the walls."""
    params, a = scenes.dam_break(1024)
    assert len(a) == 1 << 20
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.computeDensityPressure()
        _, want_counts = O.density_cells(params, a.copy())
        assert np.array_equal(s.neighbor_counts(), want_counts)
        s.upload(a)
        s.stepSimulation(nsteps=3)
        got, ids = s.download(with_ids=True)
    assert np.array_equal(np.sort(ids), np.arange(len(a)))
    want = O.step_cells(params, a.copy(), 3)
    assert orc.same_bits(got, want)
    assert got["pos"][:, 0].min() >= params.xmin and got["pos"][:, 0].max() <= params.xmax
    assert got["pos"][:, 1].min() >= params.ymin and got["pos"][:, 1].max() <= params.ymax


def test_full_size_dam_break_through_the_settling_jump():
    """BASELINE configs[1] at full size over 120 steps enqueued in ONE call: around step 76 the settling column crosses
    a cell row and the pair kernels' work list grows by 6 % between two steps (4423 -> 4676 entries), while the host,
    32 to 64 steps ahead (sph_throttle), still sizes the launches from the old length -- the entries beyond the grid
    are reached by striding inside each XCD's chunk (pairs.hip: for_each_tile).  Whole state bit for bit against the
    cell-list oracle (12 s of CPU), and the list did grow."""
    params, a = scenes.dam_break(1024)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=2)
        s.synchronize()
        before = s.tile_stats()["tiles"]
        s.upload(a)
        s.stepSimulation(nsteps=120)
        got, ids = s.download(with_ids=True)
        after = s.tile_stats()["tiles"]
    assert after > before + before // 32, (before, after)
    assert np.array_equal(np.sort(ids), np.arange(len(a)))
    assert orc.same_bits(got, O.step_cells(params, a.copy(), 120))


def test_full_size_uniform_box_bit_exact():
    """BASELINE configs[2] at full size (4M-particle box filled wall to wall, ~50 neighbours): neighbour
    counts and the whole state after 2 steps, bit for bit against the cell-list oracle."""
    params, a = scenes.uniform_box(2048)
    assert len(a) == 1 << 22
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.computeDensityPressure()
        _, want_counts = O.density_cells(params, a.copy())
        assert np.array_equal(s.neighbor_counts(), want_counts)
        assert 45 < want_counts.mean() < 52
        s.upload(a)
        s.stepSimulation(nsteps=2)
        got, ids = s.download(with_ids=True)
    assert np.array_equal(np.sort(ids), np.arange(len(a)))
    assert orc.same_bits(got, O.step_cells(params, a.copy(), 2))


def test_full_size_16m_dam_break_one_step():
    """The scene of BASELINE configs[3] (16M-particle dam break) on one GPU: one step, bit for bit."""
    params, a = scenes.dam_break(4096)
    assert len(a) == 1 << 24
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=1)
        got = s.download()
    assert orc.same_bits(got, O.step_cells(params, a, 1))


def test_full_size_8m_dam_break_one_step():
    """The per-GPU scene of BASELINE configs[4] (weak scaling, 8M particles per GPU: a 2896 x 2896 dam-break column)
    on one GPU: one step bit for bit against the cell-list oracle, and -- size-independent -- the ids stay a
    permutation and every particle stays inside the walls."""
    params, a = scenes.dam_break(2896)
    assert len(a) == 2896 * 2896
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=1)
        got, ids = s.download(with_ids=True)
    assert np.array_equal(np.sort(ids), np.arange(len(a)))
    assert got["pos"][:, 0].min() >= params.xmin and got["pos"][:, 0].max() <= params.xmax
    assert got["pos"][:, 1].min() >= params.ymin and got["pos"][:, 1].max() <= params.ymax
    assert orc.same_bits(got, O.step_cells(params, a, 1))


@pytest.mark.parametrize("kind", ["dam", "box"])
def test_scaled_scenes_tolerance_vs_serial(kind, capsys):
    """Density, acceleration and position against the reference's serial semantics (all pairs in index order,
    in-place velocity update: sph.cpp:41-55,57-100) on miniatures of the bench scenes, after 40 steps of motion.
    Stated tolerances (fp32): density rel 4e-6 (summation order only); acceleration, relative to the largest
    acceleration of the scene, 2e-6 against the same sum with snapshot velocities (summation order only) and 5e-4
    against serial (its viscosity term sees already-updated v_j for j < i); |dx| <= 1e-6 after 5 further steps.
    Measured (printed by the test): dam 4.2e-7 / 1.6e-7 / 1.3e-4 / 1.6e-7, box 3.2e-7 / 1.6e-7 / 9.8e-5 / 2.4e-7."""
    params, a = scenes.dam_break(48) if kind == "dam" else scenes.uniform_box(56)
    a = O.step_cells(params, a.copy(), 40)  # (bit-identical to the HIP path, checked elsewhere)
    dt = np.float32(params.dt)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.computeDensityPressure()
        dens = s.download()
        s.computeForces()
        forced = s.download()
        s.upload(a)
        s.stepSimulation(nsteps=5)
        stepped = s.download()
    ref_d = O.density_allpairs(params, a.copy())
    d_rel = (np.abs(dens["density"] - ref_d["density"]) / ref_d["density"]).max()
    acc = (forced["vel"] - dens["vel"]) / dt
    serial = O.forces_allpairs(params, ref_d.copy(), flags=orc.GAUSS_SEIDEL)
    snap = O.forces_allpairs(params, ref_d.copy(), flags=orc.SNAPSHOT)
    acc_serial = (serial["vel"] - ref_d["vel"]) / dt
    acc_snap = (snap["vel"] - ref_d["vel"]) / dt
    scale = np.abs(acc_serial).max()
    a_serial = np.abs(acc - acc_serial).max() / scale
    a_snap = np.abs(acc - acc_snap).max() / scale
    ref5 = O.step_allpairs(params, a.copy(), 5, flags=orc.GAUSS_SEIDEL)
    dx = np.abs(stepped["pos"] - ref5["pos"]).max()
    with capsys.disabled():
        print("\n[%s miniature, %d particles] density rel %.2e; acceleration / max|a| (%.3g): vs snapshot sum %.2e, vs serial "
              "%.2e; |dx| after 5 steps %.2e" % (kind, len(a), d_rel, scale, a_snap, a_serial, dx))
    assert d_rel <= 4e-6
    assert a_snap <= 2e-6
    assert a_serial <= 5e-4
    assert dx <= 1e-6


def _pile_up(k, seed=5):
    """The reference box with a 4096-particle lattice and k more particles piled up inside one grid cell, in
    shuffled index order."""
    base = scenes.lattice(4096)
    a = np.zeros(len(base) + k, PARTICLE_DTYPE)
    a[: len(base)] = base
    rng = np.random.default_rng(seed)
    blob = a[len(base):]
    blob["pos"][:, 0] = (1.0 + 0.1 * rng.random(k)).astype(np.float32)
    blob["pos"][:, 1] = (1.0 + 0.1 * rng.random(k)).astype(np.float32)
    blob["density"] = 1000.0
    blob["valid"] = 1
    return a[rng.permutation(len(a))].copy()


@pytest.mark.parametrize("rank_big", ["1", "0"])
def test_crowded_cell_pile_up(monkeypatch, rank_big, capsys):
    """20 000 particles in one cell (a corner pile-up late in a run): the in-cell ranking by id sorts such a cell's
    ids in LDS (k_rank_big) instead of comparing every particle with every cell-mate; SPHHIP_RANK_BIG=0 keeps the
    loop.  The sorted order must be the canonical (cell, id) order, one step must equal the oracle bit for bit,
    and a step must finish within 60 ms (measured: about 9 ms, 0.8 ms less with the sort)."""
    import time

    monkeypatch.setenv("SPHHIP_RANK_BIG", rank_big)
    a = _pile_up(20000)
    with Solver(P, n_max=len(a)) as s:
        s.upload(a)
        s.buildNeighbors()
        keys, order, cs, ce = s.cells()
        _, oorder, oskeys, ocs = O.build_cells(P, a)
        assert (ce - cs).max() >= 20000
        assert np.array_equal(keys, oskeys) and np.array_equal(order, oorder)
        assert np.array_equal(cs, ocs[:-1]) and np.array_equal(ce, ocs[1:])
        s.upload(a)
        s.stepSimulation(nsteps=1)
        got = s.download()
        s.synchronize()
        t0 = time.perf_counter()
        s.stepSimulation(nsteps=3)
        s.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
    with capsys.disabled():
        print("\n[pile-up of 20000 in one cell, SPHHIP_RANK_BIG=%s] %.2f ms/step" % (rank_big, ms))
    assert ms < 60.0
    assert orc.same_bits_nan(got, O.step_cells(P, a.copy(), 1))


def test_crowded_cell_beyond_the_lds_sort():
    """40 000 particles in one cell: more ids than k_rank_big sorts in LDS at a time (32768), so that cell is ranked
    in two chunks whose partial ranks add up, next to another crowded cell (3000 particles); the order stays
    canonical."""
    a = _pile_up(40000, seed=9)
    rng = np.random.default_rng(13)
    b = np.zeros(3000, PARTICLE_DTYPE)
    b["pos"][:, 0] = (0.52 + 0.1 * rng.random(3000)).astype(np.float32)
    b["pos"][:, 1] = (0.52 + 0.1 * rng.random(3000)).astype(np.float32)
    b["density"] = 1000.0
    b["valid"] = 1
    both = np.zeros(len(a) + len(b), PARTICLE_DTYPE)
    both[: len(a)] = a
    both[len(a):] = b
    with Solver(P, n_max=len(both)) as s:
        s.upload(both)
        s.buildNeighbors()
        keys, order, cs, ce = s.cells()
    _, oorder, oskeys, ocs = O.build_cells(P, both)
    pops = np.sort(ce - cs)
    assert pops[-1] >= 40000 and pops[-2] >= 3000
    assert np.array_equal(keys, oskeys) and np.array_equal(order, oorder)


def test_reference_box_65536():
    """The reference's own box at its largest harness size (N = 65536, serial_main.cpp:120-147: about 390 particles
    per cell, about 1200 neighbours each): 3 steps bit for bit."""
    a = scenes.lattice(65536)
    with Solver(P, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=3)
        got = s.download()
    assert orc.same_bits(got, O.step_cells(P, a.copy(), 3))


def test_long_horizon_dam_break_bit_exact():
    """800 steps of a 16k-particle dam break: the column collapses, particles reach MAX_VEL, cells fill up to
    twice their initial population (tiles split, second batches of own particles), the grid is re-sorted 800
    times -- and the state still equals the oracle's bit for bit."""
    params, a = scenes.dam_break(128)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=800)
        got = s.download()
        s.computeDensityPressure()
        counts = s.neighbor_counts()
    want = O.step_cells(params, a.copy(), 800)
    assert orc.same_bits(got, want)
    assert np.abs(got["vel"]).max() > 4.9 and counts.max() > 70  # the scene did evolve
    _, want_counts = O.density_cells(params, want.copy())
    assert np.array_equal(counts, want_counts)


def test_division_selftest():
    """exact_div.h (shared reciprocal + FMA corrections) against hipcc's correctly rounded `/`:
    every float of the safe range for 1/x, 2^31 random pairs for a/b.  Bar: zero mismatches."""
    import ctypes

    from sph_amd import lib

    L = lib.load()
    checked = (ctypes.c_ulonglong * 8)()
    bad = (ctypes.c_ulonglong * 8)()
    lib.check(L.sphhip_selftest_division(0, checked, bad))
    assert checked[0] > 1.3e9 and checked[1] >= 1 << 30
    assert list(bad)[:7] == [0] * 7  # [5], [6]: the rsq-based square root and the divisions by it


def test_error_behaviour():
    """Errors come back as negative codes with a message (never exit(), contrast SPH/sph.cuh:17-23) and leave
    the context usable: capacity overflow, calls without particles, diagnostics without their pass."""
    from sph_amd import lib

    with Solver(P, n_max=256) as s:
        with pytest.raises(lib.SphHipError, match="-3"):  # ESTATE: nothing uploaded
            s.stepSimulation(nsteps=1)
        with pytest.raises(lib.SphHipError, match="-4"):  # ECAPACITY
            s.upload(scenes.lattice(257))
        a = scenes.lattice(256)
        s.upload(a)
        with pytest.raises(lib.SphHipError, match="-3"):  # no grid yet
            s.cells()
        with pytest.raises(lib.SphHipError, match="-3"):  # no density pass yet
            s.neighbor_counts()
        with pytest.raises(lib.SphHipError, match="-1"):  # EINVAL
            s.set_tuning(1000, 0)
        with pytest.raises(lib.SphHipError, match="-1"):
            s.stepSimulation(nsteps=-1)
        s.stepSimulation(nsteps=2)  # still usable
        assert orc.same_bits(s.download(), O.step_cells(P, a.copy(), 2))


def test_tile_stats_and_long_tiles():
    """sphhip_get_tile_stats: the pristine dam-break miniature has neither long nor dense entries; the reference box at
    6000 particles (about 35 per cell: stencil ranges of 257..448 candidates) has long entries and no dense ones in the
    force pass, and steps bit-exactly through the two-pass mask walk."""
    params, a = scenes.dam_break(64)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=3)
        st = s.tile_stats()
    assert st["tiles"] > 0 and st["long"] == 0 and st["dense_force"] == 0
    a = scenes.lattice(6000)
    with Solver(P, n_max=len(a)) as s:
        s.upload(a)
        s.stepSimulation(nsteps=4)
        got = s.download()
        st = s.tile_stats()
    assert st["long"] > 0 and st["dense_force"] == 0, st
    assert orc.same_bits(got, O.step_cells(P, a.copy(), 4))


def test_profiling_period_samples_steps():
    """sphhip_set_profiling_period: phases are bracketed in steps 1, 1 + period, ... only; results do not change."""
    params, a = scenes.dam_break(64)
    with Solver(params, n_max=len(a)) as s:
        s.upload(a)
        s.set_profiling_phases(["force"])
        s.set_profiling_period(8)
        s.reset_phase_times()
        s.stepSimulation(nsteps=20)
        s.synchronize()
        ms, k = s.phase_times()["force"]
        assert k == 3 and ms > 0.0  # steps 1, 9, 17
        assert s.phase_times()["density"][1] == 0
        s.set_profiling_period(1)
        s.reset_phase_times()
        s.stepSimulation(nsteps=4)
        s.synchronize()
        assert s.phase_times()["force"][1] == 4
        s.set_profiling(False)
        got = s.download()
    assert orc.same_bits(got, O.step_cells(params, a.copy(), 24))
    with pytest.raises(Exception):
        with Solver(params, n_max=len(a)) as s:
            s.set_profiling_period(0)
