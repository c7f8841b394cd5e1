"""GPU fuzz against the banded CPU solver (oracle/banded_qp.c, "CPU-B": an independent restatement with its own assembly, band
width and solver, pinned against the dense oracle in tests/test_oracle.py): ring sizes around every switch in the kernels (partial
last chunks of the elimination, the interior point's register-resident passes up to n = 2048, the tridiagonal sweeps in LDS up to n = 2048
and on workspace vectors beyond -- rings of 2049 .. 2208 waypoints were broken in round 3 and covered by no test), one centreline
per problem, several vehicle widths.  The dense oracle needs a minute per N = 2000 problem; this route checks a few hundred
full-size-class problems in seconds."""
import numpy as np
import pytest

from global_racetrajectory_optimization_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,w_veh", [(293, 2.0), (400, 3.4), (511, 1.6), (777, 3.0), (1001, 2.6), (1500, 3.4), (2000, 2.2), (2047, 3.0), (2048, 2.6), (2049, 2.4), (2100, 3.0), (2600, 3.6), (4100, 2.8), (5000, 3.2)])
def test_random_rings_against_banded_cpu_solver(gpu_engine, n, w_veh):
    from oracle import banded_ref
    bsz = 24
    ref, nv, sc = synthetic.oval_batch(bsz, n=n, first=1000 + n, perturb_centreline=True)
    a_cpu, c_cpu, st_cpu, _, _ = banded_ref.solve_batch(ref, nv, sc, 0.5, w_veh)
    al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=ref[k], normvec=nv[k], scaling=sc[k], kappa_bound=0.5, w_veh=w_veh)
                                                 for k in range(bsz)])
    assert np.all(st_cpu == 0) and np.all(np.asarray(st) == 0)
    err = max(float(np.max(np.abs(al[k] - a_cpu[k]))) for k in range(bsz))
    assert err < 1e-7, err                      # two independent solvers on cond ~ 1e10 problems: observed ~1e-10
    assert np.max(np.abs(np.asarray(curv) - c_cpu)) < 1e-8
    assert max(i["kkt_res"] for i in info) < 1e-9


@pytest.mark.parametrize("n,w_veh", [(293, 2.0), (511, 1.6), (777, 3.0), (1001, 2.6), (2047, 3.0), (2049, 2.4)])
def test_kkt_certificate_from_the_dense_oracle_assembly(gpu_engine, n, w_veh):
    """`kkt_res` is self-reported by the engine.  Here the KKT conditions of the QP are checked ON THE
    HOST from the dense-faithful assembly (oracle/tph_ref.assemble_dense: dense 4N x 4N inverse, dense E; H = E'E, f = 2 E'k_ref) --
    stationarity on the free rows, multiplier signs on the active ones, feasibility -- for alpha as the engine returned it."""
    from oracle import tph_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    bsz = 3 if n < 1500 else 1          # (the dense 4N x 4N inverse of the assembly: about 20 s per problem at N = 2049 -- either side of the
                                        #  switch between the LDS sweeps and the long-ring route, round 5)
    ref, nv, sc = synthetic.oval_batch(bsz, n=n, first=5000 + n, perturb_centreline=True)
    al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=ref[k], normvec=nv[k], scaling=sc[k], kappa_bound=0.5, w_veh=w_veh)
                                                 for k in range(bsz)])
    assert np.all(np.asarray(st) == 0)
    for k in range(bsz):
        H, f, E, k_ref, _ = tph_ref.assemble_dense(ref[k], nv[k], cs.build_les_matrix(n, sc[k]))
        x = al[k]
        g = H @ x + f                                     # gradient of 1/2 x'Hx + f'x: the QP exactly as it is handed to quadprog
        lo, hi = -(ref[k, :, 3] - w_veh / 2), ref[k, :, 2] - w_veh / 2
        scale = np.max(np.abs(f))
        assert np.all(x >= lo - 1e-10) and np.all(x <= hi + 1e-10)
        at_lo, at_hi = x <= lo + 1e-9, x >= hi - 1e-9
        free = ~(at_lo | at_hi)
        assert np.max(np.abs(g[free])) < 1e-8 * scale, float(np.max(np.abs(g[free])) / scale)
        assert np.all(g[at_lo] > -1e-8 * scale) and np.all(g[at_hi] < 1e-8 * scale)
        assert np.max(np.abs(k_ref + E @ x)) < 0.5
        assert int(np.count_nonzero(at_lo | at_hi)) == info[k]["n_active_box"]


@pytest.mark.parametrize("n", [48, 71, 72, 73, 100, 143, 144, 145, 200])
def test_rings_around_the_halo_widths_against_the_dense_oracle(gpu_engine, n):
    """The sweeps' LDS arrays carry a halo of 72 waypoints either side (mcq_tri.inc): rings shorter than one halo wrap several times
    inside it, rings below two halos have both copies of a waypoint written by the same few threads.  Dense oracle (these sizes cost
    it nothing), two tracks per size in one ragged launch."""
    from oracle import tph_ref
    from test_emu_kernels import _small_track
    probs, want = [], []
    for m, seed in ((n, n), (n + 5, n + 300)):
        ref, nv, A, sc = _small_track(m, seed=seed)
        a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, 0.5, 2.0)
        probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.5, w_veh=2.0))
        want.append((a_ref, err_ref))
    al, curv, st, info = gpu_engine.solve_batch(probs)
    assert list(st) == [0, 0]
    for k, (a_ref, err_ref) in enumerate(want):
        assert np.max(np.abs(al[k] - a_ref)) < 1e-8, (k, float(np.max(np.abs(al[k] - a_ref))))
        assert abs(curv[k] - err_ref) < 1e-9
