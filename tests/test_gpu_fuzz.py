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


@pytest.mark.parametrize("n,w_veh", [(293, 2.0), (400, 3.4), (511, 1.6), (777, 3.0), (1001, 2.6), (1500, 3.4), (2000, 2.2), (2049, 2.4), (2100, 3.0), (2600, 3.6), (4100, 2.8)])
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
