"""CPU: the oracle's plain-C Lindblad right-hand side (oracle/csrc/fast_lindblad.c) equals the SciPy CSR
restatement of the QuTiP path (oracle/qutip_path.py:lindblad_rhs) entry by entry - global and per-atom
complex drives, every dissipator kind, bad atoms - and drives zvode to the same states."""
import numpy as np
import pytest

from helpers import DEPOL_PAULIS, load_fixture, local_problem, tight_density_matrices, with_anneal_samples
from oracle import fast_lindblad as fl
from oracle import qutip_path as qp

OPS = {
    "dephasing": [(np.sqrt(2 * 0.05), "sigma_rr")],
    "relaxation": [(np.sqrt(0.3), "sigma_gr")],
    "depolarizing": [(np.sqrt(0.2 / 4), "x"), (np.sqrt(0.2 / 4), "y"), (np.sqrt(0.2 / 4), "z")],
    "mixed": [(np.sqrt(2 * 0.05), "sigma_rr"), (np.sqrt(0.3), "sigma_gr"), (0.4, np.array([[0.2, 1j], [0.5, -0.3]]))],
    "none": [],
}


@pytest.mark.parametrize("kind", sorted(OPS))
@pytest.mark.parametrize("n", [1, 3, 5])
def test_c_rhs_equals_scipy_rhs(kind, n):
    prob = local_problem(n, seed=n, duration=61, collapse_ops=OPS[kind], paulis=DEPOL_PAULIS)
    if n >= 3:
        bad = np.zeros(n, bool)
        bad[1] = True
        prob["bad_atoms"] = bad
    f = fl.FastLindblad(qp.build_hamiltonian(prob))
    for t, seed in ((0.0, 1), (0.0313, 2), (0.06, 3)):
        assert f.check(t, seed) < 1e-12


def test_c_rhs_reproduces_the_committed_6_atom_fixture():
    prob, extra = load_fixture("cfg3_tri6_dephasing.npz")
    prob = with_anneal_samples(prob)
    ham = qp.build_hamiltonian(prob)
    f = fl.FastLindblad(ham)
    assert f.check() < 1e-12
    times = np.asarray(extra["eval_times"])[:2]  # 0 -> 0.5 us is enough to tell an integrator apart
    opts = dict(extra["aux"]["options"])
    opts.update(qp.TIGHT)
    psi0 = qp.all_ground_state(6, prob["eigenbasis"])
    ys = qp._zvode(f, np.outer(psi0, psi0.conj()).ravel(), times, opts)
    assert np.max(np.abs(ys[-1].reshape(64, 64) - np.asarray(extra["oracle_states_tight"])[1])) < 1e-10


def test_triangle_storage_round_trip():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    rho = a + a.conj().T
    out = tight_density_matrices({"oracle_states_tight_triu": np.stack([rho[np.triu_indices(8)]])}, 3)
    assert np.array_equal(out[0], rho)
