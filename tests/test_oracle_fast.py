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


def test_fuzz_oracle_fixtures_match_the_seeded_cases_and_one_case_reintegrates():
    """tests/golden/fuzz_oracle_*.npz (make_fuzz_fixtures.py): every stored ket belongs to the inputs fuzz_case draws today
    (SHA-256 of coords / amp / det / phase), is normalised, and the cheapest 8-atom case re-integrated here with the tight
    oracle reproduces its fixture (the fixtures are outputs of oracle/qutip_path.py, not of the product)."""
    import os
    import sys

    import numpy as np

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from helpers import fuzz_case
    from make_fuzz_fixtures import digest

    from oracle import qutip_path as qp

    counts = {}
    for name in ("fuzz_oracle_small.npz", "fuzz_oracle_12.npz", "fuzz_oracle_13.npz", "fuzz_oracle_14.npz"):
        fx = np.load(os.path.join(here, "golden", name))
        counts[name] = len(fx["seeds"])
        for r, (k, b) in enumerate(fx["state_owner"]):
            over = int(fx["n_override"][k])
            probs, _ = fuzz_case(int(fx["seeds"][k]), None if over < 0 else over)
            assert digest(probs[b]) == str(fx["input_sha256"][r]), (name, int(fx["seeds"][k]), b)
            n = int(fx["state_atoms"][r])
            assert n == probs[b]["n_qudits"]
            ket = fx["states"][r][: 2**n]
            assert abs(np.linalg.norm(ket) - 1.0) < 1e-9 and not np.any(fx["states"][r][2**n:])
    assert counts == {"fuzz_oracle_small.npz": 24, "fuzz_oracle_12.npz": 24, "fuzz_oracle_13.npz": 8, "fuzz_oracle_14.npz": 8}
    fx = np.load(os.path.join(here, "golden", "fuzz_oracle_small.npz"))
    r = int(np.argmin(np.where(fx["state_atoms"] == 8, fx["state_t_end"], np.inf)))
    k, b = fx["state_owner"][r]
    prob = fuzz_case(int(fx["seeds"][k]), int(fx["n_override"][k]))[0][b]
    s = prob["samples"]["Global"]["ground-rydberg"]
    opts = qp.default_options([(s["amp"], s["det"])], prob["duration"])
    opts.update(qp.TIGHT)
    fin = qp.sesolve(qp.build_hamiltonian(prob), qp.all_ground_state(8, prob["eigenbasis"]),
                     np.array([0.0, float(fx["state_t_end"][r])]), **opts)[-1]
    assert np.max(np.abs(fin - fx["states"][r][:256])) < 1e-12
