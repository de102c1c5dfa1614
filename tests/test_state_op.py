"""RydState / RydOperator (the backend's State and Operator types) against the
behaviour tests/pulser_simulation/test_qutip_state_op.py pins for
QutipState / QutipOperator: validation messages, overlaps, probabilities,
sampling, operator algebra, ``from_operator_repr`` and the JSON abstract repr."""
import json
import re

import numpy as np
import pytest

from pulser_amd.backend import RydOperator, RydState, _AbstractReprEncoder

R, G = np.array([1.0, 0.0]), np.array([0.0, 1.0])
EYE = np.eye(2, dtype=complex)
SX = np.array([[0, 1], [1, 0]], dtype=complex)
SY = np.array([[0, -1j], [1j, 0]], dtype=complex)
SZ = np.diag([1.0, -1.0]).astype(complex)


def basis(d, i):
    v = np.zeros(d, dtype=complex)
    v[i] = 1.0
    return v


def proj(v):
    return np.outer(v, np.conj(v))


@pytest.fixture
def ket_r():
    return RydState(R, eigenstates=("r", "g"))


@pytest.fixture
def dm_g():
    return RydState(proj(G), eigenstates=("r", "g"))


@pytest.fixture
def ket_plus():
    return RydState.from_state_amplitudes(eigenstates=("r", "g"),
                                          amplitudes={"r": 1 / np.sqrt(2), "g": 1 / np.sqrt(2)})


# ---------------------------------------------------------------------- state
def test_state_init():
    """test_qutip_state_op.py:50-104."""
    with pytest.raises(ValueError, match="eigenstates must be represented by single characters"):
        RydState(R, eigenstates=["ground", "rydberg"])
    with pytest.raises(ValueError, match="can't contain repeated entries"):
        RydState(R, eigenstates=["r", "g", "r"])
    with pytest.raises(TypeError, match="must be a 'collections.Sequence'"):
        RydState(R, eigenstates={"r", "g"})
    with pytest.raises(TypeError, match="must be a state vector"):
        RydState("not a state", eigenstates=["r", "g"])
    with pytest.raises(TypeError, match="must be a state vector"):
        RydState(np.zeros((4, 2)), eigenstates=["r", "g"])
    with pytest.raises(ValueError, match="incompatible with a system of 3-level qudits"):
        RydState(R, eigenstates=["r", "g", "h"])
    state = RydState(basis(3, 0).reshape(1, 3), eigenstates=["r", "g", "h"])  # a bra
    assert (state.n_qudits, state.qudit_dim, state.eigenstates) == (1, 3, ("r", "g", "h"))
    assert np.array_equal(np.asarray(state.to_qobj()).ravel(), basis(3, 0)) and state.to_qobj().isket
    with pytest.raises(RuntimeError, match="Failed to infer the 'one state'"):
        state.infer_one_state()
    three = np.kron(np.kron(G, G), G)
    state = RydState(three, eigenstates=("r", "g"))
    assert (state.n_qudits, state.qudit_dim) == (3, 2) and state.infer_one_state() == "r"
    assert np.array_equal(np.asarray(state.to_qobj()).ravel(), three)
    dm = proj(np.kron(basis(3, 0), basis(3, 0)))
    state = RydState(dm, eigenstates=["r", "g", "h"])
    assert (state.n_qudits, state.qudit_dim) == (2, 3) and np.array_equal(np.asarray(state.to_qobj()), dm)


@pytest.mark.parametrize("eigenstates", [("g", "r"), ("g", "r", "x"), ("g", "h"), ("u", "d"), ("0", "1")])
def test_infer_one_state(eigenstates):
    """:106-116."""
    assert RydState(basis(len(eigenstates), 0), eigenstates=eigenstates).infer_one_state() == eigenstates[1]


def test_get_basis_state():
    """:118-135."""
    state = RydState.from_state_amplitudes(eigenstates=("r", "g", "h"), amplitudes={"ggg": 1.0})
    for index, name in ((0, "rrr"), (1, "rrg"), (2, "rrh"), (3, "rgr"), (4, "rgg"), (9, "grr"), (26, "hhh")):
        assert state.get_basis_state_from_index(index) == name
    with pytest.raises(ValueError, match="'index' must be a non-negative integer"):
        state.get_basis_state_from_index(-1)


def test_overlap(ket_r, dm_g, ket_plus):
    """:137-180."""
    assert ket_r.overlap(ket_r) == 1.0
    assert dm_g.overlap(ket_r) == ket_r.overlap(dm_g) == 0.0
    assert ket_plus.overlap(ket_r) == ket_r.overlap(ket_plus)
    assert np.isclose(ket_plus.overlap(ket_r), 0.5)
    assert np.isclose(dm_g.overlap(ket_plus), ket_plus.overlap(dm_g))
    assert np.isclose(dm_g.overlap(ket_plus), 0.5)
    with pytest.raises(TypeError, match="expects another 'RydState'"):
        dm_g.overlap(ket_r.to_qobj())
    with pytest.raises(ValueError, match="Can't calculate the overlap between a state with 1 "
                       "2-dimensional qudits and another with 2 3-dimensional qudits"):
        ket_r.overlap(RydState.from_state_amplitudes(eigenstates=("r", "g", "h"), amplitudes={"rr": 1.0}))
    msg = "Can't calculate the overlap between states with eigenstates ('r', 'g') and {}."
    with pytest.raises(ValueError, match=re.escape(msg.format(("u", "d")))):
        ket_r.overlap(RydState(R, eigenstates=("u", "d")))
    with pytest.raises(NotImplementedError, match=re.escape(msg.format(("g", "r")))):
        ket_r.overlap(RydState(R, eigenstates=("g", "r")))


def test_probabilities(ket_plus):
    """:182-213 - an unnormalised state keeps its amplitudes; probabilities are
    normalised after the cutoff."""
    amps = {"rr": np.sqrt(0.5), "gg": 1j * np.sqrt(0.5 - 1e-12), "gr": 1e-6}
    state = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes=amps)
    assert np.asarray(state.to_qobj()).ravel()[2] == 1e-6  # not renormalised
    probs = {k: np.abs(a) ** 2 for k, a in amps.items()}
    got = state.probabilities(cutoff=9e-13)
    assert all(np.isclose(probs[k], got[k]) for k in probs)
    probs.pop("gr")
    total = sum(probs.values())
    probs = {k: v / total for k, v in probs.items()}
    got = state.probabilities()
    assert set(got) == set(probs) and all(np.isclose(probs[k], got[k]) for k in probs)
    assert state.infer_one_state() == "r"
    assert state.bitstring_probabilities() == {"11": got["rr"], "00": got["gg"]}
    assert state.bitstring_probabilities(one_state="g") == {"11": got["gg"], "00": got["rr"]}
    dm_plus = RydState(proj(np.asarray(ket_plus.to_qobj()).ravel()), eigenstates=ket_plus.eigenstates)
    assert dm_plus.probabilities() == pytest.approx({"r": 0.5, "g": 0.5})
    assert dm_plus.bitstring_probabilities() == pytest.approx({"0": 0.5, "1": 0.5})


def test_sample(ket_r, dm_g):
    """:215-226."""
    shots = 2000
    np.random.seed(12)
    assert ket_r.sample(num_shots=shots) == {"1": shots}
    assert ket_r.sample(num_shots=shots, one_state="g") == {"0": shots}
    assert ket_r.sample(num_shots=shots, p_false_pos=0.1) == {"1": shots}
    assert ket_r.sample(num_shots=shots, p_false_neg=0.1)["0"] > 0
    assert dm_g.sample(num_shots=shots) == {"0": shots}
    assert dm_g.sample(num_shots=shots, one_state="g") == {"1": shots}
    assert dm_g.sample(num_shots=shots, p_false_neg=0.1) == {"0": shots}
    assert dm_g.sample(num_shots=shots, p_false_pos=0.1)["1"] > 0


@pytest.mark.parametrize("amplitudes", [{"rrh": 1.0}, {"rr": 0.5, "rgg": np.sqrt(0.75)}])
def test_from_state_amplitudes_error(amplitudes):
    """:228-246."""
    with pytest.raises(ValueError, match=re.escape(
            "All basis states must be combinations of eigenstates with the same length. Expected "
            f"combinations of ('r', 'g'), each with {len(list(amplitudes)[0])} elements.")):
        RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes=amplitudes)


def test_from_state_amplitudes():
    """:248-270."""
    def vec(eigenstates, amplitudes):
        return np.asarray(RydState.from_state_amplitudes(eigenstates=eigenstates,
                                                         amplitudes=amplitudes).to_qobj()).ravel()

    assert np.array_equal(vec(("r", "g"), {"g": 1.0}), basis(2, 1))
    assert np.array_equal(vec(("g", "r"), {"g": 1.0}), basis(2, 0))
    assert np.array_equal(vec(("r", "g", "h"), {"g": 1.0}), basis(3, 1))
    got = vec(("r", "g"), {"rr": -0.5j, "gr": 0.5, "rg": 0.5j, "gg": -0.5})
    want = -0.5j * np.kron(R, R) + 0.5 * np.kron(G, R) + 0.5j * np.kron(R, G) - 0.5 * np.kron(G, G)
    assert np.array_equal(got, want)


def test_state_repr_eq_and_abstract_repr(ket_r, dm_g):
    """:272-315."""
    assert repr(ket_r).startswith("RydState\n--------\nEigenstates: ('r', 'g')\n")
    assert ket_r == RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"r": 1.0})
    assert dm_g != RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"g": 1.0})
    assert dm_g != proj(G)
    kwargs = dict(eigenstates=("r", "g"), amplitudes={"g": 1.0})
    state = RydState.from_state_amplitudes(**kwargs)
    assert json.dumps(state, cls=_AbstractReprEncoder) == json.dumps(kwargs)
    with pytest.raises(ValueError, match=re.escape(
            "Failed to serialize state of type 'RydState' because it was not created via "
            "'RydState.from_state_amplitudes()'")):
        json.dumps(RydState(state.to_qobj(), eigenstates=state.eigenstates), cls=_AbstractReprEncoder)
    state._state = ket_r._state  # modified in place
    with pytest.raises(ValueError, match="modified in place after its creation"):
        json.dumps(state, cls=_AbstractReprEncoder)


# ------------------------------------------------------------------- operator
@pytest.fixture
def paulis():
    return {k: RydOperator(m, eigenstates=("r", "g")) for k, m in (("i", EYE), ("x", SX), ("y", SY), ("z", SZ))}


def test_operator_init():
    """:318-343."""
    with pytest.raises(ValueError, match="eigenstates must be represented by single characters"):
        RydOperator(SZ, eigenstates=["ground", "rydberg"])
    with pytest.raises(ValueError, match="can't contain repeated entries"):
        RydOperator(SZ, eigenstates=["r", "g", "r"])
    with pytest.raises(TypeError, match="must be a square matrix"):
        RydOperator("sigmaz", eigenstates=["r", "g"])
    with pytest.raises(TypeError, match="must be a square matrix"):
        RydOperator(R.reshape(2, 1), eigenstates=["r", "g"])
    with pytest.raises(ValueError, match="incompatible with a system of 3-level qudits"):
        RydOperator(SZ, eigenstates=["r", "g", "h"])
    z = RydOperator(SZ, eigenstates=("r", "g"))
    assert z.eigenstates == ("r", "g")
    assert np.array_equal(z.to_qobj().toarray(), proj(R) - proj(G))


@pytest.mark.parametrize("op_name", ["apply_to", "expect"])
def test_operator_errors_on_state(paulis, op_name):
    """:361-383."""
    op = getattr(paulis["x"], op_name)
    with pytest.raises(TypeError, match=re.escape(f"'RydOperator.{op_name}()' expects a 'RydState' instance")):
        op(R)
    msg = (f"Can't apply RydOperator.{op_name}() between a RydOperator "
           "with eigenstates ('r', 'g') and a RydState with {}")
    with pytest.raises(ValueError, match=re.escape(msg.format(("g", "h")))):
        op(RydState(R, eigenstates=("g", "h")))
    with pytest.raises(NotImplementedError, match=re.escape(msg.format(("g", "r")))):
        op(RydState(R, eigenstates=("g", "r")))


@pytest.mark.parametrize("op_name", ["__add__", "__matmul__"])
def test_operator_errors_on_operator(paulis, op_name):
    """:385-406."""
    op = getattr(paulis["x"], op_name)
    with pytest.raises(TypeError, match=re.escape(f"'{op_name}' expects a 'RydOperator' instance")):
        op(RydState(R, eigenstates=("r", "g")))
    msg = f"Can't apply {op_name} between a RydOperator with eigenstates ('r', 'g') and a RydOperator with {{}}"
    with pytest.raises(ValueError, match=re.escape(msg.format(("g", "h")))):
        op(RydOperator(proj(R), eigenstates=("g", "h")))
    with pytest.raises(NotImplementedError, match=re.escape(msg.format(("g", "r")))):
        op(RydOperator(proj(R), eigenstates=("g", "r")))


def test_operator_apply_and_expect(paulis, ket_r, dm_g, ket_plus):
    """:408-437."""
    x, y, z = paulis["x"], paulis["y"], paulis["z"]
    assert x.apply_to(ket_r) == RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"g": 1.0})
    assert x.apply_to(dm_g) == RydState(proj(R), eigenstates=dm_g.eigenstates)
    assert x.expect(ket_r) == 0.0 and x.expect(dm_g) == 0.0
    assert np.isclose(x.expect(ket_plus), 1.0)
    assert np.isclose(x.expect(y.apply_to(ket_plus)), -1.0)
    assert z.expect(ket_r) == 1.0 and z.expect(dm_g) == -1.0
    assert np.isclose(z.expect(ket_plus), 0.0)


def test_operator_algebra(paulis, dm_g):
    """:439-475, 579-585."""
    i, x, y, z = (paulis[k] for k in "ixyz")
    ket = lambda a, b: np.outer(a, b)  # noqa: E731
    assert x + y == RydOperator((1 - 1j) * ket(R, G) + (1 + 1j) * ket(G, R), eigenstates=x.eigenstates)
    assert i + z == RydOperator(2 * proj(R), eigenstates=z.eigenstates)
    assert (1 - 2j) * i == RydOperator((1 - 2j) * EYE, eigenstates=z.eigenstates)
    assert 0.5 * (i + z) == RydOperator(proj(R), eigenstates=z.eigenstates)
    assert x @ x == y @ y == z @ z == i
    assert x @ z == -1j * y and z @ x == 1j * y
    g_proj = 0.5 * (i + (-1) * z)
    assert g_proj == RydOperator(proj(G), eigenstates=i.eigenstates)
    assert g_proj != dm_g
    assert repr(z).startswith("RydOperator\n-----------\nEigenstates: ('r', 'g')\n")


def test_from_operator_repr(paulis):
    """:477-577."""
    def build(ops, n=2, eig=("r", "g")):
        return RydOperator.from_operator_repr(eigenstates=eig, n_qudits=n, operations=ops)

    for bad in ("gggg", "hh"):
        with pytest.raises(ValueError, match=re.escape(
                "Every QuditOp key must be made up of two eigenstates among ('r', 'g'); "
                f"instead, got '{bad}'.")):
            build([(1.0, [({bad: 1.0, "rr": -1.0}, {0})])])
    with pytest.raises(ValueError, match="Got invalid indices for a system with 2 qudits"):
        build([(1.0, [({"gg": 1.0, "rr": -1.0}, {3, 5, 9})])])
    with pytest.raises(ValueError, match=re.escape("only indices {1} were still available")):
        build([(1.0, [({"gg": 1.0, "rr": -1.0}, {0}), ({"rg": 1.0}, {0})])])
    got = build([(1.0, [({"rr": 1.0, "hh": -1.0}, {0}), ({"gr": -1j}, {2})])], n=3, eig=("r", "g", "h"))
    want = np.kron(np.kron(proj(basis(3, 0)) - proj(basis(3, 2)), np.eye(3)),
                   -1j * np.outer(basis(3, 1), basis(3, 0)))
    assert got == RydOperator(want, eigenstates=("r", "g", "h"))
    assert build([(1, [])], n=1) == paulis["i"]
    assert build([(0.5, [({"rr": 1.0, "gg": -1.0}, {0})]), (0.5, [])]) == RydOperator(
        np.kron(proj(R), EYE), eigenstates=("r", "g"))


def test_operator_abstract_repr():
    """:587-605."""
    kwargs = dict(eigenstates=("r", "g"), n_qudits=3,
                  operations=[(0.5, [({"rr": 1.0, "gg": 1.0j}, {0})]), (0.5, [])])
    op = RydOperator.from_operator_repr(**kwargs)
    ser_ops = [(0.5, [({"rr": 1.0, "gg": {"real": 0.0, "imag": 1.0}}, [0])]), (0.5, [])]
    assert json.dumps(op, cls=_AbstractReprEncoder) == json.dumps({**kwargs, "operations": ser_ops})
    with pytest.raises(ValueError, match=re.escape(
            "Failed to serialize state of type 'RydOperator' because it was not created via "
            "'RydOperator.from_operator_repr()'")):
        json.dumps(RydOperator(op.to_qobj(), eigenstates=op.eigenstates), cls=_AbstractReprEncoder)


# ---------------------------------------------------------------- QutipConfig
def test_qutip_config_validation():
    """tests/pulser_simulation/test_qutip_config.py:17-147."""
    import warnings

    from pulser_amd import NoiseModel, Solver
    from pulser_amd.backend import BitStrings, QutipConfig, StateResult

    obs = lambda: [StateResult(evaluation_times=[1.0])]  # noqa: E731
    with pytest.raises(NotImplementedError, match="'QutipBackendV2' does not handle custom interaction matrices."):
        QutipConfig(observables=obs(), interaction_matrix=np.eye(4))
    with pytest.raises(ValueError, match="be greater than 0 and less than or equal to 1"):
        QutipConfig(observables=obs(), sampling_rate=1.2)
    cfg = QutipConfig(observables=obs(), sampling_rate=0.5, progress_bar=True)
    assert {"sampling_rate", "progress_bar"} <= cfg._expected_kwargs() and cfg.progress_bar
    with pytest.raises(ValueError, match="received unexpected keyword arguments"):
        QutipConfig(observables=obs(), not_an_option=3)
    with pytest.warns(UserWarning, match="The number of samples per run .* is ignored when using QutipBackendV2."):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            QutipConfig(observables=obs(), noise_model=NoiseModel(temperature=45, samples_per_run=5))
    with pytest.raises(TypeError, match=re.escape("If provided, `initial_state` must be an instance of `RydState`")):
        QutipConfig(observables=obs(), initial_state="all-ground")
    assert QutipConfig.state_type is RydState and QutipConfig.operator_type is RydOperator
    # evaluation times given as arrays
    default_times = np.array([0.0, 0.25, 0.5, 0.75, 1.0])
    t1, t2 = np.array([0.2, 0.4, 0.8]), np.array([0.15, 0.35, 0.65, 0.95])
    cfg = QutipConfig(observables=[StateResult(evaluation_times=t1),
                                   StateResult(evaluation_times=t2, tag_suffix="second")],
                      default_evaluation_times=default_times)
    np.testing.assert_almost_equal(cfg._get_legacy_evaluation_times(1000),
                                   np.union1d(np.union1d(default_times, t1), t2))
    # solver round trip through the JSON document
    for solver in Solver:
        for given in (solver, str(solver.value)):
            cfg = QutipConfig(observables=[BitStrings(evaluation_times=[1.0])], solver=given)
            text = cfg.to_abstract_repr()
            assert json.loads(text)["solver"] == str(solver.value)
            assert QutipConfig.from_abstract_repr(text).solver is solver
    with pytest.raises(ValueError, match="Invalid solver 'fakesolver'"):
        QutipConfig(observables=[BitStrings(evaluation_times=[1.0])], solver="fakesolver")


@pytest.mark.gpu
def test_density_matrix_aggregator():
    """tests/pulser_simulation/test_aggregators.py:4-47 (the mean is formed on the device: a GPU test)."""
    from pulser_amd.backend import density_matrix_aggregator

    s1 = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rgr": 1.0})
    s2 = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"grr": 1.0})
    s3 = RydState(proj(np.asarray(RydState.from_state_amplitudes(
        eigenstates=("r", "g"), amplitudes={"ggr": 1.0}).to_qobj()).ravel()), eigenstates=("r", "g"))
    acc = density_matrix_aggregator([s1, s2])  # vector and vector
    res1 = np.zeros((8, 8))
    res1[2, 2] = res1[4, 4] = 0.5  # |rgr> = index 2, |grr> = index 4
    assert np.isclose(acc.to_qobj().norm(), 1.0) and np.allclose(np.asarray(acc.to_qobj()), res1)
    acc = density_matrix_aggregator([acc, s3])  # matrix and matrix from a ket
    res2 = 0.5 * res1
    res2[6, 6] = 0.5
    assert np.isclose(acc.to_qobj().norm(), 1.0) and np.allclose(np.asarray(acc.to_qobj()), res2)
    acc = density_matrix_aggregator([acc, acc])
    assert np.isclose(acc.to_qobj().norm(), 1.0) and np.allclose(np.asarray(acc.to_qobj()), res2)


@pytest.mark.filterwarnings("ignore::DeprecationWarning")
def test_simconfig_effective_noise_and_noise_model_conversion():
    """tests/pulser_simulation/test_simconfig.py:103-172."""
    from pulser_amd import NoiseModel, SimConfig

    ket = np.array([[1.0], [2.0]])
    with pytest.raises(ValueError, match="The operators list length"):
        SimConfig(noise=("eff_noise"), eff_noise_rates=[1.0])
    with pytest.raises(TypeError, match="eff_noise_rates is a list of floats"):
        SimConfig(noise=("eff_noise"), eff_noise_rates=["0.1"], eff_noise_opers=[EYE])
    with pytest.raises(ValueError, match="The effective noise parameters have not been filled."):
        SimConfig(noise=("eff_noise"))
    with pytest.raises(TypeError, match="is not a matrix"):
        SimConfig(noise=("eff_noise"), eff_noise_opers=[2.0], eff_noise_rates=[1.0])
    with pytest.raises(TypeError, match="type 'oper'"):
        SimConfig(noise=("eff_noise"), eff_noise_opers=[ket], eff_noise_rates=[1.0])
    for bad in (EYE, np.eye(5)):
        with pytest.raises(ValueError, match="With leakage, operator's shape"):
            SimConfig(noise=("eff_noise", "leakage"), eff_noise_opers=[bad], eff_noise_rates=[1.0])
    with pytest.raises(ValueError, match="Without leakage, operator's shape"):
        SimConfig(noise=("eff_noise",), eff_noise_opers=[np.eye(4)], eff_noise_rates=[1.0])
    SimConfig(noise=("eff_noise"), eff_noise_opers=[SX, EYE], eff_noise_rates=[0.5, 0.5])
    nm = NoiseModel(p_false_neg=0.4, p_false_pos=0.1, amp_sigma=1e-3, runs=10, samples_per_run=1)
    expected = SimConfig(noise=("SPAM", "amplitude"), epsilon=0.1, epsilon_prime=0.4, eta=0.0,
                         amp_sigma=1e-3, laser_waist=float("inf"), runs=10, samples_per_run=1)
    assert SimConfig.from_noise_model(nm) == expected
    assert expected.to_noise_model() == nm


def test_mean_and_std_aggregators():
    """/tests/test_aggregators.py:15-160 of the reference (pulser.backend.aggregators)."""
    import torch

    from pulser_amd.backend import _mean_of, _std_of

    assert _mean_of([1.0, 2.0, 3.0, 4.0]) == 2.5 and _mean_of([1.0j, 2.0j, 3.0j, 4.0j]) == 2.5j
    arrays = [np.array([1.0, 2.0, 3.0]), np.array([2.0, 3.0, 4.0]), np.array([3.0, 4.0, 5.0])]
    assert np.all(_mean_of(arrays) == np.array([2.0, 3.0, 4.0]))
    lists = [a.tolist() for a in arrays]
    assert _mean_of(lists) == [2.0, 3.0, 4.0]
    assert _mean_of([[x] for x in lists]) == [[2.0, 3.0, 4.0]]
    assert torch.allclose(_mean_of([torch.tensor(x) for x in lists]), torch.tensor([2.0, 3.0, 4.0]))
    assert np.isclose(_std_of([1.0, 2.0, 3.0, 4.0]), 1.2909944487358056)
    assert np.isclose(_std_of([1.0j, 2.0j, 3.0j, 4.0j]), 1.2909944487358056)
    assert np.all(_std_of(arrays) == np.array([1.0, 1.0, 1.0]))
    assert _std_of(lists) == [1.0, 1.0, 1.0] and _std_of([[x] for x in lists]) == [[1.0, 1.0, 1.0]]
    assert torch.allclose(_std_of([torch.tensor(x) for x in lists]), torch.tensor([1.0, 1.0, 1.0]))
    for fn, name in ((_mean_of, "Mean"), (_std_of, "Std")):
        with pytest.raises(ValueError, match="Cannot process 0 samples."):
            fn([])
        with pytest.raises(ValueError, match="Cannot process list of empty lists."):
            fn([[], []])
        with pytest.raises(ValueError, match="Need to supply a list of values to process."):
            fn("abcd")
        with pytest.raises(ValueError, match=f"{name} aggregator cannot process data"):
            fn([{}, {}])
        with pytest.raises(ValueError, match=f"Cannot process list of lists of {type({})}."):
            fn([[{}], [{}]])
        with pytest.raises(ValueError, match=f"Cannot process list of matrices of {type('a')}."):
            fn([[["abcd"]], [["efgh"]]])
        with pytest.raises(ValueError, match="Cannot process list of matrices with empty columns."):
            fn([[[]], [[]]])


def test_emulation_config_semantics():
    """/tests/test_backend.py:570-802 of the reference (``EmulationConfig``), for the
    parts this backend's ``QutipConfig`` implements."""
    from pulser_amd import NoiseModel
    from pulser_amd.backend import BitStrings, QutipConfig

    with pytest.warns(UserWarning, match="'QutipConfig' was initialized without any observables"):
        QutipConfig()
    with pytest.raises(TypeError, match="All entries in 'observables' must be instances of Observable"
                       ".*at index 0.*fidelity"):
        QutipConfig(observables=["fidelity"])
    with pytest.raises(TypeError, match="All entries in 'callbacks' must not be instances of Observable"
                       ".*at index 0"):
        QutipConfig(callbacks=(BitStrings(),))
    with pytest.raises(TypeError, match="All entries in 'callbacks' must be instances of Callback"
                       ".*at index 0.*Hello"):
        QutipConfig(callbacks=("Hello",), observables=(BitStrings(),))
    with pytest.raises(ValueError, match="Some of the provided 'observables' share identical tags"):
        QutipConfig(observables=[BitStrings(), BitStrings(num_shots=200000)])
    with pytest.raises(ValueError, match="All evaluation times must be between 0. and 1."):
        QutipConfig(observables=(BitStrings(),), default_evaluation_times=[-1e15, 0.0, 0.5, 1.0])
    with pytest.raises(ValueError, match="Evaluation times must be unique up to"):
        QutipConfig(observables=(BitStrings(),), default_evaluation_times=[0.0, 0.5, 0.5 + 1e-14, 1.0])
    with pytest.raises(ValueError, match="Evaluation times must be in ascending order"):
        QutipConfig(observables=(BitStrings(),), default_evaluation_times=[0.0, 1.0, 0.5])
    with pytest.raises(TypeError, match="must be a NoiseModel"):
        QutipConfig(observables=(BitStrings(),), noise_model={"p_false_pos": 0.1})
    for bad in (0, 1.001):
        with pytest.raises(ValueError, match="strictly positive integer"):
            QutipConfig(observables=(BitStrings(),), n_trajectories=bad)
    with pytest.deprecated_call():
        runs_nm = NoiseModel(amp_sigma=0.1, runs=10)
    with pytest.raises(ValueError, match="`EmulationConfig.n_trajectories` and `NoiseModel.runs` can't be"
                       " simultaneously defined"):
        QutipConfig(observables=(BitStrings(),), noise_model=runs_nm, n_trajectories=2)
    assert QutipConfig(observables=(BitStrings(),), noise_model=runs_nm, n_trajectories=10.0).n_trajectories == 10
    assert QutipConfig(observables=(BitStrings(),), noise_model=runs_nm).n_trajectories == 10
    assert QutipConfig(observables=(BitStrings(),), noise_model=runs_nm,
                       prefer_device_noise_model=True).n_trajectories == 40
    config = QutipConfig(observables=(BitStrings(),))
    assert config.n_trajectories == 1
    assert config.with_changes(n_trajectories=10).n_trajectories == 10 and config.n_trajectories == 1
    times = np.array([0.5, 1.0])
    conf = QutipConfig(default_evaluation_times=times, observables=(BitStrings(),))
    np.testing.assert_equal(conf.default_evaluation_times, times)


def test_results_container():
    """/tests/test_backend.py:1060-1247 of the reference (``pulser.backend.Results``)."""
    from collections import Counter

    from pulser_amd.backend import BitStrings, QutipConfig, Results, StateResult

    res = Results(atom_order=(), total_duration=100)
    assert res.get_result_tags() == [] and res.get_tagged_results() == {}
    template = "\n".join(["Results", "-------", "Stored results: {stored}",
                          "Evaluation times per result: {times}",
                          "Atom order in states and bitstrings: ()", "Total sequence duration: 100 ns"])
    assert str(res) == template.format(stored=[], times={})
    with pytest.raises(AttributeError, match="'bitstrings' is not in the results"):
        res.bitstrings
    with pytest.raises(ValueError, match="'bitstrings' is not an Observable instance nor a known observable tag"):
        res.get_result_times("bitstrings")
    obs = BitStrings(num_shots=100, tag_suffix="test")
    with pytest.raises(ValueError, match=f"'bitstrings_test:{obs.uuid}' has not been stored"):
        res.get_result(obs, 1.0)
    state = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rrr": 1.0})
    ham = RydOperator.from_operator_repr(eigenstates=("r", "g"), n_qudits=3, operations=[(1.0, [])])
    obs(config=QutipConfig(observables=(obs,)), t=1.0, state=state, hamiltonian=ham, result=res)
    expected = [Counter({"111": 100})]
    assert res.get_result_tags() == ["bitstrings_test"]
    assert res.get_tagged_results() == {"bitstrings_test": expected} and res.bitstrings_test == expected
    assert res.get_result_times("bitstrings_test") == res.get_result_times(obs) == [1.0]
    assert res.get_result(obs, 1.0) == res.get_result("bitstrings_test", 1.0) == expected[0]
    with pytest.raises(ValueError, match="not available at time 0.912"):
        res.get_result(obs, 0.912)
    assert str(res) == template.format(stored=["bitstrings_test"], times={"bitstrings_test": [1.0]})
    # final bitstrings / state
    empty = Results(atom_order=(), total_duration=0)
    with pytest.raises(RuntimeError, match="final bitstrings are not available"):
        empty.final_bitstrings
    with pytest.raises(RuntimeError, match="final state is not available"):
        empty.final_state
    plain = BitStrings()
    plain(config=QutipConfig(observables=(BitStrings(),)), t=1.0, state=state, hamiltonian=ham, result=empty)
    assert empty.final_bitstrings == empty.get_result(plain, 1.0)
    so, holder = StateResult(), Results(atom_order=(), total_duration=0)
    so(config=QutipConfig(observables=(so,)), t=1.0, state=state, hamiltonian=ham, result=holder)
    assert holder.final_state == holder.get_result(so, 1.0) == state
    # from_final_bitstrings
    counts = {"000": 60, "111": 40}
    made = Results.from_final_bitstrings(atom_order=("q0", "q1", "q2"), total_duration=1000,
                                         final_bitstrings=counts)
    assert made.atom_order == ("q0", "q1", "q2") and made.total_duration == 1000
    assert made.final_bitstrings == Counter(counts) and made.get_result_times("bitstrings") == [1.0]
    with pytest.raises(TypeError, match="'final_bitstrings' is not a valid bitstrings counter"):
        Results.from_final_bitstrings(atom_order=("q0",), total_duration=100, final_bitstrings=42)
    with pytest.warns(FutureWarning, match="'bitstring_counts' is an attribute of the deprecated"):
        assert made.bitstring_counts == made.final_bitstrings
    with pytest.warns(FutureWarning, match="'bitstring_counts'"):
        with pytest.raises(RuntimeError, match="final bitstrings are not available"):
            Results(atom_order=("q0",), total_duration=100).bitstring_counts
    blank = Results(atom_order=("q0",), total_duration=100)
    for attr in ("sampling_dist", "sampling_errors", "get_samples", "get_state", "plot_histogram",
                 "n_samples", "evaluation_time", "meas_basis"):
        with pytest.raises(AttributeError, match=f"{attr} is available only in 'SampledResult'"):
            getattr(blank, attr)
    with pytest.raises(AttributeError, match="'not_an_attr' is not in the results"):
        blank.not_an_attr


def test_density_matrix_aggregator_fails_loudly_without_a_gpu():
    """The backend has no CPU path: on a host without a GPU the aggregator says so itself instead of failing
    somewhere inside torch (round-3 ADVICE)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the -m gpu test covers the aggregator")
    from pulser_amd.backend import density_matrix_aggregator

    s1 = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rgr": 1.0})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        density_matrix_aggregator([s1, s1])
