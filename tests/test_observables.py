"""The default observables of the V2 backend on host-side states and operators,
following ``TestObservables`` of the reference's /tests/test_backend.py:1250-1529."""
import uuid
from collections import Counter

import numpy as np
import pytest

from pulser_amd import NoiseModel
from pulser_amd.backend import (BitStrings, CorrelationMatrix, Energy, EnergySecondMoment,
                                EnergyVariance, Expectation, Fidelity, Occupation, QutipConfig,
                                Results, RydOperator, RydState, StateResult)


@pytest.fixture
def ghz_state():
    return RydState.from_state_amplitudes(eigenstates=("r", "g"),
                                          amplitudes={"rrr": np.sqrt(0.5), "ggg": np.sqrt(0.5)})


@pytest.fixture
def ham():
    return RydOperator.from_operator_repr(eigenstates=("r", "g"), n_qudits=3, operations=[(1.0, [])])


@pytest.fixture
def zzz():
    return RydOperator.from_operator_repr(eigenstates=("r", "g"), n_qudits=3,
                                          operations=[(1.0, [({"rr": 1.0, "gg": -1.0}, {0, 1, 2})])])


@pytest.fixture
def config():
    return QutipConfig(observables=(BitStrings(),))


@pytest.fixture
def results():
    return Results(atom_order=("q0", "q1", "q2"), total_duration=1000)


@pytest.mark.parametrize("tag_suffix", [None, "foo"])
@pytest.mark.parametrize("eval_times", [None, (0.0, 0.5, 1.0)])
def test_base_init(eval_times, tag_suffix):
    obs = StateResult(evaluation_times=eval_times, tag_suffix=tag_suffix)
    assert isinstance(obs.uuid, uuid.UUID)
    np.testing.assert_array_equal(obs.evaluation_times, eval_times)
    expected_tag = "state_foo" if tag_suffix else "state"
    assert obs.tag == expected_tag and repr(obs) == f"{expected_tag}:{obs.uuid}"
    with pytest.raises(ValueError, match="All evaluation times must be between 0. and 1."):
        StateResult(evaluation_times=[1.000001])
    with pytest.raises(ValueError, match="Evaluation times must be unique"):
        StateResult(evaluation_times=[1.0, 1.0])
    with pytest.raises(ValueError, match="Evaluation times must be in ascending order"):
        StateResult(evaluation_times=[0.0, 1.0, 0.9999])


@pytest.mark.parametrize("eval_times", [None, (0.0, 0.5, 1.0)])
def test_call(config, results, ghz_state, ham, eval_times):
    assert not results.get_result_tags()
    assert tuple(config.default_evaluation_times) == (1.0,)
    obs = StateResult(evaluation_times=eval_times)
    assert obs.apply(state=ghz_state) == ghz_state
    true_times = eval_times or config.default_evaluation_times
    assert not config.is_time_in_evaluation_times(0.1, true_times)
    obs(config, 0.1, ghz_state, ham, results)
    assert not results.get_result_tags()
    tol = 0.5 / results.total_duration
    t_minus = 1.0 - tol
    assert config.is_time_in_evaluation_times(t_minus, true_times, tol=tol)
    obs(config, t_minus, ghz_state, ham, results)
    assert results.get_result_times(obs) == [t_minus]
    assert results.get_result(obs, t_minus) == ghz_state
    assert config.is_time_in_evaluation_times(1.0, true_times)
    obs(config, 1.0, ghz_state, ham, results)
    assert results.get_result_tags() == ["state"]
    assert results.get_result_times("state") == results.get_result_times(obs) == [t_minus, 1.0]
    with pytest.raises(RuntimeError, match="A value is already stored for observable 'state' at time 1.0"):
        obs(config, 1.0, ghz_state, ham, results)
    t_plus = 1.0 + tol
    assert not config.is_time_in_evaluation_times(t_plus, true_times, tol=tol)
    obs(config, t_plus, ghz_state, ham, results)
    assert t_plus not in results.get_result_times(obs)


@pytest.mark.parametrize("p_false_pos", [0, 0.4])
@pytest.mark.parametrize("p_false_neg", [0, 0.3])
@pytest.mark.parametrize("one_state", [None, "g"])
@pytest.mark.parametrize("num_shots", [None, 100])
def test_bitstrings(config, ghz_state, num_shots, one_state, p_false_pos, p_false_neg):
    with pytest.raises(ValueError, match="greater than or equal to 1"):
        BitStrings(num_shots=0)
    obs = BitStrings(one_state=one_state, **({"num_shots": num_shots} if num_shots else {}))
    assert obs.tag == "bitstrings"
    nm = NoiseModel(p_false_pos=p_false_pos, p_false_neg=p_false_neg)
    cfg = config.with_changes(noise_model=nm, default_num_shots=2000)
    assert cfg.noise_model.noise_types == (("SPAM",) if p_false_pos or p_false_neg else ())
    np.random.seed(123)
    shots = num_shots or cfg.default_num_shots
    expected = ghz_state.sample(num_shots=shots, one_state=one_state or ghz_state.infer_one_state(),
                                p_false_pos=p_false_pos or 0, p_false_neg=p_false_neg or 0)
    np.random.seed(123)
    counts = obs.apply(config=cfg, state=ghz_state)
    assert isinstance(counts, Counter) and sum(counts.values()) == shots
    if not (p_false_pos or p_false_neg):
        assert set(counts) == {"000", "111"}
    assert counts == expected


@pytest.mark.parametrize("one_state", [None, "r", "g"])
def test_correlation_matrix_and_occupation(ghz_state, ham, one_state):
    corr, occ = CorrelationMatrix(one_state=one_state), Occupation(one_state=one_state)
    assert corr.tag == "correlation_matrix" and occ.tag == "occupation"

    def check(state, expected):
        np.testing.assert_allclose(corr.apply(state=state, hamiltonian=ham), expected)
        np.testing.assert_allclose(occ.apply(state=state, hamiltonian=ham), np.diagonal(expected))

    check(ghz_state, np.full((3, 3), 0.5))
    ggg = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"ggg": 1.0})
    check(ggg, np.ones((3, 3)) * int(one_state == "g"))
    ggr = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"ggr": 1.0})
    if one_state == "g":
        expected = np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]])
    else:
        expected = np.zeros((3, 3))
        expected[2, 2] = 1
    check(ggr, expected)


def test_energy_observables(ghz_state, ham, zzz):
    energy, var, energy2 = Energy(), EnergyVariance(), EnergySecondMoment()
    assert (energy.tag, var.tag, energy2.tag) == ("energy", "energy_variance", "energy_second_moment")
    custom = RydOperator.from_operator_repr(eigenstates=("r", "g"), n_qudits=3,
                                            operations=[(1.0, [({"gg": -1}, {0, 1, 2})])])
    for op, (e, e2, v) in ((ham, (1.0, 1.0, 0.0)), (zzz, (0.0, 1.0, 1.0)), (custom, (-0.5, 0.5, 0.25))):
        assert np.isclose(energy.apply(state=ghz_state, hamiltonian=op), e)
        assert np.isclose(energy2.apply(state=ghz_state, hamiltonian=op), e2)
        assert np.isclose(var.apply(state=ghz_state, hamiltonian=op), v)


def test_expectation_and_fidelity(ghz_state, ham, zzz):
    with pytest.raises(TypeError, match="'operator' must be an Operator instance"):
        Expectation(ham.to_qobj())
    h_exp = Expectation(ham)
    assert h_exp.tag == "expectation" and h_exp.apply(state=ghz_state) == ham.expect(ghz_state)
    z_exp = Expectation(zzz, tag_suffix="zzz")
    assert z_exp.tag == "expectation_zzz" and z_exp.apply(state=ghz_state) == zzz.expect(ghz_state)
    with pytest.raises(TypeError, match="'state' must be a State instance"):
        Fidelity(ghz_state.to_qobj())
    fid = Fidelity(RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"ggg": 1.0}),
                   tag_suffix="ggg")
    assert fid.tag == "fidelity_ggg" and np.isclose(fid.apply(state=ghz_state), 0.5)
    assert Fidelity(ghz_state).tag == "fidelity" and np.isclose(Fidelity(ghz_state).apply(state=ghz_state), 1.0)


@pytest.mark.parametrize("obs_cls, default", [(StateResult, "density_matrix"), (BitStrings, "bag_union"),
                                              (CorrelationMatrix, "mean"), (Occupation, "mean"),
                                              (Energy, "mean"), (EnergyVariance, "skip_warn"),
                                              (EnergySecondMoment, "mean")])
def test_default_aggregation(obs_cls, default):
    """test_backend.py:886-905 (StateResult: the backend's density-matrix aggregator replaces
    the reference's SKIP_WARN, qutip_backend.py:322-325)."""
    from pulser_amd.backend import AggregationMethod

    codes = {"density_matrix": AggregationMethod.SKIP_WARN, "bag_union": AggregationMethod.BAG_UNION,
             "mean": AggregationMethod.MEAN, "skip_warn": AggregationMethod.SKIP_WARN}
    assert obs_cls().default_aggregation == default
    assert obs_cls().default_aggregation_method == codes[default]
    with pytest.raises(AttributeError):  # read-only
        obs_cls().default_aggregation_method = AggregationMethod.SKIP
    overridden = obs_cls(default_aggregation_method=AggregationMethod.SKIP)
    assert overridden.default_aggregation_method == AggregationMethod.SKIP
    assert obs_cls(default_aggregation_method=4).default_aggregation == "meanstd"


def test_aggregate_with_enum_methods(config, ghz_state, ham):
    """test_backend.py:805-884: per-tag aggregation given as an AggregationMethod."""
    from pulser_amd.backend import AggregationMethod

    runs = []
    occ = Occupation()
    for shift in (0.0, 1.0):
        res = Results(atom_order=("q0", "q1", "q2"), total_duration=1000)
        res._store(observable=occ, time=1.0, value=[0.5 + shift, 0.5, 0.5])
        runs.append(res)
    assert Results.aggregate(runs).occupation == [[1.0, 0.5, 0.5]]
    both = Results.aggregate(runs, occupation=AggregationMethod.MEANSTD).occupation
    assert isinstance(both[0], tuple) and np.allclose(both[0][1], [np.sqrt(0.5), 0.0, 0.0])
    assert "occupation" not in Results.aggregate(runs, occupation=AggregationMethod.SKIP).get_result_tags()
