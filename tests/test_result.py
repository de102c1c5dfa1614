"""pulser_amd.results (Result / SampledResult / StateResult) against what the
reference's tests/test_result.py pins for pulser.result and QutipResult."""
import re
from collections import Counter

import numpy as np
import pytest

from pulser_amd.results import QState, Result, SampledResult, StateResult


def test_get_samples():
    """test_result.py:29-53 (seeded multinomial goldens)."""
    class FixedWeights(Result):
        def __init__(self, weights):
            assert weights.sum() == pytest.approx(1.0)
            self.weights = weights
            self.atom_order = tuple(f"q{x}" for x in range(int(np.log2(weights.shape[0]))))

        def _weights(self):
            return self.weights

    np.random.seed(123)
    w = np.array([0.1, 0.2, 0.3, 0.4])
    assert FixedWeights(w).get_samples(100) == Counter({"10": 41, "11": 38, "01": 15, "00": 6})
    assert FixedWeights(w).get_samples(1000) == Counter({"11": 383, "10": 310, "01": 195, "00": 112})
    one_hot = np.array([1.0 if x == 0b110101 else 0.0 for x in range(2**6)])
    assert FixedWeights(one_hot).get_samples(1000) == Counter({"110101": 1000})


def test_sampled_result():
    """test_result.py:56-120."""
    import matplotlib

    matplotlib.use("Agg")
    samples_dict = {"000": 50, "111": 50}
    from_dict = SampledResult(atom_order=("a", "b", "c"), meas_basis="ground-rydberg",
                              bitstring_counts=samples_dict)
    samples = Counter(samples_dict)
    result = SampledResult(atom_order=("a", "b", "c"), meas_basis="ground-rydberg", bitstring_counts=samples)
    assert repr(result) == str(result) == (
        "SampledResult(atom_order=('a', 'b', 'c'), meas_basis='ground-rydberg', "
        f"bitstring_counts={samples}, evaluation_time=1.0)")
    assert result.final_bitstrings == from_dict.final_bitstrings
    assert isinstance(result.final_bitstrings, Counter) and isinstance(from_dict.final_bitstrings, Counter)
    assert result.n_samples == 100
    assert result.sampling_dist == {"000": 0.5, "111": 0.5}
    err = np.sqrt(0.5**2 / 100)
    assert result.sampling_errors == {"000": err, "111": err}
    np.random.seed(3052023)
    with pytest.warns(UserWarning, match=re.escape(
            "'SampledResult.get_samples()' resamples a sampling distribution")):
        new = result.get_samples(100)
    new.subtract(samples)
    assert all(abs(d) < err * 100 for d in new.values())
    with pytest.raises(NotImplementedError, match=re.escape("`SampledResult.get_state()` is not implemented")):
        result.get_state()
    with pytest.raises(NotImplementedError, match=re.escape(
            "'SampledResult.from_final_bitstrings()' is not implemented")):
        SampledResult.from_final_bitstrings(("a", "b"), 100, {"0": 100})
    result.plot_histogram(show=False)


def _basis(d, i):
    v = np.zeros(d, dtype=complex)
    v[i] = 1.0
    return v


def test_state_result_bases_and_sampling():
    """test_result.py:123-255."""
    qutrit = np.kron(_basis(3, 0), _basis(3, 1))
    result = StateResult(atom_order=("q0", "q1"), meas_basis="ground-rydberg", state=QState(qutrit),
                         matching_meas_basis=False)
    assert result.sampling_dist == {"10": 1.0} and result.sampling_errors == {"10": 0.0}
    assert result._basis_name == "all" and result._eigenbasis == ["r", "g", "h"]
    assert np.array_equal(np.asarray(result.get_state()).ravel(), qutrit)
    qubit = np.kron(_basis(2, 0), _basis(2, 1))
    assert np.array_equal(np.asarray(result.get_state(reduce_to_basis="ground-rydberg")).ravel(), qubit)
    with pytest.raises(ValueError, match="'reduce_to_basis' must be 'ground-rydberg', 'XY', or 'digital'"):
        result.get_state("rydberg")
    with pytest.raises(ValueError, match="Can't reduce a state expressed in all into XY"):
        result.get_state("XY")
    result.meas_basis = "digital"
    assert result.sampling_dist == {"00": 1.0} and result._basis_name == "all"
    result.matching_meas_basis = True
    assert result._basis_name == "digital_with_error" and result._eigenbasis == ["g", "h", "x"]
    assert result.sampling_dist == {"01": 1.0}
    result.meas_basis = "ground-rydberg"
    assert result._basis_name == "ground-rydberg_with_error" and result._eigenbasis == ["r", "g", "x"]
    assert result.sampling_dist == {"10": 1.0}
    result.meas_basis = "XY"
    assert result._basis_name == "XY_with_error" and result._eigenbasis == ["u", "d", "x"]
    assert result.sampling_dist == {"01": 1.0}

    new = StateResult(atom_order=("q0", "q1"), meas_basis="digital", state=QState(qubit), matching_meas_basis=True)
    assert new.sampling_dist == {"01": 1.0}
    new.meas_basis = "ground-rydberg"
    assert new.sampling_dist == {"10": 1.0}
    new.matching_meas_basis = False
    assert new.sampling_dist == {"00": 1.0}
    with pytest.raises(TypeError, match="Can't reduce a system in digital to the ground-rydberg basis"):
        new.get_state(reduce_to_basis="ground-rydberg")

    qudit = np.kron(_basis(4, 0), _basis(4, 1))
    r4 = StateResult(atom_order=("q0", "q1"), meas_basis="ground-rydberg", state=QState(qudit),
                     matching_meas_basis=False)
    assert r4._dim == 4 and r4._basis_name == "all_with_error" and r4._eigenbasis == ["r", "g", "h", "x"]
    assert r4.sampling_dist == {"10": 1.0}
    r4.meas_basis = "digital"
    assert r4.sampling_dist == {"00": 1.0}
    r4.meas_basis = "XY"
    with pytest.raises(AssertionError, match="In XY, state's dimension can only be 2 or 3, not 4"):
        r4._basis_name
    wrong = StateResult(atom_order=("q0", "q1"), meas_basis="ground-rydberg",
                        state=QState(np.kron(_basis(5, 0), _basis(5, 1))), matching_meas_basis=False)
    assert wrong._dim == 5
    with pytest.raises(AssertionError, match="In Ising, state's dimension can be 2, 3 or 4, not 5."):
        wrong._basis_name
    with pytest.raises(NotImplementedError,
                       match="Cannot sample system with single-atom state vectors of dimension > 4"):
        wrong.sampling_dist
    unknown = StateResult(atom_order=("q0", "q1"), meas_basis="rydberg", state=QState(qudit),
                          matching_meas_basis=False)
    with pytest.raises(RuntimeError, match="Unknown measurement basis 'rydberg'."):
        unknown.sampling_dist


def test_state_result_density_matrices():
    """test_result.py:258-310."""
    kw = dict(atom_order=("a", "b"), meas_basis="ground-rydberg", matching_meas_basis=False)
    assert StateResult(state=QState(np.eye(16) / 16), **kw)._basis_name == "all_with_error"
    result = StateResult(state=QState(np.eye(9) / 9), **kw)
    assert result._basis_name == "all"
    with pytest.raises(NotImplementedError, match="Reduce to basis not implemented for density matrix states."):
        result.get_state(reduce_to_basis="ground-rydberg")
    result.matching_meas_basis = True
    assert result._basis_name == "ground-rydberg_with_error"
    result.meas_basis = "digital"
    assert result._basis_name == "digital_with_error"
    result.meas_basis = "XY"
    assert result._basis_name == "XY_with_error"
    result = StateResult(atom_order=("a", "b"), meas_basis="ground-rydberg", state=QState(np.eye(4) / 4),
                         matching_meas_basis=True)
    assert result.state.isoper and result._dim == 2
    assert result.sampling_dist == {"00": 0.25, "01": 0.25, "10": 0.25, "11": 0.25}
