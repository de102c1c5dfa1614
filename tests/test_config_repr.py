"""JSON abstract representation of observables and of the configuration, following
``TestObservableRepr`` / ``TestConfigRepr`` of the reference's /tests/test_backend_abstract_repr.py."""
import json

import numpy as np
import pytest

from pulser_amd import NoiseModel
from pulser_amd.backend import (AggregationMethod, BitStrings, CorrelationMatrix, Energy,
                                EnergySecondMoment, EnergyVariance, Expectation, Fidelity, Occupation,
                                QutipConfig, RydOperator, RydState, StateResult, _AbstractReprEncoder)

STATE = RydState.from_state_amplitudes(eigenstates=("0", "1"), amplitudes={"11": 1.0})
OPERATOR = RydOperator.from_operator_repr(eigenstates=("r", "g"), n_qudits=3,
                                          operations=[(0.3, [({"rr": 0.2j}, [0, 2])])])


@pytest.mark.parametrize("with_uuid", [True, False])
@pytest.mark.parametrize("observable, arg, kwargs", [
    (BitStrings, (), {"evaluation_times": [i * 0.05 for i in range(10)], "num_shots": 211, "one_state": "r",
                      "tag_suffix": "7"}),
    (BitStrings, (), {}),
    (CorrelationMatrix, (), {"one_state": "r"}),
    (Occupation, (), {"one_state": "g"}),
    (Energy, (), {"evaluation_times": [i * 0.05 for i in range(10)]}),
    (EnergyVariance, (), {"evaluation_times": np.linspace(0, 1, 13)}),
    (EnergySecondMoment, (), {"evaluation_times": [i * 0.1 for i in range(5)]}),
    (Fidelity, (STATE,), {"evaluation_times": [i / 7.2 for i in range(5)]}),
    (Expectation, (OPERATOR,), {"tag_suffix": "my_op"}),
    (Expectation, (OPERATOR,), {"default_aggregation_method": AggregationMethod.SKIP}),
])
def test_observable_repr_round_trip(observable, arg, kwargs, with_uuid):
    """test_backend_abstract_repr.py:44-170 through the configuration document."""
    obs = observable(*arg, **kwargs)
    doc = json.loads(QutipConfig(observables=[obs]).to_abstract_repr(skip_validation=True))
    rep = doc["observables"][0]
    assert rep["observable"] == obs._base_tag and rep["tag_suffix"] == kwargs.get("tag_suffix")
    if rep["evaluation_times"] is None:
        assert "evaluation_times" not in kwargs
    else:
        assert np.allclose(rep["evaluation_times"], kwargs["evaluation_times"])
    assert rep.get("one_state") == kwargs.get("one_state") and rep.get("num_shots") == kwargs.get("num_shots")
    assert rep["default_aggregation_method"] == obs.default_aggregation_method
    if not with_uuid:
        rep.pop("uuid")  # not required by the schema
    back = QutipConfig.from_abstract_repr(json.dumps(doc)).observables[0]
    assert type(back) is observable and back.tag == obs.tag
    assert (back.uuid == obs.uuid) == with_uuid
    assert back.default_aggregation_method == obs.default_aggregation_method
    again = back._to_abstract_repr()
    for key in ("observable", "tag_suffix", "evaluation_times", "default_aggregation_method"):
        assert json.dumps(again[key], cls=_AbstractReprEncoder) == json.dumps(rep[key], cls=_AbstractReprEncoder)


def test_state_result_and_unknown_observables_are_refused():
    """:265-281."""
    with pytest.raises(ValueError, match="`StateResult` observable is not supported in any remote backend"):
        QutipConfig(observables=[StateResult()]).to_abstract_repr()
    doc = json.loads(QutipConfig(observables=[Energy()]).to_abstract_repr())
    doc["observables"][0]["observable"] = "magic"
    with pytest.raises(ValueError, match="magic"):
        QutipConfig.from_abstract_repr(json.dumps(doc))


def test_config_documents():
    """:289-415."""
    with pytest.raises(TypeError, match="The serialized EmulationConfig must be given as a string. "):
        QutipConfig.from_abstract_repr(1.0)
    # pulser <= 1.8 did not serialise 'default_aggregation_method'
    obs = Energy()
    doc = json.loads(QutipConfig(observables=[obs]).to_abstract_repr())
    doc["observables"][0].pop("default_aggregation_method")
    assert (QutipConfig.from_abstract_repr(json.dumps(doc)).observables[0].default_aggregation_method
            == obs.default_aggregation_method)
    state = RydState.from_state_amplitudes(eigenstates=("0", "1"), amplitudes={"1111": 1.0})
    for observables in ((BitStrings(evaluation_times=[i * 0.01 for i in range(10)]), CorrelationMatrix()),
                        (Energy(), Occupation(one_state="0"))):
        for kwargs in ({"with_modulation": True, "initial_state": state},
                       {"default_evaluation_times": [0.1, 0.2, 0.3], "prefer_device_noise_model": True},
                       {"default_evaluation_times": "Full"},
                       {"noise_model": NoiseModel(p_false_pos=0.1, dephasing_rate=0.01)},
                       {"sampling_rate": 0.5, "solver": "MasterEquation", "progress_bar": True}):
            config = QutipConfig(observables=observables, **kwargs)
            back = QutipConfig.from_abstract_repr(config.to_abstract_repr())
            for a, b in zip(back.observables, config.observables):
                assert json.dumps(a._to_abstract_repr(), cls=_AbstractReprEncoder) == json.dumps(
                    b._to_abstract_repr(), cls=_AbstractReprEncoder)
            if isinstance(config.default_evaluation_times, str):
                assert back.default_evaluation_times == "Full"
            else:
                assert np.allclose(config.default_evaluation_times, back.default_evaluation_times)
            if config.initial_state is None:
                assert back.initial_state is None
            else:
                assert back.initial_state._to_abstract_repr() == config.initial_state._to_abstract_repr()
            assert back.with_modulation == config.with_modulation
            assert back.prefer_device_noise_model == config.prefer_device_noise_model
            assert back.noise_model == config.noise_model
            assert (back.sampling_rate, back.solver, back.progress_bar, back.n_trajectories) == (
                config.sampling_rate, config.solver, config.progress_bar, config.n_trajectories)


@pytest.mark.parametrize("use_torch", [False, True])
def test_result_serialization(use_torch):
    """test_backend_abstract_repr.py:585-666: arrays / tensors become lists, complex
    entries survive, times, tags and aggregation methods come back."""
    import torch

    from pulser_amd.backend import Results

    rng = np.random.default_rng(7)
    bitstrings, corr, energy, occ = BitStrings(), CorrelationMatrix(), Energy(), Occupation()
    results = Results(atom_order=(), total_duration=100)
    results._store(observable=bitstrings, time=0.1, value="rgrgrg")
    cor_mat = torch.from_numpy(rng.normal(size=(6, 6))) if use_torch else rng.normal(size=(6, 6))
    results._store(observable=corr, time=0.2, value=cor_mat)
    results._store(observable=energy, time=0.3, value=5.0)
    occ_vec = rng.normal(size=6).astype(complex)
    occ_vec[0] += 1j
    occ_vec = torch.from_numpy(occ_vec) if use_torch else occ_vec
    results._store(observable=occ, time=0.4, value=occ_vec)
    d = results._to_abstract_repr()
    assert d["results"][str(bitstrings.uuid)] == ["rgrgrg"] and d["results"][str(energy.uuid)] == [5.0]
    assert type(d["results"][str(corr.uuid)][0]) is type(cor_mat)
    assert d["tagmap"] == {o.tag: str(o.uuid) for o in (bitstrings, corr, energy, occ)}
    assert d["times"] == {str(bitstrings.uuid): [0.1], str(corr.uuid): [0.2], str(energy.uuid): [0.3],
                          str(occ.uuid): [0.4]}
    assert d["aggregation_methods"] == {str(bitstrings.uuid): AggregationMethod.BAG_UNION,
                                        str(corr.uuid): AggregationMethod.MEAN,
                                        str(energy.uuid): AggregationMethod.MEAN,
                                        str(occ.uuid): AggregationMethod.MEAN}
    text = results.to_abstract_repr()
    assert text == json.dumps(d, cls=_AbstractReprEncoder)
    back = Results.from_abstract_repr(text)
    assert results.energy == back.energy and results.bitstrings == back.bitstrings
    assert [x.tolist() for x in results.occupation] == back.occupation
    assert isinstance(back.occupation[0][0], complex)
    assert all(isinstance(v, float) for v in back.occupation[0][1:])
    assert [x.tolist() for x in results.correlation_matrix] == back.correlation_matrix
    for o in (bitstrings, occ, corr, energy):
        assert results.get_result_times(o) == back.get_result_times(o)
    assert results.get_result_tags() == back.get_result_tags()
    assert results._aggregation == back._aggregation
