"""pulser_amd.NoiseModel (a restatement of pulser.NoiseModel for users without
pulser-core) against the behaviour /tests/test_noise_model.py of the reference pins:
noise types from parameters, validation messages and warnings."""
import contextlib
import re

import numpy as np
import pytest

from pulser_amd import NoiseModel
from pulser_amd.noise_model import _NOISE_TYPE_PARAMS, _PARAM_TO_NOISE_TYPE

pytestmark = pytest.mark.filterwarnings("ignore:.*'NoiseModel.runs' is deprecated:DeprecationWarning")

I2, I3, I4 = np.eye(2), np.eye(3), np.eye(4)


def test_constants():
    """test_noise_model.py:32-40."""
    flat = {}
    for noise_type, params in _NOISE_TYPE_PARAMS.items():
        for p in params:
            assert p not in flat
            flat[p] = noise_type
    assert flat == _PARAM_TO_NOISE_TYPE


@pytest.mark.parametrize("params, noise_types", [
    (set(), set()),
    ({"disable_doppler"}, set()),
    ({"p_false_pos", "dephasing_rate"}, {"SPAM", "dephasing"}),
    ({"state_prep_error", "relaxation_rate", "runs", "samples_per_run"}, {"SPAM", "relaxation"}),
    ({"temperature", "depolarizing_rate", "runs", "samples_per_run"}, {"doppler", "depolarizing"}),
    ({"temperature", "depolarizing_rate", "runs", "samples_per_run", "disable_doppler"}, {"depolarizing"}),
    ({"amp_sigma", "runs", "samples_per_run"}, {"amplitude"}),
    ({"laser_waist", "hyperfine_dephasing_rate"}, {"amplitude", "dephasing"}),
    ({"detuning_sigma", "runs", "samples_per_run"}, {"detuning"}),
    ({"temperature", "trap_waist", "trap_depth", "runs", "samples_per_run"}, {"doppler", "register"}),
    ({"temperature", "trap_waist", "trap_depth", "runs", "samples_per_run", "disable_doppler"}, {"register"}),
    ({"dmm_sigma", "runs", "samples_per_run"}, {"dmm_sigma"}),
    ({"detuning_map_spot_waist"}, {"dmm_crosstalk"}),
])
def test_init(params, noise_types):
    """:44-135."""
    ctx = pytest.deprecated_call(match="NoiseModel.runs") if "runs" in params else contextlib.nullcontext()
    with ctx:
        nm = NoiseModel(**{p: (1.0 if p != "disable_doppler" else True) for p in params})
    assert set(nm.noise_types) == noise_types
    relevant = NoiseModel._find_relevant_params(noise_types, nm.state_prep_error, nm.amp_sigma, nm.laser_waist)
    assert "disable_doppler" not in relevant
    assert nm.disable_doppler == ("disable_doppler" in params)
    params = params - {"disable_doppler"}
    assert all(getattr(nm, p) == 1.0 for p in params)
    assert all(not getattr(nm, p) for p in relevant - params)


@pytest.mark.parametrize("noise_param", ["relaxation_rate", "p_false_neg", "laser_waist"])
@pytest.mark.parametrize("unused_param", ["runs", "samples_per_run"])
@pytest.mark.filterwarnings("ignore:Setting samples_per_run different to 1:DeprecationWarning")
def test_unused_params(unused_param, noise_param):
    """:137-162."""
    with pytest.warns(UserWarning, match=re.escape(
            f"'{unused_param}' is not used by any active noise type in"
            f" {(_PARAM_TO_NOISE_TYPE[noise_param],)} when the only defined parameters are {[noise_param]}")):
        NoiseModel(**{unused_param: 100, noise_param: 1.0})


def test_samples_per_run_deprecation():
    with pytest.deprecated_call(match="Setting samples_per_run different to 1 is"):
        NoiseModel(samples_per_run=5, temperature=10.0)


@pytest.mark.parametrize("param", ["runs", "samples_per_run", "laser_waist", "detuning_map_spot_waist"])
def test_init_strict_pos(param):
    """:164-178."""
    with pytest.raises(ValueError, match=f"'{param}' must be greater than zero, not 0"):
        NoiseModel(**{param: 0})


@pytest.mark.parametrize("value", [None, -1e-9, 0.0, 0.2, 1.0001])
@pytest.mark.parametrize("param, noise", [
    ("dephasing_rate", "dephasing"), ("hyperfine_dephasing_rate", "dephasing"),
    ("relaxation_rate", "relaxation"), ("depolarizing_rate", "depolarizing"),
    ("temperature", "doppler"), ("detuning_sigma", "detuning")])
def test_init_rate_like(param, noise, value):
    """:180-210."""
    if value is None:
        with pytest.raises(TypeError, match=f"{param} should be castable to float, not"):
            NoiseModel(**{param: value})
    elif value < 0:
        with pytest.raises(ValueError, match=f"'{param}' must be greater than or equal to zero, not {value}."):
            NoiseModel(**{param: value})
    else:
        nm = NoiseModel(**{param: value})
        assert getattr(nm, param) == value
        assert nm.noise_types == ((noise,) if value > 0 else ())


@pytest.mark.parametrize("value", [-1e-9, 0.0, 0.5, 1.0, 1.0001])
@pytest.mark.parametrize("param, noise", [
    ("state_prep_error", "SPAM"), ("p_false_pos", "SPAM"), ("p_false_neg", "SPAM"),
    ("amp_sigma", "amplitude"), ("dmm_sigma", "dmm_sigma")])
def test_init_prob_like(param, noise, value):
    """:212-250."""
    if 0 <= value <= 1:
        kwargs = {param: value}
        if value > 0 and param in ("amp_sigma", "state_prep_error", "dmm_sigma"):
            kwargs.update(runs=1, samples_per_run=1)
        nm = NoiseModel(**kwargs)
        assert getattr(nm, param) == value
        assert nm.noise_types == ((noise,) if value > 0 else ())
        return
    with pytest.raises(ValueError, match=f"'{param}' must be greater than or equal to zero and "
                       f"smaller than or equal to one, not {value}"):
        NoiseModel(runs=1, samples_per_run=1, **{param: value})


def test_bool_like():
    """:266-292."""
    for value in (False, True):
        nm = NoiseModel(eff_noise_rates=[0.1], eff_noise_opers=[I3 if value else I2], with_leakage=value)
        assert nm.with_leakage == value
        assert NoiseModel(disable_doppler=value).disable_doppler == value
    for value in (0, 1, 0.1):
        with pytest.raises(ValueError, match=f"'with_leakage' must be a boolean, not {value}"):
            NoiseModel(eff_noise_rates=[0.1], eff_noise_opers=[I3 if value else I2], with_leakage=value)
        with pytest.raises(ValueError, match=f"'disable_doppler' must be a boolean, not {value}"):
            NoiseModel(disable_doppler=value)


def test_effective_noise():
    """:294-355."""
    x = np.ones((2, 2)) - I2
    with pytest.raises(ValueError, match="The provided rates must be greater than 0."):
        NoiseModel(eff_noise_opers=[I2, x], eff_noise_rates=[-1.0, 0.5])
    with pytest.raises(ValueError, match="The operators list length"):
        NoiseModel(eff_noise_rates=[1.0])
    with pytest.raises(TypeError, match="eff_noise_rates is a list of floats"):
        NoiseModel(eff_noise_rates=["0.1"], eff_noise_opers=[I2])
    with pytest.raises(TypeError, match="not castable to a Numpy array"):
        NoiseModel(eff_noise_rates=[2.0], eff_noise_opers=[{(1.0, 0), (0.0, -1)}])
    with pytest.raises(ValueError, match="is not a 2D array."):
        NoiseModel(eff_noise_opers=[2.0], eff_noise_rates=[1.0])
    for bad in (I2, np.eye(5)):
        with pytest.raises(ValueError, match="With leakage, operator's shape"):
            NoiseModel(eff_noise_opers=[bad], eff_noise_rates=[1.0], with_leakage=True)
    with pytest.raises(ValueError, match="Without leakage, operator's shape"):
        NoiseModel(eff_noise_opers=[I4], eff_noise_rates=[1.0])
    nested = ((1.0, 0.0), (0.0, 1.0))
    assert NoiseModel(eff_noise_opers=[I2], eff_noise_rates=[1.0]).eff_noise_opers == (nested,)
    assert NoiseModel(eff_noise_opers=[I2.tolist()], eff_noise_rates=[1.0]).eff_noise_opers == (nested,)
    with pytest.raises(ValueError, match="At least one effective noise operator"):
        NoiseModel(with_leakage=True)


def test_hf_detuning_noise_validation():
    """:370-435."""
    for psd, om in (([1, 4, 2], [3, 6, 7]), (np.array([1, 4, 2]), np.array([3, 6, 7])), ((1, 4, 2), (3, 6, 7))):
        NoiseModel(detuning_hf_psd=psd, detuning_hf_omegas=om)
    nm = NoiseModel()
    assert nm.detuning_hf_psd == () and nm.detuning_hf_omegas == ()
    with pytest.raises(ValueError, match="empty tuples or both be provided"):
        NoiseModel(detuning_hf_psd=(1, 2, 3))
    with pytest.raises(ValueError, match="empty tuples or both be provided"):
        NoiseModel(detuning_hf_omegas=(4, 5, 6))
    with pytest.raises(ValueError, match="1D tuples"):
        NoiseModel(detuning_hf_psd=[[1, 2, 3]], detuning_hf_omegas=[3, 4, 5])
    with pytest.raises(ValueError, match="1D tuples"):
        NoiseModel(detuning_hf_psd=[1, 2, 3], detuning_hf_omegas=[[3, 4, 5]])
    with pytest.raises(ValueError, match="same length"):
        NoiseModel(detuning_hf_psd=[1, 2], detuning_hf_omegas=[3, 4, 5])
    with pytest.raises(ValueError, match="length > 1"):
        NoiseModel(detuning_hf_psd=[1], detuning_hf_omegas=[3])
    with pytest.raises(ValueError, match="positive values"):
        NoiseModel(detuning_hf_psd=[-1, 2], detuning_hf_omegas=[3, 4])
    with pytest.raises(ValueError, match="positive values"):
        NoiseModel(detuning_hf_psd=[1, 2], detuning_hf_omegas=[3, -4])
    with pytest.raises(ValueError, match="monotonously growing"):
        NoiseModel(detuning_hf_psd=[1, 2], detuning_hf_omegas=[4, 3])


def test_register_noise_parameters():
    """:661-720."""
    nm = NoiseModel(temperature=15.0, trap_depth=150.0, trap_waist=1.0, runs=1)
    assert set(nm.noise_types) == {"doppler", "register"}
    with pytest.raises(ValueError, match="trap_waist, trap_depth, and temperature must be defined"):
        NoiseModel(trap_waist=1.0, trap_depth=150.0, runs=1)


def test_register_sigma_and_shot_to_shot_classification():
    """/tests/test_hamiltonian_data.py:25-33 (sigma_xy = 0.158, sigma_z = 0.826 um at 15 uK,
    1 um waist, 150 uK depth) and :648-675 (which noise types vary from shot to shot)."""
    from types import SimpleNamespace

    from pulser_amd.noise_model import has_shot_to_shot_except_spam, register_sigma_xy_z

    sxy, sz = register_sigma_xy_z(15.0, 1.0, 150.0)
    assert 0.158 == pytest.approx(sxy, abs=1e-2) and 0.826 == pytest.approx(sz, abs=1e-2)
    for data, expected in ((dict(noise_types="doppler"), True),
                           (dict(noise_types="amplitude", amp_sigma=1), True),
                           (dict(noise_types="amplitude", amp_sigma=0), False),
                           (dict(noise_types="detuning"), True), (dict(noise_types="register"), True),
                           (dict(noise_types="dmm_sigma"), True), (dict(noise_types="SPAM"), False),
                           (dict(noise_types="other"), False),
                           (dict(noise_types={"other", "doppler"}), True)):
        assert has_shot_to_shot_except_spam(SimpleNamespace(**data)) is expected


def test_high_frequency_detuning_noise_formula():
    """/tests/test_hamiltonian_data.py:616-645: delta_hf(t) = sum_i sqrt(2 df_i S_i) cos(w_i t + phi_i)."""
    from pulser_amd.hamiltonian_data import generate_detuning_fluctuations

    psd, freqs = [1, 2, 3], [3, 4, 5]
    times = np.arange(0, 10, 0.1)
    rng = np.random.default_rng(3)
    phases = rng.uniform(0, 2 * np.pi, size=2)
    nm = NoiseModel(detuning_hf_psd=psd, detuning_hf_omegas=freqs)
    got = generate_detuning_fluctuations(nm, 0.0, phases, times)
    want = np.zeros_like(times)
    for i, s in enumerate(psd[1:]):
        want += np.sqrt(2 * (freqs[i + 1] - freqs[i]) * s) * np.cos(freqs[i + 1] * times * 1e-3 + phases[i])
    assert got.size == times.size and np.allclose(got, want)
    assert np.allclose(generate_detuning_fluctuations(nm, 0.25, phases, times), want + 0.25)
