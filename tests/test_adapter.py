"""``pulser_amd.pulser_adapter`` with duck-typed stand-ins for pulser's SequenceSamples /
Register / Device (pulser itself is a user-side dependency and absent on the test boxes):
conversion into ``SequenceInputs`` and the reference's error messages
(pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:208-303,
tests/pulser_simulation/test_simulation.py:111-135, 1748-1775)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest

from pulser_amd import QutipEmulator
from pulser_amd.pulser_adapter import channel_amp_det, sequence_inputs_from_pulser


def _samples(local_target="q1", slm_end=0, slm_targets=(), eom_open=False, basis_local="digital"):
    n = 40
    ramp = np.linspace(0.0, 4.0, n)
    glob = NS(amp=ramp, det=-ramp, phase=np.zeros(n), slots=[NS(ti=0, tf=n, targets={"q0", "q1"})],
              eom_blocks=[NS(tf=None, detuning_off=-3.5)] if eom_open else [])
    loc = NS(amp=2 * ramp, det=np.zeros(n), phase=np.full(n, 0.5), slots=[NS(ti=10, tf=30, targets={local_target})],
             eom_blocks=[])
    ch_objs = {"ryd": NS(addressing="Global", basis="ground-rydberg", propagation_dir=(1.0, 0.0, 0.0)),
               "ram": NS(addressing="Local", basis=basis_local)}
    return NS(channel_samples={"ryd": glob, "ram": loc}, samples_list=[glob, loc], channels=["ryd", "ram"],
              _ch_objs=ch_objs,
              used_bases={"ground-rydberg", basis_local}, _measurement="ground-rydberg",
              _slm_mask=NS(end=slm_end, targets=set(slm_targets)), _magnetic_field=None, max_duration=n)


def _register(ids=("q0", "q1")):
    return NS(qubits={q: np.array([4.0 * i, 0.0]) for i, q in enumerate(ids)})


def _device(slm=True, bases=("ground-rydberg", "digital")):
    return NS(validate_register=lambda reg: None, supports_slm_mask=slm, supported_bases=set(bases),
              interaction_coeff=5420158.53, interaction_coeff_xy=3700.0)


def test_conversion_into_sequence_inputs():
    inputs = sequence_inputs_from_pulser(_samples(eom_open=True), _register(), _device())
    assert inputs.qubit_ids == ("q0", "q1") and inputs.measurement == "ground-rydberg"
    assert np.array_equal(inputs.coords, [[0.0, 0.0], [4.0, 0.0]])
    ryd, ram = inputs.channels
    assert (ryd.addressing, ryd.basis, ryd.propagation_dir) == ("Global", "ground-rydberg", (1.0, 0.0, 0.0))
    assert ryd.slots[0].targets == (0, 1) and ram.slots[0].targets == (1,)
    assert (ram.slots[0].ti, ram.slots[0].tf) == (10, 30)
    assert ryd.final_detuning == -3.5 and ram.final_detuning == 0.0  # sequence left in EOM mode
    assert ryd.extend_duration(41).det[-1] == -3.5 and ryd.extend_duration(41).amp[-1] == 0.0
    amp_det = channel_amp_det(_samples())
    assert len(amp_det) == 2 and np.array_equal(amp_det[1][0], 2 * np.linspace(0.0, 4.0, 40))
    emu = QutipEmulator(_samples(), _register(), _device())  # the emulator takes the same triple
    assert emu.basis_name == "all" and emu._tot_duration == 40


def test_reference_error_messages():
    with pytest.raises(TypeError, match="sequence has to be a valid"):
        QutipEmulator.from_sequence({"pulse1": "fake"})
    with pytest.raises(TypeError):
        sequence_inputs_from_pulser(_samples(), None, None)
    with pytest.raises(ValueError, match="Samples use SLM mask but device does not have one."):
        sequence_inputs_from_pulser(_samples(slm_end=20, slm_targets=("q1",)), _register(), _device(slm=False))
    with pytest.raises(ValueError, match="The ids of qubits targeted in SLM mask"):
        sequence_inputs_from_pulser(_samples(slm_end=20, slm_targets=("q7",)), _register(), _device())
    with pytest.raises(ValueError, match="The ids of qubits targeted in Local channels"):
        sequence_inputs_from_pulser(_samples(local_target="control1"), _register(), _device())
    with pytest.raises(ValueError, match="Bases used in samples should be supported by device."):
        sequence_inputs_from_pulser(_samples(), _register(), _device(bases=("ground-rydberg",)))
