"""Pulser-native plug-in acceptance (pulser/backends.py:49-58, 72-102; backend/abc.py:88-169).

Runs only where pulser-core is importable: in the build container that is
``/root/reference/pulser-core`` + the jsonschema stand-in of SURVEY Appendix B1 (``/tmp/shim``);
a subprocess keeps ``pulser`` out of the other tests' interpreter.  Nothing here needs a GPU: the
backend is constructed from a real ``pulser.Sequence`` and a real ``pulser.backend.EmulationConfig``
and driven up to the solver call, which must fail loudly for want of a device.
"""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PULSER = "/root/reference/pulser-core"
SHIM = "/tmp/shim"

SCRIPT = textwrap.dedent('''
    import warnings
    import numpy as np
    import pulser
    from pulser.backend import EmulationConfig, Occupation, BitStrings, Energy
    import pulser.backend as pb
    import pulser_amd

    reg = pulser.Register.square(2, spacing=6.0, prefix="q")
    seq = pulser.Sequence(reg, pulser.MockDevice)
    seq.declare_channel("ryd", "rydberg_global")
    seq.add(pulser.Pulse.ConstantPulse(200, 2.0, -1.0, 0.0), "ryd")

    # the registry entry of INTEGRATION.md resolves
    import importlib
    assert getattr(importlib.import_module("pulser_amd"), "RydEmuBackend") is pulser_amd.QutipBackendV2

    seen = []

    class Recorder(pb.Callback):
        def __call__(self, config, t, state, hamiltonian, result):
            seen.append(t)

    cfg = EmulationConfig(observables=[Occupation(evaluation_times=[0.5, 1.0]),
                                       BitStrings(num_shots=50), Energy()],
                          callbacks=[Recorder()],
                          sampling_rate=0.5, noise_model=pulser.NoiseModel(dephasing_rate=0.1))
    backend = pulser_amd.RydEmuBackend(seq, config=cfg)
    mine = backend._config
    assert isinstance(mine, pulser_amd.QutipConfig)
    assert [o.tag for o in mine.observables] == [o.tag for o in cfg.observables]
    assert [str(o.uuid) for o in mine.observables] == [str(o.uuid) for o in cfg.observables]
    assert mine.sampling_rate == 0.5 and len(mine.callbacks) == 1
    assert np.allclose(mine.observables[0].evaluation_times, [0.5, 1.0])
    assert mine.noise_model.dephasing_rate == 0.1
    # virtual subclasses: pulser-side isinstance checks accept this package's state / operator
    assert issubclass(pulser_amd.RydState, pb.State) and issubclass(pulser_amd.RydOperator, pb.Operator)
    st = pulser_amd.RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rggg": 1.0})
    assert isinstance(st, pb.State)
    EmulationConfig(observables=[pb.Fidelity(st)], initial_state=st)  # accepted by pulser's own checks
    # anything that is not a configuration is still refused with the reference's message
    try:
        pulser_amd.QutipBackendV2(seq, config=object())
    except TypeError as e:
        assert "must be an instance of 'EmulationConfig'" in str(e)
    else:
        raise AssertionError("bad config accepted")
    # up to the solver call: everything host-side ran (sampling, HamiltonianData, evaluation times)
    sim = backend._sim_obj
    assert sim.dim == 2 and sim.total_duration_ns == 200 and len(sim.samples_obj.qubit_ids) == 4
    import torch
    if not torch.cuda.is_available():
        try:
            backend.run()
        except RuntimeError as e:
            assert "GPU" in str(e) or "HIP" in str(e) or "cuda" in str(e).lower(), e
        else:
            raise AssertionError("run() succeeded without a device")
    else:
        res = backend.run()
        occ = res.get_result(cfg.observables[0], 1.0)
        assert len(occ) == 4 and seen
    print("INTEROP-OK")
''')


@pytest.mark.skipif(not (os.path.isdir(PULSER) and os.path.isdir(SHIM)),
                    reason="pulser-core (reference checkout + jsonschema stand-in) only exists in the build container")
def test_backend_accepts_pulser_sequence_config_and_observables():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIM, PULSER, ROOT]), PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "INTEROP-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
