"""MI355X-native emulator backend for the pulser_simulation classical path.

``QutipEmulator`` / ``Solver`` / ``SimConfig`` / ``NoiseModel`` /
``CoherentResults`` / ``NoisyResults`` mirror the API surface of
``pulser_simulation`` (reference: pasqal-io/Pulser 1.10dev0); the solver call is
served by hand-written HIP kernels behind the C ABI of ``include/rydemu.h``.
"""

from .hamiltonian_data import (ChannelInput, HamiltonianData, SequenceInputs,
                               Slot, single_global_channel)
from .noise_model import NoiseModel
from .results import (CoherentResults, NoisyResults, QState, SampledResult,
                      SimulationResults, StateResult)
from .simulation import QutipEmulator, SimConfig, Solver
from . import backend
from .backend import (EmulatorConfig, QutipBackend, QutipBackendV2, QutipConfig, Results, RydEmuBackend,
                      RydOperator, RydState, register_with_pulser)

__version__ = "0.1.0"

__all__ = [
    "QutipEmulator", "Solver", "SimConfig", "NoiseModel", "CoherentResults",
    "NoisyResults", "SimulationResults", "StateResult", "SampledResult", "QState",
    "SequenceInputs", "ChannelInput", "Slot", "HamiltonianData", "single_global_channel",
    "QutipBackendV2", "RydEmuBackend", "register_with_pulser", "QutipBackend", "EmulatorConfig", "QutipConfig", "RydState", "RydOperator", "Results", "backend",
]
