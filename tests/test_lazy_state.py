"""CPU: the lazily materialised states of ``results.states`` (pulser_amd/results.py: SnapshotStore, LazyState).

The reference returns every state of ``result.states`` as a host ``qutip.Qobj`` (simulation.py:744-748); with its default
``evaluation_times="Full"`` that is one state per sample.  Here the stored states stay on the device and read like the
``QState`` they become; these tests use a stand-in for the device tensor (``.cpu().numpy()`` and indexing are all the store
uses), the GPU tests exercise the real thing through the emulator (tests/test_gpu_fullsize.py)."""
from __future__ import annotations

import numpy as np
import pytest

from pulser_amd.results import LazyState, QState, SnapshotStore, StateResult


class FakeTensor:
    """Counts host copies and their sizes."""

    def __init__(self, a, log):
        self.a, self.log = a, log

    def __getitem__(self, k):
        return FakeTensor(self.a[k], self.log)

    @property
    def nbytes(self):
        return self.a.nbytes

    @property
    def shape(self):
        return self.a.shape

    def cpu(self):
        self.log.append(self.a.size)
        return self

    def numpy(self):
        return self.a


def _store(n=40, batch=2, dim=8, bulk_after=16):
    rng = np.random.default_rng(1)
    data = rng.normal(size=(n, batch, dim)) + 1j * rng.normal(size=(n, batch, dim))
    log = []
    return data, log, SnapshotStore(FakeTensor(data, log), bulk_after=bulk_after)


def test_a_state_is_copied_when_it_is_read_and_a_loop_over_all_states_costs_one_bulk_copy():
    data, log, store = _store()
    states = [LazyState(store, i, 1, (8, 1)) for i in range(40)]
    assert log == [] and states[3].shape == (8, 1) and states[3].isket and not states[3].isoper  # shape / kind: no copy
    assert states[3].device_tensor is not None
    assert np.array_equal(np.asarray(states[3])[:, 0], data[3, 1]) and log == [8]
    # a materialised state has let go of the store (one kept state must not pin the run's device snapshots)
    assert states[3]._store is None and states[3].device_tensor is None and states[4]._store is store
    np.asarray(states[3])
    assert log == [8]  # materialised once
    for s in states[4:19]:
        np.asarray(s)
    assert log == [8] * 16  # sixteen single reads ...
    np.asarray(states[30])
    # ... then everything, one transfer per evaluation time (each state owns its time slice: no second copy)
    assert sum(log[16:]) == data.size and len(log) == 16 + 40
    for i, s in enumerate(states):
        assert np.array_equal(np.asarray(s)[:, 0], data[i, 1])
    assert len(log) == 16 + 40 and store.device_tensor is None and states[0].device_tensor is None


def test_lazy_state_reads_like_the_qstate_it_becomes():
    data, log, store = _store()
    s = LazyState(store, 5, 0, (8, 1))
    q = QState(data[5, 0])
    ref = np.ones((8, 1))
    assert np.allclose(s - ref, q - ref) and np.allclose(ref - s, ref - q) and np.allclose(2 * s, 2 * q) and np.allclose(s * 2, q * 2)
    assert np.allclose(-s, -q) and np.allclose(abs(s), abs(q)) and np.allclose(s / 2, q / 2)
    assert isinstance(s + ref, QState) and (s == q).all()
    assert s.norm() == pytest.approx(q.norm()) and np.allclose(s.unit(), q.unit()) and s.full().shape == (8, 1)
    assert s.dag().shape == (1, 8) and s.overlap(q) == pytest.approx(q.overlap(q)) and s.copy() is not s
    assert s[2, 0] == q[2, 0] and len(s) == 8 and np.allclose(np.vdot(s, s), np.vdot(q, q)) and np.allclose(s.conj(), q.conj())
    assert np.allclose(np.max(np.abs(s)), np.max(np.abs(q))) and s.dtype == np.dtype(complex) and s.ndim == 2
    with pytest.raises(TypeError):
        hash(s)
    with pytest.raises(AttributeError):
        s._no_such_private_attribute
    # density matrices
    rho, log2, store2 = _store(n=3, batch=1, dim=16)
    d = LazyState(store2, 1, 0, (4, 4))
    assert d.isoper and not d.isket and np.allclose(d.diag(), np.diag(rho[1, 0].reshape(4, 4))) and d.tr() == pytest.approx(np.trace(rho[1, 0].reshape(4, 4)))


def test_state_result_samples_from_a_lazy_state_like_from_a_host_state():
    """qutip_result.py:101-158 through StateResult._weights: the same Counter for the same seed."""
    data, log, store = _store(n=2, batch=1, dim=8)
    data /= np.linalg.norm(data, axis=2, keepdims=True)
    lazy = StateResult(("q0", "q1", "q2"), "ground-rydberg", LazyState(store, 1, 0, (8, 1)), True)
    host = StateResult(("q0", "q1", "q2"), "ground-rydberg", QState(data[1, 0]), True)
    np.random.seed(4)
    a = lazy.get_samples(500)
    np.random.seed(4)
    b = host.get_samples(500)
    assert a == b and sum(a.values()) == 500
    assert np.allclose(np.asarray(lazy.get_state()), np.asarray(host.get_state()))


def test_retained_device_snapshots_are_bounded_and_the_oldest_spill_first(monkeypatch):
    """results objects a sweep keeps alive must not accumulate HBM.  Every live store is
    registered; a new one spills the oldest to the host until the budget holds; spilling never changes what is read."""
    import gc

    SnapshotStore.spill_all()
    one = 40 * 2 * 8 * 16  # bytes of one stand-in store
    monkeypatch.setenv("PULSER_AMD_SNAPSHOT_GB", str(2.5 * one / 2**30))
    kept = [_store() for _ in range(2)]
    assert SnapshotStore.retained_device_bytes() == 2 * one and all(log == [] for _, log, _ in kept)
    data3, log3, store3 = _store()  # the third does not fit: the OLDEST goes to the host in one transfer
    assert kept[0][2].device_tensor is None and sum(kept[0][1]) == kept[0][0].size
    assert kept[1][2].device_tensor is not None and store3.device_tensor is not None
    assert SnapshotStore.retained_device_bytes() == 2 * one
    # the spilled store reads the same, and hands out copies (one kept state does not pin the run's host array)
    st = LazyState(kept[0][2], 5, 1, (8, 1))
    a = np.asarray(st)
    assert np.array_equal(a[:, 0], kept[0][0][5, 1]) and not any(np.shares_memory(a, h) for h in kept[0][2]._host)
    # a store nobody holds any more does not count
    del kept, st, a
    gc.collect()
    assert SnapshotStore.retained_device_bytes() == one
    assert SnapshotStore.spill_all() == one and store3.device_tensor is None and sum(log3) == data3.size
    assert SnapshotStore.retained_device_bytes() == 0


def test_results_to_host_moves_the_run_and_reads_the_same():
    from pulser_amd.results import CoherentResults

    SnapshotStore.spill_all()
    data, log, store = _store(n=5, batch=1, dim=4)
    times = np.linspace(0.0, 1.0, 6)
    qids = ("q0", "q1")
    res = [StateResult(qids, "ground-rydberg", QState(np.array([0, 0, 0, 1.0])), True, evaluation_time=0.0)]
    res += [StateResult(qids, "ground-rydberg", LazyState(store, i, 0, (4, 1)), True, evaluation_time=float(times[i + 1]))
            for i in range(5)]
    cr = CoherentResults(res, 2, "ground-rydberg", times, "ground-rydberg")
    assert SnapshotStore.retained_device_bytes() == data.nbytes
    assert cr.to_host() is cr and store.device_tensor is None and sum(log) == data.size and len(log) == 5
    assert SnapshotStore.retained_device_bytes() == 0
    for i in range(5):
        assert np.array_equal(np.asarray(cr.states[i + 1])[:, 0], data[i, 0])
    assert sum(log) == data.size and len(log) == 5


def test_expect_takes_sparse_and_qobj_like_observables_and_evaluates_diagonal_ones_from_the_snapshots():
    """simresults.py:89-132: `expect` takes qutip.Qobj or arrays.  Here: arrays, SciPy sparse matrices and anything with
    the Qobj interface (shape + full, sparse storage kept when offered); diagonal observables are evaluated from the
    (device) snapshots without materialising a state, everything else state by state - all equal to the dense
    computation."""
    import scipy.sparse as sp
    import torch

    from pulser_amd.results import CoherentResults

    rng = np.random.default_rng(0)
    n, T = 4, 40
    D = 2**n
    data = rng.normal(size=(T, 1, D)) + 1j * rng.normal(size=(T, 1, D))
    data /= np.linalg.norm(data, axis=-1, keepdims=True)
    store = SnapshotStore(torch.from_numpy(data))
    times = np.linspace(0, 1, T + 1)
    qids = tuple(f"q{i}" for i in range(n))
    psi0 = np.zeros(D, complex)
    psi0[-1] = 1
    res = [StateResult(qids, "ground-rydberg", QState(psi0), True, evaluation_time=0.0)]
    res += [StateResult(qids, "ground-rydberg", LazyState(store, i, 0, (D, 1)), True, evaluation_time=float(times[i + 1]))
            for i in range(T)]
    cr = CoherentResults(res, n, "ground-rydberg", times, "ground-rydberg")
    dvec = rng.normal(size=D)
    A = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    H = A + A.conj().T

    class FakeQobj:  # what the code needs of a qutip.Qobj
        def __init__(self, m):
            self._m = sp.csr_matrix(m)
            self.shape = m.shape

        def full(self):
            return self._m.toarray()

        def data_as(self, kind):
            assert kind == "csr_matrix"
            return self._m

    allst = [psi0] + [data[i, 0] for i in range(T)]
    ref_d = np.array([np.vdot(a, dvec * a).real for a in allst])
    out = cr.expect([np.diag(dvec).astype(complex), sp.diags(dvec).tocsr(), FakeQobj(np.diag(dvec))])
    assert all(np.max(np.abs(o - ref_d)) < 1e-14 and o.dtype == np.float64 for o in out)
    assert store.device_tensor is not None and all(s._store is store for s in cr.states[1:])  # nothing was read back
    cd = rng.normal(size=D) + 1j * rng.normal(size=D)  # a non-Hermitian diagonal: complex values
    out = cr.expect([sp.diags(cd).tocsr()])
    assert np.max(np.abs(out[0] - np.array([np.vdot(a, cd * a) for a in allst]))) < 1e-14 and out[0].dtype == np.complex128
    out = cr.expect([H, sp.csr_matrix(H), FakeQobj(H), A])
    ref_h = np.array([np.vdot(a, H @ a).real for a in allst])
    assert all(np.max(np.abs(o - ref_h)) < 1e-13 for o in out[:3])
    assert np.max(np.abs(out[3] - np.array([np.vdot(a, A @ a) for a in allst]))) < 1e-13
    with pytest.raises(TypeError, match="Incompatible type"):
        cr.expect(["sigma_z"])
    with pytest.raises(ValueError, match="Incompatible shape"):
        cr.expect([sp.eye(D // 2).tocsr()])
