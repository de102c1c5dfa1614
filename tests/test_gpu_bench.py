"""-m gpu: the multi-GPU legs of bench.py executed BEFORE an 8-GPU node ever runs them: two ranks share the one
GPU of the test box and reduce over gloo (``RYD_BENCH_BACKEND=gloo``; under RCCL only the backend of the
collectives differs).  Checked: the JSON contract line, and that what the ranks compute does not depend on the
number of ranks."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(n_gpus, *flags, backend="gloo", torchrun=False):
    env = dict(os.environ, RYD_BENCH_BACKEND=backend, RYD_BENCH_DETAIL=os.devnull)  # (keep the tree's bench_detail.json)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = ["bench.py", "--gpus", str(n_gpus), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras", *flags]
    if n_gpus == 1 and not torchrun:
        cmd = [sys.executable, *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *args]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return _parse(out.stdout)


LINE_KEYS = REQUIRED | {"roofline", "cpu_baseline", "detail"}
MAX_LINE = 8192  # the driver keeps an 8-KB tail of stdout (round 5 lost its measurement to a 21-KB line)


def _parse(stdout):
    """(contract line, detail): the LAST stdout line is the short driver line - alone in starting with '{', below the
    driver's 8-KB tail, a JSON round trip; the full record stands on an earlier line behind the DETAIL prefix."""
    rows = stdout.splitlines()
    lines = [ln for ln in rows if ln.startswith("{")]
    assert len(lines) == 1 and rows[-1] == lines[0], stdout[-2000:]  # rank 0 prints ONE contract line, last
    assert len(lines[0]) < MAX_LINE, len(lines[0])
    line = json.loads(lines[0])
    assert json.loads(json.dumps(line)) == line and LINE_KEYS <= set(line), LINE_KEYS - set(line)
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())  # scalars only: no prose trees
    assert len(line["config"]["workload"]) <= 200
    detail = [ln for ln in rows if ln.startswith("BENCH_DETAIL ")]
    assert len(detail) == 1
    detail = json.loads(detail[0][len("BENCH_DETAIL "):])
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5)
    return line, detail


def test_default_workload_with_two_ranks_prints_the_contract_line_and_the_same_ensemble():
    (one, one_d), (two, two_d) = _bench(1, "--no-legs", "--batch", "16"), _bench(2, "--no-legs", "--batch", "16")
    for line, det, n in ((one, one_d, 1), (two, two_d, 2)):
        assert line["n_gpus"] == n and line["scaling"] == "weak" and line["higher_is_better"] is True
        assert line["config"]["sequences_per_gpu"] == 16 and "14-atom" in line["config"]["workload"]
        roof = line["roofline"]
        assert roof["bound"] == "valu_f64" and 0 < roof["frac"] <= 1 and roof["kernel"] == "k_split_reg<14, 5>"
        # the time per stage and the bare algorithmic count stand beside the ISA-counted fraction
        assert 1.0 < roof["us_per_stage"] < 100.0 and "isa_flops_per_launch" in roof
        assert 0 < roof["frac_algorithmic"] < roof["frac"]
        assert roof["algorithmic_bytes_per_launch"] == 32.0 * 2**14 * 16 and "traffic" in roof
        assert det["roofline"]["algorithmic_flops_per_amplitude_per_stage"] == 62.0
        # the timed kernels' own result against the tight-oracle fixture of the headline register, in the bench process
        assert 0 <= line["config"]["parity_max_abs"] < 1e-7 and "ns_tri14_anneal.npz" in det["parity_reference"]
        assert abs(det["ensemble_mean_norm"] - 1.0) < 1e-8
    # N > 1: the line proves what the collectives ran on (under RCCL: N distinct PCI bus ids; here both ranks share the GPU)
    assert "collective" not in one
    col = two_d["collective"]
    assert col["backend"] == "gloo" and col["world_size"] == 2 and col["allreduce_of_ones"] == 2.0
    assert two["collective"] == {"backend": "gloo", "world_size": 2, "distinct_devices": 1, "allreduce_of_ones": 2.0}
    assert [r["rank"] for r in col["ranks"]] == [0, 1] and all(r["pci"] for r in col["ranks"]) and col["distinct_devices"] == 1
    one, two = one_d, two_d
    # every rank runs the same 16 sequences: the all-reduced ensemble mean is independent of the world size
    assert np.allclose(one["ensemble_mean_occupations"], two["ensemble_mean_occupations"], rtol=0, atol=1e-12)
    assert two["config"]["stages_per_sequence"] == one["config"]["stages_per_sequence"]
    # value = units of ALL ranks / max-over-ranks time (both ranks share one GPU here, so no speed-up is asserted)
    assert two["value"] == pytest.approx(2 * 16 * 3.1 / (two["ms_per_step"] * 1e-3), rel=1e-9)


def test_cfg4_workload_sharded_over_two_ranks_matches_one_rank():
    (one_l, one), (two_l, two) = (_bench(1, "--workload", "cfg4", "--trajectories", "48"),
                                  _bench(2, "--workload", "cfg4", "--trajectories", "48"))
    assert one_l["config"]["n_trajectories"] == 48 and two_l["unit"] == "trajectories/s" and two_l["roofline"] is None
    for line, n in ((one, 1), (two, 2)):
        assert REQUIRED <= set(line)
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["unit"] == "trajectories/s"
        assert line["config"]["n_trajectories"] == 48 and line["config"]["histogram_total"] == 2 * line["config"]["n_measures"]
        assert abs(line["config"]["with_density_matrix"]["trace_final"] - 1.0) < 1e-8
    # rank 0 owns every random draw: the sharded ensemble is the serial one
    assert np.allclose(one["config"]["mean_occupations_final"], two["config"]["mean_occupations_final"], atol=1e-12)


def test_the_rccl_branch_runs_at_world_size_one():
    """`--gpus 8` on the scaling box differs from a tested path only in the world size: the same launcher
    (torch.distributed.run), `init_process_group("nccl", device_id=...)`, the all-reduces of the occupation sums and of
    the max-over-ranks time on DEVICE tensors through RCCL - at world size 1 on the one GPU of this box."""
    line, det = _bench(1, "--no-legs", "--batch", "16", backend="nccl", torchrun=True)
    col = det["collective"]
    assert col["backend"] == "nccl" and col["world_size"] == 1 and col["allreduce_of_ones"] == 1.0
    assert col["distinct_devices"] == 1 and col["ranks"][0]["pci"]
    assert line["collective"] == {"backend": "nccl", "world_size": 1, "distinct_devices": 1, "allreduce_of_ones": 1.0}
    assert line["n_gpus"] == 1 and 0 <= line["config"]["parity_max_abs"] < 1e-7
    assert abs(det["ensemble_mean_norm"] - 1.0) < 1e-8
    # the sharded cfg4 workload: broadcasts of the draws, all-reduce of int64 histograms / occupation sums and of the
    # fp64 view of the ensemble density matrix (268 MB device tensor) through RCCL
    line, det = _bench(1, "--workload", "cfg4", "--trajectories", "48", backend="nccl", torchrun=True)
    assert det["collective"]["backend"] == "nccl" and line["config"]["n_trajectories"] == 48
    assert det["config"]["histogram_total"] == 2 * det["config"]["n_measures"]
    assert abs(det["config"]["with_density_matrix"]["trace_final"] - 1.0) < 1e-8
    # ... and it is the serial ensemble (rank 0 owns every draw)
    _, serial = _bench(1, "--workload", "cfg4", "--trajectories", "48")
    assert np.allclose(det["config"]["mean_occupations_final"], serial["config"]["mean_occupations_final"], atol=1e-12)
