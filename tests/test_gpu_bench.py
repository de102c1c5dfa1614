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


def _bench(n_gpus, *flags):
    env = dict(os.environ, RYD_BENCH_BACKEND="gloo")
    args = ["bench.py", "--gpus", str(n_gpus), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras", *flags]
    if n_gpus == 1:
        cmd = [sys.executable, *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *args]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_default_workload_with_two_ranks_prints_the_contract_line_and_the_same_ensemble():
    one = _bench(1, "--no-legs", "--batch", "16")
    two = _bench(2, "--no-legs", "--batch", "16")
    for line, n in ((one, 1), (two, 2)):
        assert REQUIRED <= set(line), REQUIRED - set(line)
        assert line["n_gpus"] == n and line["scaling"] == "weak" and line["higher_is_better"] is True
        assert line["config"]["sequences_per_gpu"] == 16 and "14-atom" in line["config"]["workload"]
        assert line["roofline"]["bound"] == "valu_f64" and 0 < line["roofline"]["frac"] <= 1
        # the time per stage and the bare algorithmic count stand beside the ISA-counted fraction
        assert 1.0 < line["roofline"]["us_per_stage"] < 100.0 and "isa_flops_per_launch" in line["roofline"]
        assert 0 < line["roofline"]["frac_algorithmic"] < line["roofline"]["frac"]
        assert line["roofline"]["algorithmic_flops_per_amplitude_per_stage"] == 62.0
        # the timed kernels' own result against the tight-oracle fixture of the headline register, in the bench process
        assert 0 <= line["parity_max_abs"] < 1e-7 and "ns_tri14_anneal.npz" in line["parity_reference"]
        assert abs(line["ensemble_mean_norm"] - 1.0) < 1e-8
    # N > 1: the line proves what the collectives ran on (under RCCL: N distinct PCI bus ids; here both ranks share the GPU)
    assert "collective" not in one
    col = two["collective"]
    assert col["backend"] == "gloo" and col["world_size"] == 2 and col["allreduce_of_ones"] == 2.0
    assert [r["rank"] for r in col["ranks"]] == [0, 1] and all(r["pci"] for r in col["ranks"]) and col["distinct_devices"] == 1
    # every rank runs the same 16 sequences: the all-reduced ensemble mean is independent of the world size
    assert np.allclose(one["ensemble_mean_occupations"], two["ensemble_mean_occupations"], rtol=0, atol=1e-12)
    assert two["config"]["stages_per_sequence"] == one["config"]["stages_per_sequence"]
    # value = units of ALL ranks / max-over-ranks time (both ranks share one GPU here, so no speed-up is asserted)
    assert two["value"] == pytest.approx(2 * 16 * 3.1 / (two["ms_per_step"] * 1e-3), rel=1e-9)


def test_cfg4_workload_sharded_over_two_ranks_matches_one_rank():
    one = _bench(1, "--workload", "cfg4", "--trajectories", "48")
    two = _bench(2, "--workload", "cfg4", "--trajectories", "48")
    for line, n in ((one, 1), (two, 2)):
        assert REQUIRED <= set(line)
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["unit"] == "trajectories/s"
        assert line["config"]["n_trajectories"] == 48 and line["config"]["histogram_total"] == 2 * line["config"]["n_measures"]
        assert abs(line["config"]["with_density_matrix"]["trace_final"] - 1.0) < 1e-8
    # rank 0 owns every random draw: the sharded ensemble is the serial one
    assert np.allclose(one["config"]["mean_occupations_final"], two["config"]["mean_occupations_final"], atol=1e-12)
