"""CPU: the stage layout of MIXED split-operator runs (pulser_amd/csrc/split_types.hpp: splitrun_first / _locate /
_weight - the helpers every kernel and the host go through since round 5).

A run of the 6th-order 10-stage composition whose flagged sub-steps take the 4th-order 6-stage one: stage j -> (sub-step,
stage inside it), and the weight of E0 in the D of every stage (a_i tau of its own sub-step; the first stage of a sub-step
also carries the last D of the PREVIOUS sub-step, with that sub-step's own composition; the closing D follows the last
stage).  The helpers are `__host__ __device__`: a small host program is compiled with hipcc (no GPU needed) and compared
with a restatement in Python."""
from __future__ import annotations

import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

PROGRAM = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef double2 cplx;
struct Segs { int lo[3]; int len[3]; };
#include "split_types.hpp"
int main(int argc, char** argv) {
  SplitRun R;
  std::memset(&R, 0, sizeof R);
  const int mixed = argv[1][0] == '1';
  const char* flags = argv[2];
  R.nsub = (int)std::strlen(flags); R.S = 10; R.S2 = 6; R.mixed = mixed;
  for (int i = 0; i <= 10; ++i) R.a[i] = 0.1 * (i + 1);
  for (int i = 0; i <= 6; ++i) R.a2[i] = 0.01 * (i + 1);
  int at = 0;
  for (int s = 0; s < R.nsub; ++s) {
    R.alt[s] = (mixed && flags[s] == '1') ? 1 : 0;
    R.first[s] = (short)at;
    at += R.alt[s] ? 6 : 10;
    R.tau[s] = 1.0 + 0.5 * s;
  }
  R.first[R.nsub] = (short)at;
  std::printf("%d\n", splitrun_first(R, R.nsub));
  for (int j = 0; j <= at; ++j) {
    int s = -1, st = -1;
    if (j < at) splitrun_locate(R, j, s, st);
    std::printf("%d %d %d %.17g\n", j, s, st, splitrun_weight(R, j));
  }
  return 0;
}
"""


@pytest.fixture(scope="module")
def layout_program():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    d = tempfile.mkdtemp()
    src = os.path.join(d, "t.hip")
    open(src, "w").write(PROGRAM)
    exe = os.path.join(d, "t")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "pulser_amd", "csrc"),
                    src, "-o", exe], check=True, capture_output=True, timeout=300)
    yield exe
    shutil.rmtree(d, ignore_errors=True)


def _reference(mixed, flags):
    a10 = [0.1 * (i + 1) for i in range(11)]
    a6 = [0.01 * (i + 1) for i in range(7)]
    alt = [mixed and f == "1" for f in flags]
    tau = [1.0 + 0.5 * s for s in range(len(flags))]
    rows = []
    for s, is_alt in enumerate(alt):
        S, a = (6, a6) if is_alt else (10, a10)
        for st in range(S):
            w = a[st] * tau[s]
            if st == 0 and s > 0:
                Sp, ap = (6, a6) if alt[s - 1] else (10, a10)
                w += ap[Sp] * tau[s - 1]
            rows.append((s, st, w))
    Sl, al = (6, a6) if alt[-1] else (10, a10)
    rows.append((-1, -1, al[Sl] * tau[-1]))
    return rows


@pytest.mark.parametrize("mixed, flags", [(1, "01101"), (1, "10"), (1, "1111"), (0, "01101"), (1, "0"), (1, "0" * 31 + "1" + "0" * 32)])
def test_mixed_run_layout_matches_its_restatement(layout_program, mixed, flags):
    out = subprocess.run([layout_program, str(mixed), flags], check=True, capture_output=True, text=True).stdout.split("\n")
    ref = _reference(bool(mixed), flags)
    assert int(out[0]) == len(ref) - 1  # stages of the compositions
    for j, want in enumerate(ref):
        jj, s, st, w = out[1 + j].split()
        assert (int(jj), int(s), int(st)) == (j, want[0], want[1]), (j, out[1 + j], want)
        assert float(w) == pytest.approx(want[2], rel=1e-15, abs=1e-300)
    # the weights of a run add up to its duration: sum_i a_i = 1 is NOT assumed here (test coefficients), so check the
    # bookkeeping instead: every sub-step's a_0 .. a_S times its tau appears exactly once
    total = sum(r[2] for r in ref)
    assert sum(float(l.split()[3]) for l in out[1:1 + len(ref)]) == pytest.approx(total, rel=1e-14)
