"""CPU oracle for the pulser_simulation classical-emulation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pulser_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the *checker*.

What it restates (reference = pasqal-io/Pulser 1.10dev0, paths relative to
``/root/reference``):

* ``pulser-simulation/pulser_simulation/hamiltonian.py:97-124, 145-200,
  231-244, 246-439`` - term structure of the time-dependent Hamiltonian and of
  the collapse operators (``oracle/qutip_path.py``).
* the third-party, un-vendored dependency the reference hands the arithmetic
  to: ``qutip >= 5, < 6`` (``pulser-simulation/requirements.txt:1``; no lock
  file, exact version unpinned).  Call sites: ``qutip.QobjEvo``
  ``hamiltonian.py:436-438``; ``qutip.sesolve/mesolve``
  ``simulation.py:705-735``.  Its published algorithm for this path is
  restated as: array coefficients = cubic *not-a-knot* spline through the
  knots (``scipy.interpolate.make_interp_spline(k=3)``), integration by
  ODEPACK ``zvode`` Adams (order 12, atol 1e-8, rtol 1e-6) with Pulser's
  ``max_step``/``nsteps`` (``simulation.py:768-780``).
* ``pulser_simulation/qutip_result.py:101-158`` (``_weights``),
  ``pulser-core/pulser/result.py:103-115`` + ``pulser/math/multinomial.py:18-36``
  (sampling), ``pulser_simulation/simresults.py:176-190, 522-568`` (time lookup,
  SPAM measurement flips) (``oracle/sampling.py``).

Pinning: the restatement reproduces the reference's own
seeded golden Counters ``{"0": 572, "1": 428}``
(``tests/pulser_simulation/test_simulation.py:981``) and
``{"111": 978, "110": 12, "011": 7, "101": 3}`` (``test_simulation.py:1079``)
bit-for-bit and the 3-atom golden state ``test_simulation.py:2176-2187`` within
the test's own tolerance; see ``tests/test_oracle_goldens.py``.
"""
