"""NumPy model: how long may a time step be?  4th-order commutator-free Magnus (CF4, two exponentials - what the
kernels run) against the 6th-order Magnus expansion (three Gauss-Legendre nodes, exact commutators - the accuracy a
6th-order commutator-free scheme would have, up to its error constant) on the dense 10-atom anneal Hamiltonian.

    python tools/magnus6_probe.py ROWS COLS [tri|rect] T0_NS T1_NS
"""
import sys, os
import numpy as np
from scipy.linalg import expm
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ket_split_probe import Prob

rows, cols, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
t0, t1 = float(sys.argv[4]), float(sys.argv[5])
pr = Prob(rows, cols, kind)
n = pr.n
D = 1 << n
X = np.zeros((D, D))
idx = np.arange(D)
for k in range(n):
    X[idx, idx ^ (1 << k)] += 1.0

def H(t):
    return np.diag(pr.e0 * 1e-3 - pr.det(t) * 1e-3 * pr.nexc) + 0.5 * pr.amp(t) * 1e-3 * X

S3, S15 = np.sqrt(3.0), np.sqrt(15.0)
def cf4(psi, t, h):
    c1, c2 = 0.5 - S3 / 6, 0.5 + S3 / 6
    a1, a2 = 0.25 + S3 / 6, 0.25 - S3 / 6
    H1, H2 = H(t + c1 * h), H(t + c2 * h)
    psi = expm(-1j * h * (a1 * H1 + a2 * H2)) @ psi
    return expm(-1j * h * (a2 * H1 + a1 * H2)) @ psi

def comm(a, b):
    return a @ b - b @ a

def magnus6(psi, t, h):
    A1, A2, A3 = (-1j * H(t + (0.5 - S15 / 10) * h), -1j * H(t + 0.5 * h), -1j * H(t + (0.5 + S15 / 10) * h))
    al1 = h * A2
    al2 = (S15 * h / 3) * (A3 - A1)
    al3 = (10 * h / 3) * (A3 - 2 * A2 + A1)
    C1 = comm(al1, al2)
    C2 = -(1 / 60) * comm(al1, 2 * al3 + C1)
    Om = al1 + al3 / 12 + (1 / 240) * comm(-20 * al1 - al3 + C1, al2 + C2)
    return expm(Om) @ psi

def run(method, h):
    psi = psi0.copy()
    nst = int(round((t1 - t0) / h))
    for s in range(nst):
        psi = method(psi, t0 + s * h, h)
    return psi

# start state: evolve from |g..g> to t0 with fine CF4
psi0 = np.zeros(D, complex); psi0[-1] = 1.0
t = 0.0
while t < t0 - 1e-9:
    psi0 = cf4(psi0, t, min(1.0, t0 - t)); t += 1.0
ref = run(cf4, 0.25)
nrm = max(np.max(np.abs(pr.e0)) * 1e-3, 1e-9)
print(f"{n} atoms, [{t0}, {t1}] ns, spectral scale e0_max {np.max(pr.e0):.0f} rad/us")
for h in (1.0, 2.0, 4.0):
    if (t1 - t0) / h == int((t1 - t0) / h):
        print(f"CF4      h = {h:5.1f} ns: {np.max(np.abs(run(cf4, h) - ref)):.2e}")
for h in (4.0, 8.0, 12.0, 16.0, 24.0):
    if abs((t1 - t0) / h - round((t1 - t0) / h)) < 1e-9:
        print(f"Magnus-6 h = {h:5.1f} ns: {np.max(np.abs(run(magnus6, h) - ref)):.2e}")
