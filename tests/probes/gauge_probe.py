"""Dev probe (NumPy, CPU): complex drives gauged away for the register-resident ket kernel.

H(t) = sum_k [c_k(t) |g><r|_k + h.c.] - sum_k delta_k(t) n_k + sum U_ij n_i n_j with c_k the COMPLEX cubic
spline of 0.5 Omega e^{-i phi} (hamiltonian.py:349-351).  With psi~ = prod_k exp(i theta_k(t) n_k) psi and
e^{i theta_k} = w_k = c_k / r_k (r_k real, signed, continuous) the Hamiltonian is real symmetric:
drive r_k(t), detuning delta_k + theta_k'(t), theta' = Im(c' conj c) / |c|^2.
Question: does CF4 (two exponentials per knot interval at the Gauss points) stay as accurate in that gauge?
"""
import sys, os
import numpy as np
from scipy.linalg import expm
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helpers import local_problem
from oracle import qutip_path as qp
from scipy.interpolate import CubicSpline

n = 4
prob = local_problem(n, seed=1, duration=61)
ham = qp.build_hamiltonian(prob)
D = 2**n
tl = ham.tlist
# per-atom complex drive splines and detunings from the problem's samples
loc = prob["samples"]["Local"]["ground-rydberg"]
cs = [CubicSpline(tl, 0.5 * np.asarray(loc[q]["amp"]) * np.exp(-1j * np.asarray(loc[q]["phase"])), bc_type="not-a-knot") for q in range(n)]
ds = [CubicSpline(tl, np.asarray(loc[q]["det"]), bc_type="not-a-knot") for q in range(n)]
U = prob["interaction_matrix"][-1]
idx = np.arange(D)
nk = [1 - ((idx >> (n - 1 - k)) & 1) for k in range(n)]
E0 = sum(U[i, j] * nk[i] * nk[j] for i in range(n) for j in range(i + 1, n)).astype(float)

def H_lab(t):
    H = np.diag(E0 - sum(ds[k](t) * nk[k] for k in range(n))).astype(complex)
    for k in range(n):
        c = cs[k](t)
        for s in range(D):
            if (s >> (n - 1 - k)) & 1:  # s_k = 1 (g): row g, col r
                H[s, s ^ (1 << (n - 1 - k))] += c
                H[s ^ (1 << (n - 1 - k)), s] += np.conj(c)
    return H

assert np.max(np.abs(H_lab(0.0123) - ham.matrix(0.0123).toarray())) < 1e-12

def gauge(t, k):
    c, dc = cs[k](t), cs[k](t, 1)
    r = abs(c)
    w = c / r if r > 0 else 1.0
    thdot = (dc * np.conj(c)).imag / (r * r) if r > 0 else 0.0
    return r, w, thdot

def H_rot(t):
    H = np.diag(E0).astype(complex)
    for k in range(n):
        r, w, thdot = gauge(t, k)
        H -= np.diag((ds[k](t) + thdot) * nk[k])
        for s in range(D):
            if (s >> (n - 1 - k)) & 1:
                H[s, s ^ (1 << (n - 1 - k))] += r
                H[s ^ (1 << (n - 1 - k)), s] += r
    return H

def Ug(t):
    f = np.ones(D, complex)
    for k in range(n):
        _, w, _ = gauge(t, k)
        f *= np.where(nk[k] == 1, w, 1.0)
    return f  # psi~ = f * psi

S3 = np.sqrt(3.0)
C1, C2 = 0.5 - S3 / 6, 0.5 + S3 / 6
A1, A2 = 0.25 + S3 / 6, 0.25 - S3 / 6

def cf4(Hf, psi, t0, t1, nsub):
    h = (t1 - t0) / nsub
    for s in range(nsub):
        ta = t0 + s * h
        H1, H2 = Hf(ta + C1 * h), Hf(ta + C2 * h)
        psi = expm(-1j * h * (A1 * H1 + A2 * H2)) @ psi   # first exponential acts first? CF4: exp(a2 G1 + a1 G2) exp(a1 G1 + a2 G2)
        psi = expm(-1j * h * (A2 * H1 + A1 * H2)) @ psi
    return psi

psi0 = np.zeros(D, complex); psi0[-1] = 1
T = tl[-1]
knots = tl
def run(Hf, nsub, rot):
    psi = psi0 * (Ug(0.0) if rot else 1.0)
    for a, b in zip(knots[:-1], knots[1:]):
        psi = cf4(Hf, psi, a, b, nsub)
    return psi / (Ug(T) if rot else 1.0)

ref = run(H_lab, 8, False)
for nsub in (1, 2):
    lab = run(H_lab, nsub, False)
    rot = run(H_rot, nsub, True)
    print(f"nsub {nsub}: |lab - ref| {np.max(np.abs(lab - ref)):.2e}   |rot - ref| {np.max(np.abs(rot - ref)):.2e}   |rot - lab| {np.max(np.abs(rot - lab)):.2e}")
print("max |theta'| over the sequence:", max(abs(gauge(t, k)[2]) for t in np.linspace(0, T, 400) for k in range(n)))
