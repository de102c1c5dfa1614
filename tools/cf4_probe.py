import sys, numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from scipy.linalg import expm
from test_gpu_general_free import _problem_and_state
from test_general_lowering import _coefs
from pulser_amd.general import lower_general, dense_generator
S3 = np.sqrt(3.0)
C1, C2 = 0.5 - S3/6, 0.5 + S3/6
A1, A2 = 0.25 + S3/6, 0.25 - S3/6
for fixture, mesolve, RANGE in (("noisy_xy_2.npz", False, lambda n: (0, n)), ("noises_all_0.npz", False, lambda n: (0, n)), ("noisy_xy_0.npz", True, lambda n: (0, n, 1))):
    prob, init = _problem_and_state(fixture)
    tb = lower_general(prob, mesolve=mesolve, matrix_free=True)
    T = (int(prob["duration"]) - 1) * 1e-3
    G = lambda t: dense_generator(tb, _coefs(tb, t))
    v0 = init if not mesolve else np.outer(init, init.conj()).reshape(-1)
    kn = tb.tknots
    bound = sum(abs(c) * n for c, n in zip(_coefs(tb, 0.3*T), tb.row_norm))
    def run(nsub):
        v = v0.astype(complex)
        for i in range(*RANGE(len(kn) - 1)):
            h = (kn[i+1] - kn[i]) / nsub
            for s in range(nsub):
                t = kn[i] + s*h
                G1, G2 = G(t + C1*h), G(t + C2*h)
                v = expm(h*(A2*G1 + A1*G2)) @ (expm(h*(A1*G1 + A2*G2)) @ v)
        return v
    ref = run(48 if tb.dim < 100 else 16)
    dt = kn[1] - kn[0]
    print(fixture, "dim", tb.dim, "knot dt", dt, "bound", bound, "rho per exp at nsub=1:", 0.5*dt*bound)
    for nsub in ((1, 2, 3, 4, 6, 8, 12, 16, 24) if tb.dim < 100 else (2, 3, 4, 6, 8)):
        v = run(nsub)
        print(flush=True, end=""); print(f"  nsub {nsub:3d} rho/exp {0.5*dt*bound/nsub:7.3f} err {np.max(np.abs(v-ref)):.2e}")
