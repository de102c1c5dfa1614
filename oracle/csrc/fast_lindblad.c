/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of oracle/qutip_path.py:lindblad_rhs for
 * two-level registers, matrix-free, so that the tight zvode reference of a 10-atom master equation
 * (rho = 2^20 entries) finishes in minutes instead of hours.
 *
 * What it restates (paths relative to /root/reference):
 *   d rho/dt = -i[H, rho] + sum_k sum_c (C_c^(k) rho C_c^(k)+ - 1/2 {C_c^(k)+ C_c^(k), rho})
 *   = what qutip.mesolve integrates for pulser_simulation/simulation.py:707-735 with
 *   H(t) of pulser_simulation/hamiltonian.py:246-439 (diagonal + one |g><r| flip per atom and its
 *   adjoint) and the collapse operators of hamiltonian.py:97-124 (the same local 2x2 operators on
 *   every atom).  rho is row-major, atom 0 = most significant bit, local index 0 = r, 1 = g
 *   (docs/source/conventions.md:58-73).
 *
 * It is checked entry by entry against the SciPy CSR right-hand side (tests/test_oracle_fast.py and
 * inside make_fixtures.py before every use).  Nothing of the product links or loads this file.
 */
#include <complex.h>
#include <stddef.h>
#include <stdint.h>

typedef double complex c128;

/* diag[D]   : real diagonal of H(t)
 * cflip[n]  : H[row with atom k in g, col with atom k in r] (= conj of the mirrored entry)
 * S[16]     : local dissipator superoperator on the digit pair (a_k, b_k):
 *             out_loc[i][j] = sum_{i',j'} S[(2i+j)*4 + (2i'+j')] rho_loc[i'][j']
 * has_S     : 0 -> no dissipator */
void lind_rhs_2level(int n, const c128 *rho, c128 *out, const double *diag, const c128 *cflip,
                     const c128 *S, int has_S)
{
    const int64_t D = (int64_t)1 << n;
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < D; ++a) {
        const c128 *ra = rho + a * D;
        c128 *oa = out + a * D;
        const double da = diag[a];
        for (int64_t b = 0; b < D; ++b) oa[b] = -I * ((da - diag[b]) * ra[b]);
        for (int k = 0; k < n; ++k) {
            const int64_t e = (int64_t)1 << (n - 1 - k);
            const int ak = (int)((a >> (n - 1 - k)) & 1);
            const c128 ck = cflip[k];
            /* (H rho)[a,b] += H[a, a^e] rho[a^e, b];  H[a, a^e] = c_k if a_k = g(1) else conj */
            const c128 ha = ak ? ck : conj(ck);
            const c128 *rf = rho + (a ^ e) * D;
            /* (rho H)[a,b] += rho[a, b^e] H[b^e, b];  row b^e in g <=> b_k = 0 */
            const c128 hb0 = ck, hb1 = conj(ck);
            if (!has_S) {
                for (int64_t b = 0; b < D; ++b) {
                    const int bk = (int)((b >> (n - 1 - k)) & 1);
                    oa[b] += -I * (ha * rf[b] - ra[b ^ e] * (bk ? hb1 : hb0));
                }
            } else {
                for (int64_t b = 0; b < D; ++b) {
                    const int bk = (int)((b >> (n - 1 - k)) & 1);
                    const c128 *row = S + (size_t)(2 * ak + bk) * 4;
                    /* neighbours: (a_k', b_k') = (ak,bk), (ak,1-bk), (1-ak,bk), (1-ak,1-bk) */
                    c128 acc = row[2 * ak + bk] * ra[b] + row[2 * ak + (1 - bk)] * ra[b ^ e]
                             + row[2 * (1 - ak) + bk] * rf[b] + row[2 * (1 - ak) + (1 - bk)] * rf[b ^ e];
                    oa[b] += acc - I * (ha * rf[b] - ra[b ^ e] * (bk ? hb1 : hb0));
                }
            }
        }
    }
}
