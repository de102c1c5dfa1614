// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// k_ket<N>: register-resident kets of 12-14 atoms, whole schedules in one launch
// ---------------------------------------------------------------------------
// A 14-atom ket is 256 KiB - half of a CU's register file.  A Taylor / Horner
// exponential needs two copies (the iterate and its base), which no CU can hold;
// the persistent kernel k_traj therefore stops at 13 atoms (and spills there).
// For a REAL symmetric H~ (every drive coefficient real: phase 0 or pi) the flow
// of  i psi' = H~ psi  with  psi = q + i p  is the rotation  q' = H~ p, p' = -H~ q,
// and a product of shears
//
//     q += a_1 x p;  p -= b_1 x q;  ...  p -= b_m x q;  q += a_{m+1} x p     (x = h H~)
//
// only ever adds H~ times ONE array to the OTHER: it runs in place on a single
// copy.  The palindromic coefficients (symp_coefs.hpp, fitted by
// tools/symplectic_coefs.py) make the 2x2 propagator match the rotation to
// 1e-10 .. 1e-13 for every eigenvalue inside the spectral bound the stepper
// computes anyway; m stages cost m generator applications, like a degree-m
// Taylor polynomial, and reach the accuracy of degree m + 2.
//
// Layout: 512 threads, thread t owns the amplitudes t + 512 j (j < R = 2^N / 512)
// as q[R], p[R] in registers.  Per half-stage the source array (128 KiB of
// doubles at N = 14) is published to LDS once; partners of index bits 0-8 are
// ds_read_b128 (two amplitudes per read), bits 9+ are register-to-register.
// HBM is touched for the initial load, snapshots and the final store only.
//
// Two uses:
//  * sesolve batches of 13/14-atom sequences (one workgroup per sequence);
//  * the split-operator master equation (host_rowpath.hpp): a density matrix is
//    2^N rows, each a ket that is right-multiplied by U^dagger (conj = -1); the
//    dephasing-type diagonal of the dissipator is an elementwise factor applied
//    at load / store.
#include "symp_coefs.hpp"

__constant__ SympScheme kSympDev[sizeof(kSymp) / sizeof(kSymp[0])];
#define KET_C1 0.21132486540518713  /* first Gauss node, 1/2 - sqrt(3)/6 */

struct KetStep {
  double h, u1, u2;
  double shift_a, shift_b;
  int idx;
  short sch_a, sub_a, sch_b, sub_b;  // scheme index / equal sub-exponentials
  int snap;                          // snapshot slot written after this step, or -1
  // The spectral shifts only contribute a global phase e^{-i sum h shift}: it is not applied per
  // exponential but accumulated on the host (sign of the launch folded in) and applied to
  // snapshots / the final store - cos and sin come from the host because a double-precision
  // sincos inside the kernel spills the whole register-resident state.
  double cum_phase, cum_cs, cum_sn;  // through the end of this step, from the start of the schedule
};

struct KetArgs {
  cplx* state;              // [n_rows][2^N] in/out
  cplx* snaps;              // [n_slots][n_rows][2^N] or null
  const cplx* pp;           // [n_series][n_int][4]
  const ryd_qdesc* desc;    // [B][N]
  const ryd_dterm* dterms;  // or null
  const double* e0;
  long long e0_stride;
  const KetStep* steps;
  int n_int, n_steps;
  int rows_log2;            // rows per batch entry = 2^rows_log2 (0: kets; N: density-matrix rows)
  double a1, a2;
  double conj_sign;         // +1: psi <- U psi;  -1: row <- row U^dagger (psi <- conj(U) psi)
  // elementwise real factor exp(sum_k fac[k] * n_k(row, col)) over the four (row bit, column bit)
  // counts n00, n01, n10, n11 (dissipator diagonal times a time span); row mode only
  int use_pre, use_post;
  // drive-only kick exp(-i kick H_drive(t_kick)) before the first (kick_pre) / after the last step
  // (kick_post): the commutator correction of the 4th-order operator splitting (host_ket.hpp)
  double kick_pre, kick_post;
  int kick_idx;
  double kick_u;
  double fin_cs, fin_sn;    // global phase of this launch's steps, applied at the final store
  const double* ftab;       // [2][4][16] = exp(pre[k] n), exp(post[k] n)  (host-computed, device memory)
  double gauge_eps2;        // KET_GAUGE: |c|^2 below which a drive has no direction of its own
};

// MODE of k_ket:
//  KET_PLAIN  real drive coefficients, kets (sesolve batches);
//  KET_ROWS   the rows of a density matrix (split-operator master equation): elementwise load / store
//             factors and the drive-only commutator kick;
//  KET_GAUGE  COMPLEX drives c_k(t) = 0.5 Omega e^{-i phi} (hamiltonian.py:349-351) gauged away: with
//             psi~ = prod_k exp(i theta_k(t) n_k) psi and e^{i theta_k} = w_k = c_k / r_k (r_k real, signed,
//             continuous in time) the Hamiltonian is real symmetric again - drive r_k, detuning
//             delta_k + theta_k', theta_k' = Im(c_k' conj c_k) / |c_k|^2 - so the in-place scheme applies.
//             |c| and theta' are not polynomials, so the two CF4 exponents are formed from the moments
//             int f dt and int (t - t_mid) f dt by 4-point Gauss-Legendre quadrature (with the 2 Gauss points of
//             the polynomial case the quadrature error of |c(t)| dominates: 5e-7 instead of 9e-10 on the
//             probe tests/probes/gauge_probe.py).  The state is rotated into the gauge at the load and back
//             at snapshots / the final store (per-atom unit factors w_k, no trigonometric functions).
enum { KET_PLAIN = 0, KET_ROWS = 1, KET_GAUGE = 2 };

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// DPPM: mask of the index bits 0-3 whose partners come over the DPP crossbar (quad permutes for bits 0 and
// 1, row rotate by 8 for bit 3: one v_mov_dpp per 32-bit half; bit 2 needs two); every other thread bit is a
// ds_read_b128 from the published copy.  The two resources trade against each other: a DPP-served bit costs
// 4 (bit 2: 8) vector instructions per amplitude pair, an LDS-served one 64 LDS cycles per pair and CU.
// Measured at 14 atoms, us per stage (256 / 1024 rows; round 3, after the partner addresses moved into the
// read's immediate offset): 0b1111: 10.49 / 9.42, 0b0011: 10.15 / 8.95, 0b0000: 10.93 / 9.83,
// 0b1011 (the cheap three): see profiles/r03_kket_variants.md.  Round 2 (address arithmetic per read still
// in the loop): 0b1111 9.9, 0b0000 10.7.  LOGNT: log2 of the workgroup size (1024 threads x 16 amplitudes
// spill in the hot loop: 13.5).
#ifndef RYD_KET_DPPM
#define RYD_KET_DPPM 0xB
#endif
template <int N, int MODE = KET_PLAIN, unsigned DPPM = RYD_KET_DPPM, int LOGNT = 9>
__global__ __launch_bounds__(1 << LOGNT) void k_ket(const KetArgs A) {
  constexpr bool ROWS = MODE == KET_ROWS, GAUGE = MODE == KET_GAUGE;
  static_assert(DPPM < 16u, "only index bits 0-3 can use the DPP crossbar");
  constexpr int NDPP = __builtin_popcount(DPPM);
  constexpr int NLDS = LOGNT - NDPP;  // thread bits read from the published copy
  constexpr int D = 1 << N, NTT = 1 << LOGNT;
  constexpr int R = D / NTT;   // amplitudes per thread (8, 16, 32)
  constexpr int RP = R / 2;    // pairs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2* xs = reinterpret_cast<double2*>(smem);                  // [RP][512] published array
  double* cfA = reinterpret_cast<double*>(xs + (size_t)RP * NTT);  // [16][4]
  double* cfB = cfA + 64;
  double* ehi = cfB + 64;   // [2][R]: high-bit detuning sums minus the shift, exp A / exp B
  double* hfx = ehi + 2 * R;  // [16][2]
  double* ftab = hfx + 32;    // [2][4][16] load / store factor tables
  double* ehh = ftab + 128;   // [R] static diagonal of the register-index bits
  double* cfK = ehh + R;      // [16][4] drive coefficients of the splitting kick
  double* gw = cfK + 64;      // KET_GAUGE: [16][2] unit factors w_k of the atoms at one time
  double* gj = gw + 32;       // KET_GAUGE: [R][2] products of w over the excited register-index atoms

  const int tid = threadIdx.x;
  const size_t row = blockIdx.x;
  const int b = (int)(row >> A.rows_log2);
  const unsigned rowidx = (unsigned)(row & ((1ull << A.rows_log2) - 1ull));
  cplx* st = A.state + row * D;
  const double* e0g = A.e0 + (size_t)b * A.e0_stride;

  if constexpr (ROWS) {
    if (A.use_pre || A.use_post) {
      if (tid < 128) ftab[tid] = A.ftab[tid];
      __syncthreads();
    }
  }
  // exp(sum_k fac[k] n_k(row, col)) for col = tid + 512 j: the (row bit, column bit) pair counts are sums of
  // a per-thread part (column bits 0-8) and a per-j part (register-index bits), and the tables hold
  // exp(fac n), so the factor is a product  F_thread x F_j : one double per thread and an LDS table of R
  // entries per direction (load / store) - one multiplication per element, nothing per element kept alive
  // (the first version looked four table entries up per element and cost 496 B per lane of scratch)
  double ft_pre = 1.0, ft_post = 1.0;
  double* fjt = gj;  // [2][R]: ROWS and GAUGE never coexist, the slot of the gauge tables is free here
  if constexpr (ROWS) {
    const unsigned lomask = (unsigned)NTT - 1u, hm = (1u << (N - LOGNT)) - 1u;
    const int t11 = __popc(rowidx & (unsigned)tid), t10 = __popc(rowidx & ~(unsigned)tid & lomask),
              t01 = __popc(~rowidx & (unsigned)tid & lomask), t00 = LOGNT - t11 - t10 - t01;
    if (A.use_pre) ft_pre = ftab[t00] * ftab[16 + t01] * ftab[32 + t10] * ftab[48 + t11];
    if (A.use_post) ft_post = ftab[64 + t00] * ftab[64 + 16 + t01] * ftab[64 + 32 + t10] * ftab[64 + 48 + t11];
    if (tid < 2 * R) {
      const int which = tid / R, j = tid % R;
      const unsigned rh = rowidx >> LOGNT;
      const int j11 = __popc(rh & (unsigned)j), j10 = __popc(rh & ~(unsigned)j & hm),
                j01 = __popc(~rh & (unsigned)j & hm), j00 = (N - LOGNT) - j11 - j10 - j01;
      const double* tab = ftab + 64 * which;
      const bool use = which ? A.use_post != 0 : A.use_pre != 0;
      fjt[tid] = use ? tab[j00] * tab[16 + j01] * tab[32 + j10] * tab[48 + j11] : 1.0;
    }
    __syncthreads();
  }

  // The static interaction diagonal E0 is a quadratic form of the index bits, so with
  // i = (j: register-index bits 9.., t: thread bits 0-8)
  //     E0(i) = Ehh(j) + Ell(t) + sum over the high atoms excited in j of V_a(t):
  // six doubles per thread (read back from the E0 table itself) instead of R, which is
  // what keeps q, p and the partner reads inside the 256 registers of a lane.
  constexpr int NH = N - LOGNT;               // register-index bits
  constexpr unsigned HIMASK = ((1u << NH) - 1u) << LOGNT;
  const double wst = A.a1 + A.a2;             // the static diagonal always enters with weight a1 + a2
  const double ell = wst * e0g[tid | HIMASK];
  double vhi[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) vhi[k] = wst * e0g[tid | (HIMASK ^ (1u << (LOGNT + k)))] - ell;
  if (tid < R) ehh[tid] = wst * e0g[((unsigned)tid << LOGNT) | (NTT - 1)];

  // ---- KET_GAUGE: direction w = c / r of a drive at one time, continuous along the lane's own history ----
  // lanes tid < N own atom tid; (wpx, wpy) = the direction at the previous evaluation (have_w: one exists)
  double wpx = 1.0, wpy = 0.0;
  bool have_w = false;
  ryd_qdesc gd;
  if constexpr (GAUGE) {
    if (tid < N) gd = A.desc[(size_t)b * N + tid];
  }
  // evaluates atom tid's drive at offset u inside knot interval idx: returns r (signed modulus, times the
  // scale), theta' and updates the direction
  auto gauge_eval = [&](int idx, double u, double* r_out, double* thd_out) {
    double r = 0.0, thd = 0.0;
    if (gd.drive_series >= 0 && gd.drive_scale != 0.0) {
      const cplx* pq = A.pp + ((size_t)gd.drive_series * A.n_int + idx) * 4;
      const cplx p0 = pq[0], p1 = pq[1], p2 = pq[2], p3 = pq[3];
      const double x = fma(fma(fma(p0.x, u, p1.x), u, p2.x), u, p3.x);
      const double y = fma(fma(fma(p0.y, u, p1.y), u, p2.y), u, p3.y);
      const double dx = fma(fma(3.0 * p0.x, u, 2.0 * p1.x), u, p2.x);
      const double dy = fma(fma(3.0 * p0.y, u, 2.0 * p1.y), u, p2.y);
      const double m2 = x * x + y * y;
      double wx = wpx, wy = wpy;
      bool fresh = false;
      if (m2 > A.gauge_eps2) {
        const double inv = 1.0 / sqrt(m2);
        wx = x * inv;
        wy = y * inv;
        thd = (dy * x - dx * y) / m2;
        fresh = true;
      } else if (!have_w) {  // a drive that starts from zero points along its derivative
        const double d2 = dx * dx + dy * dy;
        if (d2 > 0.0) {
          const double inv = 1.0 / sqrt(d2);
          wx = dx * inv;
          wy = dy * inv;
          fresh = true;
        }
      }
      if (have_w && fresh && wx * wpx + wy * wpy < 0.0) { wx = -wx; wy = -wy; }  // r changes sign, w does not jump
      r = gd.drive_scale * (x * wx + y * wy);  // c = r w  =>  r = Re(c conj w)
      if (fresh) { wpx = wx; wpy = wy; have_w = true; }
    }
    *r_out = r;
    *thd_out = thd;
  };
  // unit factors of the current directions -> this thread's product over its excited thread-bit atoms (T) and
  // the table over the register-index atoms (gj); psi~ = (T J) psi, psi = conj(T J) psi~
  double Tx = 1.0, Ty = 0.0;
  auto gauge_publish = [&]() {
    __syncthreads();
    if (tid < N) { gw[2 * tid] = wpx; gw[2 * tid + 1] = wpy; }
    __syncthreads();
    Tx = 1.0; Ty = 0.0;
#pragma unroll
    for (int f = 0; f < LOGNT; ++f)
      if (!((tid >> f) & 1)) {  // atom N-1-f excited
        const double gx = gw[2 * (N - 1 - f)], gy = gw[2 * (N - 1 - f) + 1];
        const double nx = Tx * gx - Ty * gy;
        Ty = Tx * gy + Ty * gx;
        Tx = nx;
      }
    if (tid < R) {
      double jx = 1.0, jy = 0.0;
#pragma unroll
      for (int f = LOGNT; f < N; ++f)
        if (!((tid >> (f - LOGNT)) & 1)) {
          const double gx = gw[2 * (N - 1 - f)], gy = gw[2 * (N - 1 - f) + 1];
          const double nx = jx * gx - jy * gy;
          jy = jx * gy + jy * gx;
          jx = nx;
        }
      gj[2 * tid] = jx;
      gj[2 * tid + 1] = jy;
    }
    __syncthreads();
  };

  double q[R], p[R];
  if constexpr (GAUGE) {
    if (tid < N && A.n_steps > 0) {
      const KetStep s0 = A.steps[0];
      double r0, t0;
      gauge_eval(s0.idx, s0.u1 - KET_C1 * s0.h, &r0, &t0);  // start of the first step
    }
    gauge_publish();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const cplx v = st[tid + j * NTT];
      const double fx = Tx * gj[2 * j] - Ty * gj[2 * j + 1], fy = Tx * gj[2 * j + 1] + Ty * gj[2 * j];
      q[j] = v.x * fx - v.y * fy;
      p[j] = v.x * fy + v.y * fx;
    }
  } else if constexpr (ROWS) {
    // loads in groups of eight (with the factors applied group by group): hoisting all 32 loads and all 32
    // factor evaluations together cost 496 B per lane of scratch, written and read back by every row
    constexpr int G = R < 8 ? R : 8;
#pragma unroll
    for (int j0 = 0; j0 < R; j0 += G) {
      cplx v[G];
#pragma unroll
      for (int j = 0; j < G; ++j) v[j] = st[tid + (j0 + j) * NTT];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const double f = ft_pre * fjt[j0 + j];
        q[j0 + j] = f * v[j].x;
        p[j0 + j] = f * v[j].y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const cplx v = st[tid + j * NTT];
      q[j] = v.x;
      p[j] = v.y;
    }
  }

  const bool has_kick = ROWS && (A.kick_pre != 0.0 || A.kick_post != 0.0);
  if (has_kick && tid < N) {
    const ryd_qdesc d = A.desc[(size_t)b * N + tid];
    double c = 0.0;
    if (d.drive_series >= 0) {
      const cplx* pq = A.pp + ((size_t)d.drive_series * A.n_int + A.kick_idx) * 4;
      c = fma(fma(fma(pq[0].x, A.kick_u, pq[1].x), A.kick_u, pq[2].x), A.kick_u, pq[3].x);
    }
    cfK[4 * tid] = d.drive_scale * c;
    cfK[4 * tid + 2] = 0.0;
  }

  for (int s = 0; s < A.n_steps; ++s) {
    const KetStep sd = A.steps[s];
    if (A.dterms) {
      // extra detuning terms (hf noise): wave w sums the list of atoms w, w + NW, ...
      constexpr int NW = NTT / 64;
      const int lane = tid & 63;
      for (int k = tid >> 6; k < N; k += NW) {
        const int ex = A.desc[(size_t)b * N + k].extra;
        double xa = 0.0, xb = 0.0;
        if (ex > 0) {
          const int count = A.dterms[ex - 1].remaining + 1;
          for (int e = lane; e < count; e += 64) {
            const ryd_dterm t = A.dterms[ex - 1 + e];
            const cplx* pq = A.pp + ((size_t)t.series * A.n_int + sd.idx) * 4;
            const double o1 = ((pq[0].x * sd.u1 + pq[1].x) * sd.u1 + pq[2].x) * sd.u1 + pq[3].x;
            const double o2 = ((pq[0].x * sd.u2 + pq[1].x) * sd.u2 + pq[2].x) * sd.u2 + pq[3].x;
            xa += t.scale * (A.a1 * o1 + A.a2 * o2);
            xb += t.scale * (A.a2 * o1 + A.a1 * o2);
          }
          for (int o = 32; o > 0; o >>= 1) {
            xa += __shfl_down(xa, o, 64);
            xb += __shfl_down(xb, o, 64);
          }
        }
        if (lane == 0) { hfx[2 * k] = xa; hfx[2 * k + 1] = xb; }
      }
      __syncthreads();
    }
    if (tid < N) {
      // same arithmetic as k_eval_coefs / k_traj (w1 * val(t1) + w2 * val(t2)); only the real
      // part of the drive is used (complex drives: KET_GAUGE below, or the other kernels)
      const ryd_qdesc d = A.desc[(size_t)b * N + tid];
      auto val = [&](int ser, double u) -> double {
        const cplx* pq = A.pp + ((size_t)ser * A.n_int + sd.idx) * 4;
        double r = pq[0].x;
        r = fma(r, u, pq[1].x);
        r = fma(r, u, pq[2].x);
        r = fma(r, u, pq[3].x);
        return r;
      };
      double c1 = 0, c2 = 0, dlA = 0, dlB = 0;
      if constexpr (!GAUGE) {
        if (d.drive_series >= 0) { c1 = val(d.drive_series, sd.u1); c2 = val(d.drive_series, sd.u2); }
      }
      if (d.det_series >= 0) {
        const double d1 = val(d.det_series, sd.u1), d2 = val(d.det_series, sd.u2);
        dlA += d.det_scale * (A.a1 * d1 + A.a2 * d2);
        dlB += d.det_scale * (A.a2 * d1 + A.a1 * d2);
      }
      if (d.off_series >= 0) {
        const double o1 = val(d.off_series, sd.u1), o2 = val(d.off_series, sd.u2);
        dlA += d.off_scale * (A.a1 * o1 + A.a2 * o2);
        dlB += d.off_scale * (A.a2 * o1 + A.a1 * o2);
      }
      if (d.extra > 0 && A.dterms) { dlA += hfx[2 * tid]; dlB += hfx[2 * tid + 1]; }
      double drA, drB;
      if constexpr (GAUGE) {
        // exponents 1/2 B0 -+ 2 B1 from the moments B0 = int f, B1 = (1/h) int (t - t_mid) f of the drive
        // modulus and of theta', 4-point Gauss-Legendre: weights w_i (1/4 -+ x_i / 2) per unit step
        const double us = sd.u1 - KET_C1 * sd.h, um = us + 0.5 * sd.h, hh = 0.5 * sd.h;
        drA = 0.0; drB = 0.0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          constexpr double X[4] = {-0.8611363115940526, -0.3399810435848563, 0.3399810435848563, 0.8611363115940526};
          constexpr double W[4] = {0.3478548451374538, 0.6521451548625461, 0.6521451548625461, 0.3478548451374538};
          double r, thd;
          gauge_eval(sd.idx, fma(hh, X[g], um), &r, &thd);
          const double wa = W[g] * (0.25 - 0.5 * X[g]), wb = W[g] * (0.25 + 0.5 * X[g]);
          drA = fma(wa, r, drA);
          drB = fma(wb, r, drB);
          dlA = fma(wa, thd, dlA);
          dlB = fma(wb, thd, dlB);
        }
      } else {
        drA = d.drive_scale * (A.a1 * c1 + A.a2 * c2);
        drB = d.drive_scale * (A.a2 * c1 + A.a1 * c2);
      }
      cfA[4 * tid + 0] = drA;
      cfA[4 * tid + 2] = dlA;
      cfB[4 * tid + 0] = drB;
      cfB[4 * tid + 2] = dlB;
    }
    __syncthreads();
    if (tid < 2 * R) {
      // diagonal carried by the register-index bits (9 ..): static part + detunings - shift
      const int ex = tid / R, j = tid % R;
      const double* cf = ex ? cfB : cfA;
      double sdet = 0.0;
      for (int qb = LOGNT; qb < N; ++qb)
        if (!((j >> (qb - LOGNT)) & 1)) sdet -= cf[4 * (N - 1 - qb) + 2];
      ehi[tid] = ehh[j] + sdet - (ex ? sd.shift_b : sd.shift_a);
    }
    __syncthreads();

    // ex = 0, 1: the two exponentials of the CF4 step; ex = -1 / 2: the drive-only kick before
    // the first / after the last step of the launch
    const int ex_lo = (ROWS && s == 0 && A.kick_pre != 0.0) ? -1 : 0;
    const int ex_hi = (ROWS && s == A.n_steps - 1 && A.kick_post != 0.0) ? 2 : 1;
#pragma unroll 1
    for (int ex = ex_lo; ex <= ex_hi; ++ex) {
      const bool kick = ROWS && (ex < 0 || ex > 1);
      const double* cf = kick ? cfK : (ex ? cfB : cfA);
      const int sch = ex == 1 ? sd.sch_b : sd.sch_a;
      const int nsub = ex == 1 ? sd.sub_b : sd.sub_a;
      // per-bit drive coefficients (bit f <-> atom N-1-f) -> scalar registers
      double cq[N];
#pragma unroll
      for (int f = 0; f < N; ++f) cq[f] = uniform_d(cf[4 * (N - 1 - f)]);
      double elo = ell;  // thread-bit part of the diagonal: static + detunings
#pragma unroll
      for (int f = 0; f < LOGNT; ++f)
        if (!((tid >> f) & 1)) elo -= cf[4 * (N - 1 - f) + 2];
      const double* ehx = ehi + (ex == 1 ? R : 0);
      const int m = kSympDev[sch].m;
      const double hs = A.conj_sign * sd.h / (double)nsub;

      // dst += coef * (H~ - shift) src, in place on the register arrays.
      // Partners: index bits 0, 1, 3 over the DPP crossbar (quad permutes, row rotate by 8),
      // bit 2 as row_half_mirror + reversed quad permute (two moves), bits 4-8 as ds_read_b128
      // from the published copy (two amplitudes per read), bits 9.. register to register.
      // One pair (2jp, 2jp+1) of  dst += coef * (H~ - shift) src ; `ra` = byte address of this
      // thread's slot of pair jp inside the published copy.
      constexpr int NLD = NLDS + 1;  // partner reads + the diagonal table entry of a pair
      // k-th LDS-served bit (ascending)
      auto lds_bit = [](int k) constexpr -> int {
        int seen = 0;
        for (int f = 0; f < LOGNT; ++f)
          if (!((DPPM >> f) & 1u)) { if (seen == k) return f; ++seen; }
        return -1;
      };
      // Partner addresses = a per-thread base per index bit (own slot with bit f flipped: +-(16 << f) bytes,
      // the sign is the thread's own bit) + a compile-time offset per pair, which rides in the 16-bit
      // immediate of ds_read_b128: no address arithmetic per read (it was one v_xor + adds per read: 100
      // of the 586 vector instructions per 16 amplitudes)
      constexpr unsigned STRIDE = NTT * 16u;  // bytes per pair plane of the published copy
      auto load_pair = [&](double2 (&pv)[NLD], int jp, const unsigned (&base)[NLDS + 1], int jp0) {
#pragma unroll
        for (int k = 0; k < NLDS; ++k)
          pv[k] = *reinterpret_cast<const double2*>(smem + base[k] + (unsigned)(jp - jp0) * STRIDE);
        pv[NLD - 1] = *reinterpret_cast<const double2*>(ehx + 2 * jp);
      };
      auto do_pair = [&](double (&dst)[R], const double (&src)[R], double coef, int jp, const double2 (&pv)[NLD]) {
        const double2 eh2 = pv[NLD - 1];
        // diagonal of the pair: high atoms excited <=> their bit is 0.  Re-derived per pair (<= 5
        // additions): hoisted out of the stage loop the R partial sums would live in scratch and
        // every pair would wait for a scratch load
        double ec = elo;
        asm volatile("" : "+v"(ec));
#pragma unroll
        for (int k = 1; k < NH; ++k)
          if (!((jp >> (k - 1)) & 1)) ec += vhi[k];
        const double s0 = src[2 * jp], s1 = src[2 * jp + 1];
        double acc0 = ((ec + vhi[0]) + eh2.x) * s0;
        double acc1 = (ec + eh2.y) * s1;
        // second chains: the register-index and DPP partners (independent of the LDS reads)
        double bcc0 = cq[LOGNT] * s1;  // bit 9 pairs (2jp, 2jp+1)
        double bcc1 = cq[LOGNT] * s0;
#pragma unroll
        for (int k = 0; k + LOGNT + 1 < N; ++k) {  // bits 10..: jp <-> jp ^ 2^k
          const int jo = jp ^ (1 << k);
          bcc0 = fma(cq[LOGNT + 1 + k], src[2 * jo], bcc0);
          bcc1 = fma(cq[LOGNT + 1 + k], src[2 * jo + 1], bcc1);
        }
        if constexpr (DPPM & 1u) {
          bcc0 = fma(cq[0], dpp_f64<0xB1>(s0), bcc0);   // xor 1
          bcc1 = fma(cq[0], dpp_f64<0xB1>(s1), bcc1);
        }
        if constexpr (DPPM & 2u) {
          bcc0 = fma(cq[1], dpp_f64<0x4E>(s0), bcc0);   // xor 2
          bcc1 = fma(cq[1], dpp_f64<0x4E>(s1), bcc1);
        }
        if constexpr (DPPM & 8u) {
          bcc0 = fma(cq[3], dpp_f64<0x128>(s0), bcc0);  // xor 8 (row rotate by 8)
          bcc1 = fma(cq[3], dpp_f64<0x128>(s1), bcc1);
        }
        if constexpr (DPPM & 4u) {
          bcc0 = fma(cq[2], dpp_f64<0x1B>(dpp_f64<0x141>(s0)), bcc0);  // xor 4 = (xor 7) o (xor 3)
          bcc1 = fma(cq[2], dpp_f64<0x1B>(dpp_f64<0x141>(s1)), bcc1);
        }
#pragma unroll
        for (int k = 0; k < NLDS; ++k) {
          acc0 = fma(cq[lds_bit(k)], pv[k].x, acc0);
          acc1 = fma(cq[lds_bit(k)], pv[k].y, acc1);
        }
        dst[2 * jp] = fma(coef, acc0 + bcc0, dst[2 * jp]);
        dst[2 * jp + 1] = fma(coef, acc1 + bcc1, dst[2 * jp + 1]);
      };

      // dst += coef * (H~ - shift) src, in place on the register arrays.
      // Partners: index bits 0, 1, 3 over the DPP crossbar (quad permutes, row rotate by 8),
      // bit 2 as row_half_mirror + reversed quad permute (two moves), bits 4-8 as ds_read_b128
      // from the published copy (two amplitudes per read), bits 9.. register to register.
      //
      // The published copy of `src` lives in two halves: A = pairs [0, H), B = pairs [H, RP).
      // Invariant on entry: A holds src[0, H).  First half: pairs [0, H) are computed from A while
      // src[H, RP) is written into B; second half: pairs [H, RP) are computed from B while the
      // finished dst[0, H) is written into A for the next half-stage (q <- p and p <- q alternate
      // inside an exponential; its first half-stage re-publishes p, see below).  The LDS stores ride along with the arithmetic instead of forming a store-only
      // phase (17 % of a stage before), and there are still two barriers per half-stage.
      auto half_stage = [&](double (&dst)[R], const double (&src)[R], double coef) {
        constexpr int H = RP / 2;
        unsigned base[NLDS + 1];  // [0, NLDS): partner bases; last: the thread's own slot
#pragma unroll
        for (int k = 0; k < NLDS; ++k) base[k] = ((unsigned)tid * 16u) ^ (16u << lds_bit(k));
        base[NLDS] = (unsigned)tid * 16u;
#pragma unroll
        for (int f = 0; f <= NLDS; ++f) asm volatile("" : "+v"(base[f]));
        if constexpr (RP >= 2) {
          // software pipeline: the LDS reads of pair jp + 1 are in flight while pair jp is computed
          // (two register sets; with two waves per SIMD the read latency was not covered otherwise)
          double2 pvs[2][NLD];
          load_pair(pvs[0], 0, base, 0);
#pragma unroll
          for (int jp = 0; jp < H; ++jp) {
            *reinterpret_cast<double2*>(smem + base[NLDS] + (unsigned)(jp + H) * STRIDE) =
                make_double2(src[2 * (jp + H)], src[2 * (jp + H) + 1]);
            if (jp + 1 < H) load_pair(pvs[(jp + 1) & 1], jp + 1, base, 0);
            __builtin_amdgcn_sched_barrier(0);  // issue order: next pair's reads, then this pair's arithmetic
            do_pair(dst, src, coef, jp, pvs[jp & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          __syncthreads();  // B complete; every read of A done
          // second half: the bases move up by H planes (64 KiB at 14 atoms: beyond the immediate's reach)
#pragma unroll
          for (int f = 0; f <= NLDS; ++f) {
            base[f] += (unsigned)H * STRIDE;
            asm volatile("" : "+v"(base[f]));
          }
          load_pair(pvs[H & 1], H, base, H);
#pragma unroll
          for (int jp = H; jp < RP; ++jp) {
            // finished dst[0, H) goes into A = H planes below this half's bases
            *reinterpret_cast<double2*>(smem + (base[NLDS] - (unsigned)H * STRIDE) + (unsigned)(jp - H) * STRIDE) =
                make_double2(dst[2 * (jp - H)], dst[2 * (jp - H) + 1]);
            if (jp + 1 < RP) load_pair(pvs[(jp + 1) & 1], jp + 1, base, H);
            __builtin_amdgcn_sched_barrier(0);
            do_pair(dst, src, coef, jp, pvs[jp & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
          __syncthreads();  // A complete (next source); every read of B done
        } else {
          __syncthreads();
          *reinterpret_cast<double2*>(smem + (unsigned)tid * 16u) = make_double2(src[0], src[1]);
          __syncthreads();
          double2 pv1[NLD];
          load_pair(pv1, 0, base, 0);
          do_pair(dst, src, coef, 0, pv1);
        }
      };
      // slow path (kernel start, kicks): make A hold arr[0, H) whatever was there
      auto republish_low = [&](const double (&arr)[R]) {
        if constexpr (RP >= 2) {
          __syncthreads();
#pragma unroll
          for (int jp = 0; jp < RP / 2; ++jp)
            *reinterpret_cast<double2*>(smem + (unsigned)tid * 16u + (unsigned)jp * NTT * 16u) =
                make_double2(arr[2 * jp], arr[2 * jp + 1]);
          __syncthreads();
        }
      };

      // The exponential as ONE loop of half-stages, so that the hot code exists twice only
      // (q <- p and p <- q): 2 m nsub + 1 shears of the in-place scheme (consecutive
      // sub-exponentials share their boundary shear a_{m+1} + a_1), or - for the splitting kick
      // exp(-i hk F), F = sum_k c_k X_k (drive only), hk ~ 1e-11 - one symmetric shear triple
      // (exact to hk^3) through the same code, whose diagonal term is taken out again afterwards.
      const double hk = ex < 0 ? A.kick_pre : A.kick_post;
      auto undo_diag = [&](double (&dst)[R], const double (&src)[R], double coef) {
        if constexpr (!ROWS) return;
#pragma unroll
        for (int jp = 0; jp < RP; ++jp) {
          const double2 eh2 = *reinterpret_cast<const double2*>(ehx + 2 * jp);
          double ec = elo;
          asm volatile("" : "+v"(ec));  // not hoisted out of the shear loop (the R / 2 partial sums would be parked in scratch)
#pragma unroll
          for (int k = 1; k < NH; ++k)
            if (!((jp >> (k - 1)) & 1)) ec += vhi[k];
          dst[2 * jp] = fma(-coef * ((ec + vhi[0]) + eh2.x), src[2 * jp], dst[2 * jp]);
          dst[2 * jp + 1] = fma(-coef * (ec + eh2.y), src[2 * jp + 1], dst[2 * jp + 1]);
        }
      };
      const int n_pairs = kick ? 1 : m * nsub;  // (q-shear, p-shear) pairs before the closing q-shear
      int i = 0, sb = 0;
#pragma unroll 1
      for (int t = 0; t <= n_pairs; ++t) {
        double ca, cb = 0.0;
        if (kick) {
          ca = 0.5 * hk;
          cb = -hk;
        } else if (t == n_pairs) {
          ca = kSympDev[sch].a[m] * hs;
        } else {
          ca = (kSympDev[sch].a[i] + ((i == 0 && sb > 0) ? kSympDev[sch].a[m] : 0.0)) * hs;
          cb = -kSympDev[sch].b[i] * hs;
        }
        if (kick || t == 0) republish_low(p);  // the exponential (or kick shear) starts from p
        half_stage(q, p, ca);
        if (kick) { undo_diag(q, p, ca); republish_low(q); }
        if (t < n_pairs) {
          half_stage(p, q, cb);
          if (kick) undo_diag(p, q, cb);
          if (++i == m) { i = 0; ++sb; }
        }
      }
    }
    if (sd.snap >= 0 && A.snaps) {
      cplx* o = A.snaps + ((size_t)sd.snap * gridDim.x + row) * D;
      int tid_sn = tid;
      asm volatile("" : "+v"(tid_sn));  // addresses recomputed here, not parked in scratch from the loads on
      if constexpr (GAUGE) {
        // back to the laboratory gauge at the END of this step: psi = conj(T J) e^{-i phase} psi~
        if (tid < N) {
          double r1, t1;
          gauge_eval(sd.idx, sd.u1 + (1.0 - KET_C1) * sd.h, &r1, &t1);
        }
        gauge_publish();
        const double ax = Tx * sd.cum_cs - Ty * sd.cum_sn, ay = Tx * sd.cum_sn + Ty * sd.cum_cs;  // T e^{+i phase}
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const double fx = ax * gj[2 * j] - ay * gj[2 * j + 1], fy = ax * gj[2 * j + 1] + ay * gj[2 * j];
          o[tid_sn + j * NTT] = make_double2(q[j] * fx + p[j] * fy, p[j] * fx - q[j] * fy);  // (q + i p) conj(f)
        }
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j)  // psi * e^{-i phase}: the accumulated spectral shifts
          o[tid_sn + j * NTT] = make_double2(fma(q[j], sd.cum_cs, p[j] * sd.cum_sn), fma(p[j], sd.cum_cs, -q[j] * sd.cum_sn));
      }
    }
    __syncthreads();  // cf / ehi are rewritten by the next step
  }
  // the store addresses are recomputed from an opaque copy of the thread index: kept alive from the loads
  // they would be parked in scratch for the whole kernel (46 doubles per lane at 14 atoms)
  int tid_st = tid;
  asm volatile("" : "+v"(tid_st));
  if constexpr (GAUGE) {
    if (tid < N && A.n_steps > 0) {
      const KetStep sl = A.steps[A.n_steps - 1];
      double r1, t1;
      gauge_eval(sl.idx, sl.u1 + (1.0 - KET_C1) * sl.h, &r1, &t1);
    }
    gauge_publish();
    const double ax = Tx * A.fin_cs - Ty * A.fin_sn, ay = Tx * A.fin_sn + Ty * A.fin_cs;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const double fx = ax * gj[2 * j] - ay * gj[2 * j + 1], fy = ax * gj[2 * j + 1] + ay * gj[2 * j];
      st[tid_st + j * NTT] = make_double2(q[j] * fx + p[j] * fy, p[j] * fx - q[j] * fy);
    }
  } else if constexpr (ROWS) {
    constexpr int G = R < 8 ? R : 8;
    const double* fj_st = fjt + R;
    asm volatile("" : "+v"(fj_st));  // read now, not hoisted to the top of the kernel and parked
#pragma unroll
    for (int j0 = 0; j0 < R; j0 += G) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const double f = ft_post * fj_st[j0 + j];
        st[tid_st + (j0 + j) * NTT] = make_double2(f * fma(q[j0 + j], A.fin_cs, p[j0 + j] * A.fin_sn),
                                               f * fma(p[j0 + j], A.fin_cs, -q[j0 + j] * A.fin_sn));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int j = 0; j < R; ++j)
      st[tid_st + j * NTT] = make_double2(fma(q[j], A.fin_cs, p[j] * A.fin_sn), fma(p[j], A.fin_cs, -q[j] * A.fin_sn));
  }
}

// out[b][c][r] = conj(in[b][r][c]) for 2^N x 2^N matrices, 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose_conj(const cplx* __restrict__ in, cplx* __restrict__ out, int N) {
  __shared__ cplx t[32][33];
  const size_t D = (size_t)1 << N;
  const size_t boff = (size_t)blockIdx.z * D * D;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t r0 = (size_t)blockIdx.y * 32, c0 = (size_t)blockIdx.x * 32;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[ty + 8 * r][tx] = in[boff + (r0 + ty + 8 * r) * D + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const cplx v = t[tx][ty + 8 * r];
    out[boff + (c0 + ty + 8 * r) * D + r0 + tx] = make_double2(v.x, -v.y);
  }
}

// ---------------------------------------------------------------------------
// k_local_exp: rho <- exp(tau * sum_atoms S_local) rho for local dissipators WITH double flips
// ---------------------------------------------------------------------------
// The dissipator of identical local collapse operators is a sum of commuting 4x4
// superoperators S on the digit pairs (a_k, b_k) (hamiltonian.py:97-124 puts the same
// operators on every atom), so exp(tau B) is the product over atoms of M = exp(tau S) - a
// "two-qubit gate" on (row bit k, column bit k).  S only has diagonal and double-flip entries
// ((a,b) <-> (1-a,1-b): relaxation, depolarizing, C rho C^+ terms), so M couples 00 <-> 11 and
// 01 <-> 10.  Tiling = the pair passes of the multi-launch Lindbladian (both bits of an atom in
// one tile): the tile is staged in LDS, every atom of the pass is applied in place (one barrier
// per atom), the tile is written back - 32 B per element per pass, three passes at 12-14 atoms.
struct LocalExpArgs {
  cplx* rho;       // in place
  Segs tile, outer;
  int nb, T, n_dbl;
  signed char dbl_qb[MAXD], dbl_qa[MAXD];
  cplx Mdiag[4];   // M[r][r],   r = 2 a + b
  cplx Mflip[4];   // M[r][3-r]
};

__global__ __launch_bounds__(512) void k_local_exp(const LocalExpArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* xs = reinterpret_cast<cplx*>(smem);
  const int tileSize = 1 << A.T;
  const unsigned long long base_idx = deposit((unsigned long long)blockIdx.x, A.outer);
  cplx* g = A.rho + ((size_t)blockIdx.y << A.nb);
  for (int l = threadIdx.x; l < tileSize; l += 512) xs[l] = g[base_idx | deposit((unsigned long long)l, A.tile)];
  for (int d = 0; d < A.n_dbl; ++d) {
    __syncthreads();
    const int qb = A.dbl_qb[d], qa = A.dbl_qa[d];
    const int lo = qb < qa ? qb : qa, hi = qb < qa ? qa : qb;
    for (int t = threadIdx.x; t < (tileSize >> 2); t += 512) {
      // insert zeros at bit positions lo and hi
      int l = ((t >> lo) << (lo + 1)) | (t & ((1 << lo) - 1));
      l = ((l >> hi) << (hi + 1)) | (l & ((1 << hi) - 1));
      const int i00 = l, i01 = l | (1 << qb), i10 = l | (1 << qa), i11 = i01 | i10;  // r = 2 a + b
      const cplx v00 = xs[i00], v01 = xs[i01], v10 = xs[i10], v11 = xs[i11];
      xs[i00] = cfma(A.Mdiag[0], v00, cmul(A.Mflip[0], v11));
      xs[i11] = cfma(A.Mdiag[3], v11, cmul(A.Mflip[3], v00));
      xs[i01] = cfma(A.Mdiag[1], v01, cmul(A.Mflip[1], v10));
      xs[i10] = cfma(A.Mdiag[2], v10, cmul(A.Mflip[2], v01));
    }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < tileSize; l += 512) g[base_idx | deposit((unsigned long long)l, A.tile)] = xs[l];
}


// index of the largest diagonal entry of every density matrix of the batch (the row the split-operator master equation
// probes its unitary sub-steps on: rows_split_probe, host_split.hpp); grid (B), 1024 lanes
__global__ __launch_bounds__(1024) void k_argmax_diag(const cplx* __restrict__ rho, int N, int* __restrict__ idx_out) {
  __shared__ double sv[1024];
  __shared__ unsigned si[1024];
  const size_t D = (size_t)1 << N;
  const cplx* m = rho + (size_t)blockIdx.x * D * D;
  double best = -1.0;
  unsigned bi = 0;
  for (size_t i = threadIdx.x; i < D; i += 1024) {
    const double v = fabs(m[i * (D + 1)].x);
    if (v > best) { best = v; bi = (unsigned)i; }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (unsigned o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o && sv[threadIdx.x + o] > sv[threadIdx.x]) { sv[threadIdx.x] = sv[threadIdx.x + o]; si[threadIdx.x] = si[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) idx_out[blockIdx.x] = (int)si[0];
}
