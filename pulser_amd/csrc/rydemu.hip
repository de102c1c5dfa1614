// rydemu.hip - MI355X (gfx950 / CDNA4) emulation core behind include/rydemu.h.
//
// What it computes (restating, from scratch, the arithmetic the reference hands
// to QuTiP at pulser-simulation/pulser_simulation/simulation.py:729-735):
//
//   sesolve:  d/dt psi = G(t) psi,   G = -i H(t)
//   mesolve:  d/dt rho = G(t) rho,   G = Lindbladian, rho as a 2N-bit vector
//
// with, in both cases, G a sum of
//   * a diagonal   g(i)           (interaction + detuning + dissipator diagonal)
//   * single-bit flips  coef_p[bit_p(i)] * x[i ^ 2^p]   (the Rabi drive)
//   * (mesolve) double flips on the digit pair (a_k, b_k)  (C rho C^+ jumps).
//
// Time stepping: commutator-free 4th-order Magnus (two exponentials per step of
// the Hamiltonian evaluated at the two Gauss points), each exponential by a
// Horner-form Taylor polynomial whose only primitive is the generator
// application  out = base + scale * (G~ x)  - the batched matrix-free
// "state-vector x Hamiltonian matvec" of the north star.
//
// Kernel design for CDNA4: one workgroup owns a tile of 2^T amplitudes
// (T = 12 -> 64 KiB of LDS, two workgroups per CU).  The tile is staged into
// LDS with coalesced 16-byte loads (each lane one complex128; the low C tile
// bits are the low address bits, so a wave reads whole 1 KiB / 256 B runs),
// every flip partner inside the tile is one ds_read_b128, and index bits that
// do not fit the tile are handled by further passes over a different tiling
// that accumulate into a partial-sum buffer.  HBM-bound by construction: the
// algorithmic traffic is 32 B per amplitude per application.
//
// No reference code is used; file:line citations name the behaviour restated.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rydemu.h"

typedef double2 cplx;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail(RYD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                  hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
struct Segs {
  int lo[3];
  int len[3];
};

__host__ __device__ __forceinline__ unsigned long long deposit(
    unsigned long long v, const Segs& s) {
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r |= (v & ((1ull << s.len[i]) - 1ull)) << s.lo[i];
    v >>= s.len[i];
  }
  return r;
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) {  // a*b + c
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)),
                      fma(a.x, b.y, fma(a.y, b.x, c.y)));
}

// wave-uniform double -> scalar registers
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

#define MAXF 16  // flips per pass (<= tile bits)
#define MAXD 8   // double flips per pass

struct PassArgs {
  const cplx* in;    // x: the vector G is applied to
  const cplx* kin;   // partial sums of earlier passes (or null)
  cplx* kout;        // partial sums out (non-final pass)
  const cplx* base;  // Horner base (or null)
  cplx* out;         // final output: post * (base + scale * (kin + partial))
  const double* coefs;  // [B][N][4] = Re c~, Im c~, delta~, 0 (time-mixed)
  const double* e0;     // [n_mats][2^N] static interaction diagonal
  long long e0_stride;  // 0 when shared by the batch
  double wmix;          // weight of the static parts (w1 + w2)
  double scale;         // h / j
  double shift;         // spectral shift of H (sesolve)
  double dec_a, dec_b;  // Monte-Carlo wavefunction: real diagonal dec_a + dec_b * popc(i) (sesolve)
  cplx post;            // final multiplier
  cplx Sd[4];           // mesolve: dissipator diagonal, index 2*a_k + b_k
  cplx J[4];            // mesolve: double-flip coefficient, by output pair
  Segs tile, outer;
  int N, nb, T;
  int n_flip, n_dbl;
  int include_diag, final_pass;
  signed char flip_q[MAXF];  // tile-local bit of each single flip
  signed char dbl_qb[MAXD], dbl_qa[MAXD];
};

// Global bit position of tile-local bit q.
__device__ __forceinline__ int tile_bit_pos(const Segs& s, int q) {
  int off = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (q < off + s.len[i]) return s.lo[i] + (q - off);
    off += s.len[i];
  }
  return -1;
}

// threads per workgroup of the apply kernel: 512 (two workgroups per CU when the
// launch has many tiles) or 1024 (launches with at most ~2 tiles per CU: more
// waves per CU to hide the load -> compute -> store latency of a lone tile)

// out = post * (base + scale * (kin + G~_pass x))        (final pass)
// kout = kin + G~_pass x                                  (other passes)
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_apply(const PassArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = A.T;
  const int tileSize = 1 << T;
  const int TL = T >> 1, TH = T - TL;
  cplx* xs = reinterpret_cast<cplx*>(smem);
  double* tabLo = reinterpret_cast<double*>(xs + tileSize);
  double* tabHi = tabLo + (1 << TL);
  cplx* c0 = reinterpret_cast<cplx*>(tabHi + (1 << TH));
  cplx* c1 = c0 + MAXF;

  const int tid = threadIdx.x;
  const int N = A.N;
  const int b = blockIdx.y;
  const unsigned long long base_idx = deposit((unsigned long long)blockIdx.x, A.outer);
  const size_t boff = (size_t)b << A.nb;
  const double* __restrict__ cf = A.coefs + (size_t)b * N * 4;
  const cplx* __restrict__ xin = A.in + boff;

  // ---- stage the tile (coalesced 16 B / lane) ----
  for (int l = tid; l < tileSize; l += NT)
    xs[l] = xin[base_idx | deposit((unsigned long long)l, A.tile)];

  // ---- per-pass coefficient tables ----
  if (tid < A.n_flip) {
    const int p = tile_bit_pos(A.tile, A.flip_q[tid]);
    cplx lo, hi;  // coefficient when the OUTPUT index has bit p = 0 / 1
    if (MODE == RYD_SESOLVE) {
      const int k = N - 1 - p;
      const double cr = cf[4 * k], ci = cf[4 * k + 1];
      // (H psi)(s_k = 1) += c psi(s_k = 0); (s_k = 0) += conj(c) psi(s_k = 1); G = -iH
      hi = make_double2(ci, -cr);    // -i * c
      lo = make_double2(-ci, -cr);   // -i * conj(c)
    } else if (p >= N) {             // row bit: -i (H rho)
      const int k = 2 * N - 1 - p;
      const double cr = cf[4 * k], ci = cf[4 * k + 1];
      hi = make_double2(ci, -cr);
      lo = make_double2(-ci, -cr);
    } else {                         // column bit: +i (rho H)
      const int k = N - 1 - p;
      const double cr = cf[4 * k], ci = cf[4 * k + 1];
      lo = make_double2(-ci, cr);    // +i * c
      hi = make_double2(ci, cr);     // +i * conj(c)
    }
    c0[tid] = lo;
    c1[tid] = hi;
  }
  double eOuter = 0.0;
  if (A.include_diag) {
    // detuning part of the diagonal, split over (outer bits) + (low/high half
    // of the tile bits): e_det(i) = sum_bits sgn * delta~_k * n_k, n_k = !bit.
    for (int e = tid; e < (1 << TL) + (1 << TH); e += NT) {
      const bool hiHalf = e >= (1 << TL);
      const int v = hiHalf ? e - (1 << TL) : e;
      const int q0 = hiHalf ? TL : 0, nq = hiHalf ? TH : TL;
      double s = 0.0;
      for (int q = 0; q < nq; ++q) {
        const int p = tile_bit_pos(A.tile, q0 + q);
        double sg;
        int k;
        if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
        else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
        else { k = N - 1 - p; sg = 1.0; }
        if (!((v >> q) & 1)) s += sg * cf[4 * k + 2];
      }
      (hiHalf ? tabHi : tabLo)[v] = s;
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < A.outer.len[i]; ++j) {
        const int p = A.outer.lo[i] + j;
        double sg;
        int k;
        if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
        else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
        else { k = N - 1 - p; sg = 1.0; }
        if (!((base_idx >> p) & 1ull)) eOuter += sg * cf[4 * k + 2];
      }
  }
  __syncthreads();

  const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;
  const unsigned Dm1 = (MODE == RYD_MESOLVE) ? ((1u << N) - 1u) : 0u;
  const int maskLo = (1 << TL) - 1;

  for (int l = tid; l < tileSize; l += NT) {
    const unsigned long long gi = base_idx | deposit((unsigned long long)l, A.tile);
    cplx acc = make_double2(0.0, 0.0);
    if (A.include_diag) {
      const cplx x = xs[l];
      double e = tabLo[l & maskLo] + tabHi[l >> TL] + eOuter;
      if (MODE == RYD_SESOLVE) {
        e += A.wmix * e0[gi] - A.shift;
        acc = make_double2(e * x.y, -e * x.x);  // -i e x
        if (A.dec_a != 0.0 || A.dec_b != 0.0) {  // -(1/2) sum C^dag C of H_eff (diagonal)
          const double dr = fma(A.dec_b, (double)__popcll(gi), A.dec_a);
          acc.x = fma(dr, x.x, acc.x);
          acc.y = fma(dr, x.y, acc.y);
        }
      } else {
        const unsigned a = (unsigned)(gi >> N), bb = (unsigned)gi & Dm1;
        e += A.wmix * (e0[a] - e0[bb]);
        const int n11 = __popc(a & bb), n10 = __popc(a & ~bb & Dm1),
                  n01 = __popc(~a & bb & Dm1), n00 = N - n11 - n10 - n01;
        const double dr = A.wmix * (A.Sd[0].x * n00 + A.Sd[1].x * n01 +
                                    A.Sd[2].x * n10 + A.Sd[3].x * n11);
        const double di = A.wmix * (A.Sd[0].y * n00 + A.Sd[1].y * n01 +
                                    A.Sd[2].y * n10 + A.Sd[3].y * n11) - e;
        acc = make_double2(dr * x.x - di * x.y, dr * x.y + di * x.x);
      }
    }
    for (int f = 0; f < A.n_flip; ++f) {
      const int q = A.flip_q[f];
      const cplx xv = xs[l ^ (1 << q)];
      const cplx cc = ((l >> q) & 1) ? c1[f] : c0[f];
      acc = cfma(cc, xv, acc);
    }
    if (MODE == RYD_MESOLVE) {
      for (int d = 0; d < A.n_dbl; ++d) {
        const int qb = A.dbl_qb[d], qa = A.dbl_qa[d];
        const int r = (((l >> qa) & 1) << 1) | ((l >> qb) & 1);
        const cplx jc = A.J[r];
        const cplx xv = xs[l ^ (1 << qb) ^ (1 << qa)];
        acc = cfma(make_double2(jc.x * A.wmix, jc.y * A.wmix), xv, acc);
      }
    }
    const size_t go = boff + gi;
    if (A.kin) {
      const cplx kv = A.kin[go];
      acc.x += kv.x;
      acc.y += kv.y;
    }
    if (A.final_pass) {
      cplx r = make_double2(A.scale * acc.x, A.scale * acc.y);
      if (A.base) {
        const cplx bv = A.base[go];
        r.x += bv.x;
        r.y += bv.y;
      }
      A.out[go] = cmul(A.post, r);
    } else {
      A.kout[go] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// k_apply12: the T = 12 specialisation of k_apply (single flips only).
// Same arithmetic, restructured for the memory system: every global load of a
// phase is issued before any of them is consumed (x and E0 up front; the
// partial sums and the Horner base together, after the flip phase), each
// thread keeps its 8 amplitudes in registers (flips of tile bits 9-11 are
// register moves), LDS partner reads of one amplitude are issued as one batch,
// and the wave-uniform flip coefficients live in scalar registers.
// ---------------------------------------------------------------------------
template <int MODE, int Q0>
__global__ __launch_bounds__(512, 4) void k_apply12(const PassArgs A) {
  constexpr int T = 12, NTT = 512, R = 8, LOGNT = 9, TL = 6;
  constexpr int NFMAX = T - Q0;  // flips are the tile-local bits Q0 .. Q0 + n_flip - 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* xs = reinterpret_cast<cplx*>(smem);
  double* tabLo = reinterpret_cast<double*>(xs + (1 << T));
  double* tabHi = tabLo + (1 << TL);
  double* cft = tabHi + (1 << TL);  // [MAXF][2]: cr, ci of every flip

  const int tid = threadIdx.x;
  const int N = A.N;
  const int b = blockIdx.y;
  const unsigned long long base_idx = deposit((unsigned long long)blockIdx.x, A.outer);
  const size_t boff = (size_t)b << A.nb;
  const double* __restrict__ cf = A.coefs + (size_t)b * N * 4;
  const cplx* __restrict__ xin = A.in + boff;
  const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;
  const unsigned Dm1 = (MODE == RYD_MESOLVE) ? ((1u << N) - 1u) : 0u;
  constexpr int q0 = Q0;
  const int nf = A.n_flip;

  // ---- phase A: all loads of x (and the E0 entries) in flight at once ----
  cplx x[R];
  double e0v[R];
#pragma unroll
  for (int j = 0; j < R; ++j)
    x[j] = xin[base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile)];
  if (A.include_diag) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const unsigned long long g = base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile);
      if (MODE == RYD_SESOLVE) {
        e0v[j] = e0[g];
      } else {
        const unsigned a = (unsigned)(g >> N), bb = (unsigned)g & Dm1;
        e0v[j] = e0[a] - e0[bb];
      }
    }
  }
  if (tid < nf) {
    const int p = tile_bit_pos(A.tile, q0 + tid);
    const int k = (MODE == RYD_SESOLVE || p < N) ? N - 1 - p : 2 * N - 1 - p;
    cft[2 * tid] = cf[4 * k];
    cft[2 * tid + 1] = cf[4 * k + 1];
  }
  double eOuter = 0.0;
  if (A.include_diag) {
    for (int e = tid; e < 2 * (1 << TL); e += NTT) {
      const bool hiHalf = e >= (1 << TL);
      const int v = hiHalf ? e - (1 << TL) : e;
      const int qb = hiHalf ? TL : 0;
      double s = 0.0;
      for (int q = 0; q < TL; ++q) {
        const int p = tile_bit_pos(A.tile, qb + q);
        double sg;
        int k;
        if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
        else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
        else { k = N - 1 - p; sg = 1.0; }
        if (!((v >> q) & 1)) s += sg * cf[4 * k + 2];
      }
      (hiHalf ? tabHi : tabLo)[v] = s;
    }
    for (int i = 0; i < 3; ++i)
      for (int jj = 0; jj < A.outer.len[i]; ++jj) {
        const int p = A.outer.lo[i] + jj;
        double sg;
        int k;
        if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
        else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
        else { k = N - 1 - p; sg = 1.0; }
        if (!((base_idx >> p) & 1ull)) eOuter += sg * cf[4 * k + 2];
      }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) xs[tid + j * NTT] = x[j];
  __syncthreads();

  // wave-uniform flip coefficients -> scalar registers.  The coefficient of a
  // flip is (sgn * ci, s2 * cr): sgn = +1 / -1 for output bit 1 / 0, s2 = -1 for
  // -iH (sesolve, row bits) and +1 for +i rho H (column bits).
  double fcr[NFMAX], fci[NFMAX];
#pragma unroll
  for (int f = 0; f < NFMAX; ++f) {
    const bool on = f < nf;
    const double cr = on ? cft[2 * f] : 0.0, ci = on ? cft[2 * f + 1] : 0.0;
    double s2 = -1.0;
    if (MODE == RYD_MESOLVE && on && tile_bit_pos(A.tile, q0 + f) < N) s2 = 1.0;
    fcr[f] = uniform_d(s2 * cr);
    fci[f] = uniform_d(ci);
  }

  // ---- phase B: diagonal first (frees the E0 registers), then the flips ----
  cplx acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int l = tid + j * NTT;
    cplx a = make_double2(0.0, 0.0);
    if (A.include_diag) {
      double e = tabLo[l & ((1 << TL) - 1)] + tabHi[l >> TL] + eOuter;
      if (MODE == RYD_SESOLVE) {
        e += A.wmix * e0v[j] - A.shift;
        a = make_double2(e * x[j].y, -e * x[j].x);
      } else {
        const unsigned long long g = base_idx | deposit((unsigned long long)l, A.tile);
        const unsigned aa = (unsigned)(g >> N), bb = (unsigned)g & Dm1;
        e += A.wmix * e0v[j];
        const int n11 = __popc(aa & bb), n10 = __popc(aa & ~bb & Dm1),
                  n01 = __popc(~aa & bb & Dm1), n00 = N - n11 - n10 - n01;
        const double dr = A.wmix * (A.Sd[0].x * n00 + A.Sd[1].x * n01 +
                                    A.Sd[2].x * n10 + A.Sd[3].x * n11);
        const double di = A.wmix * (A.Sd[0].y * n00 + A.Sd[1].y * n01 +
                                    A.Sd[2].y * n10 + A.Sd[3].y * n11) - e;
        a = make_double2(dr * x[j].x - di * x[j].y, dr * x[j].y + di * x[j].x);
      }
    }
    acc[j] = a;
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int l = tid + j * NTT;
    cplx a = acc[j];
    constexpr int NX = LOGNT > Q0 ? LOGNT - Q0 : 1;  // partners read from LDS
    cplx xv[NX];
#pragma unroll
    for (int f = 0; f < NX; ++f)
      if (Q0 + f < LOGNT && f < nf) xv[f] = xs[l ^ (1 << (Q0 + f))];
#pragma unroll
    for (int f = 0; f < NFMAX; ++f) {
      constexpr int dummy = 0;
      (void)dummy;
      const int q = Q0 + f;
      if (f >= nf) continue;  // wave-uniform
      const cplx p = q < LOGNT ? xv[f < NX ? f : 0]
                               : x[(j ^ (1 << (q >= LOGNT ? q - LOGNT : 0))) & (R - 1)];
      const double sgi = ((l >> q) & 1) ? fci[f] : -fci[f];
      a = cfma(make_double2(sgi, fcr[f]), p, a);
    }
    acc[j] = a;
  }

  // ---- phase C: partial sums / Horner base, again as one batch of loads ----
  if (A.final_pass) {
    cplx kv[R], bv[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const size_t g = boff + (base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile));
      kv[j] = A.kin ? A.kin[g] : make_double2(0.0, 0.0);
      bv[j] = A.base ? A.base[g] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const cplx r = make_double2(fma(A.scale, acc[j].x + kv[j].x, bv[j].x),
                                  fma(A.scale, acc[j].y + kv[j].y, bv[j].y));
      A.out[boff + (base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile))] = cmul(A.post, r);
    }
  } else {
    cplx kv[R];
#pragma unroll
    for (int j = 0; j < R; ++j)
      kv[j] = A.kin ? A.kin[boff + (base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile))]
                    : make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < R; ++j)
      A.kout[boff + (base_idx | deposit((unsigned long long)(tid + j * NTT), A.tile))] =
          make_double2(acc[j].x + kv[j].x, acc[j].y + kv[j].y);
  }
}

// ---------------------------------------------------------------------------
// k_apply14: 2^14-amplitude tiles held in REGISTERS (16 per thread, 1024
// threads); LDS only exchanges the 10 low tile bits, slice by slice (a slice
// = the 1024 amplitudes with equal register index, closed under flips of bits
// 0-9); flips of tile bits 10-13 are register moves.  Covers the whole state
// of a 14-atom sesolve in one pass and every column-bit flip of a 14-atom
// density-matrix row in one pass.  Tile = index bits [0, 14); flips are the
// bits 0 .. n_flip-1; blockIdx.x = the higher index bits, blockIdx.y = batch.
// ---------------------------------------------------------------------------
struct Apply14Args {
  const cplx* in;
  const cplx* base;   // Horner base (final form) or null
  cplx* out;          // final form: post * (base + scale * acc)
  cplx* kout;         // partial form (no base, no scale) when not null
  const double* coefs;
  const double* e0;
  long long e0_stride;
  double wmix, diag_scale, scale, shift;
  double dec_a, dec_b;  // see PassArgs
  cplx post;
  cplx Sd[4];
  int N, nb, n_flip;
};

template <int MODE, bool REAL, bool FULL>
__global__ __launch_bounds__(1024) void k_apply14(const Apply14Args A) {
  // FULL: all 14 tile bits are flipped (n_flip == 14) - no per-flip predicates.
  constexpr int T = 14, NTT = 1024, R = 16, LOGNT = 10, TL = 7, GS = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* xs = reinterpret_cast<cplx*>(smem);            // GS slices of 1024
  double* tabLo = reinterpret_cast<double*>(xs + GS * NTT);
  double* tabHi = tabLo + (1 << TL);
  double* cft = tabHi + (1 << TL);                     // [T][2]

  const int tid = threadIdx.x;
  const int N = A.N;
  const int nf = FULL ? T : A.n_flip;
  const size_t boff = (size_t)blockIdx.y << A.nb;
  const unsigned long long base_idx = (unsigned long long)blockIdx.x << T;
  const double* __restrict__ cf = A.coefs + (size_t)blockIdx.y * N * 4;
  const double* __restrict__ e0 = A.e0 + (size_t)blockIdx.y * A.e0_stride;
  const cplx* __restrict__ xin = A.in + boff + base_idx;
  const unsigned Dm1 = (MODE == RYD_MESOLVE) ? ((1u << N) - 1u) : 0u;

  cplx x[R];
#pragma unroll
  for (int j = 0; j < R; ++j) x[j] = xin[tid + j * NTT];

  if (tid < T) {
    double cr = 0.0, ci = 0.0;
    if (tid < nf) {
      const int p = tid;  // tile-local bit = global bit
      const int k = (MODE == RYD_SESOLVE || p < N) ? N - 1 - p : 2 * N - 1 - p;
      const double s2 = (MODE == RYD_MESOLVE && p < N) ? 1.0 : -1.0;
      cr = s2 * cf[4 * k];
      ci = cf[4 * k + 1];
    }
    cft[2 * tid] = cr;      // s2 * cr
    cft[2 * tid + 1] = ci;
  }
  // detuning part of the diagonal: two 128-entry tables + the outer bits
  for (int e = tid; e < 2 * (1 << TL); e += NTT) {
    const bool hiHalf = e >= (1 << TL);
    const int v = hiHalf ? e - (1 << TL) : e;
    const int qb = hiHalf ? TL : 0;
    double s = 0.0;
    for (int q = 0; q < TL; ++q) {
      const int p = qb + q;
      if (p >= A.nb) continue;
      double sg;
      int k;
      if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
      else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
      else { k = N - 1 - p; sg = 1.0; }
      if (!((v >> q) & 1)) s += sg * cf[4 * k + 2];
    }
    (hiHalf ? tabHi : tabLo)[v] = s;
  }
  double eOuter = 0.0;
  for (int p = T; p < A.nb; ++p) {
    double sg;
    int k;
    if (MODE == RYD_SESOLVE) { k = N - 1 - p; sg = -1.0; }
    else if (p >= N) { k = 2 * N - 1 - p; sg = -1.0; }
    else { k = N - 1 - p; sg = 1.0; }
    if (!((base_idx >> p) & 1ull)) eOuter += sg * cf[4 * k + 2];
  }
  // mesolve: row-dependent part of the dissipator diagonal is tile-constant
  const double dsw = A.diag_scale * A.wmix;

#pragma unroll
  for (int g0 = 0; g0 < R; g0 += GS) {
    __syncthreads();  // tables ready / previous group's partner reads done
#pragma unroll
    for (int jj = 0; jj < GS; ++jj) xs[jj * NTT + tid] = x[g0 + jj];
    __syncthreads();
    // diagonal of the GS amplitudes of this group
    cplx acc[GS];
#pragma unroll
    for (int jj = 0; jj < GS; ++jj) {
      const int j = g0 + jj;
      const int l = tid + j * NTT;
      const unsigned long long gi = base_idx | (unsigned long long)l;
      const cplx xo = x[j];
      double e = tabLo[l & ((1 << TL) - 1)] + tabHi[l >> TL] + eOuter;
      if (MODE == RYD_SESOLVE) {
        e = A.diag_scale * (e + A.wmix * e0[gi]) - A.shift;
        acc[jj] = make_double2(e * xo.y, -e * xo.x);
        if (A.dec_a != 0.0 || A.dec_b != 0.0) {
          const double dr = fma(A.dec_b, (double)__popcll(gi), A.dec_a);
          acc[jj].x = fma(dr, xo.x, acc[jj].x);
          acc[jj].y = fma(dr, xo.y, acc[jj].y);
        }
      } else {
        const unsigned aa = (unsigned)(gi >> N), bb = (unsigned)gi & Dm1;
        e += A.wmix * (e0[aa] - e0[bb]);
        const int n11 = __popc(aa & bb), n10 = __popc(aa & ~bb & Dm1),
                  n01 = __popc(~aa & bb & Dm1), n00 = N - n11 - n10 - n01;
        const double dr = dsw * (A.Sd[0].x * n00 + A.Sd[1].x * n01 + A.Sd[2].x * n10 + A.Sd[3].x * n11);
        const double di = dsw * (A.Sd[0].y * n00 + A.Sd[1].y * n01 + A.Sd[2].y * n10 + A.Sd[3].y * n11) -
                          A.diag_scale * e;
        acc[jj] = make_double2(dr * xo.x - di * xo.y, dr * xo.y + di * xo.x);
      }
    }
    // flips: one coefficient fetch per flip serves the GS amplitudes
#pragma unroll
    for (int f = 0; f < T; ++f) {
      if (!FULL && f >= nf) continue;  // wave-uniform
      const double fcr = cft[2 * f];   // LDS broadcast read
      const double fci = REAL ? 0.0 : cft[2 * f + 1];
      cplx pv[GS];
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) {
        const int j = g0 + jj;
        if (f < LOGNT) pv[jj] = xs[jj * NTT + (tid ^ (1 << f))];
        else pv[jj] = x[(j ^ (1 << (f >= LOGNT ? f - LOGNT : 0))) & (R - 1)];
      }
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) {
        const int l = tid + (g0 + jj) * NTT;
        if (REAL) {
          acc[jj] = make_double2(fma(-fcr, pv[jj].y, acc[jj].x), fma(fcr, pv[jj].x, acc[jj].y));
        } else {
          const double sgi = ((l >> f) & 1) ? fci : -fci;
          acc[jj] = cfma(make_double2(sgi, fcr), pv[jj], acc[jj]);
        }
      }
    }
    // epilogue of the group
    if (A.kout) {
#pragma unroll
      for (int jj = 0; jj < GS; ++jj)
        A.kout[boff + (base_idx | (unsigned long long)(tid + (g0 + jj) * NTT))] = acc[jj];
    } else {
      cplx bv[GS];
#pragma unroll
      for (int jj = 0; jj < GS; ++jj)
        bv[jj] = A.base ? A.base[boff + (base_idx | (unsigned long long)(tid + (g0 + jj) * NTT))]
                        : make_double2(0.0, 0.0);
#pragma unroll
      for (int jj = 0; jj < GS; ++jj) {
        const cplx r = make_double2(fma(A.scale, acc[jj].x, bv[jj].x), fma(A.scale, acc[jj].y, bv[jj].y));
        A.out[boff + (base_idx | (unsigned long long)(tid + (g0 + jj) * NTT))] = cmul(A.post, r);
      }
    }
  }
}

// out = base + scale * (P + P^dagger) for Hermitian-preserving generators:
// 32 x 32 tile pairs (A <= B); both mirror tiles are written (coalesced, via an
// LDS transpose), only the upper one is read from `base`.
struct SymmArgs {
  const cplx* P;
  const cplx* base;
  cplx* out;
  double scale;
  int N;
};

__global__ __launch_bounds__(256) void k_symm(const SymmArgs A) {
  __shared__ cplx tAB[32][33];
  __shared__ cplx tBA[32][33];
  const int TA = blockIdx.y, TB = blockIdx.x;
  if (TA > TB) return;
  const size_t D = (size_t)1 << A.N;
  const size_t boff = (size_t)blockIdx.z * D * D;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty: 0..7
  const cplx* __restrict__ P = A.P + boff;
  const cplx* __restrict__ base = A.base + boff;
  cplx* __restrict__ out = A.out + boff;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ty + 8 * r;
    tAB[i][tx] = P[((size_t)TA * 32 + i) * D + (size_t)TB * 32 + tx];
    tBA[i][tx] = P[((size_t)TB * 32 + i) * D + (size_t)TA * 32 + tx];
  }
  __syncthreads();
  cplx v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ty + 8 * r;
    const cplx pab = tAB[i][tx], pba = tBA[tx][i];
    const size_t g = ((size_t)TA * 32 + i) * D + (size_t)TB * 32 + tx;
    const cplx b = base[g];
    v[r] = make_double2(fma(A.scale, pab.x + pba.x, b.x), fma(A.scale, pab.y - pba.y, b.y));
    out[g] = v[r];
  }
  if (TA == TB) return;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) tAB[ty + 8 * r][tx] = v[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ty + 8 * r;
    const cplx w = tAB[tx][i];
    out[((size_t)TB * 32 + i) * D + (size_t)TA * 32 + tx] = make_double2(w.x, -w.y);
  }
}

// coefs[b][k] = w1 * val(t1) + w2 * val(t2) for the drive (complex) and the
// detuning (real) of atom k of trajectory b.  pp: [n_series][n_int][4] complex.
__global__ void k_eval_coefs(const cplx* __restrict__ pp, int n_int,
                             const ryd_qdesc* __restrict__ desc, int total,
                             int idx1, double u1, double w1, int idx2, double u2,
                             double w2, double* __restrict__ coefs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const ryd_qdesc d = desc[i];
  auto val = [&](int s, int idx, double u) -> cplx {
    const cplx* p = pp + ((size_t)s * n_int + idx) * 4;
    cplx r = p[0];
    r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
    r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
    r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
    return r;
  };
  double cr = 0, ci = 0, dl = 0;
  if (d.drive_series >= 0) {
    const cplx a = val(d.drive_series, idx1, u1), b2 = val(d.drive_series, idx2, u2);
    cr = d.drive_scale * (w1 * a.x + w2 * b2.x);
    ci = d.drive_scale * (w1 * a.y + w2 * b2.y);
  }
  if (d.det_series >= 0)
    dl += d.det_scale * (w1 * val(d.det_series, idx1, u1).x + w2 * val(d.det_series, idx2, u2).x);
  if (d.off_series >= 0)
    dl += d.off_scale * (w1 * val(d.off_series, idx1, u1).x + w2 * val(d.off_series, idx2, u2).x);
  coefs[4 * (size_t)i + 0] = cr;
  coefs[4 * (size_t)i + 1] = ci;
  coefs[4 * (size_t)i + 2] = dl;
  coefs[4 * (size_t)i + 3] = 0.0;
}

// E0[m][s] = sum_{i<j} U[m][i][j] n_i(s) n_j(s), n_k(s) = 1 - bit_{N-1-k}(s)
// (hamiltonian.py:260-274, 308-331; coefficient U/2 doubled by H + H^dagger).
__global__ void k_build_e0(const double* __restrict__ U, int N, double* __restrict__ e0) {
  const size_t D = (size_t)1 << N;
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= D) return;
  const int m = blockIdx.y;
  const double* u = U + (size_t)m * N * N;
  double e = 0.0;
  for (int i = 0; i < N; ++i) {
    if ((s >> (N - 1 - i)) & 1) continue;
    for (int j = i + 1; j < N; ++j)
      if (!((s >> (N - 1 - j)) & 1)) e += u[i * N + j];
  }
  e0[(size_t)m * D + s] = e;
}

// w[b][i'] = |psi_i|^2 (ket) or Re rho_ii (dm); i' = D-1-i when reverse.
__global__ void k_probabilities(const cplx* __restrict__ st, int N, int is_dm,
                                int reverse, double* __restrict__ w) {
  const size_t D = (size_t)1 << N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D) return;
  const int b = blockIdx.y;
  double p;
  if (is_dm) {
    p = st[((size_t)b << (2 * N)) + i * D + i].x;
  } else {
    const cplx v = st[((size_t)b << N) + i];
    p = v.x * v.x + v.y * v.y;
  }
  w[(size_t)b * D + (reverse ? D - 1 - i : i)] = p;
}

// out[b][k] += sum_i p_i n_k(i) (k < N), out[b][N] += sum_i p_i.
__global__ __launch_bounds__(256) void k_occupations(const cplx* __restrict__ st,
                                                     int N, int is_dm,
                                                     double* __restrict__ out) {
  const size_t D = (size_t)1 << N;
  const int b = blockIdx.y;
  double acc[RYD_MAX_QUBITS + 1];
#pragma unroll
  for (int k = 0; k <= RYD_MAX_QUBITS; ++k) acc[k] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < D;
       i += (size_t)gridDim.x * blockDim.x) {
    double p;
    if (is_dm) {
      p = st[((size_t)b << (2 * N)) + i * D + i].x;
    } else {
      const cplx v = st[((size_t)b << N) + i];
      p = v.x * v.x + v.y * v.y;
    }
#pragma unroll
    for (int k = 0; k < RYD_MAX_QUBITS; ++k)
      if (k < N && !((i >> (N - 1 - k)) & 1)) acc[k] += p;
    acc[RYD_MAX_QUBITS] += p;
  }
  __shared__ double red[4][RYD_MAX_QUBITS + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= RYD_MAX_QUBITS; ++k) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= N) {
    const int k = threadIdx.x == N ? RYD_MAX_QUBITS : threadIdx.x;
    const double v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    atomicAdd(&out[(size_t)b * (N + 1) + threadIdx.x], v);
  }
}

__global__ void k_ket_to_dm(const cplx* __restrict__ psi, int N, cplx* __restrict__ rho) {
  const size_t D = (size_t)1 << N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // a*D + b
  if (i >= D * D) return;
  const int bt = blockIdx.y;
  const cplx pa = psi[((size_t)bt << N) + (i >> N)];
  const cplx pb = psi[((size_t)bt << N) + (i & (D - 1))];
  rho[((size_t)bt << (2 * N)) + i] =
      make_double2(pa.x * pb.x + pa.y * pb.y, pa.y * pb.x - pa.x * pb.y);
}

__global__ void k_outer_acc(const cplx* __restrict__ psi, int N, int B,
                            const double* __restrict__ wts, cplx* __restrict__ acc) {
  const size_t D = (size_t)1 << N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D * D) return;
  const size_t a = i >> N, b = i & (D - 1);
  double sr = 0, si = 0;
  for (int t = 0; t < B; ++t) {
    const cplx pa = psi[((size_t)t << N) + a], pb = psi[((size_t)t << N) + b];
    const double w = wts ? wts[t] : 1.0;
    sr += w * (pa.x * pb.x + pa.y * pb.y);
    si += w * (pa.y * pb.x - pa.x * pb.y);
  }
  acc[i].x += sr;
  acc[i].y += si;
}

// ---------------------------------------------------------------------------
// Monte-Carlo wavefunction (quantum-jump) kernels - the work qutip.mcsolve does
// between and at the collapses (pulser-simulation/pulser_simulation/
// simulation.py:705-735: solver_fn = qutip.mcsolve, c_ops = one local operator
// per (spec, atom), hamiltonian.py:97-124).  The unnormalised ket evolves under
// H_eff = H - (i/2) sum C^dag C; when its squared norm has dropped below a
// uniform threshold, a collapse operator is drawn with weights ||C psi||^2,
// applied, and the ket renormalised.  Everything runs on the device without
// host synchronisation: one norm reduction per step, and - only for the
// trajectories that jump - the single-atom reduced density matrices, the
// selection and the 2x2 local update.  Random numbers: Philox4x32-10 keyed by a
// per-trajectory 64-bit seed, counter = jump index, so a trajectory's history
// does not depend on how the batch is split over launches or GPUs.
// ---------------------------------------------------------------------------
#define MC_MAX_OPS 16

__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// (threshold uniform, selection uniform) of jump number j: 53-bit doubles in [0, 1)
__device__ __forceinline__ void mc_uniforms(unsigned long long seed, unsigned j, double* ut,
                                            double* us) {
  unsigned c[4] = {j, 0u, 0u, 0u};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  *ut = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
  *us = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6)) * (1.0 / 9007199254740992.0);
}

struct McState {
  double* norm2;     // [2][B] squared norms (slot = step parity), zero between uses
  double* red;       // [B][N][4]: rho_rr, rho_gg, Re rho_rg, Im rho_rg of each atom
  double* target;    // [B] current threshold uniform
  double* refnorm;   // [B] squared norm right after the last jump (or at the start)
  double* lastnorm;  // [B] squared norm after the last completed step
  double* scale;     // [B] 1 / ||C psi|| of the selected collapse
  int* flag;         // [B] this step jumps
  int* sel;          // [B] atom * MC_MAX_OPS + op
  int* count;        // [B] jumps so far
  const unsigned long long* seeds;  // [B]
  const cplx* ops;   // [n_ops][4] local collapse operators, row-major (index 0 = r)
  int n_ops;
};

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_mc_norm(const cplx* __restrict__ st, int nb,
                                                 double* __restrict__ norm2) {
  __shared__ double sh[4];
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const cplx* __restrict__ x = st + ((size_t)b << nb);
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = x[i];
    s = fma(v.x, v.x, fma(v.y, v.y, s));
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) atomicAdd(&norm2[b], s);
}

// start of a Monte-Carlo solve: thresholds of jump 0, reference norms
__global__ void k_mc_init(McState M, int B, int N) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double ut, us;
  mc_uniforms(M.seeds[b], 0u, &ut, &us);
  M.target[b] = ut;
  M.refnorm[b] = M.norm2[b];  // slot 0 holds the initial squared norm
  M.lastnorm[b] = M.norm2[b];
  M.norm2[b] = 0.0;
  M.norm2[B + b] = 0.0;
  M.count[b] = 0;
  M.flag[b] = 0;
  for (int i = 0; i < 4 * N; ++i) M.red[(size_t)b * 4 * N + i] = 0.0;
}

// reduced single-atom density matrices of the trajectories that jump this step
__global__ __launch_bounds__(256) void k_mc_reduced(const cplx* __restrict__ st, int N,
                                                    McState M, int B, int slot) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  if (!(M.norm2[(size_t)slot * B + b] <= M.target[b] * M.refnorm[b])) return;  // block-uniform
  const size_t D = (size_t)1 << N;
  const cplx* __restrict__ x = st + ((size_t)b << N);
  for (int a = 0; a < N; ++a) {
    const size_t bit = (size_t)1 << (N - 1 - a);
    double rr = 0.0, gg = 0.0, cr = 0.0, ci = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
      const cplx v = x[i];
      const double m = v.x * v.x + v.y * v.y;
      if (i & bit) {
        gg += m;
      } else {
        const cplx w = x[i | bit];
        rr += m;
        cr += v.x * w.x + v.y * w.y;  // psi_r conj(psi_g)
        ci += v.y * w.x - v.x * w.y;
      }
    }
    rr = block_sum256(rr, sh);
    gg = block_sum256(gg, sh);
    cr = block_sum256(cr, sh);
    ci = block_sum256(ci, sh);
    if (threadIdx.x == 0) {
      double* r = M.red + ((size_t)b * N + a) * 4;
      atomicAdd(r + 0, rr);
      atomicAdd(r + 1, gg);
      atomicAdd(r + 2, cr);
      atomicAdd(r + 3, ci);
    }
  }
}

// ||C psi||^2 = Tr(C rho_atom C^dag) for a local 2x2 operator
__device__ __forceinline__ double mc_weight(const cplx* C, const double* r) {
  double p = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const cplx c0 = C[2 * i], c1 = C[2 * i + 1];
    // c0 conj(c1) rho_rg
    const double xr = c0.x * c1.x + c0.y * c1.y, xi = c0.y * c1.x - c0.x * c1.y;
    p += (c0.x * c0.x + c0.y * c0.y) * r[0] + (c1.x * c1.x + c1.y * c1.y) * r[1] +
         2.0 * (xr * r[2] - xi * r[3]);
  }
  return p;
}

__global__ void k_mc_select(McState M, int B, int N, int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double n2 = M.norm2[(size_t)slot * B + b];
  M.norm2[(size_t)(slot ^ 1) * B + b] = 0.0;  // the next step accumulates there
  M.norm2[(size_t)slot * B + b] = 0.0;
  int flag = 0;
  double last = n2;
  if (n2 <= M.target[b] * M.refnorm[b]) {
    double* red = M.red + (size_t)b * N * 4;
    double total = 0.0;
    for (int a = 0; a < N; ++a)
      for (int k = 0; k < M.n_ops; ++k) total += fmax(mc_weight(M.ops + 4 * k, red + 4 * a), 0.0);
    if (total > 0.0) {
      const unsigned j = (unsigned)M.count[b];
      double ut, us;
      mc_uniforms(M.seeds[b], j, &ut, &us);
      const double x = us * total;
      double cum = 0.0, psel = 0.0, plast = 0.0;
      int sel = -1, lastpos = -1;
      for (int a = 0; a < N; ++a)
        for (int k = 0; k < M.n_ops; ++k) {
          const double p = fmax(mc_weight(M.ops + 4 * k, red + 4 * a), 0.0);
          cum += p;
          if (p > 0.0) { lastpos = a * MC_MAX_OPS + k; plast = p; }
          if (sel < 0 && p > 0.0 && cum > x) { sel = a * MC_MAX_OPS + k; psel = p; }
        }
      if (sel < 0) { sel = lastpos; psel = plast; }  // rounding left x >= cum
      M.sel[b] = sel;
      M.scale[b] = 1.0 / sqrt(psel);
      M.count[b] = (int)j + 1;
      mc_uniforms(M.seeds[b], j + 1u, &ut, &us);
      M.target[b] = ut;
      M.refnorm[b] = 1.0;
      last = 1.0;
      flag = 1;
    }
    for (int i = 0; i < 4 * N; ++i) red[i] = 0.0;
  }
  M.flag[b] = flag;
  M.lastnorm[b] = last;
}

// psi <- C_k^(atom) psi / ||C psi|| for the flagged trajectories (in place, by pairs)
__global__ __launch_bounds__(256) void k_mc_jump(cplx* __restrict__ st, int N, McState M) {
  const int b = blockIdx.y;
  if (!M.flag[b]) return;
  const int sel = M.sel[b];
  const int p = N - 1 - sel / MC_MAX_OPS;
  const cplx* C = M.ops + 4 * (sel % MC_MAX_OPS);
  const cplx c00 = C[0], c01 = C[1], c10 = C[2], c11 = C[3];
  const double s = M.scale[b];
  const size_t half = (size_t)1 << (N - 1), bit = (size_t)1 << p;
  cplx* __restrict__ x = st + ((size_t)b << N);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (size_t)gridDim.x * 256) {
    const size_t l0 = ((i >> p) << (p + 1)) | (i & (bit - 1)), l1 = l0 | bit;
    const cplx v0 = x[l0], v1 = x[l1];
    const cplx o0 = cfma(c00, v0, cmul(c01, v1)), o1 = cfma(c10, v0, cmul(c11, v1));
    x[l0] = make_double2(s * o0.x, s * o0.y);
    x[l1] = make_double2(s * o1.x, s * o1.y);
  }
}

// dst = src / ||src|| using the norm recorded after the last step (dst may be src)
__global__ __launch_bounds__(256) void k_mc_normalize(const cplx* __restrict__ src,
                                                      cplx* __restrict__ dst, int nb,
                                                      const double* __restrict__ lastnorm) {
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const double s = rsqrt(lastnorm[b]);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = src[((size_t)b << nb) + i];
    dst[((size_t)b << nb) + i] = make_double2(s * v.x, s * v.y);
  }
}

// ---------------------------------------------------------------------------
// Persistent trajectory kernel (sesolve, N <= 12)
// ---------------------------------------------------------------------------
// One workgroup evolves one state vector through a whole schedule of CF4 steps
// in a single launch: psi lives in registers (thread t owns amplitudes
// t + j*NT), the Horner iterate lives in LDS for the flip-partner reads, and
// HBM is touched only for the initial load, the snapshots and the final store.
// Flip partners of the high index bits (>= log2 NT) are register-to-register.
// 1/j for the Horner scale h/j (orders are capped at 32)
__constant__ double kInvInt[33] = {
    0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10,
    1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19,
    1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28,
    1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32};

struct StepDesc {
  double h, u1, u2;
  double shift_a, shift_b;
  int idx;
  int order_a, order_b;
  int snap;  // snapshot slot written after this step, or -1
  int pad;
};

struct TrajArgs {
  cplx* state;         // [B][2^N] in/out
  cplx* snaps;         // [n_slots][B][2^N] or null
  const cplx* pp;      // [n_series][n_int][4]
  const ryd_qdesc* desc;
  const double* e0;
  long long e0_stride;
  const StepDesc* steps;
  int n_int, n_steps, B;
  double a1, a2;
  // Monte-Carlo wavefunction instantiations (MC = true)
  McState mc;
  double mc_a, mc_b;  // real diagonal of G_eff: mc_a + mc_b * popc(index)
  int mc_jumps;       // 0: no-jump evolution under H_eff only
};

// MODEL 0: per-atom complex drive coefficients (local addressing, noise).
// MODEL 1: one real drive coefficient shared by the driven atoms of the
//          trajectory (global channel with constant zero phase; bad atoms are
//          masked out) - the flip partners are summed first, 2 DADD each.
// MC: the generator is G_eff (adds the real decay diagonal); with A.mc_jumps the
//     norm threshold is tested after every step and collapses are applied in
//     place (same arithmetic and random stream as the k_mc_* kernels).
template <int N, int NTT, int MODEL, bool MC>
__global__ __launch_bounds__(NTT) void k_traj(const TrajArgs A) {
  constexpr int D = 1 << N;
  constexpr int R = D / NTT > 0 ? D / NTT : 1;
  constexpr int LOGNT = NTT == 1024 ? 10 : (NTT == 512 ? 9 : (NTT == 256 ? 8 : (NTT == 128 ? 7 : 6)));
  constexpr int NLDS = N < LOGNT ? N : LOGNT;  // bits whose partner is read from LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* ws0 = reinterpret_cast<cplx*>(smem);      // two buffers: the Horner iterate
  cplx* ws1 = ws0 + D;                            // ping-pongs, one barrier per stage
  double* cfA = reinterpret_cast<double*>(ws1 + D);  // [16][4]: cr, ci, delta, 0 for exp A
  double* cfB = cfA + 64;                            // same for exp B
  double* mcred = cfB + 64;                          // [16][4] per-wave partial sums
  double* mcrho = mcred + 64;                        // [16][4] reduced density matrices

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const bool active = tid < D;
  cplx* st = A.state + (size_t)b * D;
  const double* e0g = A.e0 + (size_t)b * A.e0_stride;

  cplx psi[R];
  double e0r[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int l = tid + j * NTT;
    psi[j] = active ? st[l] : make_double2(0.0, 0.0);
    e0r[j] = active ? e0g[l] : 0.0;
  }
  // Monte-Carlo bookkeeping (block-uniform)
  const bool jumps = MC && A.mc_jumps != 0;
  double mc_target = 0.0, mc_ref = 1.0, mc_n2 = 1.0;
  int mc_count = 0;
  unsigned long long mc_seed = 0;
  if (jumps) {
    mc_target = A.mc.target[b];
    mc_ref = A.mc.refnorm[b];
    mc_n2 = A.mc.lastnorm[b];
    mc_count = A.mc.count[b];
    mc_seed = A.mc.seeds[b];
  }

  for (int s = 0; s < A.n_steps; ++s) {
    const StepDesc sd = A.steps[s];
    if (tid < N) {
      const ryd_qdesc d = A.desc[(size_t)b * N + tid];
      auto val = [&](int ser, double u) -> cplx {
        const cplx* p = A.pp + ((size_t)ser * A.n_int + sd.idx) * 4;
        cplx r = p[0];
        r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
        r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
        r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
        return r;
      };
      // same arithmetic as k_eval_coefs (w1 * val(t1) + w2 * val(t2))
      double c1r = 0, c1i = 0, c2r = 0, c2i = 0, dlA = 0, dlB = 0;
      if (d.drive_series >= 0) {
        const cplx v1 = val(d.drive_series, sd.u1), v2 = val(d.drive_series, sd.u2);
        c1r = v1.x; c1i = v1.y; c2r = v2.x; c2i = v2.y;
      }
      if (d.det_series >= 0) {
        const double d1 = val(d.det_series, sd.u1).x, d2 = val(d.det_series, sd.u2).x;
        dlA += d.det_scale * (A.a1 * d1 + A.a2 * d2);
        dlB += d.det_scale * (A.a2 * d1 + A.a1 * d2);
      }
      if (d.off_series >= 0) {
        const double o1 = val(d.off_series, sd.u1).x, o2 = val(d.off_series, sd.u2).x;
        dlA += d.off_scale * (A.a1 * o1 + A.a2 * o2);
        dlB += d.off_scale * (A.a2 * o1 + A.a1 * o2);
      }
      cfA[4 * tid + 0] = d.drive_scale * (A.a1 * c1r + A.a2 * c2r);
      cfA[4 * tid + 1] = d.drive_scale * (A.a1 * c1i + A.a2 * c2i);
      cfA[4 * tid + 2] = dlA;
      cfA[4 * tid + 3] = d.drive_series >= 0 ? 1.0 : 0.0;
      cfB[4 * tid + 0] = d.drive_scale * (A.a2 * c1r + A.a1 * c2r);
      cfB[4 * tid + 1] = d.drive_scale * (A.a2 * c1i + A.a1 * c2i);
      cfB[4 * tid + 2] = dlB;
      cfB[4 * tid + 3] = d.drive_series >= 0 ? 1.0 : 0.0;
    }
    __syncthreads();

#pragma unroll 1
    for (int ex = 0; ex < 2; ++ex) {
      const double* cf = ex ? cfB : cfA;
      const int order = ex ? sd.order_b : sd.order_a;
      const double shift = ex ? sd.shift_b : sd.shift_a;
      const double wmix = A.a1 + A.a2;
      // all per-atom values of this exponential in one batch of LDS reads
      // (bit q <-> atom N-1-q), then wave-uniform ones -> scalar registers
      double craw[N], ciraw[N], draw[N], mraw[N];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const double2 a = *reinterpret_cast<const double2*>(cf + 4 * (N - 1 - q));
        const double2 c = *reinterpret_cast<const double2*>(cf + 4 * (N - 1 - q) + 2);
        craw[q] = a.x; ciraw[q] = a.y; draw[q] = c.x; mraw[q] = c.y;
      }
      double cr[MODEL == 0 ? N : 1], ci[MODEL == 0 ? N : 1];
      double mq[MODEL == 1 ? N : 1];  // 1.0 for driven atoms, 0.0 otherwise
      double cuni = 0.0;
      if (MODEL == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
          cr[q] = uniform_d(craw[q]);
          ci[q] = uniform_d(ciraw[q]);
        }
      } else {
        double cv = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q) cv = mraw[q] != 0.0 ? craw[q] : cv;
        cuni = uniform_d(cv);
#pragma unroll
        for (int q = 0; q < N; ++q) mq[q] = uniform_d(mraw[q]);
      }
      double eg[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int l = tid + j * NTT;
        double sdet = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q)
          if (!((l >> q) & 1)) sdet -= draw[q];
        eg[j] = sdet + (wmix * e0r[j] - shift);
      }
      double er[MC ? R : 1];  // centred real part of the G_eff diagonal
      if (MC) {
#pragma unroll
        for (int j = 0; j < R; ++j)
          er[j] = wmix * A.mc_b * ((double)__popc(tid + j * NTT) - 0.5 * N);
      }
      cplx w[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        w[j] = psi[j];
        if (active) ws0[tid + j * NTT] = w[j];
      }
      __syncthreads();
      const cplx* rd = ws0;
      cplx* wr = ws1;
      for (int jj = order; jj >= 1; --jj) {
        const double sc = sd.h * kInvInt[jj];  // a v_div_f64 costs ~15 VALU issue slots per stage
        cplx acc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int l = tid + j * NTT;
          // issue every LDS partner read of this element before using any
          cplx xv[NLDS > 0 ? NLDS : 1];
#pragma unroll
          for (int q = 0; q < NLDS; ++q) xv[q] = rd[(l ^ (1 << q)) & (D - 1)];
          if (MODEL == 0) {
            cplx a = make_double2(eg[j] * w[j].y, -eg[j] * w[j].x);  // -i e x
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              // coefficient -i c (output bit 1) or -i conj(c) (output bit 0)
              const double sgi = ((l >> q) & 1) ? ci[q] : -ci[q];
              a = cfma(make_double2(sgi, -cr[q]), x, a);
            }
            acc[j] = a;
          } else {
            double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0;  // two chains
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              if (q & 1) { s1x = fma(mq[q], x.x, s1x); s1y = fma(mq[q], x.y, s1y); }
              else { s0x = fma(mq[q], x.x, s0x); s0y = fma(mq[q], x.y, s0y); }
            }
            const double sx = s0x + s1x, sy = s0y + s1y;
            // -i (e w + c sum)
            acc[j] = make_double2(fma(cuni, sy, eg[j] * w[j].y), -fma(cuni, sx, eg[j] * w[j].x));
          }
        }
        if (MC) {
#pragma unroll
          for (int j = 0; j < R; ++j)
            acc[j] = make_double2(fma(er[j], w[j].x, acc[j].x), fma(er[j], w[j].y, acc[j].y));
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
          w[j] = make_double2(fma(sc, acc[j].x, psi[j].x), fma(sc, acc[j].y, psi[j].y));
        if (jj > 1) {
#pragma unroll
          for (int j = 0; j < R; ++j)
            if (active) wr[tid + j * NTT] = w[j];
          __syncthreads();  // one barrier per stage: reads of `rd` done, `wr` visible
          const cplx* t = rd;
          rd = wr;
          wr = const_cast<cplx*>(t);
        }
      }
      const double mag = MC ? exp(sd.h * wmix * (A.mc_a + 0.5 * N * A.mc_b)) : 1.0;
      const cplx post = make_double2(mag * cos(sd.h * shift), -mag * sin(sd.h * shift));
#pragma unroll
      for (int j = 0; j < R; ++j) psi[j] = cmul(post, w[j]);
      __syncthreads();  // last-stage reads done before ws0 / cf are rewritten
    }
    if (jumps) {
      constexpr int NW = NTT / 64;
      const int lane = tid & 63, wave = tid >> 6;
      double s2 = 0.0;
#pragma unroll
      for (int j = 0; j < R; ++j) s2 = fma(psi[j].x, psi[j].x, fma(psi[j].y, psi[j].y, s2));
      if (!active) s2 = 0.0;  // lanes beyond a small state hold garbage
      for (int o = 32; o > 0; o >>= 1) s2 += __shfl_down(s2, o, 64);
      if (lane == 0) mcred[wave] = s2;
      __syncthreads();
      double n2 = 0.0;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) n2 += mcred[wv];
      mc_n2 = n2;
      __syncthreads();
      if (n2 <= mc_target * mc_ref) {  // block-uniform: this trajectory jumps now
#pragma unroll
        for (int j = 0; j < R; ++j)
          if (active) ws0[tid + j * NTT] = psi[j];
        __syncthreads();
        for (int a = 0; a < N; ++a) {
          const int bit = 1 << (N - 1 - a);
          double rr = 0.0, gg = 0.0, cr = 0.0, ci = 0.0;
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int l = tid + j * NTT;
            const cplx v = psi[j];
            const double m = v.x * v.x + v.y * v.y;
            if (l & bit) {
              gg += m;
            } else {
              const cplx wv = ws0[(l | bit) & (D - 1)];
              rr += m;
              cr += v.x * wv.x + v.y * wv.y;
              ci += v.y * wv.x - v.x * wv.y;
            }
          }
          if (!active) rr = gg = cr = ci = 0.0;
          for (int o = 32; o > 0; o >>= 1) {
            rr += __shfl_down(rr, o, 64);
            gg += __shfl_down(gg, o, 64);
            cr += __shfl_down(cr, o, 64);
            ci += __shfl_down(ci, o, 64);
          }
          if (lane == 0) {
            mcred[4 * wave + 0] = rr;
            mcred[4 * wave + 1] = gg;
            mcred[4 * wave + 2] = cr;
            mcred[4 * wave + 3] = ci;
          }
          __syncthreads();
          if (tid < 4) {
            double t = 0.0;
            for (int wv = 0; wv < NW; ++wv) t += mcred[4 * wv + tid];
            mcrho[4 * a + tid] = t;
          }
          __syncthreads();
        }
        // selection: every thread repeats the (uniform) arithmetic of k_mc_select
        double total = 0.0;
        for (int a = 0; a < N; ++a)
          for (int k = 0; k < A.mc.n_ops; ++k)
            total += fmax(mc_weight(A.mc.ops + 4 * k, mcrho + 4 * a), 0.0);
        if (total > 0.0) {
          double ut, us;
          mc_uniforms(mc_seed, (unsigned)mc_count, &ut, &us);
          const double x = us * total;
          double cum = 0.0, psel = 0.0, plast = 0.0;
          int sel = -1, lastpos = -1;
          for (int a = 0; a < N; ++a)
            for (int k = 0; k < A.mc.n_ops; ++k) {
              const double p = fmax(mc_weight(A.mc.ops + 4 * k, mcrho + 4 * a), 0.0);
              cum += p;
              if (p > 0.0) { lastpos = a * MC_MAX_OPS + k; plast = p; }
              if (sel < 0 && p > 0.0 && cum > x) { sel = a * MC_MAX_OPS + k; psel = p; }
            }
          if (sel < 0) { sel = lastpos; psel = plast; }
          const int pbit = N - 1 - sel / MC_MAX_OPS;
          const cplx* C = A.mc.ops + 4 * (sel % MC_MAX_OPS);
          const double sc2 = 1.0 / sqrt(psel);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int l = tid + j * NTT;
            const int row = (l >> pbit) & 1;
            const cplx v0 = ws0[(l & ~(1 << pbit)) & (D - 1)], v1 = ws0[(l | (1 << pbit)) & (D - 1)];
            const cplx o = cfma(C[2 * row], v0, cmul(C[2 * row + 1], v1));
            if (active) psi[j] = make_double2(sc2 * o.x, sc2 * o.y);
          }
          ++mc_count;
          mc_uniforms(mc_seed, (unsigned)mc_count, &ut, &us);
          mc_target = ut;
          mc_ref = 1.0;
          mc_n2 = 1.0;
        }
        __syncthreads();  // partner reads of ws0 done before the next step rewrites it
      }
    }
    if (sd.snap >= 0 && A.snaps && active) {
      cplx* o = A.snaps + ((size_t)sd.snap * A.B + b) * D;
      const double ns = jumps ? rsqrt(mc_n2) : 1.0;  // stored kets are normalised
#pragma unroll
      for (int j = 0; j < R; ++j) o[tid + j * NTT] = make_double2(ns * psi[j].x, ns * psi[j].y);
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < R; ++j) st[tid + j * NTT] = psi[j];
  }
  if (jumps && tid == 0) {
    A.mc.target[b] = mc_target;
    A.mc.refnorm[b] = mc_ref;
    A.mc.lastnorm[b] = mc_n2;
    A.mc.count[b] = mc_count;
  }
}

// ---------------------------------------------------------------------------
// General path: G(t) = sum_t coef_t(t) A_t with explicit CSR terms (any local
// dimension; small systems).  One thread per (row, batch entry).
// ---------------------------------------------------------------------------
struct GenTermDev {
  const int* row_ptr;
  const int* col;
  const cplx* val;
};

#define MAX_GEN_TERMS 96

struct GenArgs {
  const cplx* in;
  const cplx* base;
  cplx* out;
  const cplx* tcoef;  // [n_terms] time-mixed coefficients
  const GenTermDev* terms;
  long long dim;
  int n_terms;
  double scale;
};

__global__ void k_gen_coefs(const cplx* __restrict__ pp, int n_int, const int* __restrict__ series,
                            const int* __restrict__ conjf, const cplx* __restrict__ scale,
                            int n_terms, int idx, double u1, double w1, double u2, double w2,
                            cplx* __restrict__ tcoef) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  cplx v = make_double2(w1 + w2, 0.0);
  if (series[t] >= 0) {
    auto val = [&](double u) -> cplx {
      const cplx* p = pp + ((size_t)series[t] * n_int + idx) * 4;
      cplx r = p[0];
      r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
      r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
      r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
      return r;
    };
    const cplx a = val(u1), b = val(u2);
    v = make_double2(w1 * a.x + w2 * b.x, w1 * a.y + w2 * b.y);
    if (conjf[t]) v.y = -v.y;
  }
  tcoef[t] = cmul(scale[t], v);
}

__global__ __launch_bounds__(256) void k_gen_apply(const GenArgs A) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= A.dim) return;
  const size_t boff = (size_t)blockIdx.y * A.dim;
  const cplx* __restrict__ x = A.in + boff;
  cplx acc = make_double2(0.0, 0.0);
  for (int t = 0; t < A.n_terms; ++t) {
    const GenTermDev T = A.terms[t];
    const int lo = T.row_ptr[row], hi = T.row_ptr[row + 1];
    cplx s = make_double2(0.0, 0.0);
    for (int e = lo; e < hi; ++e) s = cfma(T.val[e], x[T.col[e]], s);
    acc = cfma(A.tcoef[t], s, acc);
  }
  cplx r = make_double2(A.scale * acc.x, A.scale * acc.y);
  if (A.base) {
    const cplx b = A.base[boff + row];
    r.x += b.x;
    r.y += b.y;
  }
  A.out[boff + row] = r;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct Pass {
  Segs tile, outer;
  int T = 0;
  int n_outer_bits = 0;
  std::vector<int> flip_q;
  std::vector<std::pair<int, int>> dbl;  // (qb, qa)
  bool include_diag = false;
  bool use14 = false;  // pass 0 on 2^14 register tiles (k_apply14)
};

struct GenTermHost {
  GenTermDev dev{nullptr, nullptr, nullptr};
  int series = -1, conj = 0;
  std::complex<double> scale{1.0, 0.0};
  double row_norm = 0.0;
};

struct ryd_handle {
  ryd_config cfg{};
  int N = 0, nb = 0, B = 1, T = 12;
  size_t dim = 0;  // elements per state (2^nb)
  // tables
  int n_series = 0, n_knots = 0;
  std::vector<double> tknots;
  std::vector<std::complex<double>> pp_host;  // [series][int][4]
  std::vector<double> s_abs, s_pos, s_neg;    // per series, per interval bounds
  std::vector<double> s_curv;                 // |quadratic| dt^2 + |cubic| dt^3 per interval
  cplx* pp_dev = nullptr;
  std::vector<ryd_qdesc> desc_host;
  ryd_qdesc* desc_dev = nullptr;
  std::vector<double> bd_drive, bd_pos, bd_neg;  // per interval, max over batch
  std::vector<double> bd_curv;                   // per interval: non-linearity of H(t)
  bool bounds_valid = false;
  double* e0_dev = nullptr;
  int e0_mats = 0;
  double e0_min = 0, e0_max = 0;
  double* coefs_dev = nullptr;
  cplx Sd[4]{}, J[4]{};
  double diss_norm = 0.0;
  bool has_dbl = false;
  // work vectors
  cplx *wA = nullptr, *wB = nullptr, *kbuf = nullptr;
  std::vector<Pass> passes;
  bool passes_valid = false;
  StepDesc* sched_dev = nullptr;
  size_t sched_cap = 0;
  // general path (explicit CSR terms)
  bool general = false;
  std::vector<GenTermHost> gen_host;
  cplx* gen_tcoef = nullptr;
  GenTermDev* gen_terms_dev = nullptr;
  int* gen_series_dev = nullptr;
  int* gen_conj_dev = nullptr;
  cplx* gen_scale_dev = nullptr;
  bool auto_tile = true;       // tile_bits == 0: tile size chosen by the library
  bool force_generic = false;
  bool no_fast_apply = false;  // test hook: use the generic k_apply for T = 12 too
  bool no_tile14 = false;      // test hook: disable k_apply14 / the Hermitian mesolve path
  bool force_tile14 = false;   // test hook: use them even when too few tiles fill the GPU
  bool drive_real = false;     // every drive series is real-valued
  bool uniform_real_drive = false;  // persistent-kernel MODEL 1 applies
  // Monte-Carlo wavefunction mode (sesolve handles with ryd_set_collapse)
  bool mc = false;         // collapse operators set: H_eff carries -(i/2) sum C^dag C
  bool mc_active = false;  // inside ryd_mc_solve: jump bookkeeping after every step
  int mc_n_ops = 0;
  double mc_a = 0.0, mc_b = 0.0;  // real diagonal of G_eff: mc_a + mc_b * popc(index)
  void* mc_pool = nullptr;        // one allocation behind McState
  McState mcs{};
  unsigned long long* mc_seeds_dev = nullptr;
  cplx* mc_ops_dev = nullptr;
  int mc_slot = 0;
  ryd_stats stats{};
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
  double timing_ms = 0;
  int64_t timing_launches = 0;
};

static bool hermitian_path(const ryd_handle* h);

static Segs make_segs(std::vector<std::pair<int, int>> v) {
  Segs s;
  for (int i = 0; i < 3; ++i) {
    s.lo[i] = i < (int)v.size() ? v[i].first : 0;
    s.len[i] = i < (int)v.size() ? v[i].second : 0;
  }
  return s;
}

// Build a pass whose tile consists of the bit ranges in `tile` (ascending,
// disjoint); the outer segments are the complement within [0, nb).
static Pass make_pass(int nb, std::vector<std::pair<int, int>> tile) {
  Pass p;
  std::vector<std::pair<int, int>> t2, outer;
  for (auto& s : tile)
    if (s.second > 0) t2.push_back(s);
  // merge adjacent ranges
  std::vector<std::pair<int, int>> merged;
  for (auto& s : t2) {
    if (!merged.empty() && merged.back().first + merged.back().second == s.first)
      merged.back().second += s.second;
    else
      merged.push_back(s);
  }
  int pos = 0;
  for (auto& s : merged) {
    if (s.first > pos) outer.push_back({pos, s.first - pos});
    pos = s.first + s.second;
    p.T += s.second;
  }
  if (pos < nb) outer.push_back({pos, nb - pos});
  // at most 3 outer segments by construction (<= 3 tile segments, first at 0)
  p.tile = make_segs(merged);
  p.outer = make_segs(outer);
  p.n_outer_bits = nb - p.T;
  return p;
}

// tile-local index of global bit p (or -1)
static int local_of(const Segs& s, int p) {
  int off = 0;
  for (int i = 0; i < 3; ++i) {
    if (p >= s.lo[i] && p < s.lo[i] + s.len[i]) return off + p - s.lo[i];
    off += s.len[i];
  }
  return -1;
}

// A 2^14 register tile keeps one whole CU busy per 16384 amplitudes; it only
// pays when the launch has enough tiles for the 256 CUs (measured: 64 tiles of a
// 20-atom ket are 1.4x slower than 256 LDS tiles of 2^12).
static bool tile14_pays(const ryd_handle* h) {
  if (h->no_tile14 || h->nb < 14) return false;
  if (h->force_tile14) return true;
  const long long tiles = (long long)h->B << (h->nb - 14);
  if (tiles >= 512) return true;
  // fewer tiles: only when the bigger tile saves a whole pass (e.g. 14-atom kets)
  const int p12 = 1 + (std::max(h->nb - 12, 0) + 7) / 8, p14 = 1 + (h->nb - 14 + 7) / 8;
  // measured on 14-atom kets: 16 tiles lose to two tiled passes (44 vs 58 sim-us/s),
  // 64 tiles win (136 vs 103), 256 tiles win 2.2x
  return p14 < p12 && tiles >= 48;
}

static void plan_passes(ryd_handle* h) {
  h->passes.clear();
  const int nb = h->nb, N = h->N, T = std::min(h->T, nb);
  const int C = 4;  // run bits (256 B contiguous) kept in every tile
  if (h->cfg.mode == RYD_MESOLVE && h->has_dbl) {
    // pair passes: both bits of each atom in the same tile
    int done = 0;  // atoms (counted from the low bit end) handled so far
    bool first = true;
    while (done < N) {
      int g;
      Pass p;
      if (first) {
        g = std::min(N, std::max(1, T / 2));
        p = make_pass(nb, {{0, g}, {N, g}});
      } else {
        const int c = std::max(0, std::min(std::min(C, done), T - 2));
        g = std::min(N - done, std::max(1, (T - c) / 2));
        p = make_pass(nb, {{0, c}, {done, g}, {N + done, g}});
      }
      for (int j = 0; j < g; ++j) {
        const int pb = done + j, pa = N + done + j;
        const int qb = local_of(p.tile, pb), qa = local_of(p.tile, pa);
        p.flip_q.push_back(qb);
        p.flip_q.push_back(qa);
        p.dbl.push_back({qb, qa});
      }
      p.include_diag = first;
      h->passes.push_back(p);
      done += std::max(g, 1);
      first = false;
    }
  } else {
    int done = 0;
    bool first = true;
    while (done < nb) {
      int g;
      Pass p;
      if (first) {
        g = T;
        if (h->auto_tile && nb >= 14 && tile14_pays(h)) {
          g = 14;
          p = make_pass(nb, {{0, g}});
          p.use14 = true;
        } else {
          p = make_pass(nb, {{0, g}});
        }
      } else {
        // the remaining bits are spread evenly over the passes they need, and
        // the rest of each tile is filled with contiguous run bits (longer
        // coalesced runs, full-size tiles) as long as >= 1024 workgroups remain
        const int cmin = std::max(0, std::min(std::min(C, done), T - 1));
        const int cap = std::max(1, T - cmin);
        const int left = nb - done;
        const int k = (left + cap - 1) / cap;
        g = std::max(1, (left + k - 1) / k);
        int c = cmin;
        int logB = 0;
        while ((1 << (logB + 1)) <= h->B) ++logB;
        while (c + g < T && c < done && (nb - (c + 1 + g)) + logB >= 10) ++c;
        p = make_pass(nb, {{0, c}, {done, g}});
      }
      for (int j = 0; j < g; ++j) p.flip_q.push_back(local_of(p.tile, done + j));
      p.include_diag = first;
      h->passes.push_back(p);
      done += std::max(g, 1);
      first = false;
    }
  }
  h->stats.passes = (int)h->passes.size();
  h->passes_valid = true;
}

extern "C" const char* ryd_last_error(void) { return g_err.c_str(); }
extern "C" int ryd_abi_version(void) { return RYD_ABI_VERSION; }

extern "C" int ryd_create(const ryd_config* cfg, ryd_handle** out) {
  if (!cfg || !out) return fail(RYD_ERR_INVALID, "null argument");
  if (cfg->abi_version != RYD_ABI_VERSION)
    return fail(RYD_ERR_INVALID, "ABI version mismatch: caller %d, library %d",
                cfg->abi_version, RYD_ABI_VERSION);
  if (cfg->mode != RYD_SESOLVE && cfg->mode != RYD_MESOLVE)
    return fail(RYD_ERR_INVALID, "unknown mode %d", cfg->mode);
  const int nb = cfg->mode == RYD_MESOLVE ? 2 * cfg->n_qubits : cfg->n_qubits;
  if (cfg->n_qubits < 1 || nb > RYD_MAX_QUBITS)
    return fail(RYD_ERR_INVALID, "n_qubits=%d out of range for mode %d (index bits %d > %d)",
                cfg->n_qubits, cfg->mode, nb, RYD_MAX_QUBITS);
  if (cfg->batch < 1 || cfg->batch > 65535)
    return fail(RYD_ERR_INVALID, "batch=%d out of range [1, 65535]", cfg->batch);
  int T = cfg->tile_bits ? cfg->tile_bits : 12;
  if (T < 2 || T > 13) return fail(RYD_ERR_INVALID, "tile_bits=%d out of range [2, 13]", T);
  HIPCHK(hipSetDevice(cfg->device));
  ryd_handle* h = new ryd_handle();
  h->cfg = *cfg;
  h->N = cfg->n_qubits;
  h->nb = nb;
  h->B = cfg->batch;
  h->T = T;
  h->auto_tile = cfg->tile_bits == 0;
  if (h->auto_tile) {
    // Small states: the fewest passes first, then enough tiles to occupy the
    // 256 CUs (a 2^12 tile keeps one CU busy for ~8 us; a 14-atom ket would
    // run on 4 CUs).  passes(T) = 1 + ceil((nb - T) / (T - 4)).
    auto passes = [&](int t) { return nb <= t ? 1 : 1 + (nb - t + (t - 5)) / (t - 4); };
    int best = 12;
    for (int t = 12; t >= 8; --t) {
      if (passes(t) > passes(12)) break;
      best = t;
      if (((long long)cfg->batch << std::max(nb - t, 0)) >= 256) break;
    }
    h->T = best;
  }
  h->dim = (size_t)1 << nb;
  // k_apply12 (register-resident tile kernel) measured 8-14 % slower than the
  // generic kernel on MI355X in round 1 (profiles/r01): opt-in until it wins.
  h->no_fast_apply = true;
  if (const char* ev = std::getenv("RYD_FAST_APPLY")) h->no_fast_apply = ev[0] != '1';
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  hipError_t e;
  if ((e = hipMalloc((void**)&h->wA, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->wB, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->kbuf, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->coefs_dev, (size_t)h->B * h->N * 4 * sizeof(double))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc of work vectors (%zu B each) failed: %s", bytes,
                hipGetErrorString(e));
  }
  // default: no interaction, no dissipator
  h->e0_mats = 1;
  if ((e = hipMalloc((void**)&h->e0_dev, ((size_t)1 << h->N) * sizeof(double))) != hipSuccess ||
      (e = hipMemset(h->e0_dev, 0, ((size_t)1 << h->N) * sizeof(double))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc e0 failed: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < 4; ++i) h->Sd[i] = h->J[i] = make_double2(0, 0);
  // the 2^12-amplitude tile needs 64 KiB + tables of dynamic LDS (CDNA4: 160 KiB/CU)
  if ((e = hipFuncSetAttribute((const void*)k_apply<RYD_SESOLVE, 512>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_MESOLVE, 512>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_SESOLVE, 1024>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_MESOLVE, 1024>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, true, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, true, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply12<RYD_SESOLVE, 0>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply12<RYD_MESOLVE, 0>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply12<RYD_SESOLVE, 4>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply12<RYD_MESOLVE, 4>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
  }
  plan_passes(h);
  *out = h;
  return RYD_OK;
}

extern "C" void ryd_destroy(ryd_handle* h) {
  if (!h) return;
  hipFree(h->wA);
  hipFree(h->wB);
  hipFree(h->kbuf);
  hipFree(h->coefs_dev);
  hipFree(h->e0_dev);
  hipFree(h->pp_dev);
  hipFree(h->desc_dev);
  hipFree(h->sched_dev);
  hipFree(h->gen_tcoef);
  hipFree(h->gen_terms_dev);
  hipFree(h->gen_series_dev);
  hipFree(h->gen_conj_dev);
  hipFree(h->gen_scale_dev);
  hipFree(h->mc_pool);
  for (auto& t : h->gen_host) {
    hipFree((void*)t.dev.row_ptr);
    hipFree((void*)t.dev.col);
    hipFree((void*)t.dev.val);
  }
  for (auto& p : h->ev_used) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  for (auto& p : h->ev_free) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  delete h;
}

extern "C" int ryd_set_series(ryd_handle* h, int32_t n_series, int32_t n_knots,
                              const double* tknots, const double* pp) {
  if (!h || !tknots || !pp) return fail(RYD_ERR_INVALID, "null argument");
  if (n_series < 1 || n_knots < 2) return fail(RYD_ERR_INVALID, "need >= 1 series and >= 2 knots");
  for (int i = 1; i < n_knots; ++i)
    if (!(tknots[i] > tknots[i - 1]))
      return fail(RYD_ERR_INVALID, "tknots must be strictly increasing (index %d)", i);
  HIPCHK(hipSetDevice(h->cfg.device));
  const int n_int = n_knots - 1;
  h->n_series = n_series;
  h->n_knots = n_knots;
  h->tknots.assign(tknots, tknots + n_knots);
  const size_t cnt = (size_t)n_series * n_int * 4;
  h->pp_host.resize(cnt);
  for (size_t i = 0; i < cnt; ++i) h->pp_host[i] = {pp[2 * i], pp[2 * i + 1]};
  // per-interval bounds of each series: |S|, max(Re S, 0), max(-Re S, 0)
  h->s_abs.assign((size_t)n_series * n_int, 0.0);
  h->s_pos.assign((size_t)n_series * n_int, 0.0);
  h->s_neg.assign((size_t)n_series * n_int, 0.0);
  h->s_curv.assign((size_t)n_series * n_int, 0.0);
  for (int s = 0; s < n_series; ++s)
    for (int i = 0; i < n_int; ++i) {
      const std::complex<double>* p = &h->pp_host[((size_t)s * n_int + i) * 4];
      const double dt = tknots[i + 1] - tknots[i];
      // value at the left knot is p[3]; deviation bounded by the other terms
      const double dev = std::abs(p[2]) * dt + std::abs(p[1]) * dt * dt + std::abs(p[0]) * dt * dt * dt;
      h->s_abs[(size_t)s * n_int + i] = std::abs(p[3]) + dev;
      h->s_pos[(size_t)s * n_int + i] = std::max(p[3].real() + dev, 0.0);
      h->s_neg[(size_t)s * n_int + i] = std::max(-p[3].real() + dev, 0.0);
      h->s_curv[(size_t)s * n_int + i] = std::abs(p[1]) * dt * dt + std::abs(p[0]) * dt * dt * dt;
    }
  if (h->pp_dev) hipFree(h->pp_dev);
  h->pp_dev = nullptr;
  HIPCHK(hipMalloc((void**)&h->pp_dev, cnt * sizeof(cplx)));
  HIPCHK(hipMemcpy(h->pp_dev, h->pp_host.data(), cnt * sizeof(cplx), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  return RYD_OK;
}

extern "C" int ryd_set_qubit_desc(ryd_handle* h, const ryd_qdesc* desc) {
  if (!h || !desc) return fail(RYD_ERR_INVALID, "null argument");
  if (h->n_series == 0) return fail(RYD_ERR_STATE, "ryd_set_series must be called first");
  const size_t cnt = (size_t)h->B * h->N;
  for (size_t i = 0; i < cnt; ++i) {
    const int idx[3] = {desc[i].drive_series, desc[i].det_series, desc[i].off_series};
    for (int j = 0; j < 3; ++j)
      if (idx[j] < -1 || idx[j] >= h->n_series)
        return fail(RYD_ERR_INVALID, "series index %d out of range at entry %zu", idx[j], i);
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  h->desc_host.assign(desc, desc + cnt);
  if (!h->desc_dev) HIPCHK(hipMalloc((void**)&h->desc_dev, cnt * sizeof(ryd_qdesc)));
  HIPCHK(hipMemcpy(h->desc_dev, desc, cnt * sizeof(ryd_qdesc), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  return RYD_OK;
}

static void compute_bounds(ryd_handle* h) {
  const int n_int = h->n_knots - 1;
  h->bd_drive.assign(n_int, 0.0);
  h->bd_pos.assign(n_int, 0.0);
  h->bd_neg.assign(n_int, 0.0);
  h->bd_curv.assign(n_int, 0.0);
  std::vector<double> dr(n_int), po(n_int), ne(n_int), cu(n_int);
  for (int b = 0; b < h->B; ++b) {
    std::fill(dr.begin(), dr.end(), 0.0);
    std::fill(po.begin(), po.end(), 0.0);
    std::fill(ne.begin(), ne.end(), 0.0);
    std::fill(cu.begin(), cu.end(), 0.0);
    for (int k = 0; k < h->N; ++k) {
      const ryd_qdesc& d = h->desc_host[(size_t)b * h->N + k];
      if (d.drive_series >= 0) {
        const double* a = &h->s_abs[(size_t)d.drive_series * n_int];
        const double sc = std::fabs(d.drive_scale);
        for (int i = 0; i < n_int; ++i) dr[i] += sc * a[i];
        const double* cv = &h->s_curv[(size_t)d.drive_series * n_int];
        for (int i = 0; i < n_int; ++i) cu[i] += sc * cv[i];
      }
      auto add_det = [&](int s, double sc) {
        if (s < 0 || sc == 0.0) return;
        const double* P = &h->s_pos[(size_t)s * n_int];
        const double* M = &h->s_neg[(size_t)s * n_int];
        const double* cv = &h->s_curv[(size_t)s * n_int];
        for (int i = 0; i < n_int; ++i) cu[i] += std::fabs(sc) * cv[i];
        for (int i = 0; i < n_int; ++i) {
          if (sc > 0) { po[i] += sc * P[i]; ne[i] += sc * M[i]; }
          else { po[i] += -sc * M[i]; ne[i] += -sc * P[i]; }
        }
      };
      add_det(d.det_series, d.det_scale);
      add_det(d.off_series, d.off_scale);
    }
    for (int i = 0; i < n_int; ++i) {
      h->bd_drive[i] = std::max(h->bd_drive[i], dr[i]);
      h->bd_pos[i] = std::max(h->bd_pos[i], po[i]);
      h->bd_neg[i] = std::max(h->bd_neg[i], ne[i]);
      h->bd_curv[i] = std::max(h->bd_curv[i], cu[i]);
    }
  }
  // MODEL 1 of the persistent kernel: inside every trajectory all driven atoms
  // share (series, scale) and that series is real-valued
  const int n_int2 = h->n_knots - 1;
  std::vector<char> series_real(h->n_series, 1);
  for (int sidx = 0; sidx < h->n_series; ++sidx)
    for (size_t i = 0; i < (size_t)n_int2 * 4; ++i)
      if (h->pp_host[(size_t)sidx * n_int2 * 4 + i].imag() != 0.0) { series_real[sidx] = 0; break; }
  bool uni = true;
  for (int b = 0; b < h->B && uni; ++b) {
    int ser = -2;
    double sc = 0.0;
    for (int k = 0; k < h->N; ++k) {
      const ryd_qdesc& d = h->desc_host[(size_t)b * h->N + k];
      if (d.drive_series < 0) continue;
      if (ser == -2) { ser = d.drive_series; sc = d.drive_scale; }
      else if (ser != d.drive_series || sc != d.drive_scale) { uni = false; break; }
      if (!series_real[d.drive_series]) { uni = false; break; }
    }
  }
  h->uniform_real_drive = uni;
  bool dreal = true;
  for (const ryd_qdesc& d : h->desc_host)
    if (d.drive_series >= 0 && !series_real[d.drive_series]) { dreal = false; break; }
  h->drive_real = dreal;
  h->bounds_valid = true;
}

extern "C" int ryd_set_interaction(ryd_handle* h, const double* U, int32_t n_mats) {
  if (!h || !U) return fail(RYD_ERR_INVALID, "null argument");
  if (n_mats != 1 && n_mats != h->B)
    return fail(RYD_ERR_INVALID, "n_mats must be 1 or batch (%d), got %d", h->B, n_mats);
  HIPCHK(hipSetDevice(h->cfg.device));
  const int N = h->N;
  const size_t D = (size_t)1 << N;
  double lo = 0, hi = 0;
  for (int m = 0; m < n_mats; ++m) {
    double l = 0, u = 0;
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j) {
        const double v = U[((size_t)m * N + i) * N + j];
        if (v != U[((size_t)m * N + j) * N + i])
          return fail(RYD_ERR_INVALID, "interaction matrix %d not symmetric at (%d,%d)", m, i, j);
        if (v > 0) u += v; else l += v;
      }
    lo = std::min(lo, l);
    hi = std::max(hi, u);
  }
  h->e0_min = lo;
  h->e0_max = hi;
  double* Udev = nullptr;
  HIPCHK(hipMalloc((void**)&Udev, (size_t)n_mats * N * N * sizeof(double)));
  hipError_t e = hipMemcpy(Udev, U, (size_t)n_mats * N * N * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (h->e0_dev) hipFree(h->e0_dev);
    h->e0_dev = nullptr;
    e = hipMalloc((void**)&h->e0_dev, (size_t)n_mats * D * sizeof(double));
  }
  if (e == hipSuccess) {
    dim3 grid((unsigned)((D + 255) / 256), n_mats);
    hipLaunchKernelGGL(k_build_e0, grid, dim3(256), 0, 0, Udev, N, h->e0_dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  hipFree(Udev);
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "building E0 failed: %s", hipGetErrorString(e));
  h->e0_mats = n_mats;
  return RYD_OK;
}

extern "C" int ryd_set_dissipator(ryd_handle* h, const double* S) {
  if (!h || !S) return fail(RYD_ERR_INVALID, "null argument");
  if (h->cfg.mode != RYD_MESOLVE)
    return fail(RYD_ERR_INVALID, "dissipator only valid for a mesolve handle");
  bool dbl = false;
  double norm = 0.0;
  for (int r = 0; r < 4; ++r) {
    double row = 0.0;
    for (int c = 0; c < 4; ++c) {
      const double re = S[2 * (4 * r + c)], im = S[2 * (4 * r + c) + 1];
      const double a = std::hypot(re, im);
      row += a;
      if (a == 0.0) continue;
      if (c == r) continue;
      if (c == 3 - r) { dbl = true; continue; }
      return fail(RYD_ERR_UNSUPPORTED,
                  "dissipator entry S[%d][%d] (single flip with pair-dependent coefficient) "
                  "is not supported by this ABI version", r, c);
    }
    norm = std::max(norm, row);
  }
  for (int r = 0; r < 4; ++r) {
    h->Sd[r] = make_double2(S[2 * (4 * r + r)], S[2 * (4 * r + r) + 1]);
    h->J[r] = make_double2(S[2 * (4 * r + (3 - r))], S[2 * (4 * r + (3 - r)) + 1]);
  }
  h->diss_norm = norm * h->N;
  const bool replan = dbl != h->has_dbl;
  h->has_dbl = dbl;
  if (replan) plan_passes(h);
  return RYD_OK;
}

// ---------------------------------------------------------------------------
// generator application
// ---------------------------------------------------------------------------
struct MixPoint {
  int idx1, idx2;
  double u1, u2, w1, w2;
};

static int find_interval(const ryd_handle* h, double t) {
  const int n_int = h->n_knots - 1;
  int i = int(std::upper_bound(h->tknots.begin(), h->tknots.end(), t) - h->tknots.begin()) - 1;
  return std::min(std::max(i, 0), n_int - 1);
}

static int launch_eval(ryd_handle* h, const MixPoint& m, hipStream_t st) {
  const int total = h->B * h->N;
  hipLaunchKernelGGL(k_eval_coefs, dim3((total + 127) / 128), dim3(128), 0, st, h->pp_dev,
                     h->n_knots - 1, h->desc_dev, total, m.idx1, m.u1, m.w1, m.idx2, m.u2, m.w2,
                     h->coefs_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

static int timing_begin(ryd_handle* h, hipStream_t st, std::pair<hipEvent_t, hipEvent_t>& ev) {
  if (h->ev_free.empty()) {
    HIPCHK(hipEventCreate(&ev.first));
    HIPCHK(hipEventCreate(&ev.second));
  } else {
    ev = h->ev_free.back();
    h->ev_free.pop_back();
  }
  HIPCHK(hipEventRecord(ev.first, st));
  return RYD_OK;
}

static int launch_apply14(ryd_handle* h, const Apply14Args& B, hipStream_t st) {
  const size_t lds = (size_t)4 * 1024 * sizeof(cplx) + 2 * 128 * sizeof(double) + 2 * 16 * sizeof(double);
  dim3 grid((unsigned)(1ull << (h->nb - 14)), h->B);
  const bool se = h->cfg.mode == RYD_SESOLVE;
  const bool full = B.n_flip == 14;
#define RYD_LAUNCH14(M, RL, FL) \
  hipLaunchKernelGGL((k_apply14<M, RL, FL>), grid, dim3(1024), lds, st, B)
  if (h->drive_real) {
    if (se) { if (full) RYD_LAUNCH14(RYD_SESOLVE, true, true); else RYD_LAUNCH14(RYD_SESOLVE, true, false); }
    else { if (full) RYD_LAUNCH14(RYD_MESOLVE, true, true); else RYD_LAUNCH14(RYD_MESOLVE, true, false); }
  } else {
    if (se) { if (full) RYD_LAUNCH14(RYD_SESOLVE, false, true); else RYD_LAUNCH14(RYD_SESOLVE, false, false); }
    else { if (full) RYD_LAUNCH14(RYD_MESOLVE, false, true); else RYD_LAUNCH14(RYD_MESOLVE, false, false); }
  }
#undef RYD_LAUNCH14
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

// out = post * (base + scale * G~ in); all passes.  `in` must differ from `out`
// unless single-element hazards are impossible (never used in place here).
static int apply_generator(ryd_handle* h, const cplx* in, const cplx* base, cplx* out,
                           double wmix, double scale, double shift, cplx post,
                           hipStream_t st, bool with_decay = false) {
  if (!h->passes_valid) plan_passes(h);
  // Monte-Carlo H_eff: real diagonal, centred (the centre is a scalar factor
  // of the exponential, applied by exp_step through `post`)
  const double dec_b = with_decay ? wmix * h->mc_b : 0.0;
  const double dec_a = -0.5 * h->N * dec_b;
  const int np = (int)h->passes.size();
  for (int pi = 0; pi < np; ++pi) {
    const Pass& p = h->passes[pi];
    PassArgs A;
    std::memset(&A, 0, sizeof A);
    A.in = in;
    A.kin = pi > 0 ? h->kbuf : nullptr;
    A.kout = h->kbuf;
    A.final_pass = pi == np - 1;
    A.base = A.final_pass ? base : nullptr;
    A.out = out;
    A.coefs = h->coefs_dev;
    A.e0 = h->e0_dev;
    A.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
    A.wmix = wmix;
    A.scale = scale;
    A.shift = shift;
    A.dec_a = dec_a;
    A.dec_b = dec_b;
    A.post = post;
    for (int i = 0; i < 4; ++i) { A.Sd[i] = h->Sd[i]; A.J[i] = h->J[i]; }
    A.tile = p.tile;
    A.outer = p.outer;
    A.N = h->N;
    A.nb = h->nb;
    A.T = p.T;
    A.n_flip = (int)p.flip_q.size();
    A.n_dbl = (int)p.dbl.size();
    A.include_diag = p.include_diag;
    for (int i = 0; i < A.n_flip; ++i) A.flip_q[i] = (signed char)p.flip_q[i];
    for (int i = 0; i < A.n_dbl; ++i) {
      A.dbl_qb[i] = (signed char)p.dbl[i].first;
      A.dbl_qa[i] = (signed char)p.dbl[i].second;
    }
    const int TL = p.T >> 1, TH = p.T - TL;
    const size_t lds = ((size_t)1 << p.T) * sizeof(cplx) + ((1 << TL) + (1 << TH)) * sizeof(double) +
                       2 * MAXF * sizeof(cplx);
    dim3 grid((unsigned)(1ull << p.n_outer_bits), h->B);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (h->timing) { int rc = timing_begin(h, st, ev); if (rc) return rc; }
    if (p.use14) {
      Apply14Args B;
      std::memset(&B, 0, sizeof B);
      B.in = in;
      B.base = A.final_pass ? base : nullptr;
      B.out = out;
      B.kout = A.final_pass ? nullptr : h->kbuf;
      B.coefs = h->coefs_dev;
      B.e0 = h->e0_dev;
      B.e0_stride = A.e0_stride;
      B.wmix = wmix;
      B.diag_scale = 1.0;
      B.scale = scale;
      B.shift = shift;
      B.dec_a = dec_a;
      B.dec_b = dec_b;
      B.post = post;
      for (int i = 0; i < 4; ++i) B.Sd[i] = h->Sd[i];
      B.N = h->N;
      B.nb = h->nb;
      B.n_flip = std::min(14, h->nb);
      int rc14 = launch_apply14(h, B, st);
      if (rc14) return rc14;
      if (h->timing) {
        HIPCHK(hipEventRecord(ev.second, st));
        h->ev_used.push_back(ev);
      }
      h->stats.n_launches++;
      continue;
    }
    // contiguous single-flip range on a full 2^12 tile -> specialised kernel
    bool fast = p.T == 12 && A.n_dbl == 0 && A.n_flip >= 1 && !h->no_fast_apply && !with_decay &&
                (A.flip_q[0] == 0 || A.flip_q[0] == 4);
    for (int i = 1; i < A.n_flip && fast; ++i) fast = A.flip_q[i] == A.flip_q[0] + i;
    if (fast) {
      const size_t lds12 = ((size_t)1 << 12) * sizeof(cplx) + 2 * 64 * sizeof(double) +
                           2 * MAXF * sizeof(double);
      const bool se = h->cfg.mode == RYD_SESOLVE;
      if (A.flip_q[0] == 0) {
        if (se) hipLaunchKernelGGL((k_apply12<RYD_SESOLVE, 0>), grid, dim3(512), lds12, st, A);
        else hipLaunchKernelGGL((k_apply12<RYD_MESOLVE, 0>), grid, dim3(512), lds12, st, A);
      } else {
        if (se) hipLaunchKernelGGL((k_apply12<RYD_SESOLVE, 4>), grid, dim3(512), lds12, st, A);
        else hipLaunchKernelGGL((k_apply12<RYD_MESOLVE, 4>), grid, dim3(512), lds12, st, A);
      }
    } else {
      const bool wide = (size_t)grid.x * grid.y <= 512 && p.T >= 10;
      if (h->cfg.mode == RYD_SESOLVE) {
        if (wide) hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 1024>), grid, dim3(1024), lds, st, A);
        else hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 512>), grid, dim3(512), lds, st, A);
      } else {
        if (wide) hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 1024>), grid, dim3(1024), lds, st, A);
        else hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 512>), grid, dim3(512), lds, st, A);
      }
    }
    HIPCHK(hipGetLastError());
    if (h->timing) {
      HIPCHK(hipEventRecord(ev.second, st));
      h->ev_used.push_back(ev);
    }
    h->stats.n_launches++;
  }
  h->stats.n_applications++;
  return RYD_OK;
}

static int check_ready(const ryd_handle* h) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  if (!h->pp_dev) return fail(RYD_ERR_STATE, "ryd_set_series has not been called");
  if (h->general) {
    if (h->gen_host.empty()) return fail(RYD_ERR_STATE, "ryd_general_add_term has not been called");
    return RYD_OK;
  }
  if (!h->desc_dev) return fail(RYD_ERR_STATE, "ryd_set_qubit_desc has not been called");
  return RYD_OK;
}

struct MixPoint;
static int launch_eval_general(ryd_handle* h, const MixPoint& m, hipStream_t st);
static int apply_general(ryd_handle* h, const MixPoint& m, const cplx* in, const cplx* base,
                         cplx* out, double scale, hipStream_t st);

extern "C" int ryd_apply_generator(ryd_handle* h, const void* in_dev, void* out_dev, double t,
                                   void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!in_dev || !out_dev || in_dev == out_dev)
    return fail(RYD_ERR_INVALID, "in/out must be distinct non-null device pointers");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  MixPoint m;
  m.idx1 = m.idx2 = find_interval(h, t);
  m.u1 = m.u2 = t - h->tknots[m.idx1];
  m.w1 = 1.0;
  m.w2 = 0.0;
  if (h->general) {
    if ((rc = launch_eval_general(h, m, st))) return rc;
    return apply_general(h, m, (const cplx*)in_dev, nullptr, (cplx*)out_dev, 1.0, st);
  }
  if ((rc = launch_eval(h, m, st))) return rc;
  return apply_generator(h, (const cplx*)in_dev, nullptr, (cplx*)out_dev, 1.0, 1.0, 0.0,
                         make_double2(1.0, 0.0), st);
}

// ---------------------------------------------------------------------------
// General path (explicit CSR terms) - host side
// ---------------------------------------------------------------------------
extern "C" int ryd_general_create(const ryd_general_config* cfg, ryd_handle** out) {
  if (!cfg || !out) return fail(RYD_ERR_INVALID, "null argument");
  if (cfg->abi_version != RYD_ABI_VERSION)
    return fail(RYD_ERR_INVALID, "ABI version mismatch: caller %d, library %d", cfg->abi_version,
                RYD_ABI_VERSION);
  if (cfg->dim < 1 || cfg->dim > ((int64_t)1 << 26))
    return fail(RYD_ERR_INVALID, "dim=%lld out of range", (long long)cfg->dim);
  if (cfg->batch < 1 || cfg->batch > 65535) return fail(RYD_ERR_INVALID, "batch out of range");
  HIPCHK(hipSetDevice(cfg->device));
  ryd_handle* h = new ryd_handle();
  h->general = true;
  h->cfg.abi_version = cfg->abi_version;
  h->cfg.device = cfg->device;
  h->cfg.mode = RYD_SESOLVE;
  h->cfg.batch = cfg->batch;
  h->B = cfg->batch;
  h->dim = (size_t)cfg->dim;
  h->N = 0;
  h->nb = 0;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  hipError_t e;
  if ((e = hipMalloc((void**)&h->wA, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->wB, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_tcoef, MAX_GEN_TERMS * sizeof(cplx))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_terms_dev, MAX_GEN_TERMS * sizeof(GenTermDev))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_series_dev, MAX_GEN_TERMS * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_conj_dev, MAX_GEN_TERMS * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_scale_dev, MAX_GEN_TERMS * sizeof(cplx))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc (general path) failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return RYD_OK;
}

extern "C" int ryd_general_add_term(ryd_handle* h, int64_t nnz, const int32_t* row_ptr,
                                    const int32_t* col, const double* val, int32_t series,
                                    int32_t conj, double scale_re, double scale_im,
                                    double row_norm) {
  if (!h || !h->general) return fail(RYD_ERR_INVALID, "not a general-path handle");
  if (!row_ptr || (nnz > 0 && (!col || !val))) return fail(RYD_ERR_INVALID, "null argument");
  if ((int)h->gen_host.size() >= MAX_GEN_TERMS)
    return fail(RYD_ERR_INVALID, "too many terms (max %d)", MAX_GEN_TERMS);
  if (series < -1 || series >= std::max(h->n_series, 1) || (series >= 0 && h->n_series == 0))
    return fail(RYD_ERR_INVALID, "series index %d out of range (call ryd_set_series first)", series);
  if (row_ptr[0] != 0 || row_ptr[h->dim] != nnz)
    return fail(RYD_ERR_INVALID, "row_ptr does not describe %lld non-zeros", (long long)nnz);
  for (int64_t e = 0; e < nnz; ++e)
    if (col[e] < 0 || (size_t)col[e] >= h->dim)
      return fail(RYD_ERR_INVALID, "column index out of range at entry %lld", (long long)e);
  HIPCHK(hipSetDevice(h->cfg.device));
  GenTermHost t;
  t.series = series;
  t.conj = conj;
  t.scale = std::complex<double>(scale_re, scale_im);
  t.row_norm = row_norm;
  HIPCHK(hipMalloc((void**)&t.dev.row_ptr, (h->dim + 1) * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.col, std::max<int64_t>(nnz, 1) * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.val, std::max<int64_t>(nnz, 1) * sizeof(cplx)));
  HIPCHK(hipMemcpy((void*)t.dev.row_ptr, row_ptr, (h->dim + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (nnz > 0) {
    HIPCHK(hipMemcpy((void*)t.dev.col, col, nnz * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((void*)t.dev.val, val, nnz * sizeof(cplx), hipMemcpyHostToDevice));
  }
  h->gen_host.push_back(t);
  const int n = (int)h->gen_host.size();
  std::vector<GenTermDev> devs(n);
  std::vector<int> ser(n), cj(n);
  std::vector<cplx> sc(n);
  for (int i = 0; i < n; ++i) {
    devs[i] = h->gen_host[i].dev;
    ser[i] = h->gen_host[i].series;
    cj[i] = h->gen_host[i].conj;
    sc[i] = make_double2(h->gen_host[i].scale.real(), h->gen_host[i].scale.imag());
  }
  HIPCHK(hipMemcpy(h->gen_terms_dev, devs.data(), n * sizeof(GenTermDev), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_series_dev, ser.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_conj_dev, cj.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_scale_dev, sc.data(), n * sizeof(cplx), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  return RYD_OK;
}

static void compute_bounds_general(ryd_handle* h) {
  const int n_int = h->n_knots - 1;
  h->bd_drive.assign(n_int, 0.0);
  h->bd_pos.assign(n_int, 0.0);
  h->bd_neg.assign(n_int, 0.0);
  h->bd_curv.assign(n_int, 0.0);
  for (const GenTermHost& t : h->gen_host) {
    const double w = std::abs(t.scale) * t.row_norm;
    for (int i = 0; i < n_int; ++i) {
      if (t.series >= 0) {
        h->bd_drive[i] += w * h->s_abs[(size_t)t.series * n_int + i];
        h->bd_curv[i] += w * h->s_curv[(size_t)t.series * n_int + i];
      } else {
        h->bd_drive[i] += w;
      }
    }
  }
  h->e0_min = h->e0_max = 0.0;
  h->uniform_real_drive = false;
  h->bounds_valid = true;
}

static int apply_general(ryd_handle* h, const MixPoint& m, const cplx* in, const cplx* base,
                         cplx* out, double scale, hipStream_t st) {
  const int n = (int)h->gen_host.size();
  if (n == 0) return fail(RYD_ERR_STATE, "no terms: call ryd_general_add_term first");
  GenArgs A;
  A.in = in;
  A.base = base;
  A.out = out;
  A.tcoef = h->gen_tcoef;
  A.terms = h->gen_terms_dev;
  A.dim = (long long)h->dim;
  A.n_terms = n;
  A.scale = scale;
  dim3 grid((unsigned)((h->dim + 255) / 256), h->B);
  hipLaunchKernelGGL(k_gen_apply, grid, dim3(256), 0, st, A);
  HIPCHK(hipGetLastError());
  h->stats.n_launches++;
  h->stats.n_applications++;
  (void)m;
  return RYD_OK;
}

static int launch_eval_general(ryd_handle* h, const MixPoint& m, hipStream_t st) {
  const int n = (int)h->gen_host.size();
  hipLaunchKernelGGL(k_gen_coefs, dim3((n + 63) / 64), dim3(64), 0, st, h->pp_dev, h->n_knots - 1,
                     h->gen_series_dev, h->gen_conj_dev, h->gen_scale_dev, n, m.idx1, m.u1, m.w1,
                     m.u2, m.w2, h->gen_tcoef);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

// ---------------------------------------------------------------------------
// stepping: schedule (host) -> generic multi-launch path or persistent kernel
// ---------------------------------------------------------------------------
static const double kS3 = 1.7320508075688772;
static const double kC1 = 0.5 - kS3 / 6.0, kC2 = 0.5 + kS3 / 6.0;  // Gauss nodes
static const double kA1 = 0.25 + kS3 / 6.0, kA2 = 0.25 - kS3 / 6.0;  // CF4 weights

// Taylor order and spectral shift of one exponential exp(h (w1 G(t1) + w2 G(t2)))
// with both Gauss points inside knot interval `idx`.
static void plan_exp(ryd_handle* h, int idx, double hstep, double w1, double w2,
                     const ryd_opts& o, int* order_out, double* shift_out) {
  const double wmix = w1 + w2;
  const double drive = wmix * h->bd_drive[idx];
  const double dpos = wmix * h->bd_pos[idx], dneg = wmix * h->bd_neg[idx];
  const double lo = wmix * h->e0_min - dpos, hi = wmix * h->e0_max + dneg;
  double bound, shift = 0.0;
  if (h->general) {
    bound = drive;  // sum_t |coef_t| ||A_t||_inf; no spectral shift
  } else if (h->cfg.mode == RYD_SESOLVE) {
    shift = 0.5 * (lo + hi);  // H' = H - shift: halves the spectral radius
    bound = 0.5 * (hi - lo) + drive;
    if (h->mc) bound += wmix * std::fabs(h->mc_b) * 0.5 * h->N;  // centred decay diagonal
  } else {
    bound = 2.0 * (0.5 * (hi - lo) + drive) + wmix * h->diss_norm;
  }
  h->stats.norm_bound = bound / std::max(wmix, 1e-300);
  const double rho = std::fabs(hstep) * bound;
  int order = o.taylor_order;
  if (order <= 0) {
    const int cap = std::min(o.max_order > 0 ? o.max_order : 24, 32);
    const double tol = o.tol > 0 ? o.tol : 1e-12;
    double term = rho;  // rho^(m+1)/(m+1)! for m = 0
    order = 1;
    while (order < cap) {
      term *= rho / (order + 1);  // now rho^(order+1)/(order+1)!
      if (term <= tol) break;
      ++order;
    }
  }
  if (order < 2) order = 2;
  if (order > 32) order = 32;
  h->stats.last_order = order;
  *order_out = order;
  *shift_out = shift;
}

// CF4 steps covering [t0, t1]: never straddling a spline knot (inside a knot
// interval every coefficient is a single cubic), optionally capped by max_step.
static void build_schedule(ryd_handle* h, double t0, double t1, const ryd_opts& o,
                           std::vector<StepDesc>& out) {
  const double eps = 1e-12;
  double t = t0;
  while (t < t1 - eps) {
    const int idx = find_interval(h, t + eps);
    double tend = t1;  // the last interval extends to t1 (extrapolation, as scipy does)
    if (idx < h->n_knots - 2) tend = std::min(t1, h->tknots[idx + 1]);
    if (tend <= t + eps) tend = t1;
    const double len = tend - t;
    int nsub = 1;
    if (o.max_step > 0) nsub = std::max(1, (int)std::ceil(len / o.max_step - 1e-9));
    {
      // The local error of the 4th-order Magnus step is dominated by the
      // non-linear (quadratic + cubic) part of the spline inside the interval -
      // large only where it rings next to a kink of the waveform.  Calibrated
      // against converged references: err ~ 1e-5 * h * curvature,
      // and it falls as n^-4 with n equal sub-steps.
      const double dtk = h->tknots[idx + 1] - h->tknots[idx];
      const double frac = dtk > 0 ? std::min(1.0, len / dtk) : 1.0;
      const double est = 1e-5 * len * h->bd_curv[idx] * frac * frac;
      const double mtol = o.magnus_tol > 0 ? o.magnus_tol : 1e-10;
      if (est > mtol) {
        const int nm = (int)std::ceil(std::pow(est / mtol, 0.25));
        nsub = std::max(nsub, std::min(nm, 256));
      }
    }
    {
      // keep the Taylor argument rho = h * ||G~|| near 1: beyond that the
      // polynomial degree grows faster than the step (and cancellation sets in)
      int ord;
      double sh;
      plan_exp(h, idx, len / nsub, kA1, kA2, o, &ord, &sh);
      const double rho = (len / nsub) * h->stats.norm_bound * (kA1 + kA2);
      if (rho > 1.5) nsub *= (int)std::ceil(rho / 1.0);
    }
    const double hs = len / nsub;
    for (int s = 0; s < nsub; ++s) {
      const double ta = t + s * hs;
      StepDesc d;
      std::memset(&d, 0, sizeof d);
      d.h = hs;
      d.idx = idx;
      d.u1 = ta + kC1 * hs - h->tknots[idx];
      d.u2 = ta + kC2 * hs - h->tknots[idx];
      plan_exp(h, idx, hs, kA1, kA2, o, &d.order_a, &d.shift_a);
      plan_exp(h, idx, hs, kA2, kA1, o, &d.order_b, &d.shift_b);
      d.snap = -1;
      out.push_back(d);
    }
    t = tend;
  }
}

static bool hermitian_path(const ryd_handle* h) {
  return !h->general && h->cfg.mode == RYD_MESOLVE && !h->has_dbl && h->N >= 7 && h->N <= 14 &&
         h->auto_tile && tile14_pays(h);
}

// One exponential  state <- exp(h * G~) state  on the generic multi-launch path.
static int exp_step(ryd_handle* h, cplx* state, double hstep, const MixPoint& m, int order,
                    double shift, hipStream_t st) {
  int rc;
  if (h->general) {
    if ((rc = launch_eval_general(h, m, st))) return rc;
    const cplx* gin = state;
    cplx* gbufs[2] = {h->wA, h->wB};
    int gw = 0;
    for (int j = order; j >= 1; --j) {
      cplx* out = j == 1 ? state : gbufs[gw];
      if ((rc = apply_general(h, m, gin, state, out, hstep / j, st))) return rc;
      gin = out;
      gw ^= 1;
    }
    return RYD_OK;
  }
  if ((rc = launch_eval(h, m, st))) return rc;
  const double wmix = m.w1 + m.w2;
  if (hermitian_path(h)) {
    // rho stays Hermitian, so G rho = P + P^dagger with P = (1/2) D.rho + the
    // column-bit flips only: one register-tile pass over rows + one tile-pair
    // symmetrisation instead of three tiled passes.
    const cplx* hin = state;
    cplx* hb[2] = {h->wA, h->wB};
    int hw = 0;
    for (int j = order; j >= 1; --j) {
      cplx* out = j == 1 ? state : hb[hw];
      Apply14Args B;
      std::memset(&B, 0, sizeof B);
      B.in = hin;
      B.kout = h->kbuf;
      B.coefs = h->coefs_dev;
      B.e0 = h->e0_dev;
      B.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
      B.wmix = wmix;
      B.diag_scale = 0.5;
      B.scale = 1.0;
      B.post = make_double2(1.0, 0.0);
      for (int i = 0; i < 4; ++i) B.Sd[i] = h->Sd[i];
      B.N = h->N;
      B.nb = h->nb;
      B.n_flip = h->N;
      std::pair<hipEvent_t, hipEvent_t> ev;
      if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
      if ((rc = launch_apply14(h, B, st))) return rc;
      if (h->timing) { HIPCHK(hipEventRecord(ev.second, st)); h->ev_used.push_back(ev); }
      SymmArgs S;
      S.P = h->kbuf;
      S.base = state;
      S.out = out;
      S.scale = hstep / j;
      S.N = h->N;
      const unsigned nt = 1u << (h->N - 5);
      if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
      hipLaunchKernelGGL(k_symm, dim3(nt, nt, h->B), dim3(256), 0, st, S);
      HIPCHK(hipGetLastError());
      if (h->timing) { HIPCHK(hipEventRecord(ev.second, st)); h->ev_used.push_back(ev); }
      h->stats.n_launches += 2;
      h->stats.n_applications++;
      hin = out;
      hw ^= 1;
    }
    return RYD_OK;
  }
  // Horner: w_m = psi; w_{j-1} = psi + (h/j) G' w_j; result w_0, times e^{-i h shift}
  const cplx one = make_double2(1.0, 0.0);
  const cplx* in = state;
  cplx* bufs[2] = {h->wA, h->wB};
  int which = 0;
  for (int j = order; j >= 1; --j) {
    cplx* out = j == 1 ? state : bufs[which];
    cplx post = one;
    if (j == 1) {
      // e^{-i h shift}, and for H_eff the centre of the decay diagonal
      const double mag = h->mc ? std::exp(hstep * wmix * (h->mc_a + 0.5 * h->N * h->mc_b)) : 1.0;
      post = make_double2(mag * std::cos(hstep * shift), -mag * std::sin(hstep * shift));
    }
    if ((rc = apply_generator(h, in, state, out, wmix, hstep / j, shift, post, st, h->mc))) return rc;
    in = out;
    which ^= 1;
  }
  return RYD_OK;
}

static unsigned mc_blocks(const ryd_handle* h) {
  return (unsigned)std::min<size_t>(std::max<size_t>(h->dim >> 10, 1), 128);
}

// Jump bookkeeping after one CF4 step of a Monte-Carlo solve (all on `st`).
static int mc_after_step(ryd_handle* h, cplx* state, hipStream_t st) {
  const unsigned nblk = mc_blocks(h);
  hipLaunchKernelGGL(k_mc_norm, dim3(nblk, h->B), dim3(256), 0, st, state, h->nb, h->mcs.norm2);
  hipLaunchKernelGGL(k_mc_reduced, dim3(nblk, h->B), dim3(256), 0, st, state, h->N, h->mcs, h->B, 0);
  hipLaunchKernelGGL(k_mc_select, dim3((h->B + 127) / 128), dim3(128), 0, st, h->mcs, h->B, h->N, 0);
  hipLaunchKernelGGL(k_mc_jump, dim3(nblk, h->B), dim3(256), 0, st, state, h->N, h->mcs);
  HIPCHK(hipGetLastError());
  h->stats.n_launches += 4;
  return RYD_OK;
}

// Snapshot of the state: a plain copy, or the normalised ket in a Monte-Carlo solve.
static int snapshot_copy(ryd_handle* h, const cplx* state, cplx* dst, hipStream_t st) {
  if (h->mc_active) {
    hipLaunchKernelGGL(k_mc_normalize, dim3(mc_blocks(h), h->B), dim3(256), 0, st, state, dst, h->nb,
                       h->mcs.lastnorm);
    HIPCHK(hipGetLastError());
    return RYD_OK;
  }
  HIPCHK(hipMemcpyAsync(dst, state, h->dim * (size_t)h->B * sizeof(cplx), hipMemcpyDeviceToDevice, st));
  return RYD_OK;
}

static int run_generic(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                       cplx* snaps, hipStream_t st) {
  int rc;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  for (const StepDesc& d : sched) {
    MixPoint m;
    m.idx1 = m.idx2 = d.idx;
    m.u1 = d.u1;
    m.u2 = d.u2;
    m.w1 = kA1; m.w2 = kA2;
    if ((rc = exp_step(h, state, d.h, m, d.order_a, d.shift_a, st))) return rc;
    m.w1 = kA2; m.w2 = kA1;
    if ((rc = exp_step(h, state, d.h, m, d.order_b, d.shift_b, st))) return rc;
    h->stats.n_steps++;
    if (h->mc_active && (rc = mc_after_step(h, state, st))) return rc;
    if (d.snap >= 0 && snaps && (rc = snapshot_copy(h, state, snaps + (size_t)d.snap * h->dim * h->B, st)))
      return rc;
  }
  (void)bytes;
  return RYD_OK;
}

template <int N, int MODEL, bool MC>
static int launch_traj2(ryd_handle* h, const TrajArgs& A, hipStream_t st) {
  constexpr int D = 1 << N;
  constexpr int NTT = D < 64 ? 64 : (N >= 11 ? 1024 : (D > 512 ? 512 : D));
  const size_t lds = 2 * (size_t)D * sizeof(cplx) + 4 * 16 * sizeof(double) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute((const void*)k_traj<N, NTT, MODEL, MC>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_traj<N, NTT, MODEL, MC>), dim3(h->B), dim3(NTT), lds, st, A);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

template <int N>
static int launch_traj(ryd_handle* h, const TrajArgs& A, hipStream_t st) {
  if (h->mc)
    return h->uniform_real_drive ? launch_traj2<N, 1, true>(h, A, st)
                                 : launch_traj2<N, 0, true>(h, A, st);
  return h->uniform_real_drive ? launch_traj2<N, 1, false>(h, A, st)
                               : launch_traj2<N, 0, false>(h, A, st);
}

// Persistent path (sesolve, N <= 12): one workgroup per trajectory keeps its
// state vector in LDS/registers for the whole schedule; one launch.
static int run_persistent(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                          cplx* snaps, hipStream_t st) {
  if (sched.empty()) return RYD_OK;
  const size_t bytes = sched.size() * sizeof(StepDesc);
  if (h->sched_cap < sched.size()) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_cap = 0;
    HIPCHK(hipMalloc((void**)&h->sched_dev, bytes * 2));
    h->sched_cap = sched.size() * 2;
  }
  // the schedule buffer may still be read by an earlier launch on `st`
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), bytes, hipMemcpyHostToDevice, st));
  TrajArgs A;
  A.state = state;
  A.snaps = snaps;
  A.pp = h->pp_dev;
  A.n_int = h->n_knots - 1;
  A.desc = h->desc_dev;
  A.e0 = h->e0_dev;
  A.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
  A.steps = h->sched_dev;
  A.n_steps = (int)sched.size();
  A.B = h->B;
  A.a1 = kA1;
  A.a2 = kA2;
  A.mc = h->mcs;
  A.mc_a = h->mc_a;
  A.mc_b = h->mc_b;
  A.mc_jumps = h->mc_active ? 1 : 0;
  int rc = RYD_ERR_INVALID;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
  switch (h->N) {
    case 1: rc = launch_traj<1>(h, A, st); break;
    case 2: rc = launch_traj<2>(h, A, st); break;
    case 3: rc = launch_traj<3>(h, A, st); break;
    case 4: rc = launch_traj<4>(h, A, st); break;
    case 5: rc = launch_traj<5>(h, A, st); break;
    case 6: rc = launch_traj<6>(h, A, st); break;
    case 7: rc = launch_traj<7>(h, A, st); break;
    case 8: rc = launch_traj<8>(h, A, st); break;
    case 9: rc = launch_traj<9>(h, A, st); break;
    case 10: rc = launch_traj<10>(h, A, st); break;
    case 11: rc = launch_traj<11>(h, A, st); break;
    case 12: rc = launch_traj<12>(h, A, st); break;
    default: return fail(RYD_ERR_INVALID, "persistent path needs N <= 12");
  }
  if (rc) return rc;
  if (h->timing) {
    HIPCHK(hipEventRecord(ev.second, st));
    h->ev_used.push_back(ev);
  }
  for (const StepDesc& d : sched) {
    h->stats.n_applications += d.order_a + d.order_b;
    h->stats.n_steps++;
  }
  h->stats.n_launches++;
  return RYD_OK;
}

static bool use_persistent(const ryd_handle* h) {
  return !h->general && h->cfg.mode == RYD_SESOLVE && h->N <= 12 && !h->force_generic;
}

extern "C" int ryd_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                         void* out_dev, const ryd_opts* opts, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!state_dev || !times || n_times < 2) return fail(RYD_ERR_INVALID, "need a state and >= 2 times");
  for (int i = 1; i < n_times; ++i)
    if (!(times[i] >= times[i - 1])) return fail(RYD_ERR_INVALID, "times must be non-decreasing");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->bounds_valid) { if (h->general) compute_bounds_general(h); else compute_bounds(h); }
  ryd_opts o;
  std::memset(&o, 0, sizeof o);
  if (opts) o = *opts;
  hipStream_t st = (hipStream_t)stream;
  cplx* state = (cplx*)state_dev;
  cplx* snaps = (cplx*)out_dev;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  std::vector<StepDesc> sched;
  // snapshot slot i-1 receives the state at times[i]
  for (int i = 1; i < n_times; ++i) {
    const size_t before = sched.size();
    build_schedule(h, times[i - 1], times[i], o, sched);
    if (snaps) {
      if (sched.size() > before) {
        sched.back().snap = i - 1;
      } else {  // zero-length interval: the state is unchanged
        if (before == 0) {
          if ((rc = snapshot_copy(h, state, snaps + (size_t)(i - 1) * h->dim * h->B, st))) return rc;
        } else {
          // duplicate time after at least one step: flush what we have, copy, continue
          if ((rc = use_persistent(h) ? run_persistent(h, state, sched, snaps, st)
                                      : run_generic(h, state, sched, snaps, st)))
            return rc;
          sched.clear();
          if ((rc = snapshot_copy(h, state, snaps + (size_t)(i - 1) * h->dim * h->B, st))) return rc;
        }
      }
    }
  }
  return use_persistent(h) ? run_persistent(h, state, sched, snaps, st)
                           : run_generic(h, state, sched, snaps, st);
}

extern "C" int ryd_evolve(ryd_handle* h, void* state_dev, double t0, double t1,
                          const ryd_opts* opts, void* stream) {
  if (!(t1 >= t0)) return fail(RYD_ERR_INVALID, "t1 < t0");
  const double times[2] = {t0, t1};
  return ryd_solve(h, state_dev, 2, times, nullptr, opts, stream);
}

extern "C" int ryd_set_collapse(ryd_handle* h, int32_t n_ops, const double* ops) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  if (h->general || h->cfg.mode != RYD_SESOLVE)
    return fail(RYD_ERR_INVALID, "collapse operators need a ket (sesolve) handle of the tuned path");
  if (n_ops < 0 || n_ops > MC_MAX_OPS || (n_ops > 0 && !ops))
    return fail(RYD_ERR_INVALID, "n_ops=%d out of range [0, %d]", n_ops, MC_MAX_OPS);
  HIPCHK(hipSetDevice(h->cfg.device));
  if (n_ops == 0) {
    h->mc = false;
    h->mc_n_ops = 0;
    h->mc_a = h->mc_b = 0.0;
    return RYD_OK;
  }
  // M = sum C^dag C must be diagonal: H_eff then only gains a real diagonal
  double m00 = 0, m11 = 0, m01r = 0, m01i = 0;
  for (int k = 0; k < n_ops; ++k) {
    const std::complex<double> c00(ops[8 * k + 0], ops[8 * k + 1]), c01(ops[8 * k + 2], ops[8 * k + 3]),
        c10(ops[8 * k + 4], ops[8 * k + 5]), c11(ops[8 * k + 6], ops[8 * k + 7]);
    m00 += std::norm(c00) + std::norm(c10);
    m11 += std::norm(c01) + std::norm(c11);
    const std::complex<double> x = std::conj(c00) * c01 + std::conj(c10) * c11;
    m01r += x.real();
    m01i += x.imag();
  }
  if (std::hypot(m01r, m01i) > 1e-13 * std::max(std::max(m00, m11), 1e-300))
    return fail(RYD_ERR_UNSUPPORTED,
                "sum C^dag C of the local collapse operators is not diagonal; use the "
                "master-equation solver for this noise model");
  const size_t B = (size_t)h->B, N = (size_t)h->N;
  const size_t n_dbl = 2 * B + 4 * N * B + 4 * B;
  const size_t bytes = n_dbl * sizeof(double) + B * sizeof(unsigned long long) +
                       MC_MAX_OPS * 4 * sizeof(cplx) + 3 * B * sizeof(int);
  if (!h->mc_pool) {
    HIPCHK(hipMalloc(&h->mc_pool, bytes));
    HIPCHK(hipMemset(h->mc_pool, 0, bytes));
    char* p = (char*)h->mc_pool;
    h->mcs.norm2 = (double*)p;    p += 2 * B * sizeof(double);
    h->mcs.red = (double*)p;      p += 4 * N * B * sizeof(double);
    h->mcs.target = (double*)p;   p += B * sizeof(double);
    h->mcs.refnorm = (double*)p;  p += B * sizeof(double);
    h->mcs.lastnorm = (double*)p; p += B * sizeof(double);
    h->mcs.scale = (double*)p;    p += B * sizeof(double);
    h->mc_seeds_dev = (unsigned long long*)p; p += B * sizeof(unsigned long long);
    h->mc_ops_dev = (cplx*)p;     p += MC_MAX_OPS * 4 * sizeof(cplx);
    h->mcs.flag = (int*)p;        p += B * sizeof(int);
    h->mcs.sel = (int*)p;         p += B * sizeof(int);
    h->mcs.count = (int*)p;
    h->mcs.seeds = h->mc_seeds_dev;
    h->mcs.ops = h->mc_ops_dev;
  }
  HIPCHK(hipMemcpy(h->mc_ops_dev, ops, (size_t)n_ops * 4 * sizeof(cplx), hipMemcpyHostToDevice));
  h->mcs.n_ops = n_ops;
  h->mc_n_ops = n_ops;
  h->mc_a = -0.5 * m00 * h->N;
  h->mc_b = 0.5 * (m00 - m11);
  h->mc = true;
  return RYD_OK;
}

extern "C" int ryd_mc_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                            void* out_dev, const uint64_t* seeds, const ryd_opts* opts,
                            void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->mc) return fail(RYD_ERR_STATE, "ryd_set_collapse has not been called");
  if (!state_dev || !seeds) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(h->mc_seeds_dev, seeds, (size_t)h->B * sizeof(unsigned long long),
                        hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(h->mcs.norm2, 0, 2 * (size_t)h->B * sizeof(double), st));
  hipLaunchKernelGGL(k_mc_norm, dim3(mc_blocks(h), h->B), dim3(256), 0, st, (const cplx*)state_dev,
                     h->nb, h->mcs.norm2);
  hipLaunchKernelGGL(k_mc_init, dim3((h->B + 127) / 128), dim3(128), 0, st, h->mcs, h->B, h->N);
  HIPCHK(hipGetLastError());
  h->mc_active = true;
  rc = ryd_solve(h, state_dev, n_times, times, out_dev, opts, stream);
  if (rc == RYD_OK) rc = snapshot_copy(h, (const cplx*)state_dev, (cplx*)state_dev, st);
  h->mc_active = false;
  return rc;
}

extern "C" int ryd_mc_get_jumps(ryd_handle* h, int32_t* counts, void* stream) {
  if (!h || !counts) return fail(RYD_ERR_INVALID, "null argument");
  if (!h->mc) return fail(RYD_ERR_STATE, "ryd_set_collapse has not been called");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(counts, h->mcs.count, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost));
  return RYD_OK;
}

extern "C" int ryd_set_path(ryd_handle* h, int32_t force_generic) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  h->force_generic = (force_generic & 1) != 0;
  h->no_fast_apply = (force_generic & 2) != 0;
  {
    const bool nt = (force_generic & 4) != 0, ft = (force_generic & 8) != 0;
    if (nt != h->no_tile14 || ft != h->force_tile14) {
      h->no_tile14 = nt;
      h->force_tile14 = ft;
      plan_passes(h);
    }
  }
  return RYD_OK;
}

// ---------------------------------------------------------------------------
// observables / marshalling
// ---------------------------------------------------------------------------
extern "C" int ryd_probabilities(ryd_handle* h, const void* state_dev, double* w_dev,
                                 int32_t reverse, void* stream) {
  if (!h || !state_dev || !w_dev) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t D = (size_t)1 << h->N;
  dim3 grid((unsigned)((D + 255) / 256), h->B);
  hipLaunchKernelGGL(k_probabilities, grid, dim3(256), 0, (hipStream_t)stream,
                     (const cplx*)state_dev, h->N, h->cfg.mode == RYD_MESOLVE, reverse, w_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

extern "C" int ryd_occupations(ryd_handle* h, const void* state_dev, double* out_dev,
                               void* stream) {
  if (!h || !state_dev || !out_dev) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)h->B * (h->N + 1) * sizeof(double), st));
  const size_t D = (size_t)1 << h->N;
  const unsigned nblk = (unsigned)std::min<size_t>((D + 255) / 256, 1024);
  hipLaunchKernelGGL(k_occupations, dim3(nblk, h->B), dim3(256), 0, st, (const cplx*)state_dev,
                     h->N, h->cfg.mode == RYD_MESOLVE, out_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

extern "C" int ryd_ket_to_dm(ryd_handle* h, const void* psi_dev, void* rho_dev, void* stream) {
  if (!h || !psi_dev || !rho_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (2 * h->N > RYD_MAX_QUBITS) return fail(RYD_ERR_INVALID, "2N exceeds %d", RYD_MAX_QUBITS);
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t DD = (size_t)1 << (2 * h->N);
  dim3 grid((unsigned)((DD + 255) / 256), h->B);
  hipLaunchKernelGGL(k_ket_to_dm, grid, dim3(256), 0, (hipStream_t)stream, (const cplx*)psi_dev,
                     h->N, (cplx*)rho_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

extern "C" int ryd_outer_accumulate(ryd_handle* h, const void* psi_dev, const double* weights,
                                    void* acc_dev, void* stream) {
  if (!h || !psi_dev || !acc_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (2 * h->N > RYD_MAX_QUBITS) return fail(RYD_ERR_INVALID, "2N exceeds %d", RYD_MAX_QUBITS);
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  double* wdev = nullptr;
  if (weights) {
    HIPCHK(hipMalloc((void**)&wdev, h->B * sizeof(double)));
    hipError_t e = hipMemcpyAsync(wdev, weights, h->B * sizeof(double), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { hipFree(wdev); return fail(RYD_ERR_HIP, "weights upload: %s", hipGetErrorString(e)); }
  }
  const size_t DD = (size_t)1 << (2 * h->N);
  hipLaunchKernelGGL(k_outer_acc, dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, st,
                     (const cplx*)psi_dev, h->N, h->B, wdev, (cplx*)acc_dev);
  hipError_t e = hipGetLastError();
  if (wdev) { hipStreamSynchronize(st); hipFree(wdev); }
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "k_outer_acc: %s", hipGetErrorString(e));
  return RYD_OK;
}

extern "C" int ryd_get_stats(const ryd_handle* h, ryd_stats* out) {
  if (!h || !out) return fail(RYD_ERR_INVALID, "null argument");
  *out = h->stats;
  if (hermitian_path(h)) out->passes = 2;  // row pass + symmetrisation
  return RYD_OK;
}

extern "C" int ryd_reset_stats(ryd_handle* h) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  const int passes = h->stats.passes;
  std::memset(&h->stats, 0, sizeof h->stats);
  h->stats.passes = passes;
  return RYD_OK;
}

extern "C" int ryd_set_kernel_timing(ryd_handle* h, int32_t enable) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  h->timing = enable != 0;
  if (enable) { h->timing_ms = 0; h->timing_launches = 0; }
  return RYD_OK;
}

extern "C" int ryd_get_kernel_timing(ryd_handle* h, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  for (auto& ev : h->ev_used) {
    HIPCHK(hipEventSynchronize(ev.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
    h->timing_ms += ms;
    h->timing_launches++;
    h->ev_free.push_back(ev);
  }
  h->ev_used.clear();
  *total_ms = h->timing_ms;
  *launches = h->timing_launches;
  return RYD_OK;
}
