// Part of librydemu (included by rydemu.hip, one translation unit).
#define MAXF 16  // flips per pass (<= tile bits)
#define MAXD 8   // double flips per pass
#define MAXO 10  // outer flips per pass (partner tiles streamed from global memory)

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
  // Lanczos (round 6, host_krylov.hpp): with `kry_acc` set the final pass also reduces <x | out> (x = the vector G is applied to:
  // the tile is in LDS) and |out|^2 into kry_acc[(4 b + 0 .. 2) 8 + (block & 7)] - what k_kry_dot and the norm pass of k_kry_update computed from
  // two more reads of both vectors (stride 4 per batch entry)
  double* kry_acc;
  signed char flip_q[MAXF];  // tile-local bit of each single flip
  signed char dbl_qb[MAXD], dbl_qa[MAXD];
  int n_oflip;                // single flips on bits outside the tile: the partner
  signed char oflip_p[MAXO];  // amplitude sits at the same place of another tile
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
// PRE (single-launch plan with at most one amplitude per thread): the partner
// amplitudes of the outer bits are fetched into registers before the barrier, so
// their global-memory latency overlaps the LDS staging instead of following it.
template <int MODE, int NT, bool PRE = false>
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
  cplx* oc = c1 + MAXF;  // outer flips: one coefficient each (the bit is fixed per tile)

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

  cplx pre[PRE ? MAXO : 1];
  if (PRE && tid < tileSize) {
    const unsigned long long g0 = base_idx | deposit((unsigned long long)tid, A.tile);
#pragma unroll
    for (int f = 0; f < MAXO; ++f)
      if (f < A.n_oflip) pre[f] = xin[g0 ^ (1ull << A.oflip_p[f])];
  }

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
  if (tid >= 64 && tid - 64 < A.n_oflip) {
    // same coefficients as above for a global bit p of the output index `base_idx`
    const int p = A.oflip_p[tid - 64];
    const bool bit = (base_idx >> p) & 1ull;
    cplx v;
    if (MODE == RYD_SESOLVE || p >= N) {
      const int k = (MODE == RYD_SESOLVE ? N : 2 * N) - 1 - p;
      const double cr = cf[4 * k], ci = cf[4 * k + 1];
      v = bit ? make_double2(ci, -cr) : make_double2(-ci, -cr);
    } else {
      const int k = N - 1 - p;
      const double cr = cf[4 * k], ci = cf[4 * k + 1];
      v = bit ? make_double2(ci, cr) : make_double2(-ci, cr);
    }
    oc[tid - 64] = v;
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

  double kd_re = 0.0, kd_im = 0.0, kd_nn = 0.0;  // Lanczos reductions (PassArgs.kry_acc)
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
    if (PRE) {
#pragma unroll
      for (int f = 0; f < MAXO; ++f)
        if (f < A.n_oflip) acc = cfma(oc[f], pre[f], acc);
    } else {
      for (int f = 0; f < A.n_oflip; ++f)  // coalesced: the partner tile has the same layout
        acc = cfma(oc[f], xin[gi ^ (1ull << A.oflip_p[f])], acc);
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
      const cplx o = cmul(A.post, r);
      A.out[go] = o;
      if (A.kry_acc) {
        const cplx x = xs[l];
        kd_re = fma(x.x, o.x, fma(x.y, o.y, kd_re));   // <x | o>, conjugate-linear in x
        kd_im = fma(x.x, o.y, fma(-x.y, o.x, kd_im));
        kd_nn = fma(o.x, o.x, fma(o.y, o.y, kd_nn));
      }
    } else {
      A.kout[go] = acc;
    }
  }
  if (A.kry_acc && A.final_pass) {
    // wave sums, one LDS slot per wave (the coefficient tables c0 / c1 / oc are dead: every thread is past its last read of
    // them only after the barrier), one atomic per workgroup and accumulator
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kd_re += __shfl_down(kd_re, o, 64);
      kd_im += __shfl_down(kd_im, o, 64);
      kd_nn += __shfl_down(kd_nn, o, 64);
    }
    __syncthreads();
    double* red = reinterpret_cast<double*>(xs);  // (the tile is dead too; NT / 64 <= 16 waves x 3 doubles)
    if ((tid & 63) == 0) {
      red[3 * (tid >> 6)] = kd_re;
      red[3 * (tid >> 6) + 1] = kd_im;
      red[3 * (tid >> 6) + 2] = kd_nn;
    }
    __syncthreads();
    if (tid < 3) {
      double sum = 0.0;
      for (int wv = 0; wv < NT / 64; ++wv) sum += red[3 * wv + tid];
      atomicAdd(A.kry_acc + (4 * b + tid) * 8 + (blockIdx.x & 7), sum);  // (8 sub-accumulators: atomics on one word serialise at ~12 ns)
    }
  }
}
