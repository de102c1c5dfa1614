// Part of librydemu (included by rydemu.hip, one translation unit).
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

// value of `v` held by the lane selected by the DPP control (full row / bank masks)
template <int CTRL>
__device__ __forceinline__ cplx dpp_cplx(cplx v) {
  const int a = __builtin_amdgcn_mov_dpp(__double2loint(v.x), CTRL, 0xF, 0xF, true);
  const int b = __builtin_amdgcn_mov_dpp(__double2hiint(v.x), CTRL, 0xF, 0xF, true);
  const int c = __builtin_amdgcn_mov_dpp(__double2loint(v.y), CTRL, 0xF, 0xF, true);
  const int d = __builtin_amdgcn_mov_dpp(__double2hiint(v.y), CTRL, 0xF, 0xF, true);
  return make_double2(__hiloint2double(b, a), __hiloint2double(d, c));
}

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
        // partners inside a 16-lane row come over the DPP crossbar (xor 1, 2: quad
        // permutes; xor 8: row rotate by 8) and never touch LDS, the other lane
        // bits are ds_read_b128, tile bits 10-13 are register moves
        if (f == 0) pv[jj] = dpp_cplx<0xB1>(x[j]);
        else if (f == 1) pv[jj] = dpp_cplx<0x4E>(x[j]);
        else if (f == 3) pv[jj] = dpp_cplx<0x128>(x[j]);
        else if (f < LOGNT) pv[jj] = xs[jj * NTT + (tid ^ (1 << f))];
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
