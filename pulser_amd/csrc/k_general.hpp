// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// General path: G(t) = sum_t coef_t(t) A_t (any local dimension).  One thread per (row, batch
// entry).  A term is an explicit CSR matrix, or MATRIX-FREE:
//   local:    A = sum_g w_g embed(M on the digits with strides s_g[0..n_per))   (d x d or d^2 x d^2 M given
//             by its non-zeros; the embedding is never formed: digit decode + at most nnz gathers per group)
//   diagonal: A = diag(v)                                          (interaction, detuning projectors)
// which is what every operator of the reference's Hamiltonian and collapse lists is (hamiltonian.py:97-124,
// 246-439): sums of one- and two-site operators.
// ---------------------------------------------------------------------------
struct GenTermDev {
  const int* row_ptr;   // CSR (kind 0)
  const int* col;
  const cplx* val;      // CSR values | local: [nnz] values sorted by row of M | diagonal: [dim]
  int kind;             // 0 CSR, 1 local, 2 diagonal
  int d, n_per, n_groups, nnz;
  const long long* strides;  // [n_groups][n_per]
  const int* shifts;         // [n_groups][n_per]: bit position of the digit in the packed digits of a row
  const double* weights;     // [n_groups]
  const int* rstart;         // [d^n_per + 1]: entries of row R of M are rstart[R] .. rstart[R + 1]
  const int* ecol;           // [nnz]: their columns
};

// The base-d digits of a row, packed (2 bits per digit for d <= 4, else 4): digit p (0 = most
// significant of n_dig) sits at bit (n_dig - 1 - p) * bits.  Done once per row - in the one-launch
// kernel once per kernel.
__device__ __forceinline__ unsigned long long gen_pack_digits(long long row, int d, int n_dig) {
  unsigned r = (unsigned)row;  // dim <= 2^26
  const int bits = d <= 4 ? 2 : 4;
  unsigned long long packed = 0;
  for (int p = 0; p < n_dig; ++p) {  // least significant digit first
    unsigned q, a;
    switch (d) {
      case 2: q = r >> 1; a = r & 1u; break;
      case 3: q = r / 3u; a = r - 3u * q; break;
      case 4: q = r >> 2; a = r & 3u; break;
      default: q = r / (unsigned)d; a = r - (unsigned)d * q; break;
    }
    packed |= (unsigned long long)a << (p * bits);
    r = q;
  }
  return packed;
}

// (A_t x)[row]
__device__ __forceinline__ cplx gen_term_row(const GenTermDev& T, const cplx* __restrict__ x, long long row,
                                             unsigned long long digits) {
  cplx s = make_double2(0.0, 0.0);
  if (T.kind == 0) {
    const int lo = T.row_ptr[row], hi = T.row_ptr[row + 1];
    for (int e = lo; e < hi; ++e) s = cfma(T.val[e], x[T.col[e]], s);
    return s;
  }
  if (T.kind == 2) return cmul(T.val[row], x[row]);
  const int d = T.d;
  const unsigned mask = d <= 4 ? 3u : 15u;
  for (int g = 0; g < T.n_groups; ++g) {
    const int a = (int)((digits >> T.shifts[g * T.n_per]) & mask);
    int b = 0, R = a;
    long long s0 = T.strides[(size_t)g * T.n_per], s1 = 0;
    if (T.n_per == 2) {
      b = (int)((digits >> T.shifts[g * 2 + 1]) & mask);
      s1 = T.strides[(size_t)g * 2 + 1];
      R = a * d + b;
    }
    const int lo = T.rstart[R], hi = T.rstart[R + 1];
    if (lo == hi) continue;
    cplx sg = make_double2(0.0, 0.0);
    for (int e = lo; e < hi; ++e) {
      const int Cc = T.ecol[e];
      long long j;
      if (T.n_per == 2) {
        const int c0 = Cc / d;
        j = row + (long long)(c0 - a) * s0 + (long long)(Cc - c0 * d - b) * s1;
      } else {
        j = row + (long long)(Cc - a) * s0;
      }
      sg = cfma(T.val[e], x[j], sg);
    }
    const double w = T.weights[g];
    s.x = fma(w, sg.x, s.x);
    s.y = fma(w, sg.y, s.y);
  }
  return s;
}

// ---------------------------------------------------------------------------
// Site-fused application (round 3): every matrix-free term is a sum over SITES (one digit, or a digit pair
// such as (row_k, col_k) of a Liouvillian or the two atoms of an exchange term) of a small matrix, so all terms
// that act on a site are added up ONCE per exponential into one matrix per site,
//     M_s = sum_{t, g on s} coef_t(t) w_g M_t      (k_gen_sitevals, a few hundred entries),
// and a row then costs  sum_s nnz(M_s row)  independent gathers instead of a loop over terms x groups with
// a descriptor load each (3-level register of 9 atoms, 19 683 amplitudes: 38 -> ~9 us per application).
// ---------------------------------------------------------------------------
struct GenSite {
  long long s0, s1;    // strides of the digit(s)
  int shift0, shift1;  // bit positions in the packed digits of a row
  int n_per, ld;       // digits, local dimension (d or d^2)
  int rs_off;          // this site's row starts: pat_rstart[rs_off .. rs_off + ld]
};

struct GenSitesDev {
  const GenSite* sites;
  const int* pat_rstart;      // per site: ld + 1 entries (offsets into pat_col / mvals)
  const int* pat_col;         // [P] local column of every pattern entry
  const int* contrib_start;   // [P + 1]
  const int* contrib_term;    // [K]
  const cplx* contrib_val;    // [K] weight x matrix entry
  cplx* mvals;                // [P] site-matrix entries of the current exponential
  int n_sites, P, n_rs;
};

__global__ void k_gen_sitevals(const GenSitesDev S, const cplx* __restrict__ tcoef) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S.P) return;
  cplx m = make_double2(0.0, 0.0);
  for (int k = S.contrib_start[p]; k < S.contrib_start[p + 1]; ++k) m = cfma(tcoef[S.contrib_term[k]], S.contrib_val[k], m);
  S.mvals[p] = m;
}

#define GEN_SITES_LDS_MAX 1536  /* pattern entries staged in LDS (24 KiB of values) */
struct GenSiteArgs {
  const cplx* in;
  const cplx* base;
  cplx* out;
  const cplx* tcoef;
  const GenTermDev* terms;   // only the diagonal (kind 2) terms are read here
  const int* diag_terms;     // [n_diag] indices into terms / tcoef
  GenSitesDev S;
  long long dim;
  int n_diag, d, n_dig;
  double scale;
};

__global__ __launch_bounds__(256) void k_gen_apply_sites(const GenSiteArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* mv = reinterpret_cast<cplx*>(smem);                                   // [P]
  int* pcol = reinterpret_cast<int*>(mv + A.S.P);                             // [P]
  int* prs = pcol + A.S.P;                                                    // [n_rs]
  GenSite* sites = reinterpret_cast<GenSite*>(prs + ((A.S.n_rs + 3) & ~3));   // [n_sites]
  for (int i = threadIdx.x; i < A.S.P; i += blockDim.x) { mv[i] = A.S.mvals[i]; pcol[i] = A.S.pat_col[i]; }
  for (int i = threadIdx.x; i < A.S.n_rs; i += blockDim.x) prs[i] = A.S.pat_rstart[i];
  for (int i = threadIdx.x; i < A.S.n_sites; i += blockDim.x) sites[i] = A.S.sites[i];
  __syncthreads();
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= A.dim) return;
  const size_t boff = (size_t)blockIdx.y * A.dim;
  const cplx* __restrict__ x = A.in + boff;
  const unsigned long long digits = gen_pack_digits(row, A.d, A.n_dig);
  const unsigned mask = A.d <= 4 ? 3u : 15u;
  const cplx xr = x[row];
  cplx dsum = make_double2(0.0, 0.0);
  for (int k = 0; k < A.n_diag; ++k) {
    const int t = A.diag_terms[k];
    dsum = cfma(A.tcoef[t], A.terms[t].val[row], dsum);
  }
  cplx acc = cmul(dsum, xr);
  for (int s = 0; s < A.S.n_sites; ++s) {
    const GenSite si = sites[s];
    const int a = (int)((digits >> si.shift0) & mask);
    int b = 0, R = a;
    if (si.n_per == 2) {
      b = (int)((digits >> si.shift1) & mask);
      R = a * A.d + b;
    }
    const int lo = prs[si.rs_off + R], hi = prs[si.rs_off + R + 1];
    for (int e = lo; e < hi; ++e) {
      const int Cc = pcol[e];
      long long j;
      if (si.n_per == 2) {
        const int c0 = Cc / A.d;
        j = row + (long long)(c0 - a) * si.s0 + (long long)(Cc - c0 * A.d - b) * si.s1;
      } else {
        j = row + (long long)(Cc - a) * si.s0;
      }
      acc = cfma(mv[e], x[j], acc);
    }
  }
  cplx r = make_double2(A.scale * acc.x, A.scale * acc.y);
  if (A.base) {
    const cplx bb = A.base[boff + row];
    r.x += bb.x;
    r.y += bb.y;
  }
  A.out[boff + row] = r;
}

// ---------------------------------------------------------------------------
// Padded site tables (round 6).  k_gen_apply_sites above chases four dependent LDS loads per site and a data-dependent
// entry loop, one wave per SIMD: 9 us per application on 3^9 amplitudes, 35 us on the 66 exchange pairs of a 12-atom XY
// register - pure latency.  Here every site carries, per local row R, its DIAGONAL entry and exactly K off-diagonal
// entries (K = the site's fullest row; short rows are padded with (value 0, offset 0)), stored as (value, row offset):
//     acc += M_s[R][k] * x[row + delta_s[R][k]],  k < K        dsum += Mdiag_s[R]
// No branch, no pattern walk: the gathers of a site are independent of each other and of the next sites', the site loop
// is unrolled and sites of equal K are contiguous (K is a template argument of the loop).  Vectors of up to ~6 000
// amplitudes (XY on 12 atoms, 3-level on 7, 4-level on 6) are staged whole in LDS by every workgroup and gathered from
// there (XLDS); larger ones gather from L2.  Coefficients and site matrices of an exponential come from ONE small launch
// (k_gen_coefs_fused) instead of two.
// ---------------------------------------------------------------------------
struct GenSiteF {
  int shift0, shift1;  // bit positions of the digit(s) in the packed digits of a row
  int mask1, mul;      // pair sites: digit mask and d; single sites: 0 and 1  (R = a * mul + (b & mask1))
  int ent_off;         // first padded entry of local row 0 (row R: ent_off + R * K)
  int diag_off;        // diagonal entry of local row 0
};

#define GEN_FUSED_MAX_GROUPS 8
struct GenFusedDev {
  const GenSiteF* sites;
  const int* delta;           // [E] row offset of every padded off-diagonal entry
  const int* contrib_start;   // [E + Dg + 1]: contributions of every entry (off-diagonal entries first, then diagonal ones)
  const int* contrib_term;
  const cplx* contrib_val;
  cplx* mvals;                // [E + Dg] entries of the current exponential
  int n_sites, E, Dg, n_groups;
  int gK[GEN_FUSED_MAX_GROUPS], gBegin[GEN_FUSED_MAX_GROUPS], gEnd[GEN_FUSED_MAX_GROUPS];
};

// tcoef[t] (as k_gen_coefs) and then the site-matrix entries, one workgroup
__global__ __launch_bounds__(256) void k_gen_coefs_fused(const cplx* __restrict__ pp, int n_int, const int* __restrict__ series,
                                                         const int* __restrict__ conjf, const cplx* __restrict__ scale,
                                                         int n_terms, int idx, double u1, double w1, double u2, double w2,
                                                         cplx* __restrict__ tcoef, const GenFusedDev F) {
  __shared__ double tc_raw[2 * 96];  // MAX_GEN_TERMS coefficients
  cplx* tc = reinterpret_cast<cplx*>(tc_raw);
  for (int t = threadIdx.x; t < n_terms; t += blockDim.x) {
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
    v = cmul(scale[t], v);
    tc[t] = v;
    tcoef[t] = v;
  }
  __syncthreads();
  const int P = F.E + F.Dg;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    cplx m = make_double2(0.0, 0.0);
    for (int k = F.contrib_start[p]; k < F.contrib_start[p + 1]; ++k) m = cfma(tc[F.contrib_term[k]], F.contrib_val[k], m);
    F.mvals[p] = m;
  }
}

struct GenFusedArgs {
  const cplx* in;
  const cplx* base;
  cplx* out;
  const cplx* tcoef;
  const GenTermDev* terms;   // only the diagonal (kind 2) terms are read here
  const int* diag_terms;     // [n_diag] indices into terms / tcoef
  const cplx* diag_val[4];   // the first four diagonal terms' vectors and term indices inline (no descriptor load on the way)
  int diag_idx[4];
  GenFusedDev F;
  long long dim;
  int n_diag, d, n_dig;
  double scale;
};

// One workgroup = 64 rows x 4 waves: wave w takes the sites s0 + w, s0 + w + 4, ... of every group for the SAME 64 rows and
// the four partial sums meet in LDS.  (First version, one row per thread and all sites per row: the 78 sites of a 12-atom
// XY register read 5.9 KB of LDS per row - 16 workgroups, LDS-bandwidth-bound at 10 - 14 us; a 3-level register of 9
// atoms ran 9 sites in series on 77 CUs.)  Site descriptors are wave-uniform: scalar loads from global memory, not LDS.
template <int K>
__device__ __forceinline__ void gen_fused_sites(const GenSiteF* __restrict__ sites, int s0, int s1, int wave,
                                                const cplx* __restrict__ mv, const cplx* __restrict__ mvd,
                                                const int* __restrict__ dl, const cplx* __restrict__ x, int row,
                                                unsigned long long digits, unsigned mask, int kdyn, cplx& acc, cplx& dsum) {
  if constexpr (K > 0) {
    constexpr int UNR = K <= 2 ? 4 : 2;  // sites in flight per wave (x 4 waves)
    for (int sb = s0 + wave; sb < s1; sb += 4 * UNR) {
      cplx mvv[UNR * K], xv[UNR * K];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool on = sb + 4 * u < s1;
        const int s = on ? sb + 4 * u : sb;
        const GenSiteF si = sites[s];  // (s is wave-uniform: scalar loads)
        const int R = (int)((digits >> si.shift0) & mask) * si.mul + (int)((digits >> si.shift1) & (unsigned)si.mask1);
        const cplx dg = mvd[si.diag_off + R];
        dsum.x += on ? dg.x : 0.0;
        dsum.y += on ? dg.y : 0.0;
        const int e0 = si.ent_off + R * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const cplx m = mv[e0 + k];
          mvv[u * K + k] = on ? m : make_double2(0.0, 0.0);
          xv[u * K + k] = x[row + dl[e0 + k]];
        }
      }
#pragma unroll
      for (int j = 0; j < UNR * K; ++j) acc = cfma(mvv[j], xv[j], acc);
    }
  } else {  // (K > 4: a dense two-digit superoperator)
    for (int s = s0 + wave; s < s1; s += 4) {
      const GenSiteF si = sites[s];
      const int R = (int)((digits >> si.shift0) & mask) * si.mul + (int)((digits >> si.shift1) & (unsigned)si.mask1);
      const cplx dg = mvd[si.diag_off + R];
      dsum.x += dg.x;
      dsum.y += dg.y;
      const int e0 = si.ent_off + R * kdyn;
#pragma unroll 4
      for (int k = 0; k < kdyn; ++k) acc = cfma(mv[e0 + k], x[row + dl[e0 + k]], acc);
    }
  }
}

#define GEN_FUSED_ROWS 64
template <bool XLDS>
__global__ __launch_bounds__(256) void k_gen_apply_fused(const GenFusedArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int E = A.F.E, Dg = A.F.Dg;
  cplx* mv = reinterpret_cast<cplx*>(smem);                          // [E] off-diagonal, [Dg] diagonal
  cplx* mvd = mv + E;
  cplx* red = mvd + Dg;                                              // [4][64][2]: partial sums of the waves
  int* dl = reinterpret_cast<int*>(red + 4 * GEN_FUSED_ROWS * 2);    // [E]
  cplx* xs = reinterpret_cast<cplx*>(dl + ((E + 3) & ~3));           // [dim] (XLDS)
  const size_t boff = (size_t)blockIdx.y * A.dim;
  const cplx* __restrict__ xg = A.in + boff;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long row = (long long)blockIdx.x * GEN_FUSED_ROWS + lane;
  const bool live = row < A.dim;
  // wave 0 owns the row's own amplitude, the dense diagonal terms and the result: its loads go out first and are in
  // flight while the tables are staged
  cplx xr = make_double2(0.0, 0.0), dsum = make_double2(0.0, 0.0);
  if (live && wave == 0) {
    xr = xg[row];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < A.n_diag) dsum = cfma(A.tcoef[A.diag_idx[k]], A.diag_val[k][row], dsum);
    for (int k = 4; k < A.n_diag; ++k) {
      const int t = A.diag_terms[k];
      dsum = cfma(A.tcoef[t], A.terms[t].val[row], dsum);
    }
  }
  for (int i = threadIdx.x; i < E + Dg; i += 256) mv[i] = A.F.mvals[i];
  for (int i = threadIdx.x; i < E; i += 256) dl[i] = A.F.delta[i];
  if (XLDS) {
    // the whole vector into LDS, eight independent 16-byte loads per lane in flight
    const int n = (int)A.dim;
    for (int b0 = threadIdx.x; b0 < n; b0 += 256 * 8) {
      cplx r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b0 + u * 256;
        r[u] = i < n ? xg[i] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b0 + u * 256;
        if (i < n) xs[i] = r[u];
      }
    }
  }
  __syncthreads();
  cplx acc = make_double2(0.0, 0.0);
  if (live) {
    const cplx* __restrict__ x = XLDS ? xs : xg;
    const unsigned long long digits = gen_pack_digits(row, A.d, A.n_dig);
    const unsigned mask = A.d <= 4 ? 3u : 15u;
    const GenSiteF* __restrict__ sites = A.F.sites;
    for (int g = 0; g < A.F.n_groups; ++g) {
      const int K = A.F.gK[g], s0 = A.F.gBegin[g], s1 = A.F.gEnd[g];
      switch (K) {
        case 0: for (int s = s0 + wave; s < s1; s += 4) {  // diagonal-only sites
                  const GenSiteF si = sites[s];
                  const int R = (int)((digits >> si.shift0) & mask) * si.mul + (int)((digits >> si.shift1) & (unsigned)si.mask1);
                  const cplx dg = mvd[si.diag_off + R];
                  dsum.x += dg.x; dsum.y += dg.y;
                } break;
        case 1: gen_fused_sites<1>(sites, s0, s1, wave, mv, mvd, dl, x, (int)row, digits, mask, K, acc, dsum); break;
        case 2: gen_fused_sites<2>(sites, s0, s1, wave, mv, mvd, dl, x, (int)row, digits, mask, K, acc, dsum); break;
        case 3: gen_fused_sites<3>(sites, s0, s1, wave, mv, mvd, dl, x, (int)row, digits, mask, K, acc, dsum); break;
        case 4: gen_fused_sites<4>(sites, s0, s1, wave, mv, mvd, dl, x, (int)row, digits, mask, K, acc, dsum); break;
        default: gen_fused_sites<0>(sites, s0, s1, wave, mv, mvd, dl, x, (int)row, digits, mask, K, acc, dsum); break;
      }
    }
  }
  if (wave != 0) {
    red[(wave * GEN_FUSED_ROWS + lane) * 2] = acc;
    red[(wave * GEN_FUSED_ROWS + lane) * 2 + 1] = dsum;
  }
  __syncthreads();
  if (wave != 0 || !live) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const cplx a = red[(w * GEN_FUSED_ROWS + lane) * 2], dd = red[(w * GEN_FUSED_ROWS + lane) * 2 + 1];
    acc.x += a.x; acc.y += a.y;
    dsum.x += dd.x; dsum.y += dd.y;
  }
  acc = cfma(dsum, xr, acc);
  cplx r = make_double2(A.scale * acc.x, A.scale * acc.y);
  if (A.base) {
    const cplx bb = A.base[boff + row];
    r.x += bb.x;
    r.y += bb.y;
  }
  A.out[boff + row] = r;
}

#define MAX_GEN_TERMS 96

struct GenArgs {
  const cplx* in;
  const cplx* base;
  cplx* out;
  const cplx* tcoef;  // [n_terms] time-mixed coefficients
  const GenTermDev* terms;
  long long dim;
  int n_terms;
  int d, n_dig;  // local dimension and digits of the vector index (0 when every term is CSR)
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
  const unsigned long long digits = A.d ? gen_pack_digits(row, A.d, A.n_dig) : 0ull;
  for (int t = 0; t < A.n_terms; ++t) {
    const GenTermDev T = A.terms[t];
    acc = cfma(A.tcoef[t], gen_term_row(T, x, row, digits), acc);
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
// k_gen_traj: the whole schedule of a small general-path system (dim <= 4096:
// 3- / 4-level registers, XY mode, exotic collapse operators on a few atoms) in
// ONE launch - the vector lives in registers (rows row = tid + j * 1024) and in
// two LDS copies for the gathers x[col]; the CSR terms stream from L2 every
// stage.  Same arithmetic as k_gen_coefs + k_gen_apply (CF4 steps, Horner Taylor).
// ---------------------------------------------------------------------------
struct GenTrajArgs {
  cplx* state;   // [dim] in/out
  cplx* snaps;   // [n_slots][dim] or null
  const cplx* pp;
  const int* series;
  const int* conjf;
  const cplx* scale;
  const GenTermDev* terms;
  const StepDesc* steps;
  int n_int, n_steps, n_terms, dim;
  int d, n_dig;
  double a1, a2;
};

__device__ __forceinline__ void gen_traj_body(const GenTrajArgs& A) {
  constexpr int NTT = 1024, R = 4;  // dim <= 4096
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* ws0 = reinterpret_cast<cplx*>(smem);
  cplx* ws1 = ws0 + 4096;
  cplx* tcA = ws1 + 4096;           // [MAX_GEN_TERMS] coefficients of exponential A
  cplx* tcB = tcA + MAX_GEN_TERMS;  // ... and B
  const int tid = threadIdx.x;
  const int dim = A.dim;

  cplx psi[R];
  unsigned long long digits[R];  // of this thread's rows: decoded once for the whole schedule
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int row = tid + j * NTT;
    psi[j] = row < dim ? A.state[row] : make_double2(0.0, 0.0);
    digits[j] = (A.d && row < dim) ? gen_pack_digits(row, A.d, A.n_dig) : 0ull;
  }
  for (int s = 0; s < A.n_steps; ++s) {
    const StepDesc sd = A.steps[s];
    if (tid < A.n_terms) {
      cplx va = make_double2(A.a1 + A.a2, 0.0), vb = va;
      const int ser = A.series[tid];
      if (ser >= 0) {
        auto val = [&](double u) -> cplx {
          const cplx* p = A.pp + ((size_t)ser * A.n_int + sd.idx) * 4;
          cplx r = p[0];
          r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
          r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
          r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
          return r;
        };
        const cplx a = val(sd.u1), b = val(sd.u2);
        va = make_double2(A.a1 * a.x + A.a2 * b.x, A.a1 * a.y + A.a2 * b.y);
        vb = make_double2(A.a2 * a.x + A.a1 * b.x, A.a2 * a.y + A.a1 * b.y);
        if (A.conjf[tid]) { va.y = -va.y; vb.y = -vb.y; }
      }
      tcA[tid] = cmul(A.scale[tid], va);
      tcB[tid] = cmul(A.scale[tid], vb);
    }
    __syncthreads();
#pragma unroll 1
    for (int ex = 0; ex < 2; ++ex) {
      const cplx* tc = ex ? tcB : tcA;
      const int order = ex ? sd.order_b : sd.order_a;
      cplx w[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        w[j] = psi[j];
        const int row = tid + j * NTT;
        if (row < dim) ws0[row] = w[j];
      }
      __syncthreads();
      const cplx* rd = ws0;
      cplx* wr = ws1;
      for (int jj = order; jj >= 1; --jj) {
        const double sc = sd.h * kInvInt[jj];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int row = tid + j * NTT;
          if (row >= dim) continue;
          cplx acc = make_double2(0.0, 0.0);
          for (int t = 0; t < A.n_terms; ++t) {
            const GenTermDev T = A.terms[t];
            acc = cfma(tc[t], gen_term_row(T, rd, row, digits[j]), acc);
          }
          w[j] = make_double2(fma(sc, acc.x, psi[j].x), fma(sc, acc.y, psi[j].y));
        }
        if (jj > 1) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int row = tid + j * NTT;
            if (row < dim) wr[row] = w[j];
          }
          __syncthreads();
          const cplx* t = rd;
          rd = wr;
          wr = const_cast<cplx*>(t);
        }
      }
#pragma unroll
      for (int j = 0; j < R; ++j) psi[j] = w[j];
      __syncthreads();
    }
    if (sd.snap >= 0 && A.snaps) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int row = tid + j * NTT;
        if (row < dim) A.snaps[(size_t)sd.snap * dim + row] = psi[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int row = tid + j * NTT;
    if (row < dim) A.state[row] = psi[j];
  }
}

__global__ __launch_bounds__(1024) void k_gen_traj(const GenTrajArgs A) { gen_traj_body(A); }

// A batch of INDEPENDENT small general-path problems in one launch: workgroup b runs the whole schedule of
// problem b from its own tables (the noise trajectories of a multi-level run, hamiltonian_data.py:913-931
// bases under simulation.py:903-915: bad atoms and detuning offsets change the term list per trajectory, so
// every trajectory brings its own terms - the systems are <= 4096 entries and their tables a few KB).
__global__ __launch_bounds__(1024) void k_gen_traj_many(const GenTrajArgs* __restrict__ args) {
  const GenTrajArgs A = args[blockIdx.x];
  gen_traj_body(A);
}
