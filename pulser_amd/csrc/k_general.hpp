// Part of librydemu (included by rydemu.hip, one translation unit).
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
