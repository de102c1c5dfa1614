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
  double a1, a2;
};

__global__ __launch_bounds__(1024) void k_gen_traj(const GenTrajArgs A) {
  constexpr int NTT = 1024, R = 4;  // dim <= 4096
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* ws0 = reinterpret_cast<cplx*>(smem);
  cplx* ws1 = ws0 + 4096;
  cplx* tcA = ws1 + 4096;           // [MAX_GEN_TERMS] coefficients of exponential A
  cplx* tcB = tcA + MAX_GEN_TERMS;  // ... and B
  const int tid = threadIdx.x;
  const int dim = A.dim;

  cplx psi[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int row = tid + j * NTT;
    psi[j] = row < dim ? A.state[row] : make_double2(0.0, 0.0);
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
            const int lo = T.row_ptr[row], hi = T.row_ptr[row + 1];
            cplx sum = make_double2(0.0, 0.0);
            for (int e = lo; e < hi; ++e) sum = cfma(T.val[e], rd[T.col[e]], sum);
            acc = cfma(tc[t], sum, acc);
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
