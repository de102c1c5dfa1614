// Part of librydemu (included by rydemu.hip, one translation unit).
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
