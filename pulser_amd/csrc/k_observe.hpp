// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// ryd_observe: the V2 observables of one state in one call, on the device
// ---------------------------------------------------------------------------
// Occupations <n_k>, correlations <n_k n_l> (default_observables.py:291-435) and the energy
// moments <H>, <H^2> (:437-580, through one generator application w = -i H x:
// <H> = -Im <x|w>, <H^2> = |w|^2).  Replaces N(N+1)/2 + 2 qutip.expect calls and the per-time
// materialisation of H(t) (qutip_backend.py:259-264).

// out[b][0..N-1] = <n_k>, out[b][N] = sum p, out[b][N+1 + k*N + l] = <n_k n_l>.
// One block stages a chunk of probabilities in LDS; thread <-> (k, l) pair (several per thread when
// N(N+1)/2 > blockDim); wave-uniform chunk index -> LDS broadcast reads.
__global__ __launch_bounds__(256) void k_obs_pairs(const cplx* __restrict__ st, int N, int is_dm,
                                                   double* __restrict__ out, int out_stride) {
  constexpr int CH = 2048;
  __shared__ double ps[CH];
  const size_t D = (size_t)1 << N;
  const int b = blockIdx.y;
  const int npair = N * (N + 1) / 2;
  const size_t base = (size_t)blockIdx.x * CH;
  for (int i = threadIdx.x; i < CH; i += blockDim.x) {
    const size_t g = base + i;
    double p = 0.0;
    if (g < D) {
      if (is_dm) p = st[((size_t)b << (2 * N)) + g * D + g].x;
      else { const cplx v = st[((size_t)b << N) + g]; p = v.x * v.x + v.y * v.y; }
    }
    ps[i] = p;
  }
  __syncthreads();
  double* o = out + (size_t)b * out_stride;
  for (int pr = threadIdx.x; pr <= npair; pr += blockDim.x) {
    if (pr == npair) {  // the norm
      double s = 0.0;
      for (int i = 0; i < CH; ++i) s += ps[i];
      atomicAdd(o + N, s);
      continue;
    }
    // pair index -> (k <= l)
    int k = 0, rem = pr;
    while (rem >= N - k) { rem -= N - k; ++k; }
    const int l = k + rem;
    const unsigned mk = 1u << (N - 1 - k), ml = 1u << (N - 1 - l);
    double s = 0.0;
    for (int i = 0; i < CH; ++i) {
      const unsigned g = (unsigned)(base + i);
      if (!(g & mk) && !(g & ml)) s += ps[i];  // n = 1 <=> bit 0 (local state 0 = r)
    }
    if (k == l) atomicAdd(o + k, s);
    atomicAdd(o + N + 1 + k * N + l, s);
    if (k != l) atomicAdd(o + N + 1 + l * N + k, s);
  }
}

// o[0] += -Im <x|w>, o[1] += |w|^2
__global__ __launch_bounds__(256) void k_obs_energy(const cplx* __restrict__ x, const cplx* __restrict__ w, int nb,
                                                    double* __restrict__ out, int out_stride, int off) {
  const size_t D = (size_t)1 << nb;
  const size_t boff = (size_t)blockIdx.y * D;
  double e = 0.0, e2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx a = x[boff + i], c = w[boff + i];
    e -= a.x * c.y - a.y * c.x;
    e2 = fma(c.x, c.x, fma(c.y, c.y, e2));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o, 64); e2 += __shfl_down(e2, o, 64); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out + (size_t)blockIdx.y * out_stride + off, e);
    atomicAdd(out + (size_t)blockIdx.y * out_stride + off + 1, e2);
  }
}

// Density matrices: <H> = Tr(H rho) and <H^2> = Tr(H^2 rho) only touch the matrix elements of H
// and H^2 around the diagonal: with h_k(a) = H_{a, a^k} = (bit_k(a) ? c_k : conj c_k),
//   (H)_{aa} = E(a);            (H^2)_{aa}      = E(a)^2 + sum_k |c_k|^2
//   (H)_{a,a^k} = h_k(a);       (H^2)_{a,a^k}   = h_k(a) (E(a) + E(a^k))
//                               (H^2)_{a,a^k^l} = 2 h_k(a) h_l(a)      (k != l)
// so one thread per row index a gathers 1 + N + N(N-1)/2 elements of column a.  No H(t) is built
// (qutip_backend.py:259-264 materialises it) and rho is not multiplied by anything dense.
__global__ __launch_bounds__(256) void k_obs_energy_dm(const cplx* __restrict__ rho, int N,
                                                       const double* __restrict__ coefs,
                                                       const double* __restrict__ e0, long long e0_stride,
                                                       double* __restrict__ out, int out_stride, int off) {
  const size_t D = (size_t)1 << N;
  const int b = blockIdx.y;
  const cplx* r = rho + ((size_t)b << (2 * N));
  const double* cf = coefs + (size_t)b * N * 4;
  const double* e0b = e0 + (size_t)b * e0_stride;
  double e1 = 0.0, e2 = 0.0;
  for (size_t a = (size_t)blockIdx.x * 256 + threadIdx.x; a < D; a += (size_t)gridDim.x * 256) {
    auto energy = [&](size_t s) {
      double e = e0b[s];
      for (int k = 0; k < N; ++k)
        if (!((s >> (N - 1 - k)) & 1)) e -= cf[4 * k + 2];
      return e;
    };
    const double Ea = energy(a);
    double c2 = 0.0;
    for (int k = 0; k < N; ++k) c2 += cf[4 * k] * cf[4 * k] + cf[4 * k + 1] * cf[4 * k + 1];
    const double raa = r[a * D + a].x;
    e1 += Ea * raa;
    e2 += (Ea * Ea + c2) * raa;
    for (int k = 0; k < N; ++k) {
      const size_t ak = a ^ ((size_t)1 << (N - 1 - k));
      const bool bk = (a >> (N - 1 - k)) & 1;
      const cplx hk = make_double2(cf[4 * k], bk ? cf[4 * k + 1] : -cf[4 * k + 1]);
      const cplx x = r[ak * D + a];  // rho_{a^k, a}
      const double re = hk.x * x.x - hk.y * x.y;  // Re(h_k rho_{a^k,a}); the imaginary parts cancel in the trace
      e1 += re;
      e2 += re * (Ea + energy(ak));
      for (int l = k + 1; l < N; ++l) {
        const bool bl = (a >> (N - 1 - l)) & 1;
        const cplx hl = make_double2(cf[4 * l], bl ? cf[4 * l + 1] : -cf[4 * l + 1]);
        const cplx hh = cmul(hk, hl);
        const cplx y = r[(ak ^ ((size_t)1 << (N - 1 - l))) * D + a];
        e2 += 2.0 * (hh.x * y.x - hh.y * y.y);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { e1 += __shfl_down(e1, o, 64); e2 += __shfl_down(e2, o, 64); }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out + (size_t)b * out_stride + off, e1);
    atomicAdd(out + (size_t)b * out_stride + off + 1, e2);
  }
}
