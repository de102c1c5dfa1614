// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Lanczos (Krylov-subspace) exponential for kets: the "batched zgemv" kernels
// ---------------------------------------------------------------------------
// BASELINE configs[4] names a Krylov-subspace sesolve with batched zgemv; the
// reference itself never calls one, so this is the north-star's
// alternative to the Taylor polynomial behind the same seam (simulation.py:729-735).
// One exponential  psi <- exp(h G~) psi,  G~ = -i (H~ - sigma),  H~ Hermitian:
//   v_0 = psi / |psi|;  for j < m:  w = G~ v_j  (the generator kernels),
//   alpha_j = <v_j| i w>,  u = i w - alpha_j v_j - beta_{j-1} v_{j-1},  beta_j = |u|,
//   v_{j+1} = u / beta_j;   psi' = |psi| e^{-i h sigma} V exp(-i h T_m) e_1.
// The inner products (V^H w) and the combination (V c) are the zgemv-shaped parts:
// wave64 __shfl_down reductions + one atomic per block, coefficients read from
// device memory so that a whole exponential needs no host synchronisation.

#define KRY_MAX_M 40

struct KryScalars {
  double* alpha;   // [B][KRY_MAX_M]
  double* beta;    // [B][KRY_MAX_M]
  double* dotre;   // [B] scratch accumulators (zeroed by the consumer)
  double* dotim;   // [B]
  double* nrm2;    // [B]
  double* norm0;   // [B] |psi| of the exponential
  cplx* coef;      // [B][KRY_MAX_M] combination coefficients
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Sum over the workgroup (256 lanes = 4 waves), valid on lane 0.  Round 5: one atomic per WORKGROUP.  With one per wave the
// 4 096 waves of a 20-atom reduction queued on two addresses (fp64 atomics to one word serialise at ~12 ns each): k_kry_dot
// took 103 us and k_kry_update 57 us on 16-MiB vectors - three times the generator application they bracket.
__device__ __forceinline__ double kry_block_sum(double v, double* lds4) {
  v = wave_sum(v);
  const unsigned w = threadIdx.x >> 6;
  __syncthreads();  // (the scratch may still be read from a previous call)
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// nrm2[b] += sum |x|^2
__global__ __launch_bounds__(256) void k_kry_norm(const cplx* __restrict__ x, int nb, double* nrm2) {
  const size_t D = (size_t)1 << nb;
  const cplx* xb = x + (size_t)blockIdx.y * D;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = xb[i];
    s = fma(v.x, v.x, fma(v.y, v.y, s));
  }
  __shared__ double red[4];
  s = kry_block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(nrm2 + blockIdx.y, s);
}

// out = x / sqrt(nrm2[b]); norm0[b] = sqrt(nrm2[b]) when store_norm (read before the reset below)
__global__ __launch_bounds__(256) void k_kry_scale(const cplx* __restrict__ x, cplx* __restrict__ out, int nb,
                                                   const double* nrm2, double* norm0, int store_norm) {
  const size_t D = (size_t)1 << nb;
  const size_t boff = (size_t)blockIdx.y * D;
  const double n2 = nrm2[blockIdx.y];
  const double inv = n2 > 1e-300 ? rsqrt(n2) : 0.0;
  if (store_norm && blockIdx.x == 0 && threadIdx.x == 0) norm0[blockIdx.y] = sqrt(n2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = x[boff + i];
    out[boff + i] = make_double2(v.x * inv, v.y * inv);
  }
}

// dot[b] += <v | w>  (conjugate-linear in v)
__global__ __launch_bounds__(256) void k_kry_dot(const cplx* __restrict__ v, const cplx* __restrict__ w, int nb,
                                                 double* dre, double* dim) {
  const size_t D = (size_t)1 << nb;
  const size_t boff = (size_t)blockIdx.y * D;
  double sr = 0.0, si = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx a = v[boff + i], c = w[boff + i];
    sr = fma(a.x, c.x, fma(a.y, c.y, sr));
    si = fma(a.x, c.y, fma(-a.y, c.x, si));
  }
  __shared__ double red[4];
  sr = kry_block_sum(sr, red);
  si = kry_block_sum(si, red);
  if (threadIdx.x == 0) { atomicAdd(dre + blockIdx.y, sr); atomicAdd(dim + blockIdx.y, si); }
}

// alpha_j = Re <v_j | i w> = -Im <v_j | w>;  u = i w - alpha_j v_j - beta_{j-1} v_{j-1}  (in place
// over w);  nrm2[b] += |u|^2.  One thread per batch entry publishes alpha_j (every block computes the
// same value from the finished dot product).
__global__ __launch_bounds__(256) void k_kry_update(cplx* __restrict__ w, const cplx* __restrict__ vj,
                                                    const cplx* __restrict__ vprev, int nb, int j,
                                                    KryScalars S) {
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const size_t boff = (size_t)b * D;
  const double alpha = -S.dotim[b];
  const double bprev = j > 0 ? S.beta[(size_t)b * KRY_MAX_M + j - 1] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) S.alpha[(size_t)b * KRY_MAX_M + j] = alpha;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx x = w[boff + i], a = vj[boff + i];
    cplx u = make_double2(-x.y - alpha * a.x, x.x - alpha * a.y);  // i w - alpha v_j
    if (vprev) {
      const cplx p = vprev[boff + i];
      u.x -= bprev * p.x;
      u.y -= bprev * p.y;
    }
    w[boff + i] = u;
    s = fma(u.x, u.x, fma(u.y, u.y, s));
  }
  __shared__ double red[4];
  s = kry_block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(S.nrm2 + b, s);
}

// beta_j = sqrt(nrm2); v_{j+1} = u / beta_j in place; resets the accumulators for the next
// iteration (stream order: every reader of them has finished)
__global__ __launch_bounds__(256) void k_kry_normalize(cplx* __restrict__ u, int nb, int j, KryScalars S) {
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const size_t boff = (size_t)b * D;
  const double n2 = S.nrm2[b];
  const double beta = sqrt(n2);
  const double inv = beta > 1e-14 ? 1.0 / beta : 0.0;  // happy breakdown: the subspace is invariant
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx x = u[boff + i];
    u[boff + i] = make_double2(x.x * inv, x.y * inv);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) S.beta[(size_t)b * KRY_MAX_M + j] = beta > 1e-14 ? beta : 0.0;
}

__global__ void k_kry_reset(KryScalars S, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { S.dotre[b] = 0.0; S.dotim[b] = 0.0; S.nrm2[b] = 0.0; }
}

// coef[b][:] = norm0 * e^{-i h sigma} * exp(-i h T_m) e_1 for the real symmetric tridiagonal T_m
// (alpha, beta): sub-stepped Taylor series on the m-vector (m <= 40; ||h T|| <= rho).
__global__ void k_kry_small(KryScalars S, int B, int m, double h, double sigma, double rho) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* al = S.alpha + (size_t)b * KRY_MAX_M;
  const double* be = S.beta + (size_t)b * KRY_MAX_M;
  double cr[KRY_MAX_M], ci[KRY_MAX_M], tr[KRY_MAX_M], ti[KRY_MAX_M], ur[KRY_MAX_M], ui[KRY_MAX_M];
  for (int k = 0; k < m; ++k) { cr[k] = ci[k] = 0.0; }
  cr[0] = 1.0;
  // centre of the spectrum estimate: mean of alpha (keeps the series argument small)
  double mu = 0.0;
  for (int k = 0; k < m; ++k) mu += al[k];
  mu /= m;
  const int nsub = (int)ceil(fmax(rho, 1e-3));
  const double hs = h / nsub;
  for (int sub = 0; sub < nsub; ++sub) {
    for (int k = 0; k < m; ++k) { tr[k] = cr[k]; ti[k] = ci[k]; }
    for (int term = 1; term <= 30; ++term) {
      // t <- (-i hs / term) (T - mu) t
      for (int k = 0; k < m; ++k) {
        double xr = (al[k] - mu) * tr[k], xi = (al[k] - mu) * ti[k];
        if (k > 0) { xr += be[k - 1] * tr[k - 1]; xi += be[k - 1] * ti[k - 1]; }
        if (k + 1 < m) { xr += be[k] * tr[k + 1]; xi += be[k] * ti[k + 1]; }
        const double f = hs / term;
        ur[k] = f * xi;
        ui[k] = -f * xr;
      }
      double mag = 0.0;
      for (int k = 0; k < m; ++k) {
        tr[k] = ur[k]; ti[k] = ui[k];
        cr[k] += ur[k]; ci[k] += ui[k];
        mag = fmax(mag, fabs(ur[k]) + fabs(ui[k]));
      }
      if (mag < 1e-18) break;
    }
  }
  double sn, cs;
  sincos(-h * (sigma + mu), &sn, &cs);
  const double n0 = S.norm0[b];
  for (int k = 0; k < m; ++k)
    S.coef[(size_t)b * KRY_MAX_M + k] = make_double2(n0 * (cr[k] * cs - ci[k] * sn), n0 * (cr[k] * sn + ci[k] * cs));
}

// out = sum_j coef[b][j] V[j]   (V[j] at V + j * stride)
__global__ __launch_bounds__(256) void k_kry_combine(const cplx* __restrict__ V, size_t stride, int nb, int m,
                                                     const cplx* __restrict__ coef, cplx* __restrict__ out) {
  __shared__ cplx cs[KRY_MAX_M];
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const size_t boff = (size_t)b * D;
  if (threadIdx.x < m) cs[threadIdx.x] = coef[(size_t)b * KRY_MAX_M + threadIdx.x];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    cplx acc = make_double2(0.0, 0.0);
    for (int j = 0; j < m; ++j) acc = cfma(cs[j], V[(size_t)j * stride + boff + i], acc);
    out[boff + i] = acc;
  }
}
