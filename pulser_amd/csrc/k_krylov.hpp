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
  double* acc;     // [2][32 B] (8 sub-accumulators x stride 4) fused iteration, by parity: Re <v~_j | w~>, Im <v~_j | w~>, |w~|^2 (k_apply's epilogue)
  double* sq;      // [B][KRY_MAX_M + 1] fused iteration: |v~_j|^2 of the stored basis vectors (null: the basis is normalised)
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

// Fused iteration (round 6): the generator kernel's final pass has left acc[4 b + 0 .. 2] = Re <x | w~>, Im <x | w~>, |w~|^2 for
// the STORED vector x = v~_j (PassArgs.kry_acc, stride 4).  The stored basis vectors are not normalised exactly: v~_{j+1} = u / b_f
// with a PROVISIONAL scale b_f (from |u|^2 = |w|^2 - alpha^2 - beta_{j-1}^2, which holds in exact arithmetic but cancels by a
// factor |w|^2 / |u|^2 ~ (diagonal / drive)^2 in floating point - used as the true norm it made the recurrence unstable: 2e-5 on
// the 12-atom anneal), while the kernel sums the TRUE |v~_{j+1}|^2 into sq[b][j + 1] on the way out.  The next iteration divides
// by that known scale (v_j = v~_j / s_j, w = w~ / s_j, true beta_j = b_f s_{j+1}), k_kry_small does the same for the tridiagonal
// matrix and the combination coefficients: exactly the Lanczos process, in ONE pass per iteration beside the generator's
// (3 reads + 1 write per amplitude where dot + update + normalise took 6 + 3, and 2 launches instead of 5).
// `acc_next`: the accumulators of the next iteration (the other parity), zeroed here.
__global__ __launch_bounds__(256) void k_kry_update_fused(cplx* __restrict__ w, const cplx* __restrict__ vj,
                                                          const cplx* __restrict__ vprev, int nb, int j, KryScalars S,
                                                          const double* __restrict__ acc, double* __restrict__ acc_next) {
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const size_t boff = (size_t)b * D;
  double* sq = S.sq + (size_t)b * (KRY_MAX_M + 1);
  const double sj2 = j == 0 ? 1.0 : sq[j];
  double a_im = 0.0, a_nn = 0.0;  // (8 sub-accumulators per sum: k_apply's epilogue)
#pragma unroll
  for (int q = 0; q < 8; ++q) { a_im += acc[(4 * b + 1) * 8 + q]; a_nn += acc[(4 * b + 2) * 8 + q]; }
  const bool alive = sj2 > 1e-290 && a_nn > 0.0;
  const double inv_s2 = alive ? 1.0 / sj2 : 0.0, inv_s = sqrt(inv_s2);
  const double alpha = -a_im * inv_s2;
  const double ww = a_nn * inv_s2;
  const double bprev = j > 0 ? S.beta[(size_t)b * KRY_MAX_M + j - 1] * sqrt(sj2) : 0.0;  // the true beta_{j-1}
  const double n2 = ww - alpha * alpha - bprev * bprev;
  const double bf = !alive ? 0.0 : (n2 > 1e-8 * ww ? sqrt(n2) : sqrt(ww));
  const double inv = bf > 0.0 ? 1.0 / bf : 0.0;
  const double cw = inv_s * inv, ca = alpha * inv_s * inv;
  const double sp2 = j > 1 ? sq[j - 1] : 1.0;
  const double cp = (j > 0 && sp2 > 1e-290) ? bprev * inv / sqrt(sp2) : 0.0;
  double sum = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx x = w[boff + i], a = vj[boff + i];
    cplx u = make_double2(-x.y * cw - ca * a.x, x.x * cw - ca * a.y);  // (i w - alpha v_j) / b_f
    if (vprev) {
      const cplx p = vprev[boff + i];
      u.x -= cp * p.x;
      u.y -= cp * p.y;
    }
    w[boff + i] = u;
    sum = fma(u.x, u.x, fma(u.y, u.y, sum));
  }
  __shared__ double red[4];
  sum = kry_block_sum(sum, red);
  if (threadIdx.x == 0) atomicAdd(sq + j + 1, sum);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    S.alpha[(size_t)b * KRY_MAX_M + j] = alpha;
    S.beta[(size_t)b * KRY_MAX_M + j] = bf;  // provisional: the true beta_j is b_f sqrt(sq[j + 1])
    for (int q = 0; q < 24; ++q) acc_next[32 * b + q] = 0.0;
  }
}

__global__ void k_kry_reset(KryScalars S, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { S.dotre[b] = 0.0; S.dotim[b] = 0.0; S.nrm2[b] = 0.0; }
}

// coef[b][:] = norm0 * e^{-i h sigma} * exp(-i h T_m) e_1 for the real symmetric tridiagonal T_m
// (alpha, beta): sub-stepped Taylor series on the m-vector (m <= 40; ||h T|| <= rho).
// (round 6: one WAVE per batch entry, lane k = component k, neighbours over __shfl - the one-thread version walked six
// 40-entry scratch arrays and took 70 us per exponential at m = 11, a sixth of the Lanczos process it closes)
__global__ __launch_bounds__(64) void k_kry_small(KryScalars S, int B, int m, double h, double sigma, double rho) {
  const int b = blockIdx.x;
  const int k = threadIdx.x;
  if (b >= B) return;
  const bool on = k < m;
  // fused iteration: the stored beta_k are provisional scales and the stored basis vectors have the norms s_k = sqrt(sq[k])
  const double s0 = (S.sq && on && k > 0) ? S.sq[(size_t)b * (KRY_MAX_M + 1) + k] : 1.0;
  const double s1 = (S.sq && on) ? S.sq[(size_t)b * (KRY_MAX_M + 1) + k + 1] : 1.0;
  const double isc = s0 > 1e-290 ? 1.0 / sqrt(s0) : 0.0;
  const double al = on ? S.alpha[(size_t)b * KRY_MAX_M + k] : 0.0;
  // be_up couples k and k + 1 (zero on the last component), be_dn couples k and k - 1
  const double be_up = (on && k + 1 < m) ? S.beta[(size_t)b * KRY_MAX_M + k] * sqrt(s1 > 0.0 ? s1 : 0.0) : 0.0;
  double be_dn = __shfl_up(be_up, 1, 64);
  if (k == 0) be_dn = 0.0;
  // centre of the spectrum estimate: mean of alpha (keeps the series argument small)
  double mu = al;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mu += __shfl_xor(mu, o, 64);
  mu /= m;
  const double d = on ? al - mu : 0.0;
  double cr = k == 0 ? 1.0 : 0.0, ci = 0.0;
  const int nsub = (int)ceil(fmax(rho, 1e-3));
  const double hs = h / nsub;
  for (int sub = 0; sub < nsub; ++sub) {
    double tr = cr, ti = ci;
    for (int term = 1; term <= 30; ++term) {
      // t <- (-i hs / term) (T - mu) t
      const double tr_dn = __shfl_up(tr, 1, 64), ti_dn = __shfl_up(ti, 1, 64);
      const double tr_up = __shfl_down(tr, 1, 64), ti_up = __shfl_down(ti, 1, 64);
      const double xr = fma(be_up, tr_up, fma(be_dn, tr_dn, d * tr));
      const double xi = fma(be_up, ti_up, fma(be_dn, ti_dn, d * ti));
      const double f = hs / term;
      tr = on ? f * xi : 0.0;
      ti = on ? -f * xr : 0.0;
      cr += tr;
      ci += ti;
      double mag = fabs(tr) + fabs(ti);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mag = fmax(mag, __shfl_xor(mag, o, 64));
      if (mag < 1e-18) break;
    }
  }
  double sn, cs;
  sincos(-h * (sigma + mu), &sn, &cs);
  const double n0 = S.norm0[b];
  if (on)
    S.coef[(size_t)b * KRY_MAX_M + k] = make_double2(n0 * isc * (cr * cs - ci * sn), n0 * isc * (cr * sn + ci * cs));
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
