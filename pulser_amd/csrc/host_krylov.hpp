// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Lanczos exponential on the tiled generator kernels (ryd_opts.method = 1)
// ---------------------------------------------------------------------------
// Smallest Krylov dimension with  2 (rho/2)^m / m! <= tol  (the a-priori bound of the
// Lanczos approximation of exp(-i h H) for a spectrum of half-width rho / h).
static int krylov_dim(double rho, double tol) {
  double term = 2.0;
  int m = 0;
  while (m < KRY_MAX_M) {
    ++m;
    term *= 0.5 * rho / m;
    if (m >= 2 && term <= tol) break;
  }
  return std::min(m, KRY_MAX_M);
}

static int ensure_krylov(ryd_handle* h, int m) {
  const size_t per = h->dim * (size_t)h->B;
  if (h->kry_cap < m + 1) {
    if (h->kry_V) hipFree(h->kry_V);
    h->kry_V = nullptr;
    h->kry_cap = 0;
    HIPCHK(hipMalloc((void**)&h->kry_V, (size_t)(m + 1) * per * sizeof(cplx)));
    h->kry_cap = m + 1;
  }
  if (!h->kry_pool) {
    const size_t B = (size_t)h->B;
    const size_t n_dbl = 2 * B * KRY_MAX_M + 4 * B + 64 * B + B * (KRY_MAX_M + 1);
    const size_t bytes = B * KRY_MAX_M * sizeof(cplx) + n_dbl * sizeof(double);
    HIPCHK(hipMalloc(&h->kry_pool, bytes));
    HIPCHK(hipMemset(h->kry_pool, 0, bytes));
    char* p = (char*)h->kry_pool;
    h->kry.coef = (cplx*)p;   p += B * KRY_MAX_M * sizeof(cplx);
    h->kry.alpha = (double*)p; p += B * KRY_MAX_M * sizeof(double);
    h->kry.beta = (double*)p;  p += B * KRY_MAX_M * sizeof(double);
    h->kry.dotre = (double*)p; p += B * sizeof(double);
    h->kry.dotim = (double*)p; p += B * sizeof(double);
    h->kry.nrm2 = (double*)p;  p += B * sizeof(double);
    h->kry.norm0 = (double*)p; p += B * sizeof(double);
    h->kry.acc = (double*)p;   p += 64 * B * sizeof(double);
    h->kry_sq = (double*)p;
  }
  return RYD_OK;
}

// (two workgroups per CU: 512 atomics per accumulator and reduction instead of the 4 096 of round 4)
static unsigned kry_blocks(const ryd_handle* h) {
  return (unsigned)std::min<size_t>(std::max<size_t>(h->dim >> 10, 1), 512);
}

// state <- exp(h (w1 G(t1) + w2 G(t2))) state  by an m-dimensional Lanczos process
static int exp_step_krylov(ryd_handle* h, cplx* state, double hstep, const MixPoint& mp, double rho,
                           double shift, double tol, hipStream_t st) {
  int rc;
  const int m = krylov_dim(rho, tol);
  if ((rc = ensure_krylov(h, m))) return rc;
  if ((rc = launch_eval(h, mp, st))) return rc;
  const double wmix = mp.w1 + mp.w2;
  const size_t per = h->dim * (size_t)h->B;
  const dim3 grid(kry_blocks(h), h->B), blk(256);
  const unsigned gb = (unsigned)((h->B + 127) / 128);
  cplx* V = h->kry_V;
  hipLaunchKernelGGL(k_kry_reset, dim3(gb), dim3(128), 0, st, h->kry, h->B);
  hipLaunchKernelGGL(k_kry_norm, grid, blk, 0, st, (const cplx*)state, h->nb, h->kry.nrm2);
  hipLaunchKernelGGL(k_kry_scale, grid, blk, 0, st, (const cplx*)state, V, h->nb, h->kry.nrm2, h->kry.norm0, 1);
  h->stats.n_launches += 3;
  const cplx one = make_double2(1.0, 0.0);
  // Round 6: where the generator is ONE k_apply pass (sesolve kets up to 20 atoms and beyond, not the 2^14 register tiles) the
  // inner product and |w|^2 ride on that pass's store epilogue and the three-term recurrence + normalisation are one kernel:
  // 2 launches and 6 vector transfers per iteration instead of 5 launches and 11 ("weak" 6).
  static const bool fuse_env = dev_env_flag("RYD_KRY_FUSE", true);
  if (!h->passes_valid) plan_passes(h);
  const bool fused = fuse_env && h->passes.size() == 1 && !h->passes[0].use14;
  if (fused) HIPCHK(hipMemsetAsync(h->kry.acc, 0, ((size_t)64 * h->B + (size_t)h->B * (KRY_MAX_M + 1)) * sizeof(double), st));  // acc + sq (adjacent)
  h->kry.sq = fused ? h->kry_sq : nullptr;
  for (int j = 0; j < m; ++j) {
    cplx* vj = V + (size_t)j * per;
    cplx* w = V + (size_t)(j + 1) * per;
    if (fused) {
      double* acc = h->kry.acc + (size_t)(j & 1) * 32 * h->B;
      double* acc_next = h->kry.acc + (size_t)((j + 1) & 1) * 32 * h->B;
      if ((rc = apply_generator(h, vj, nullptr, w, wmix, 1.0, shift, one, st, false, acc))) return rc;
      hipLaunchKernelGGL(k_kry_update_fused, grid, blk, 0, st, w, (const cplx*)vj,
                         j > 0 ? (const cplx*)(V + (size_t)(j - 1) * per) : (const cplx*)nullptr, h->nb, j, h->kry,
                         (const double*)acc, acc_next);
      h->stats.n_launches += 1;
      continue;
    }
    hipLaunchKernelGGL(k_kry_reset, dim3(gb), dim3(128), 0, st, h->kry, h->B);
    if ((rc = apply_generator(h, vj, nullptr, w, wmix, 1.0, shift, one, st, false))) return rc;
    hipLaunchKernelGGL(k_kry_dot, grid, blk, 0, st, (const cplx*)vj, (const cplx*)w, h->nb, h->kry.dotre, h->kry.dotim);
    hipLaunchKernelGGL(k_kry_update, grid, blk, 0, st, w, (const cplx*)vj,
                       j > 0 ? (const cplx*)(V + (size_t)(j - 1) * per) : (const cplx*)nullptr, h->nb, j, h->kry);
    hipLaunchKernelGGL(k_kry_normalize, grid, blk, 0, st, w, h->nb, j, h->kry);
    h->stats.n_launches += 4;
  }
  hipLaunchKernelGGL(k_kry_small, dim3(h->B), dim3(64), 0, st, h->kry, h->B, m, hstep, shift, rho);  // one wave per batch entry
  hipLaunchKernelGGL(k_kry_combine, grid, blk, 0, st, (const cplx*)V, per, h->nb, m, (const cplx*)h->kry.coef, state);
  HIPCHK(hipGetLastError());
  h->stats.n_launches += 2;
  h->stats.last_order = m;
  return RYD_OK;
}

static bool krylov_selected(const ryd_handle* h, const ryd_opts& o) {
  return o.method == 1 && !h->general && h->cfg.mode == RYD_SESOLVE && !h->mc;
}
