// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// observables / marshalling
// ---------------------------------------------------------------------------
extern "C" int ryd_probabilities(ryd_handle* h, const void* state_dev, double* w_dev,
                                 int32_t reverse, void* stream) {
  if (!h || !state_dev || !w_dev) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t D = (size_t)1 << h->N;
  dim3 grid((unsigned)((D + 255) / 256), h->B);
  hipLaunchKernelGGL(k_probabilities, grid, dim3(256), 0, (hipStream_t)stream,
                     (const cplx*)state_dev, h->N, h->cfg.mode == RYD_MESOLVE, reverse, w_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

extern "C" int ryd_occupations(ryd_handle* h, const void* state_dev, double* out_dev,
                               void* stream) {
  if (!h || !state_dev || !out_dev) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)h->B * (h->N + 1) * sizeof(double), st));
  const size_t D = (size_t)1 << h->N;
  const unsigned nblk = (unsigned)std::min<size_t>((D + 255) / 256, 1024);
  hipLaunchKernelGGL(k_occupations, dim3(nblk, h->B), dim3(256), 0, st, (const cplx*)state_dev,
                     h->N, h->cfg.mode == RYD_MESOLVE, out_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

extern "C" int ryd_observe(ryd_handle* h, const void* state_dev, double t, int32_t what,
                           double* out_dev, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!state_dev || !out_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (h->general) return fail(RYD_ERR_INVALID, "not available on a general-path handle");
  const bool dm = h->cfg.mode == RYD_MESOLVE || (what & RYD_OBS_DENSITY) != 0;
  if (2 * h->N > RYD_MAX_QUBITS && dm) return fail(RYD_ERR_INVALID, "2N exceeds %d", RYD_MAX_QUBITS);
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  const int N = h->N;
  const int stride = N * N + N + 3;
  HIPCHK(hipMemsetAsync(out_dev, 0, (size_t)h->B * stride * sizeof(double), st));
  const size_t D = (size_t)1 << N;
  if (what & (RYD_OBS_OCCUPATION | RYD_OBS_CORRELATION)) {
    hipLaunchKernelGGL(k_obs_pairs, dim3((unsigned)((D + 2047) / 2048), h->B), dim3(256), 0, st,
                       (const cplx*)state_dev, N, dm ? 1 : 0, out_dev, stride);
    HIPCHK(hipGetLastError());
    h->stats.n_launches++;
  }
  if (what & RYD_OBS_ENERGY) {
    if (!h->bounds_valid) compute_bounds(h);
    MixPoint m;
    m.idx1 = m.idx2 = find_interval(h, t);
    m.u1 = m.u2 = t - h->tknots[m.idx1];
    m.w1 = 1.0;
    m.w2 = 0.0;
    if ((rc = launch_eval(h, m, st))) return rc;
    if (dm) {
      // Tr(H rho), Tr(H^2 rho) from the elements of rho within two bit flips of the diagonal
      const unsigned nblk = (unsigned)std::min<size_t>(std::max<size_t>(D >> 8, 1), 1024);
      hipLaunchKernelGGL(k_obs_energy_dm, dim3(nblk, h->B), dim3(256), 0, st, (const cplx*)state_dev, N,
                         (const double*)h->coefs_dev, (const double*)h->e0_dev,
                         h->e0_mats == 1 ? 0ll : (long long)D, out_dev, stride, N * N + N + 1);
      HIPCHK(hipGetLastError());
      h->stats.n_launches++;
      return RYD_OK;
    }
    if (h->cfg.mode != RYD_SESOLVE)
      return fail(RYD_ERR_UNSUPPORTED, "ket energy moments need a sesolve handle");
    if ((rc = apply_generator(h, (const cplx*)state_dev, nullptr, h->wA, 1.0, 1.0, 0.0,
                              make_double2(1.0, 0.0), st)))
      return rc;
    const unsigned nblk = (unsigned)std::min<size_t>(std::max<size_t>(D >> 10, 1), 1024);
    hipLaunchKernelGGL(k_obs_energy, dim3(nblk, h->B), dim3(256), 0, st, (const cplx*)state_dev,
                       (const cplx*)h->wA, h->nb, out_dev, stride, N * N + N + 1);
    HIPCHK(hipGetLastError());
    h->stats.n_launches++;
  }
  return RYD_OK;
}

extern "C" int ryd_ket_to_dm(ryd_handle* h, const void* psi_dev, void* rho_dev, void* stream) {
  if (!h || !psi_dev || !rho_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (2 * h->N > RYD_MAX_QUBITS) return fail(RYD_ERR_INVALID, "2N exceeds %d", RYD_MAX_QUBITS);
  HIPCHK(hipSetDevice(h->cfg.device));
  const size_t DD = (size_t)1 << (2 * h->N);
  dim3 grid((unsigned)((DD + 255) / 256), h->B);
  hipLaunchKernelGGL(k_ket_to_dm, grid, dim3(256), 0, (hipStream_t)stream, (const cplx*)psi_dev,
                     h->N, (cplx*)rho_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

// acc[D][D] += sum_b w_b |psi_b><psi_b| for any state dimension: the fp64 matrix cores on 64 x 64
// upper-triangle tiles when D is a multiple of 64 (every two-level register of 6+ atoms, 4-level
// registers of 3+), one thread per entry otherwise.
static int outer_accumulate_impl(const void* psi_dev, int64_t B, int64_t D, const double* weights,
                                 void* acc_dev, hipStream_t st) {
  if (B <= 0 || D <= 0 || B > INT32_MAX) return fail(RYD_ERR_INVALID, "batch %lld, dim %lld", (long long)B, (long long)D);
  double* wdev = nullptr;
  if (weights) {
    HIPCHK(hipMalloc((void**)&wdev, B * sizeof(double)));
    hipError_t e = hipMemcpyAsync(wdev, weights, B * sizeof(double), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { hipFree(wdev); return fail(RYD_ERR_HIP, "weights upload: %s", hipGetErrorString(e)); }
  }
  if (D % 64 == 0) {
    constexpr int KT = 16;
    const unsigned nt = (unsigned)(D / 64);
    hipLaunchKernelGGL(k_outer_mfma<KT>, dim3(nt, nt), dim3(256), 2 * KT * 128 * sizeof(double), st,
                       (const cplx*)psi_dev, (size_t)D, (int)B, wdev, (cplx*)acc_dev);
  } else {
    const size_t DD = (size_t)D * (size_t)D;
    hipLaunchKernelGGL(k_outer_acc, dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, st,
                       (const cplx*)psi_dev, (size_t)D, (int)B, wdev, (cplx*)acc_dev);
  }
  hipError_t e = hipGetLastError();
  if (wdev) { hipStreamSynchronize(st); hipFree(wdev); }
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "k_outer_acc: %s", hipGetErrorString(e));
  return RYD_OK;
}

extern "C" int ryd_outer_accumulate(ryd_handle* h, const void* psi_dev, const double* weights,
                                    void* acc_dev, void* stream) {
  if (!h || !psi_dev || !acc_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (2 * h->N > RYD_MAX_QUBITS) return fail(RYD_ERR_INVALID, "2N exceeds %d", RYD_MAX_QUBITS);
  HIPCHK(hipSetDevice(h->cfg.device));
  return outer_accumulate_impl(psi_dev, h->B, (int64_t)1 << h->N, weights, acc_dev, (hipStream_t)stream);
}

extern "C" int ryd_outer_accumulate_dim(const void* psi_dev, int64_t batch, int64_t dim,
                                        const double* weights, void* acc_dev, int32_t device,
                                        void* stream) {
  if (!psi_dev || !acc_dev) return fail(RYD_ERR_INVALID, "null argument");
  if (dim > ((int64_t)1 << (RYD_MAX_QUBITS / 2))) return fail(RYD_ERR_INVALID, "dim %lld too large", (long long)dim);
  HIPCHK(hipSetDevice(device));
  return outer_accumulate_impl(psi_dev, batch, dim, weights, acc_dev, (hipStream_t)stream);
}

extern "C" int ryd_accumulate(const void* x_dev, double weight, int64_t count, void* acc_dev,
                              int32_t device, void* stream) {
  if (!x_dev || !acc_dev || count <= 0) return fail(RYD_ERR_INVALID, "null argument or empty array");
  HIPCHK(hipSetDevice(device));
  const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 8192);
  hipLaunchKernelGGL(k_axpy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const cplx*)x_dev, weight,
                     (size_t)count, (cplx*)acc_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "k_axpy: %s", hipGetErrorString(e));
  return RYD_OK;
}

extern "C" int ryd_get_stats(const ryd_handle* h, ryd_stats* out) {
  if (!h || !out) return fail(RYD_ERR_INVALID, "null argument");
  *out = h->stats;
  if (hermitian_path(h)) out->passes = 2;  // row pass + symmetrisation
  return RYD_OK;
}

extern "C" int ryd_reset_stats(ryd_handle* h) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  const int passes = h->stats.passes;
  std::memset(&h->stats, 0, sizeof h->stats);
  h->stats.passes = passes;
  return RYD_OK;
}

extern "C" int ryd_set_kernel_timing(ryd_handle* h, int32_t enable) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  h->timing = enable != 0;
  if (enable) { h->timing_ms = 0; h->timing_launches = 0; }
  return RYD_OK;
}

extern "C" int ryd_get_kernel_timing(ryd_handle* h, double* total_ms, int64_t* launches) {
  if (!h || !total_ms || !launches) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  for (auto& ev : h->ev_used) {
    HIPCHK(hipEventSynchronize(ev.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
    h->timing_ms += ms;
    h->timing_launches++;
    h->ev_free.push_back(ev);
  }
  h->ev_used.clear();
  *total_ms = h->timing_ms;
  *launches = h->timing_launches;
  return RYD_OK;
}
