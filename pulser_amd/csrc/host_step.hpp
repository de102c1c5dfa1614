// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// stepping: schedule (host) -> generic multi-launch path or persistent kernel
// ---------------------------------------------------------------------------
static bool hermitian_path(const ryd_handle* h) {
  return !h->general && h->cfg.mode == RYD_MESOLVE && !h->has_dbl && h->N >= 7 && h->N <= 14 &&
         h->auto_tile && tile14_pays(h);
}

// One exponential  state <- exp(h * G~) state  on the generic multi-launch path.
static int exp_step(ryd_handle* h, cplx* state, double hstep, const MixPoint& m, int order,
                    double shift, hipStream_t st) {
  int rc;
  if (h->general) {
    if ((rc = launch_eval_general(h, m, st))) return rc;
    const cplx* gin = state;
    cplx* gbufs[2] = {h->wA, h->wB};
    int gw = 0;
    for (int j = order; j >= 1; --j) {
      cplx* out = j == 1 ? state : gbufs[gw];
      if ((rc = apply_general(h, m, gin, state, out, hstep / j, st))) return rc;
      gin = out;
      gw ^= 1;
    }
    return RYD_OK;
  }
  if ((rc = launch_eval(h, m, st))) return rc;
  const double wmix = m.w1 + m.w2;
  if (hermitian_path(h)) {
    // rho stays Hermitian, so G rho = P + P^dagger with P = (1/2) D.rho + the
    // column-bit flips only: one register-tile pass over rows + one tile-pair
    // symmetrisation instead of three tiled passes.
    const cplx* hin = state;
    cplx* hb[2] = {h->wA, h->wB};
    int hw = 0;
    for (int j = order; j >= 1; --j) {
      cplx* out = j == 1 ? state : hb[hw];
      Apply14Args B;
      std::memset(&B, 0, sizeof B);
      B.in = hin;
      B.kout = h->kbuf;
      B.coefs = h->coefs_dev;
      B.e0 = h->e0_dev;
      B.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
      B.wmix = wmix;
      B.diag_scale = 0.5;
      B.scale = 1.0;
      B.post = make_double2(1.0, 0.0);
      for (int i = 0; i < 4; ++i) B.Sd[i] = h->Sd[i];
      B.N = h->N;
      B.nb = h->nb;
      B.n_flip = h->N;
      std::pair<hipEvent_t, hipEvent_t> ev;
      if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
      if ((rc = launch_apply14(h, B, st))) return rc;
      if (h->timing) { HIPCHK(hipEventRecord(ev.second, st)); h->ev_used.push_back(ev); }
      SymmArgs S;
      S.P = h->kbuf;
      S.base = state;
      S.out = out;
      S.scale = hstep / j;
      S.N = h->N;
      const unsigned nt = 1u << (h->N - 5);
      if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
      hipLaunchKernelGGL(k_symm, dim3(nt, nt, h->B), dim3(256), 0, st, S);
      HIPCHK(hipGetLastError());
      if (h->timing) { HIPCHK(hipEventRecord(ev.second, st)); h->ev_used.push_back(ev); }
      h->stats.n_launches += 2;
      h->stats.n_applications++;
      hin = out;
      hw ^= 1;
    }
    return RYD_OK;
  }
  // Horner: w_m = psi; w_{j-1} = psi + (h/j) G' w_j; result w_0, times e^{-i h shift}
  const cplx one = make_double2(1.0, 0.0);
  const cplx* in = state;
  cplx* bufs[2] = {h->wA, h->wB};
  int which = 0;
  for (int j = order; j >= 1; --j) {
    cplx* out = j == 1 ? state : bufs[which];
    cplx post = one;
    if (j == 1) {
      // e^{-i h shift}, and for H_eff the centre of the decay diagonal
      const double mag = h->mc ? std::exp(hstep * wmix * (h->mc_a + 0.5 * h->N * h->mc_b)) : 1.0;
      post = make_double2(mag * std::cos(hstep * shift), -mag * std::sin(hstep * shift));
    }
    if ((rc = apply_generator(h, in, state, out, wmix, hstep / j, shift, post, st, h->mc))) return rc;
    in = out;
    which ^= 1;
  }
  return RYD_OK;
}

static unsigned mc_blocks(const ryd_handle* h) {
  return (unsigned)std::min<size_t>(std::max<size_t>(h->dim >> 10, 1), 128);
}

// Jump bookkeeping after one CF4 step of a Monte-Carlo solve (all on `st`).
static int mc_after_step(ryd_handle* h, cplx* state, hipStream_t st) {
  const unsigned nblk = mc_blocks(h);
  hipLaunchKernelGGL(k_mc_norm, dim3(nblk, h->B), dim3(256), 0, st, state, h->nb, h->mcs.norm2);
  hipLaunchKernelGGL(k_mc_reduced, dim3(nblk, h->B), dim3(256), 0, st, state, h->N, h->mcs, h->B, 0);
  hipLaunchKernelGGL(k_mc_select, dim3((h->B + 127) / 128), dim3(128), 0, st, h->mcs, h->B, h->N, 0);
  hipLaunchKernelGGL(k_mc_jump, dim3(nblk, h->B), dim3(256), 0, st, state, h->N, h->mcs);
  HIPCHK(hipGetLastError());
  h->stats.n_launches += 4;
  return RYD_OK;
}

// Snapshot of the state: a plain copy, or the normalised ket in a Monte-Carlo solve.
static int snapshot_copy(ryd_handle* h, const cplx* state, cplx* dst, hipStream_t st) {
  if (h->mc_active) {
    hipLaunchKernelGGL(k_mc_normalize, dim3(mc_blocks(h), h->B), dim3(256), 0, st, state, dst, h->nb,
                       h->mcs.lastnorm);
    HIPCHK(hipGetLastError());
    return RYD_OK;
  }
  HIPCHK(hipMemcpyAsync(dst, state, h->dim * (size_t)h->B * sizeof(cplx), hipMemcpyDeviceToDevice, st));
  return RYD_OK;
}

static int run_generic(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                       cplx* snaps, hipStream_t st, const ryd_opts& o) {
  int rc;
  const bool kry = krylov_selected(h, o);
  const double ktol = o.tol > 0 ? o.tol : kDefaultTol * budget_scale(h);
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  for (const StepDesc& d : sched) {
    MixPoint m;
    m.idx1 = m.idx2 = d.idx;
    m.u1 = d.u1;
    m.u2 = d.u2;
    m.w1 = kA1; m.w2 = kA2;
    if (kry) {
      double bnd, sh;
      ket_bound(h, d.idx, kA1, kA2, &bnd, &sh, d.pad);
      if ((rc = exp_step_krylov(h, state, d.h, m, std::fabs(d.h) * bnd, sh, ktol, st))) return rc;
      m.w1 = kA2; m.w2 = kA1;
      ket_bound(h, d.idx, kA2, kA1, &bnd, &sh, d.pad);
      if ((rc = exp_step_krylov(h, state, d.h, m, std::fabs(d.h) * bnd, sh, ktol, st))) return rc;
    } else {
      if ((rc = exp_step(h, state, d.h, m, d.order_a, d.shift_a, st))) return rc;
      m.w1 = kA2; m.w2 = kA1;
      if ((rc = exp_step(h, state, d.h, m, d.order_b, d.shift_b, st))) return rc;
    }
    h->stats.n_steps++;
    if (h->mc_active && (rc = mc_after_step(h, state, st))) return rc;
    if (d.snap >= 0 && snaps && (rc = snapshot_copy(h, state, snaps + (size_t)d.snap * h->dim * h->B, st)))
      return rc;
  }
  (void)bytes;
  return RYD_OK;
}

template <int N, int MODEL, bool MC>
static int launch_traj2(ryd_handle* h, const TrajArgs& A, hipStream_t st) {
  constexpr int D = 1 << N;
  constexpr int NTT = D < 64 ? 64 : (N >= 11 ? 1024 : (D > 512 ? 512 : D));
  const size_t copies = (size_t)2 * D * sizeof(cplx) > 144 * 1024 ? 1 : 2;  // see k_traj: SINGLE
  const size_t lds = copies * (size_t)D * sizeof(cplx) + 4 * 16 * sizeof(double) * 4 + 2 * 16 * sizeof(double);
  // the dynamic-LDS limit is a per-device function attribute
  static bool attr_set[64] = {};
  const int dev = h->cfg.device;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIPCHK(hipFuncSetAttribute((const void*)k_traj<N, NTT, MODEL, MC>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((k_traj<N, NTT, MODEL, MC>), dim3(h->B), dim3(NTT), lds, st, A);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

template <int N>
static int launch_traj(ryd_handle* h, const TrajArgs& A, hipStream_t st) {
  if (h->mc)
    return h->uniform_real_drive ? launch_traj2<N, 1, true>(h, A, st)
                                 : launch_traj2<N, 0, true>(h, A, st);
  return h->uniform_real_drive ? launch_traj2<N, 1, false>(h, A, st)
                               : launch_traj2<N, 0, false>(h, A, st);
}

// Persistent path (sesolve, N <= 13): one workgroup per trajectory keeps its
// state vector in LDS/registers for the whole schedule; one launch.
static int run_persistent(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                          cplx* snaps, hipStream_t st) {
  if (sched.empty()) return RYD_OK;
  const size_t bytes = sched.size() * sizeof(StepDesc);
  if (h->sched_cap < sched.size()) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_cap = 0;
    HIPCHK(hipMalloc((void**)&h->sched_dev, bytes * 2));
    h->sched_cap = sched.size() * 2;
  }
  // the schedule buffer may still be read by an earlier launch on `st`
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), bytes, hipMemcpyHostToDevice, st));
  TrajArgs A;
  A.state = state;
  A.snaps = snaps;
  A.pp = h->pp_dev;
  A.n_int = h->n_knots - 1;
  A.desc = h->desc_dev;
  A.dterms = h->dterms_dev;
  A.e0 = h->e0_dev;
  A.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
  A.steps = h->sched_dev;
  A.n_steps = (int)sched.size();
  A.B = h->B;
  A.a1 = kA1;
  A.a2 = kA2;
  A.mc = h->mcs;
  A.mc_a = h->mc_a;
  A.mc_b = h->mc_b;
  A.mc_jumps = h->mc_active ? 1 : 0;
  int rc = RYD_ERR_INVALID;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
  switch (h->N) {
    case 1: rc = launch_traj<1>(h, A, st); break;
    case 2: rc = launch_traj<2>(h, A, st); break;
    case 3: rc = launch_traj<3>(h, A, st); break;
    case 4: rc = launch_traj<4>(h, A, st); break;
    case 5: rc = launch_traj<5>(h, A, st); break;
    case 6: rc = launch_traj<6>(h, A, st); break;
    case 7: rc = launch_traj<7>(h, A, st); break;
    case 8: rc = launch_traj<8>(h, A, st); break;
    case 9: rc = launch_traj<9>(h, A, st); break;
    case 10: rc = launch_traj<10>(h, A, st); break;
    case 11: rc = launch_traj<11>(h, A, st); break;
    case 12: rc = launch_traj<12>(h, A, st); break;
    case 13: rc = launch_traj<13>(h, A, st); break;
    default: return fail(RYD_ERR_INVALID, "persistent path needs N <= 13");
  }
  if (rc) return rc;
  if (h->timing) {
    HIPCHK(hipEventRecord(ev.second, st));
    h->ev_used.push_back(ev);
  }
  for (const StepDesc& d : sched) {
    h->stats.n_applications += d.order_a + d.order_b;
    h->stats.n_steps++;
  }
  h->stats.n_launches++;
  return RYD_OK;
}

template <int N, bool DBL, bool REALU>
static int launch_traj_dm(ryd_handle* h, const TrajDmArgs& A, hipStream_t st) {
  constexpr int D = 1 << (2 * N);
  constexpr int NTT = D < 64 ? 64 : (2 * N >= 11 ? 1024 : (D > 512 ? 512 : D));
  const size_t lds = 2 * (size_t)D * sizeof(cplx) + 2 * 32 * sizeof(double);
  static bool attr_set[64] = {};
  const int dev = h->cfg.device;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIPCHK(hipFuncSetAttribute((const void*)k_traj_dm<N, NTT, DBL, REALU>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((k_traj_dm<N, NTT, DBL, REALU>), dim3(h->B), dim3(NTT), lds, st, A);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

// Persistent path for small density matrices (mesolve, N <= 6): one launch.
static int run_persistent_dm(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                             cplx* snaps, hipStream_t st) {
  if (sched.empty()) return RYD_OK;
  const size_t bytes = sched.size() * sizeof(StepDesc);
  if (h->sched_cap < sched.size()) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_cap = 0;
    HIPCHK(hipMalloc((void**)&h->sched_dev, bytes * 2));
    h->sched_cap = sched.size() * 2;
  }
  HIPCHK(hipStreamSynchronize(st));  // the buffer may still be read by an earlier launch
  HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), bytes, hipMemcpyHostToDevice, st));
  TrajDmArgs A;
  A.state = state;
  A.snaps = snaps;
  A.pp = h->pp_dev;
  A.desc = h->desc_dev;
  A.dterms = h->dterms_dev;
  A.e0 = h->e0_dev;
  A.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
  A.steps = h->sched_dev;
  A.n_int = h->n_knots - 1;
  A.n_steps = (int)sched.size();
  A.B = h->B;
  A.a1 = kA1;
  A.a2 = kA2;
  for (int i = 0; i < 4; ++i) { A.Sd[i] = h->Sd[i]; A.J[i] = h->J[i]; }
  int rc = RYD_ERR_INVALID;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
#define RYD_DM_CASE(NN)                                                                        \
  case NN:                                                                                     \
    rc = h->has_dbl ? (h->uniform_real_drive ? launch_traj_dm<NN, true, true>(h, A, st)         \
                                             : launch_traj_dm<NN, true, false>(h, A, st))       \
                    : (h->uniform_real_drive ? launch_traj_dm<NN, false, true>(h, A, st)        \
                                             : launch_traj_dm<NN, false, false>(h, A, st));     \
    break;
  switch (h->N) {
    RYD_DM_CASE(1) RYD_DM_CASE(2) RYD_DM_CASE(3) RYD_DM_CASE(4) RYD_DM_CASE(5) RYD_DM_CASE(6)
    default: return fail(RYD_ERR_INVALID, "persistent density-matrix path needs N <= 6");
  }
#undef RYD_DM_CASE
  if (rc) return rc;
  if (h->timing) {
    HIPCHK(hipEventRecord(ev.second, st));
    h->ev_used.push_back(ev);
  }
  for (const StepDesc& d : sched) {
    h->stats.n_applications += d.order_a + d.order_b;
    h->stats.n_steps++;
  }
  h->stats.n_launches++;
  return RYD_OK;
}

// Persistent path of the general (explicit CSR terms) engine for small vectors.
static int run_persistent_general(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched,
                                  cplx* snaps, hipStream_t st) {
  if (sched.empty()) return RYD_OK;
  const size_t bytes = sched.size() * sizeof(StepDesc);
  if (h->sched_cap < sched.size()) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_cap = 0;
    HIPCHK(hipMalloc((void**)&h->sched_dev, bytes * 2));
    h->sched_cap = sched.size() * 2;
  }
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), bytes, hipMemcpyHostToDevice, st));
  GenTrajArgs A;
  A.state = state;
  A.snaps = snaps;
  A.pp = h->pp_dev;
  A.series = h->gen_series_dev;
  A.conjf = h->gen_conj_dev;
  A.scale = h->gen_scale_dev;
  A.terms = h->gen_terms_dev;
  A.steps = h->sched_dev;
  A.n_int = h->n_knots - 1;
  A.n_steps = (int)sched.size();
  A.n_terms = (int)h->gen_host.size();
  A.dim = (int)h->dim;
  A.d = h->gen_d;
  A.n_dig = h->gen_ndig;
  A.a1 = kA1;
  A.a2 = kA2;
  const size_t lds = 2 * 4096 * sizeof(cplx) + 2 * MAX_GEN_TERMS * sizeof(cplx);
  static bool attr_set[64] = {};
  const int dev = h->cfg.device;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIPCHK(hipFuncSetAttribute((const void*)k_gen_traj, hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  std::pair<hipEvent_t, hipEvent_t> ev;
  int rc;
  if (h->timing) { if ((rc = timing_begin(h, st, ev))) return rc; }
  hipLaunchKernelGGL(k_gen_traj, dim3(1), dim3(1024), lds, st, A);
  HIPCHK(hipGetLastError());
  if (h->timing) {
    HIPCHK(hipEventRecord(ev.second, st));
    h->ev_used.push_back(ev);
  }
  for (const StepDesc& d : sched) {
    h->stats.n_applications += d.order_a + d.order_b;
    h->stats.n_steps++;
  }
  h->stats.n_launches++;
  return RYD_OK;
}

// GenTrajArgs of the persistent general kernel for `sched` (uploaded into the handle's schedule buffer)
static int fill_gen_traj_args(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched, cplx* snaps,
                              hipStream_t st, GenTrajArgs& A) {
  const size_t bytes = sched.size() * sizeof(StepDesc);
  if (h->sched_cap < sched.size()) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_cap = 0;
    HIPCHK(hipMalloc((void**)&h->sched_dev, bytes * 2));
    h->sched_cap = sched.size() * 2;
  }
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpyAsync(h->sched_dev, sched.data(), bytes, hipMemcpyHostToDevice, st));
  A.state = state;
  A.snaps = snaps;
  A.pp = h->pp_dev;
  A.series = h->gen_series_dev;
  A.conjf = h->gen_conj_dev;
  A.scale = h->gen_scale_dev;
  A.terms = h->gen_terms_dev;
  A.steps = h->sched_dev;
  A.n_int = h->n_knots - 1;
  A.n_steps = (int)sched.size();
  A.n_terms = (int)h->gen_host.size();
  A.dim = (int)h->dim;
  A.d = h->gen_d;
  A.n_dig = h->gen_ndig;
  A.a1 = kA1;
  A.a2 = kA2;
  return RYD_OK;
}

// The one-workgroup kernel k_gen_traj walks its term list once per application: ~0.5 us per (term, group) and 1 024 rows
// (measured, round 6: XY exchange on 8 atoms - 256 amplitudes, 4 pair terms x 28 pairs + 4 site terms x 8 sites - 72 us per
// application; the multi-launch padded-site kernel k_gen_apply_fused applies the same generator in 6 us, launch included:
// profiles/r06_general_path.md).  It keeps the systems whose application is cheaper than a launch: a dozen groups on
// up to 1 024 rows (the 2 - 3 atom sequences of the reference's own tests).
static bool use_persistent_general(const ryd_handle* h) {
  if (!(h->general && h->B == 1 && h->dim <= 4096 && !h->force_generic && !h->gen_host.empty())) return false;
  const double rows = std::max(1.0, (double)h->dim / 1024.0);
  double groups = 0.0;
  for (const GenTermHost& t : h->gen_host) groups += t.dev.kind == 1 ? (double)t.dev.n_groups : 1.0;
  return groups * rows <= 13.0;
}

static bool use_persistent_dm(const ryd_handle* h) {
  return !h->general && h->cfg.mode == RYD_MESOLVE && h->N <= 6 && !h->force_generic;
}

static int run_steps(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched, cplx* snaps,
                     hipStream_t st, const ryd_opts& o);

static bool use_persistent(const ryd_handle* h) {
  return !h->general && h->cfg.mode == RYD_SESOLVE && h->N <= 13 && !h->force_generic;
}

static int run_steps(ryd_handle* h, cplx* state, const std::vector<StepDesc>& sched, cplx* snaps,
                     hipStream_t st, const ryd_opts& o) {
  if (krylov_selected(h, o)) return run_generic(h, state, sched, snaps, st, o);
  if (split_selected(h, o)) return run_split(h, state, sched, snaps, o, st);
  if (ket_path(h)) return run_ket(h, state, sched, snaps, o, st);
  if (use_persistent(h)) return run_persistent(h, state, sched, snaps, st);
  if (use_persistent_dm(h)) return run_persistent_dm(h, state, sched, snaps, st);
  if (use_persistent_general(h)) return run_persistent_general(h, state, sched, snaps, st);
  if (row_path(h)) return run_rows(h, state, sched, snaps, o, st);
  return run_generic(h, state, sched, snaps, st, o);
}

// Self-check ("kernel self-checks - norm drift, NaN scan"), enabled by the environment
// variable RYD_CHECK=1 and off by default (it synchronises the stream): after a solve every state must be
// finite, and where the dynamics conserve it (no quantum-jump decay) the squared norm / trace must not
// have moved by more than 1e-6 - a stepper that leaves its stability region or a kernel that reads a
// stale buffer shows up here long before a parity test looks at amplitudes.
static bool selfcheck_enabled() {
  static const int on = [] {
    const char* e = std::getenv("RYD_CHECK");
    return (e && e[0] && std::strcmp(e, "0") != 0) ? 1 : 0;
  }();
  return on != 0;
}

static int selfcheck_measure(ryd_handle* h, const cplx* state, hipStream_t st, std::vector<double>& out) {
  double* dev = nullptr;
  HIPCHK(hipMalloc((void**)&dev, 2 * h->B * sizeof(double)));
  hipMemsetAsync(dev, 0, 2 * h->B * sizeof(double), st);
  const bool dm = h->cfg.mode == RYD_MESOLVE;
  const size_t D = dm ? (size_t)std::llround(std::sqrt((double)h->dim)) : 0;
  const unsigned blocks = (unsigned)std::min<size_t>((h->dim + 255) / 256, 2048);
  hipLaunchKernelGGL(k_selfcheck, dim3(blocks, h->B), dim3(256), 0, st, state, (size_t)h->dim, dm ? D + 1 : (size_t)0, dev);
  out.assign(2 * h->B, 0.0);
  hipError_t e = hipMemcpyAsync(out.data(), dev, 2 * h->B * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(dev);
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "self-check: %s", hipGetErrorString(e));
  return RYD_OK;
}

static int solve_impl(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                      void* out_dev, const ryd_opts* opts, void* stream);

extern "C" int ryd_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                         void* out_dev, const ryd_opts* opts, void* stream) {
  if (!selfcheck_enabled() || !h || !state_dev) return solve_impl(h, state_dev, n_times, times, out_dev, opts, stream);
  int rc = check_ready(h);
  if (rc) return rc;
  HIPCHK(hipSetDevice(h->cfg.device));
  std::vector<double> before, after;
  if ((rc = selfcheck_measure(h, (const cplx*)state_dev, (hipStream_t)stream, before))) return rc;
  if ((rc = solve_impl(h, state_dev, n_times, times, out_dev, opts, stream))) return rc;
  if ((rc = selfcheck_measure(h, (const cplx*)state_dev, (hipStream_t)stream, after))) return rc;
  for (int b = 0; b < h->B; ++b) {
    if (after[2 * b + 1] != 0.0)
      return fail(RYD_ERR_NUMERIC, "self-check: %.0f non-finite entries in state %d after the solve to t = %g us",
                  after[2 * b + 1], b, times[n_times - 1]);
    const double n0 = before[2 * b], n1 = after[2 * b];
    if (!h->mc && before[2 * b + 1] == 0.0 && std::fabs(n1 - n0) > 1e-6 * std::max(std::fabs(n0), 1e-300))
      return fail(RYD_ERR_NUMERIC, "self-check: %s of state %d moved from %.12g to %.12g over [%g, %g] us",
                  h->cfg.mode == RYD_MESOLVE ? "trace" : "squared norm", b, n0, n1, times[0], times[n_times - 1]);
  }
  return RYD_OK;
}

static int solve_impl(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                      void* out_dev, const ryd_opts* opts, void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!state_dev || !times || n_times < 2) return fail(RYD_ERR_INVALID, "need a state and >= 2 times");
  for (int i = 1; i < n_times; ++i)
    if (!(times[i] >= times[i - 1])) return fail(RYD_ERR_INVALID, "times must be non-decreasing");
  HIPCHK(hipSetDevice(h->cfg.device));
  if (!h->bounds_valid) { if (h->general) compute_bounds_general(h); else compute_bounds(h); }
  ryd_opts o;
  std::memset(&o, 0, sizeof o);
  if (opts) o = *opts;
  hipStream_t st = (hipStream_t)stream;
  cplx* state = (cplx*)state_dev;
  cplx* snaps = (cplx*)out_dev;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  // 14-atom batches of real-drive sequences (the headline): the register-resident split-operator kernel
  // (k_split14_loop) instead of the register-resident polynomial one (k_ket) when this call's schedule is mostly
  // multi-knot steps - 6th order over 8-knot sub-steps needs a third of k_ket's stages there, at 2.2x the work per
  // stage; with evaluation times at every knot k_ket keeps the job
  auto merged_share = [&]() {
    std::vector<StepDesc> trial;
    for (int i = 1; i < n_times; ++i) build_schedule(h, times[i - 1], times[i], o, trial, true, kSplitMergeMax, true);
    double all = 0.0, merged = 0.0;
    for (const StepDesc& d : trial) { all += d.h; if (d.pad > 1) merged += d.h; }
    return all > 0.0 ? merged / all : 0.0;
  };
  h->split14_auto = false;
  h->stats.reserved[0] = 0.0;  // accumulated local-error estimate of THIS solve (the split-operator paths book it)
  // RYD_HOST_TIMING=1 (dev): host milliseconds of a solve before its first launch (schedule building) on stderr
  static const bool host_timing = [] { const char* e = std::getenv("RYD_HOST_TIMING"); return e && e[0] == '1'; }();
  const auto host_t0 = std::chrono::steady_clock::now();
  double share = -1.0;
  // (round 4: k_split_reg covers 12 - 14 atoms and any batch size, so the same choice is made for 12 / 13 atoms against
  // k_traj / k_ket / the tiled kernels)
  const bool reg_shape = split_reg_shape(h);
  if ((h->N == 14 || reg_shape) && o.method == 0 && o.taylor_order <= 0 && !h->force_generic && !h->no_split && !h->force_ket &&
      !h->no_split14 && split_capable(h) && (reg_shape || ket_path(h)) && (reg_shape || h->drive_real) && (reg_shape || split_loop14(h))) {
    if (split_s10_allowed(h)) {
      share = merged_share();
      h->split14_auto = share >= 0.5;
    }
    // Waveforms with nothing to merge (noise series, modulated local drives): since the complex drives run on the real
    // kernel too (SplitRun.gauge) the 6-stage composition with one-knot sub-steps beats the polynomial kernels at 12 - 14
    // atoms as well - 256 sequences with per-atom complex modulated drives, 400 ns: 4 150 against 3 480 sim-us/s (12
    // atoms, k_traj), 2 640 against 1 790 (13), 1 520 against 810 (14, gauged k_ket); tools/cplx_bench.py.  Not for calls
    // with evaluation times at (nearly) every knot - the persistent kernels take their snapshots inside ONE launch where
    // this path closes a run per evaluation time - and not for quantum jumps (jumps on the device, one launch).
    // RYD_SPLIT_ALWAYS=0: dev A/B.
    static const bool always_env = dev_env_flag("RYD_SPLIT_ALWAYS", true);
    const double span_knots = h->n_knots > 1 ? (times[n_times - 1] - times[0]) / ((h->tknots.back() - h->tknots.front()) / (h->n_knots - 1)) : 0.0;
    // Round 5: evaluation times no longer close a run of k_split_reg (snapshots are stored from the registers inside it,
    // k_split_reg<.., SNAP>), so a call with evaluation times at every knot - evaluation_times="Full", the reference's
    // default - stays here too: 6 stages per knot interval in ONE launch per 64 knots (14 atoms, one sequence: 281 ms on
    // a closed run per knot, 364 ms on k_ket, tools/full_probe.py).  With the round-4 hook (snaps_outside) the old rule.
    if (!h->split14_auto && reg_shape && !h->mc && !h->split_fixed && always_env &&
        (!h->snaps_outside || 8.0 * (n_times - 1) <= span_knots))
      h->split14_auto = true;
  }
  // the in-place schemes split an exponential themselves and a Lanczos process takes whole
  // steps: both skip build_schedule's Taylor sub-stepping
  const bool in_place = ket_path(h) || (row_path(h) && !use_persistent_dm(h)) || krylov_selected(h, o) ||
                        split_selected(h, o);
  // multi-knot CF4 steps: not under the split-operator ket passes (their sub-steps stay inside a knot
  // interval); at most half a block of the split-operator master equation
  h->gauge_active = !h->general && !h->drive_real && h->gauge_ok && ket_path(h) && !krylov_selected(h, o) &&
                    !split_selected(h, o);
  // split-operator ket passes: sub-steps may span knots only with the 6th-order scheme (host_split.hpp); how far is
  // the controller's business (it measures the error), so the a-priori Magnus estimate is skipped there
  bool split_merge = split_selected(h, o) && split_s10_allowed(h);
  if (split_merge) {
    // ... and only where this call's schedule has something to merge: multi-knot steps over half of its time
    if (share < 0.0) share = merged_share();
    split_merge = share >= 0.5;
  }
  if (split_selected(h, o) && split_merge != h->split_s10) {
    h->split_s10 = split_merge;
    h->split_known = false;  // the controller's sub-step belongs to the other scheme
  }
  const int merge_cap = split_selected(h, o) ? (split_merge ? kSplitMergeMax : 1)
                        : (row_path(h) && !use_persistent_dm(h)) ? row_half_knots(h, o) : kMergeMax;
  std::vector<StepDesc> sched;
  h->sched_for_split = split_selected(h, o);
  // snapshot slot i-1 receives the state at times[i]
  for (int i = 1; i < n_times; ++i) {
    const size_t before = sched.size();
    build_schedule(h, times[i - 1], times[i], o, sched, in_place, merge_cap, split_merge);
    if (snaps) {
      if (sched.size() > before) {
        sched.back().snap = i - 1;
      } else {  // zero-length interval: the state is unchanged
        if (before == 0) {
          if ((rc = snapshot_copy(h, state, snaps + (size_t)(i - 1) * h->dim * h->B, st))) return rc;
        } else {
          // duplicate time after at least one step: flush what we have, copy, continue
          if ((rc = run_steps(h, state, sched, snaps, st, o))) return rc;
          sched.clear();
          if ((rc = snapshot_copy(h, state, snaps + (size_t)(i - 1) * h->dim * h->B, st))) return rc;
        }
      }
    }
  }
  if (host_timing)
    std::fprintf(stderr, "[ryd] solve over [%g, %g] us: %zu schedule steps built in %.3f ms on the host\n", times[0],
                 times[n_times - 1], sched.size(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
  return run_steps(h, state, sched, snaps, st, o);
}

extern "C" int ryd_evolve(ryd_handle* h, void* state_dev, double t0, double t1,
                          const ryd_opts* opts, void* stream) {
  if (!(t1 >= t0)) return fail(RYD_ERR_INVALID, "t1 < t0");
  const double times[2] = {t0, t1};
  return ryd_solve(h, state_dev, 2, times, nullptr, opts, stream);
}

extern "C" int ryd_set_collapse(ryd_handle* h, int32_t n_ops, const double* ops) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  if (h->general || h->cfg.mode != RYD_SESOLVE)
    return fail(RYD_ERR_INVALID, "collapse operators need a ket (sesolve) handle of the tuned path");
  if (n_ops < 0 || n_ops > MC_MAX_OPS || (n_ops > 0 && !ops))
    return fail(RYD_ERR_INVALID, "n_ops=%d out of range [0, %d]", n_ops, MC_MAX_OPS);
  HIPCHK(hipSetDevice(h->cfg.device));
  if (n_ops == 0) {
    h->mc = false;
    h->mc_n_ops = 0;
    h->mc_a = h->mc_b = 0.0;
    return RYD_OK;
  }
  // M = sum C^dag C must be diagonal: H_eff then only gains a real diagonal
  double m00 = 0, m11 = 0, m01r = 0, m01i = 0;
  for (int k = 0; k < n_ops; ++k) {
    const std::complex<double> c00(ops[8 * k + 0], ops[8 * k + 1]), c01(ops[8 * k + 2], ops[8 * k + 3]),
        c10(ops[8 * k + 4], ops[8 * k + 5]), c11(ops[8 * k + 6], ops[8 * k + 7]);
    m00 += std::norm(c00) + std::norm(c10);
    m11 += std::norm(c01) + std::norm(c11);
    const std::complex<double> x = std::conj(c00) * c01 + std::conj(c10) * c11;
    m01r += x.real();
    m01i += x.imag();
  }
  if (std::hypot(m01r, m01i) > 1e-13 * std::max(std::max(m00, m11), 1e-300))
    return fail(RYD_ERR_UNSUPPORTED,
                "sum C^dag C of the local collapse operators is not diagonal; use the "
                "master-equation solver for this noise model");
  const size_t B = (size_t)h->B, N = (size_t)h->N;
  const size_t n_dbl = 2 * B + 4 * N * B + 4 * B;
  const size_t bytes = n_dbl * sizeof(double) + B * sizeof(unsigned long long) +
                       MC_MAX_OPS * 4 * sizeof(cplx) + 3 * B * sizeof(int);
  if (!h->mc_pool) {
    HIPCHK(hipMalloc(&h->mc_pool, bytes));
    HIPCHK(hipMemset(h->mc_pool, 0, bytes));
    char* p = (char*)h->mc_pool;  // 16-byte objects first, then 8-byte, then 4-byte ones
    h->mc_ops_dev = (cplx*)p;     p += MC_MAX_OPS * 4 * sizeof(cplx);
    h->mcs.norm2 = (double*)p;    p += 2 * B * sizeof(double);
    h->mcs.red = (double*)p;      p += 4 * N * B * sizeof(double);
    h->mcs.target = (double*)p;   p += B * sizeof(double);
    h->mcs.refnorm = (double*)p;  p += B * sizeof(double);
    h->mcs.lastnorm = (double*)p; p += B * sizeof(double);
    h->mcs.scale = (double*)p;    p += B * sizeof(double);
    h->mc_seeds_dev = (unsigned long long*)p; p += B * sizeof(unsigned long long);
    h->mcs.flag = (int*)p;        p += B * sizeof(int);
    h->mcs.sel = (int*)p;         p += B * sizeof(int);
    h->mcs.count = (int*)p;
    h->mcs.seeds = h->mc_seeds_dev;
    h->mcs.ops = h->mc_ops_dev;
  }
  HIPCHK(hipMemcpy(h->mc_ops_dev, ops, (size_t)n_ops * 4 * sizeof(cplx), hipMemcpyHostToDevice));
  h->mcs.n_ops = n_ops;
  h->mc_n_ops = n_ops;
  h->mc_a = -0.5 * m00 * h->N;
  h->mc_b = 0.5 * (m00 - m11);
  h->mc = true;
  return RYD_OK;
}

extern "C" int ryd_mc_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                            void* out_dev, const uint64_t* seeds, const ryd_opts* opts,
                            void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!h->mc) return fail(RYD_ERR_STATE, "ryd_set_collapse has not been called");
  if (!state_dev || !seeds) return fail(RYD_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(h->mc_seeds_dev, seeds, (size_t)h->B * sizeof(unsigned long long),
                        hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(h->mcs.norm2, 0, 2 * (size_t)h->B * sizeof(double), st));
  hipLaunchKernelGGL(k_mc_norm, dim3(mc_blocks(h), h->B), dim3(256), 0, st, (const cplx*)state_dev,
                     h->nb, h->mcs.norm2);
  hipLaunchKernelGGL(k_mc_init, dim3((h->B + 127) / 128), dim3(128), 0, st, h->mcs, h->B, h->N);
  HIPCHK(hipGetLastError());
  h->mc_active = true;
  rc = ryd_solve(h, state_dev, n_times, times, out_dev, opts, stream);
  if (rc == RYD_OK) rc = snapshot_copy(h, (const cplx*)state_dev, (cplx*)state_dev, st);
  h->mc_active = false;
  return rc;
}

extern "C" int ryd_mc_get_jumps(ryd_handle* h, int32_t* counts, void* stream) {
  if (!h || !counts) return fail(RYD_ERR_INVALID, "null argument");
  if (!h->mc) return fail(RYD_ERR_STATE, "ryd_set_collapse has not been called");
  HIPCHK(hipSetDevice(h->cfg.device));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(counts, h->mcs.count, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost));
  return RYD_OK;
}

extern "C" int ryd_set_path(ryd_handle* h, int32_t force_generic) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  h->force_generic = (force_generic & 1) != 0;
  {
    const bool no = (force_generic & 2) != 0, nt = (force_generic & 4) != 0,
               ft = (force_generic & 8) != 0, fo = (force_generic & 16) != 0;
    h->no_ket = (force_generic & 32) != 0;
    h->force_ket = (force_generic & 64) != 0;
    h->no_split = (force_generic & 128) != 0;
    h->split_fixed = (force_generic & 256) != 0;
    h->split_no_loop = (force_generic & 512) != 0;
    h->no_merge = (force_generic & 1024) != 0;
    h->no_split14 = (force_generic & 16384) != 0;  // 14-atom batches stay on k_ket
    h->split_turns = (force_generic & 32768) != 0; // one-launch runs on the round-3 kernel (A/B)
    h->rows_ket = (force_generic & 65536) != 0;    // master-equation row passes on k_ket (A/B)
    h->snaps_outside = (force_generic & 131072) != 0;  // k_split_reg: a closed run per evaluation time (round 4, A/B)
    {
      const bool s6 = (force_generic & 8192) != 0;  // split-operator passes: S6, sub-steps end at every knot
      if (s6 != h->split_s6_only) { h->split_s6_only = s6; h->split_known = false; }
    }
    {
      const bool ns = (force_generic & 4096) != 0;  // general path: term-by-term kernel instead of the site-fused one
      if (ns != h->gen_no_sites) { h->gen_no_sites = ns; h->gen_sites_valid = false; }
      const bool nf = (force_generic & 262144) != 0;  // general path: k_gen_apply_sites (round 3) instead of the padded site tables
      if (nf != h->gen_no_fused) { h->gen_no_fused = nf; h->gen_sites_valid = false; }
    }
    {
      const bool small = (force_generic & 2048) != 0;
      if (small != h->split_small_tiles) { h->split_small_tiles = small; h->split_tilings.clear(); }
    }
    if (nt != h->no_tile14 || ft != h->force_tile14 || no != h->no_outer || fo != h->force_outer) {
      h->no_tile14 = nt;
      h->force_tile14 = ft;
      h->no_outer = no;
      h->force_outer = fo;
      plan_passes(h);
    }
  }
  return RYD_OK;
}

// ---------------------------------------------------------------------------
// ryd_general_solve_many: n independent small general-path problems, ONE launch
// ---------------------------------------------------------------------------
extern "C" int ryd_general_solve_many(ryd_handle** hs, int32_t n, void* const* states_dev, int32_t n_times,
                                      const double* times, void* const* outs_dev, const ryd_opts* opts,
                                      void* stream) {
  if (!hs || n < 1 || !states_dev || !times || n_times < 2) return fail(RYD_ERR_INVALID, "null argument / no problems");
  for (int i = 1; i < n_times; ++i)
    if (!(times[i] > times[i - 1])) return fail(RYD_ERR_INVALID, "times must be strictly increasing");
  ryd_opts o;
  std::memset(&o, 0, sizeof o);
  if (opts) o = *opts;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  for (int b = 0; b < n; ++b) {
    ryd_handle* h = hs[b];
    if ((rc = check_ready(h))) return rc;
    if (!h->general || h->B != 1 || h->dim > 4096 || h->gen_host.empty())
      return fail(RYD_ERR_UNSUPPORTED, "problem %d: ryd_general_solve_many takes general handles of one state "
                  "with at most 4096 entries", b);
    if (h->cfg.device != hs[0]->cfg.device) return fail(RYD_ERR_INVALID, "problem %d lives on another device", b);
    if (!states_dev[b]) return fail(RYD_ERR_INVALID, "problem %d: null state", b);
  }
  HIPCHK(hipSetDevice(hs[0]->cfg.device));
  std::vector<GenTrajArgs> args(n);
  for (int b = 0; b < n; ++b) {
    ryd_handle* h = hs[b];
    if (!h->bounds_valid) compute_bounds_general(h);
    std::vector<StepDesc> sched;
    for (int i = 1; i < n_times; ++i) {
      build_schedule(h, times[i - 1], times[i], o, sched, false, kMergeMax);
      if (outs_dev && outs_dev[b]) sched.back().snap = i - 1;
    }
    std::memset(&args[b], 0, sizeof(GenTrajArgs));
    if ((rc = fill_gen_traj_args(h, (cplx*)states_dev[b], sched, outs_dev ? (cplx*)outs_dev[b] : nullptr, st, args[b])))
      return rc;
    for (const StepDesc& d : sched) {
      h->stats.n_applications += d.order_a + d.order_b;
      h->stats.n_steps++;
    }
  }
  ryd_handle* h0 = hs[0];
  if (h0->many_cap < (size_t)n) {
    if (h0->many_args_dev) hipFree(h0->many_args_dev);
    h0->many_args_dev = nullptr;
    h0->many_cap = 0;
    HIPCHK(hipMalloc((void**)&h0->many_args_dev, (size_t)n * sizeof(GenTrajArgs)));
    h0->many_cap = (size_t)n;
  }
  HIPCHK(hipStreamSynchronize(st));  // an earlier launch may still read the argument table
  HIPCHK(hipMemcpyAsync(h0->many_args_dev, args.data(), (size_t)n * sizeof(GenTrajArgs), hipMemcpyHostToDevice, st));
  const size_t lds = 2 * 4096 * sizeof(cplx) + 2 * MAX_GEN_TERMS * sizeof(cplx);
  static bool attr_set[64] = {};
  const int dev = h0->cfg.device;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIPCHK(hipFuncSetAttribute((const void*)k_gen_traj_many, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(k_gen_traj_many, dim3((unsigned)n), dim3(1024), lds, st, (const GenTrajArgs*)h0->many_args_dev);
  HIPCHK(hipGetLastError());
  h0->stats.n_launches++;
  return RYD_OK;
}
