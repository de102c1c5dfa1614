// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// generator application
// ---------------------------------------------------------------------------
struct MixPoint {
  int idx1, idx2;
  double u1, u2, w1, w2;
};

static int find_interval(const ryd_handle* h, double t) {
  const int n_int = h->n_knots - 1;
  int i = int(std::upper_bound(h->tknots.begin(), h->tknots.end(), t) - h->tknots.begin()) - 1;
  return std::min(std::max(i, 0), n_int - 1);
}

static int launch_eval(ryd_handle* h, const MixPoint& m, hipStream_t st) {
  const int total = h->B * h->N;
  hipLaunchKernelGGL(k_eval_coefs, dim3((total + 3) / 4), dim3(256), 0, st, h->pp_dev,
                     h->n_knots - 1, h->desc_dev, h->dterms_dev, total, m.idx1, m.u1, m.w1, m.idx2, m.u2, m.w2,
                     h->coefs_dev);
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

static int timing_begin(ryd_handle* h, hipStream_t st, std::pair<hipEvent_t, hipEvent_t>& ev) {
  if (h->ev_free.empty()) {
    HIPCHK(hipEventCreate(&ev.first));
    HIPCHK(hipEventCreate(&ev.second));
  } else {
    ev = h->ev_free.back();
    h->ev_free.pop_back();
  }
  HIPCHK(hipEventRecord(ev.first, st));
  return RYD_OK;
}

static int launch_apply14(ryd_handle* h, const Apply14Args& B, hipStream_t st) {
  const size_t lds = (size_t)4 * 1024 * sizeof(cplx) + 2 * 128 * sizeof(double) + 2 * 16 * sizeof(double);
  dim3 grid((unsigned)(1ull << (h->nb - 14)), h->B);
  const bool se = h->cfg.mode == RYD_SESOLVE;
  const bool full = B.n_flip == 14;
#define RYD_LAUNCH14(M, RL, FL) \
  hipLaunchKernelGGL((k_apply14<M, RL, FL>), grid, dim3(1024), lds, st, B)
  if (h->drive_real) {
    if (se) { if (full) RYD_LAUNCH14(RYD_SESOLVE, true, true); else RYD_LAUNCH14(RYD_SESOLVE, true, false); }
    else { if (full) RYD_LAUNCH14(RYD_MESOLVE, true, true); else RYD_LAUNCH14(RYD_MESOLVE, true, false); }
  } else {
    if (se) { if (full) RYD_LAUNCH14(RYD_SESOLVE, false, true); else RYD_LAUNCH14(RYD_SESOLVE, false, false); }
    else { if (full) RYD_LAUNCH14(RYD_MESOLVE, false, true); else RYD_LAUNCH14(RYD_MESOLVE, false, false); }
  }
#undef RYD_LAUNCH14
  HIPCHK(hipGetLastError());
  return RYD_OK;
}

// out = post * (base + scale * G~ in); all passes.  `in` must differ from `out`
// unless single-element hazards are impossible (never used in place here).
// `kry_acc` (Lanczos, host_krylov.hpp): [4 B] accumulators (stride 4) of <in | out> (re, im) and |out|^2, reduced by the final pass -
// only for plans of ONE k_apply pass (kry_fusable)
static int apply_generator(ryd_handle* h, const cplx* in, const cplx* base, cplx* out,
                           double wmix, double scale, double shift, cplx post,
                           hipStream_t st, bool with_decay = false, double* kry_acc = nullptr) {
  if (!h->passes_valid) plan_passes(h);
  // Monte-Carlo H_eff: real diagonal, centred (the centre is a scalar factor
  // of the exponential, applied by exp_step through `post`)
  const double dec_b = with_decay ? wmix * h->mc_b : 0.0;
  const double dec_a = -0.5 * h->N * dec_b;
  const int np = (int)h->passes.size();
  for (int pi = 0; pi < np; ++pi) {
    const Pass& p = h->passes[pi];
    PassArgs A;
    std::memset(&A, 0, sizeof A);
    A.in = in;
    A.kin = pi > 0 ? h->kbuf : nullptr;
    A.kout = h->kbuf;
    A.final_pass = pi == np - 1;
    A.kry_acc = (A.final_pass && !p.use14) ? kry_acc : nullptr;
    A.base = A.final_pass ? base : nullptr;
    A.out = out;
    A.coefs = h->coefs_dev;
    A.e0 = h->e0_dev;
    A.e0_stride = h->e0_mats == 1 ? 0 : ((long long)1 << h->N);
    A.wmix = wmix;
    A.scale = scale;
    A.shift = shift;
    A.dec_a = dec_a;
    A.dec_b = dec_b;
    A.post = post;
    for (int i = 0; i < 4; ++i) { A.Sd[i] = h->Sd[i]; A.J[i] = h->J[i]; }
    A.tile = p.tile;
    A.outer = p.outer;
    A.N = h->N;
    A.nb = h->nb;
    A.T = p.T;
    A.n_flip = (int)p.flip_q.size();
    A.n_dbl = (int)p.dbl.size();
    A.include_diag = p.include_diag;
    for (int i = 0; i < A.n_flip; ++i) A.flip_q[i] = (signed char)p.flip_q[i];
    A.n_oflip = (int)p.oflip.size();
    for (int i = 0; i < A.n_oflip; ++i) A.oflip_p[i] = (signed char)p.oflip[i];
    for (int i = 0; i < A.n_dbl; ++i) {
      A.dbl_qb[i] = (signed char)p.dbl[i].first;
      A.dbl_qa[i] = (signed char)p.dbl[i].second;
    }
    const int TL = p.T >> 1, TH = p.T - TL;
    const size_t lds = ((size_t)1 << p.T) * sizeof(cplx) + ((1 << TL) + (1 << TH)) * sizeof(double) +
                       (2 * MAXF + MAXO) * sizeof(cplx);
    dim3 grid((unsigned)(1ull << p.n_outer_bits), h->B);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (h->timing) { int rc = timing_begin(h, st, ev); if (rc) return rc; }
    if (p.use14) {
      Apply14Args B;
      std::memset(&B, 0, sizeof B);
      B.in = in;
      B.base = A.final_pass ? base : nullptr;
      B.out = out;
      B.kout = A.final_pass ? nullptr : h->kbuf;
      B.coefs = h->coefs_dev;
      B.e0 = h->e0_dev;
      B.e0_stride = A.e0_stride;
      B.wmix = wmix;
      B.diag_scale = 1.0;
      B.scale = scale;
      B.shift = shift;
      B.dec_a = dec_a;
      B.dec_b = dec_b;
      B.post = post;
      for (int i = 0; i < 4; ++i) B.Sd[i] = h->Sd[i];
      B.N = h->N;
      B.nb = h->nb;
      B.n_flip = std::min(14, h->nb);
      int rc14 = launch_apply14(h, B, st);
      if (rc14) return rc14;
      if (h->timing) {
        HIPCHK(hipEventRecord(ev.second, st));
        h->ev_used.push_back(ev);
      }
      h->stats.n_launches++;
      continue;
    }
    {
      const bool wide = (size_t)grid.x * grid.y <= 512 && p.T >= 10;
      const bool pre = A.n_oflip > 0 && (1 << p.T) <= (wide ? 1024 : 512);
      if (h->cfg.mode == RYD_SESOLVE) {
        if (pre && wide) hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 1024, true>), grid, dim3(1024), lds, st, A);
        else if (pre) hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 512, true>), grid, dim3(512), lds, st, A);
        else if (wide) hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 1024>), grid, dim3(1024), lds, st, A);
        else hipLaunchKernelGGL((k_apply<RYD_SESOLVE, 512>), grid, dim3(512), lds, st, A);
      } else {
        if (pre && wide) hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 1024, true>), grid, dim3(1024), lds, st, A);
        else if (pre) hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 512, true>), grid, dim3(512), lds, st, A);
        else if (wide) hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 1024>), grid, dim3(1024), lds, st, A);
        else hipLaunchKernelGGL((k_apply<RYD_MESOLVE, 512>), grid, dim3(512), lds, st, A);
      }
    }
    HIPCHK(hipGetLastError());
    if (h->timing) {
      HIPCHK(hipEventRecord(ev.second, st));
      h->ev_used.push_back(ev);
    }
    h->stats.n_launches++;
  }
  h->stats.n_applications++;
  return RYD_OK;
}

static int check_ready(const ryd_handle* h) {
  if (!h) return fail(RYD_ERR_INVALID, "null handle");
  if (!h->pp_dev) return fail(RYD_ERR_STATE, "ryd_set_series has not been called");
  if (h->general) {
    if (h->gen_host.empty()) return fail(RYD_ERR_STATE, "ryd_general_add_term has not been called");
    return RYD_OK;
  }
  if (!h->desc_dev) return fail(RYD_ERR_STATE, "ryd_set_qubit_desc has not been called");
  return validate_extras(h);
}

struct MixPoint;
static int launch_eval_general(ryd_handle* h, const MixPoint& m, hipStream_t st);
static int apply_general(ryd_handle* h, const MixPoint& m, const cplx* in, const cplx* base,
                         cplx* out, double scale, hipStream_t st);

extern "C" int ryd_apply_generator(ryd_handle* h, const void* in_dev, void* out_dev, double t,
                                   void* stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!in_dev || !out_dev || in_dev == out_dev)
    return fail(RYD_ERR_INVALID, "in/out must be distinct non-null device pointers");
  HIPCHK(hipSetDevice(h->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  MixPoint m;
  m.idx1 = m.idx2 = find_interval(h, t);
  m.u1 = m.u2 = t - h->tknots[m.idx1];
  m.w1 = 1.0;
  m.w2 = 0.0;
  if (h->general) {
    if ((rc = launch_eval_general(h, m, st))) return rc;
    return apply_general(h, m, (const cplx*)in_dev, nullptr, (cplx*)out_dev, 1.0, st);
  }
  if ((rc = launch_eval(h, m, st))) return rc;
  return apply_generator(h, (const cplx*)in_dev, nullptr, (cplx*)out_dev, 1.0, 1.0, 0.0,
                         make_double2(1.0, 0.0), st);
}
