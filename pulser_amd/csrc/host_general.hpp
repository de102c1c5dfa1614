// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// General path (explicit CSR terms) - host side
// ---------------------------------------------------------------------------
extern "C" int ryd_general_create(const ryd_general_config* cfg, ryd_handle** out) {
  if (!cfg || !out) return fail(RYD_ERR_INVALID, "null argument");
  if (cfg->abi_version != RYD_ABI_VERSION)
    return fail(RYD_ERR_INVALID, "ABI version mismatch: caller %d, library %d", cfg->abi_version,
                RYD_ABI_VERSION);
  if (cfg->dim < 1 || cfg->dim > ((int64_t)1 << 26))
    return fail(RYD_ERR_INVALID, "dim=%lld out of range", (long long)cfg->dim);
  if (cfg->batch < 1 || cfg->batch > 65535) return fail(RYD_ERR_INVALID, "batch out of range");
  HIPCHK(hipSetDevice(cfg->device));
  ryd_handle* h = new ryd_handle();
  h->general = true;
  h->cfg.abi_version = cfg->abi_version;
  h->cfg.device = cfg->device;
  h->cfg.mode = RYD_SESOLVE;
  h->cfg.batch = cfg->batch;
  h->B = cfg->batch;
  h->dim = (size_t)cfg->dim;
  h->N = 0;
  h->nb = 0;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  hipError_t e;
  if ((e = hipMalloc((void**)&h->wA, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->wB, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_tcoef, MAX_GEN_TERMS * sizeof(cplx))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_terms_dev, MAX_GEN_TERMS * sizeof(GenTermDev))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_series_dev, MAX_GEN_TERMS * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_conj_dev, MAX_GEN_TERMS * sizeof(int))) != hipSuccess ||
      (e = hipMalloc((void**)&h->gen_scale_dev, MAX_GEN_TERMS * sizeof(cplx))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc (general path) failed: %s", hipGetErrorString(e));
  }
  *out = h;
  return RYD_OK;
}

// device copies of the per-term tables after a term was appended
static int gen_publish_terms(ryd_handle* h) {
  const int n = (int)h->gen_host.size();
  std::vector<GenTermDev> devs(n);
  std::vector<int> ser(n), cj(n);
  std::vector<cplx> sc(n);
  for (int i = 0; i < n; ++i) {
    devs[i] = h->gen_host[i].dev;
    ser[i] = h->gen_host[i].series;
    cj[i] = h->gen_host[i].conj;
    sc[i] = make_double2(h->gen_host[i].scale.real(), h->gen_host[i].scale.imag());
  }
  HIPCHK(hipMemcpy(h->gen_terms_dev, devs.data(), n * sizeof(GenTermDev), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_series_dev, ser.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_conj_dev, cj.data(), n * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->gen_scale_dev, sc.data(), n * sizeof(cplx), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  h->gen_sites_valid = false;
  return RYD_OK;
}

extern "C" int ryd_general_add_term(ryd_handle* h, int64_t nnz, const int32_t* row_ptr,
                                    const int32_t* col, const double* val, int32_t series,
                                    int32_t conj, double scale_re, double scale_im,
                                    double row_norm) {
  if (!h || !h->general) return fail(RYD_ERR_INVALID, "not a general-path handle");
  if (!row_ptr || (nnz > 0 && (!col || !val))) return fail(RYD_ERR_INVALID, "null argument");
  if ((int)h->gen_host.size() >= MAX_GEN_TERMS)
    return fail(RYD_ERR_INVALID, "too many terms (max %d)", MAX_GEN_TERMS);
  if (series < -1 || series >= std::max(h->n_series, 1) || (series >= 0 && h->n_series == 0))
    return fail(RYD_ERR_INVALID, "series index %d out of range (call ryd_set_series first)", series);
  if (row_ptr[0] != 0 || row_ptr[h->dim] != nnz)
    return fail(RYD_ERR_INVALID, "row_ptr does not describe %lld non-zeros", (long long)nnz);
  for (int64_t e = 0; e < nnz; ++e)
    if (col[e] < 0 || (size_t)col[e] >= h->dim)
      return fail(RYD_ERR_INVALID, "column index out of range at entry %lld", (long long)e);
  HIPCHK(hipSetDevice(h->cfg.device));
  GenTermHost t;
  t.series = series;
  t.conj = conj;
  t.scale = std::complex<double>(scale_re, scale_im);
  t.row_norm = row_norm;
  t.step_norm = row_norm;
  HIPCHK(hipMalloc((void**)&t.dev.row_ptr, (h->dim + 1) * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.col, std::max<int64_t>(nnz, 1) * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.val, std::max<int64_t>(nnz, 1) * sizeof(cplx)));
  HIPCHK(hipMemcpy((void*)t.dev.row_ptr, row_ptr, (h->dim + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (nnz > 0) {
    HIPCHK(hipMemcpy((void*)t.dev.col, col, nnz * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((void*)t.dev.val, val, nnz * sizeof(cplx), hipMemcpyHostToDevice));
  }
  h->gen_host.push_back(t);
  return gen_publish_terms(h);
}

// Matrix-free term  A = sum_g w_g embed(M on the digits with strides s_g[0 .. n_per))  of the vector
// index (digit = (index / stride) % local_dim): M is (local_dim^n_per)^2, given by its non-zeros.
extern "C" int ryd_general_add_local_term(ryd_handle* h, int32_t local_dim, int32_t n_per, int32_t n_groups,
                                          const int64_t* strides, const double* weights, int32_t nnz,
                                          const int32_t* rows, const int32_t* cols, const double* vals,
                                          int32_t series, int32_t conj, double scale_re, double scale_im,
                                          double row_norm) {
  if (!h || !h->general) return fail(RYD_ERR_INVALID, "not a general-path handle");
  if (!strides || !weights || !rows || !cols || !vals) return fail(RYD_ERR_INVALID, "null argument");
  if (local_dim < 2 || local_dim > 8 || (n_per != 1 && n_per != 2) || n_groups < 1 || nnz < 1)
    return fail(RYD_ERR_INVALID, "local term: local_dim=%d n_per=%d n_groups=%d nnz=%d out of range", local_dim,
                n_per, n_groups, nnz);
  if ((int)h->gen_host.size() >= MAX_GEN_TERMS)
    return fail(RYD_ERR_INVALID, "too many terms (max %d)", MAX_GEN_TERMS);
  if (series < -1 || series >= std::max(h->n_series, 1) || (series >= 0 && h->n_series == 0))
    return fail(RYD_ERR_INVALID, "series index %d out of range (call ryd_set_series first)", series);
  const int ld = n_per == 2 ? local_dim * local_dim : local_dim;
  for (int e = 0; e < nnz; ++e)
    if (rows[e] < 0 || rows[e] >= ld || cols[e] < 0 || cols[e] >= ld)
      return fail(RYD_ERR_INVALID, "local term: entry %d outside the %d x %d matrix", e, ld, ld);
  for (int g = 0; g < n_groups * n_per; ++g)
    if (strides[g] < 1 || (size_t)strides[g] * local_dim > h->dim)
      return fail(RYD_ERR_INVALID, "local term: stride %lld does not fit the vector", (long long)strides[g]);
  HIPCHK(hipSetDevice(h->cfg.device));
  GenTermHost t;
  t.series = series;
  t.conj = conj;
  t.scale = std::complex<double>(scale_re, scale_im);
  t.row_norm = row_norm;
  t.dev.kind = 1;
  t.dev.d = local_dim;
  t.dev.n_per = n_per;
  t.dev.n_groups = n_groups;
  t.dev.nnz = nnz;
  // digits of the vector index: dim = local_dim^n_dig
  int n_dig = 0;
  for (size_t v = 1; v < h->dim; v *= (size_t)local_dim) ++n_dig;
  {
    size_t v = 1;
    for (int i = 0; i < n_dig; ++i) v *= (size_t)local_dim;
    if (v != h->dim) return fail(RYD_ERR_INVALID, "local term: dim is not a power of local_dim=%d", local_dim);
  }
  if (h->gen_d && h->gen_d != local_dim)
    return fail(RYD_ERR_INVALID, "local term: local_dim=%d differs from the earlier terms' %d", local_dim, h->gen_d);
  const int bits = local_dim <= 4 ? 2 : 4;
  if (n_dig * bits > 64) return fail(RYD_ERR_UNSUPPORTED, "local term: %d digits do not fit the packed row digits", n_dig);
  std::vector<int> shifts((size_t)n_groups * n_per);
  for (int g = 0; g < n_groups * n_per; ++g) {
    int p = 0;
    int64_t v = 1;
    while (v < strides[g]) { v *= local_dim; ++p; }
    if (v != strides[g]) return fail(RYD_ERR_INVALID, "local term: stride %lld is not a power of local_dim", (long long)strides[g]);
    shifts[g] = p * bits;
  }
  // entries sorted by the row of M, with row starts
  std::vector<int> order(nnz), rstart(ld + 1, 0), ecol(nnz);
  for (int e = 0; e < nnz; ++e) { order[e] = e; rstart[rows[e] + 1]++; }
  for (int r = 0; r < ld; ++r) rstart[r + 1] += rstart[r];
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a] < rows[b]; });
  std::vector<cplx> sval(nnz);
  for (int e = 0; e < nnz; ++e) {
    ecol[e] = cols[order[e]];
    sval[e] = make_double2(vals[2 * order[e]], vals[2 * order[e] + 1]);
  }
  HIPCHK(hipMalloc((void**)&t.dev.strides, (size_t)n_groups * n_per * sizeof(long long)));
  HIPCHK(hipMalloc((void**)&t.dev.shifts, shifts.size() * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.weights, (size_t)n_groups * sizeof(double)));
  HIPCHK(hipMalloc((void**)&t.dev.rstart, rstart.size() * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.ecol, (size_t)nnz * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t.dev.val, (size_t)nnz * sizeof(cplx)));
  HIPCHK(hipMemcpy((void*)t.dev.strides, strides, (size_t)n_groups * n_per * sizeof(long long), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((void*)t.dev.shifts, shifts.data(), shifts.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((void*)t.dev.weights, weights, (size_t)n_groups * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((void*)t.dev.rstart, rstart.data(), rstart.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((void*)t.dev.ecol, ecol.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((void*)t.dev.val, sval.data(), (size_t)nnz * sizeof(cplx), hipMemcpyHostToDevice));
  h->gen_d = local_dim;
  h->gen_ndig = n_dig;
  t.h_strides.assign(strides, strides + (size_t)n_groups * n_per);
  t.h_weights.assign(weights, weights + n_groups);
  t.h_rows.assign(rows, rows + nnz);
  t.h_cols.assign(cols, cols + nnz);
  t.h_shifts = shifts;
  t.h_vals.resize(nnz);
  for (int e = 0; e < nnz; ++e) t.h_vals[e] = std::complex<double>(vals[2 * e], vals[2 * e + 1]);
  {
    // the term's own infinity norm (every group on its fullest local row at once): the caller's `row_norm` may be smaller
    // when it has bounded several time-independent terms JOINTLY (pulser_amd/general.py: _tighten_static_norms)
    std::vector<double> rs(ld, 0.0);
    for (int e = 0; e < nnz; ++e) rs[rows[e]] += std::abs(t.h_vals[e]);
    double wsum = 0.0;
    for (int g = 0; g < n_groups; ++g) wsum += std::fabs(weights[g]);
    t.step_norm = std::max(row_norm, wsum * *std::max_element(rs.begin(), rs.end()));
  }
  h->gen_host.push_back(t);
  return gen_publish_terms(h);
}

// Matrix-free diagonal term  A = diag(values)  (complex128[dim]).
extern "C" int ryd_general_add_diag_term(ryd_handle* h, const double* values, int32_t series, int32_t conj,
                                         double scale_re, double scale_im, double row_norm) {
  if (!h || !h->general) return fail(RYD_ERR_INVALID, "not a general-path handle");
  if (!values) return fail(RYD_ERR_INVALID, "null argument");
  if ((int)h->gen_host.size() >= MAX_GEN_TERMS)
    return fail(RYD_ERR_INVALID, "too many terms (max %d)", MAX_GEN_TERMS);
  if (series < -1 || series >= std::max(h->n_series, 1) || (series >= 0 && h->n_series == 0))
    return fail(RYD_ERR_INVALID, "series index %d out of range (call ryd_set_series first)", series);
  HIPCHK(hipSetDevice(h->cfg.device));
  GenTermHost t;
  t.series = series;
  t.conj = conj;
  t.scale = std::complex<double>(scale_re, scale_im);
  t.row_norm = row_norm;
  t.step_norm = row_norm;
  t.dev.kind = 2;
  HIPCHK(hipMalloc((void**)&t.dev.val, h->dim * sizeof(cplx)));
  HIPCHK(hipMemcpy((void*)t.dev.val, values, h->dim * sizeof(cplx), hipMemcpyHostToDevice));
  h->gen_host.push_back(t);
  return gen_publish_terms(h);
}

static void compute_bounds_general(ryd_handle* h) {
  const int n_int = h->n_knots - 1;
  h->bd_drive.assign(n_int, 0.0);
  h->bd_pos.assign(n_int, 0.0);
  h->bd_neg.assign(n_int, 0.0);
  h->bd_curv.assign(n_int, 0.0);
  // Two bounds per interval.  bd_drive (sum of the callers' row norms, jointly tightened for the time-independent terms):
  // the argument of the Taylor series, i.e. its degree.  bd_step (sum of every term's OWN infinity norm, what bd_drive was
  // up to round 5): the size of the CF4 steps - the general path has no estimate of the 4th-order Magnus error, the rule
  // "argument near 1 against the crude bound" is what its oracle tests were passed with (host_sched.hpp), so a tighter
  // norm lowers the polynomial degree and leaves the steps where they were.
  h->bd_step.assign(n_int, 0.0);
  for (const GenTermHost& t : h->gen_host) {
    const double w = std::abs(t.scale) * t.row_norm, ws = std::abs(t.scale) * std::max(t.step_norm, t.row_norm);
    for (int i = 0; i < n_int; ++i) {
      if (t.series >= 0) {
        h->bd_drive[i] += w * h->s_abs[(size_t)t.series * n_int + i];
        h->bd_step[i] += ws * h->s_abs[(size_t)t.series * n_int + i];
        h->bd_curv[i] += w * h->s_curv[(size_t)t.series * n_int + i];
      } else {
        h->bd_drive[i] += w;
        h->bd_step[i] += ws;
      }
    }
  }
  h->e0_min = h->e0_max = 0.0;
  h->uniform_real_drive = false;
  h->bounds_valid = true;
}

struct GenSiteBuild {
  GenSite g;
  std::map<std::pair<int, int>, std::vector<std::pair<int, std::complex<double>>>> ent;  // (R, C) -> contributions
};

// LDS budget of k_gen_apply_fused (CDNA4: 160 KiB per CU; tables + optionally the whole vector)
static const size_t kGenFusedLds = 150 * 1024;

// Padded site tables of k_gen_apply_fused from the per-site entry maps: per site and local row the diagonal entry
// and K = (fullest row's off-diagonal count) padded (value, row offset) entries; sites ordered by K.
static int gen_build_fused(ryd_handle* h, const std::vector<GenSiteBuild>& sb, const std::vector<int>& diag_terms) {
  h->gen_fused_ok = false;
  if (h->gen_no_fused || sb.empty() || h->dim > ((size_t)1 << 30)) return RYD_OK;
  const int d = h->gen_d;
  struct Row { std::vector<std::pair<int, const std::vector<std::pair<int, std::complex<double>>>*>> off;
               const std::vector<std::pair<int, std::complex<double>>>* diag = nullptr; };
  std::vector<int> order(sb.size()), Ks(sb.size(), 0);
  std::vector<std::vector<Row>> rows(sb.size());
  for (size_t i = 0; i < sb.size(); ++i) {
    order[i] = (int)i;
    rows[i].resize(sb[i].g.ld);
    for (auto& kv : sb[i].ent) {
      Row& r = rows[i][kv.first.first];
      if (kv.first.first == kv.first.second) r.diag = &kv.second;
      else r.off.push_back({kv.first.second, &kv.second});
    }
    for (const Row& r : rows[i]) Ks[i] = std::max(Ks[i], (int)r.off.size());
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return Ks[a] < Ks[b]; });
  std::vector<GenSiteF> sites;
  std::vector<int> delta, cstart(1, 0), cterm;
  std::vector<cplx> cval;
  std::vector<std::vector<std::pair<int, std::complex<double>>>> diag_contribs;
  GenFusedDev F{};
  int E = 0, Dg = 0;
  for (int oi = 0; oi < (int)order.size(); ++oi) {
    const int i = order[oi];
    const GenSite& g = sb[i].g;
    const int K = Ks[i];
    if (F.n_groups == 0 || F.gK[F.n_groups - 1] != K) {
      if (F.n_groups == GEN_FUSED_MAX_GROUPS) return RYD_OK;  // (more distinct row lengths than a Hamiltonian of site sums has)
      F.gK[F.n_groups] = K;
      F.gBegin[F.n_groups] = oi;
      F.gEnd[F.n_groups] = oi;
      F.n_groups++;
    }
    F.gEnd[F.n_groups - 1] = oi + 1;
    GenSiteF sf;
    sf.shift0 = g.shift0;
    sf.shift1 = g.n_per == 2 ? g.shift1 : 0;
    sf.mask1 = g.n_per == 2 ? (d <= 4 ? 3 : 15) : 0;
    sf.mul = g.n_per == 2 ? d : 1;
    sf.ent_off = E;
    sf.diag_off = Dg;
    for (int R = 0; R < g.ld; ++R) {
      const Row& r = rows[i][R];
      const int a = g.n_per == 2 ? R / d : R, b = g.n_per == 2 ? R % d : 0;
      for (int k = 0; k < K; ++k) {
        if (k < (int)r.off.size()) {
          const int Cc = r.off[k].first;
          const int c0 = g.n_per == 2 ? Cc / d : Cc, c1 = g.n_per == 2 ? Cc % d : 0;
          const long long dl = (long long)(c0 - a) * g.s0 + (long long)(c1 - b) * g.s1;
          delta.push_back((int)dl);
          for (auto& c : *r.off[k].second) {
            cterm.push_back(c.first);
            cval.push_back(make_double2(c.second.real(), c.second.imag()));
          }
        } else {
          delta.push_back(0);  // padding: value 0 (no contributions) x the row's own amplitude
        }
        cstart.push_back((int)cterm.size());
      }
      diag_contribs.push_back(r.diag ? *r.diag : std::vector<std::pair<int, std::complex<double>>>());
    }
    E += g.ld * K;
    Dg += g.ld;
    sites.push_back(sf);
  }
  for (auto& dc : diag_contribs) {  // the diagonal entries follow the E off-diagonal ones
    for (auto& c : dc) {
      cterm.push_back(c.first);
      cval.push_back(make_double2(c.second.real(), c.second.imag()));
    }
    cstart.push_back((int)cterm.size());
  }
  // LDS: entries, the four waves' partial sums (4 x 64 x 2 complex), row offsets (site descriptors are scalar loads)
  const size_t lds_tables = (size_t)(E + Dg) * sizeof(cplx) + 4 * 64 * 2 * sizeof(cplx) + (size_t)((E + 3) & ~3) * sizeof(int) + 16;
  if (lds_tables > kGenFusedLds) return RYD_OK;
  const bool xlds = lds_tables + h->dim * sizeof(cplx) <= kGenFusedLds;
  const size_t b_cval = std::max<size_t>(cval.size(), 1) * sizeof(cplx), b_mv = (size_t)(E + Dg) * sizeof(cplx),
               b_sites = sites.size() * sizeof(GenSiteF), b_delta = std::max<size_t>(delta.size(), 1) * sizeof(int),
               b_cs = cstart.size() * sizeof(int), b_ct = std::max<size_t>(cterm.size(), 1) * sizeof(int);
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  if (h->gen_fused_pool) hipFree(h->gen_fused_pool);
  h->gen_fused_pool = nullptr;
  HIPCHK(hipMalloc(&h->gen_fused_pool, up16(b_cval) + up16(b_mv) + up16(b_sites) + up16(b_delta) + up16(b_cs) + up16(b_ct)));
  char* p = (char*)h->gen_fused_pool;
  auto put = [&](const void* src, size_t have, size_t bytes) -> void* {
    void* dst = p;
    if (src && have) hipMemcpy(dst, src, have, hipMemcpyHostToDevice);
    p += up16(bytes);
    return dst;
  };
  F.contrib_val = (const cplx*)put(cval.data(), cval.size() * sizeof(cplx), b_cval);
  F.mvals = (cplx*)put(nullptr, 0, b_mv);
  F.sites = (const GenSiteF*)put(sites.data(), b_sites, b_sites);
  F.delta = (const int*)put(delta.data(), delta.size() * sizeof(int), b_delta);
  F.contrib_start = (const int*)put(cstart.data(), b_cs, b_cs);
  F.contrib_term = (const int*)put(cterm.data(), cterm.size() * sizeof(int), b_ct);
  F.n_sites = (int)sites.size();
  F.E = E;
  F.Dg = Dg;
  h->gen_fused = F;
  h->gen_fused_xlds = xlds;
  h->gen_fused_lds = lds_tables + (xlds ? h->dim * sizeof(cplx) : 0);
  if (h->gen_diag_terms_dev) hipFree(h->gen_diag_terms_dev);
  h->gen_diag_terms_dev = nullptr;
  h->gen_n_diag = (int)diag_terms.size();
  h->gen_diag_host = diag_terms;
  HIPCHK(hipMalloc((void**)&h->gen_diag_terms_dev, std::max<size_t>(diag_terms.size(), 1) * sizeof(int)));
  if (!diag_terms.empty())
    HIPCHK(hipMemcpy(h->gen_diag_terms_dev, diag_terms.data(), diag_terms.size() * sizeof(int), hipMemcpyHostToDevice));
  {
    static bool attr[64] = {};
    const int dev = h->cfg.device;
    if (dev < 0 || dev >= 64 || !attr[dev]) {
      HIPCHK(hipFuncSetAttribute((const void*)k_gen_apply_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(hipFuncSetAttribute((const void*)k_gen_apply_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev >= 0 && dev < 64) attr[dev] = true;
    }
  }
  h->gen_fused_ok = true;
  return RYD_OK;
}

// Site table of a matrix-free handle (every term local or diagonal): which terms / groups act on which
// site, the union of their sparsity patterns per site and, per pattern entry, the list of (term, weight x
// entry) contributions that k_gen_sitevals adds up per exponential.
static int gen_build_sites(ryd_handle* h) {
  h->gen_sites_valid = true;
  h->gen_sites_ok = false;
  h->gen_fused_ok = false;
  if (h->gen_no_sites || h->gen_d == 0) return RYD_OK;
  for (const GenTermHost& t : h->gen_host)
    if (t.dev.kind == 0) return RYD_OK;  // explicit CSR terms: the term-by-term kernel
  std::vector<GenSiteBuild> sb;
  std::map<std::tuple<int, long long, long long>, int> index;
  std::vector<int> diag_terms;
  for (int ti = 0; ti < (int)h->gen_host.size(); ++ti) {
    const GenTermHost& t = h->gen_host[ti];
    if (t.dev.kind == 2) { diag_terms.push_back(ti); continue; }
    const int np = t.dev.n_per, ld = np == 2 ? t.dev.d * t.dev.d : t.dev.d;
    for (int g = 0; g < t.dev.n_groups; ++g) {
      const long long s0 = t.h_strides[(size_t)g * np], s1 = np == 2 ? t.h_strides[(size_t)g * np + 1] : 0;
      auto key = std::make_tuple(np, s0, s1);
      auto it = index.find(key);
      if (it == index.end()) {
        GenSiteBuild b;
        b.g.s0 = s0; b.g.s1 = s1;
        b.g.shift0 = t.h_shifts[(size_t)g * np];
        b.g.shift1 = np == 2 ? t.h_shifts[(size_t)g * np + 1] : 0;
        b.g.n_per = np; b.g.ld = ld; b.g.rs_off = 0;
        it = index.emplace(key, (int)sb.size()).first;
        sb.push_back(std::move(b));
      }
      GenSiteBuild& b = sb[it->second];
      for (size_t e = 0; e < t.h_vals.size(); ++e)
        b.ent[{t.h_rows[e], t.h_cols[e]}].push_back({ti, t.h_weights[g] * t.h_vals[e]});
    }
  }
  {
    int rc = gen_build_fused(h, sb, diag_terms);
    if (rc) return rc;
    if (h->gen_fused_ok) return RYD_OK;  // (the round-3 tables are not needed)
  }
  std::vector<GenSite> sites;
  std::vector<int> prs, pcol, cstart(1, 0), cterm;
  std::vector<cplx> cval;
  for (GenSiteBuild& b : sb) {
    b.g.rs_off = (int)prs.size();
    int R = 0;
    prs.push_back((int)pcol.size());
    for (auto& kv : b.ent) {  // sorted by (R, C)
      while (R < kv.first.first) { prs.push_back((int)pcol.size()); ++R; }
      pcol.push_back(kv.first.second);
      for (auto& c : kv.second) {
        cterm.push_back(c.first);
        cval.push_back(make_double2(c.second.real(), c.second.imag()));
      }
      cstart.push_back((int)cterm.size());
    }
    while (R < b.g.ld) { prs.push_back((int)pcol.size()); ++R; }
    sites.push_back(b.g);
  }
  const int P = (int)pcol.size();
  if (P == 0 || P > GEN_SITES_LDS_MAX || sites.size() > 512) return RYD_OK;  // (huge pattern: keep the generic kernel)
  // one device pool: 16-byte objects first
  const size_t b_cval = cval.size() * sizeof(cplx), b_mv = (size_t)P * sizeof(cplx), b_sites = sites.size() * sizeof(GenSite),
               b_prs = prs.size() * sizeof(int), b_pcol = (size_t)P * sizeof(int), b_cs = cstart.size() * sizeof(int),
               b_ct = cterm.size() * sizeof(int);
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t total = up16(b_cval) + up16(b_mv) + up16(b_sites) + up16(b_prs) + up16(b_pcol) + up16(b_cs) + up16(b_ct);
  if (h->gen_sites_pool) hipFree(h->gen_sites_pool);
  h->gen_sites_pool = nullptr;
  HIPCHK(hipMalloc(&h->gen_sites_pool, total));
  char* p = (char*)h->gen_sites_pool;
  auto put = [&](const void* src, size_t bytes) -> void* {
    void* dst = p;
    if (src && bytes) hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    p += up16(bytes);
    return dst;
  };
  GenSitesDev& S = h->gen_sites;
  S.contrib_val = (const cplx*)put(cval.data(), b_cval);
  S.mvals = (cplx*)put(nullptr, b_mv);
  S.sites = (const GenSite*)put(sites.data(), b_sites);
  S.pat_rstart = (const int*)put(prs.data(), b_prs);
  S.pat_col = (const int*)put(pcol.data(), b_pcol);
  S.contrib_start = (const int*)put(cstart.data(), b_cs);
  S.contrib_term = (const int*)put(cterm.data(), b_ct);
  S.n_sites = (int)sites.size();
  S.P = P;
  S.n_rs = (int)prs.size();
  if (h->gen_diag_terms_dev) hipFree(h->gen_diag_terms_dev);
  h->gen_diag_terms_dev = nullptr;
  h->gen_n_diag = (int)diag_terms.size();
  HIPCHK(hipMalloc((void**)&h->gen_diag_terms_dev, std::max<size_t>(diag_terms.size(), 1) * sizeof(int)));
  if (!diag_terms.empty())
    HIPCHK(hipMemcpy(h->gen_diag_terms_dev, diag_terms.data(), diag_terms.size() * sizeof(int), hipMemcpyHostToDevice));
  h->gen_sites_ok = true;
  return RYD_OK;
}

static int apply_general(ryd_handle* h, const MixPoint& m, const cplx* in, const cplx* base,
                         cplx* out, double scale, hipStream_t st) {
  const int n = (int)h->gen_host.size();
  if (n == 0) return fail(RYD_ERR_STATE, "no terms: call ryd_general_add_term first");
  if (h->gen_fused_ok) {
    GenFusedArgs A;
    A.in = in;
    A.base = base;
    A.out = out;
    A.tcoef = h->gen_tcoef;
    A.terms = h->gen_terms_dev;
    A.diag_terms = h->gen_diag_terms_dev;
    A.F = h->gen_fused;
    for (int k = 0; k < 4; ++k) {
      const bool have = k < h->gen_n_diag;
      A.diag_idx[k] = have ? h->gen_diag_host[k] : 0;
      A.diag_val[k] = have ? h->gen_host[h->gen_diag_host[k]].dev.val : nullptr;
    }
    A.dim = (long long)h->dim;
    A.n_diag = h->gen_n_diag;
    A.d = h->gen_d;
    A.n_dig = h->gen_ndig;
    A.scale = scale;
    dim3 grid((unsigned)((h->dim + GEN_FUSED_ROWS - 1) / GEN_FUSED_ROWS), h->B);
    if (h->gen_fused_xlds) hipLaunchKernelGGL(k_gen_apply_fused<true>, grid, dim3(256), h->gen_fused_lds, st, A);
    else hipLaunchKernelGGL(k_gen_apply_fused<false>, grid, dim3(256), h->gen_fused_lds, st, A);
    HIPCHK(hipGetLastError());
    h->stats.n_launches++;
    h->stats.n_applications++;
    return RYD_OK;
  }
  if (h->gen_sites_ok) {
    GenSiteArgs A;
    A.in = in;
    A.base = base;
    A.out = out;
    A.tcoef = h->gen_tcoef;
    A.terms = h->gen_terms_dev;
    A.diag_terms = h->gen_diag_terms_dev;
    A.S = h->gen_sites;
    A.dim = (long long)h->dim;
    A.n_diag = h->gen_n_diag;
    A.d = h->gen_d;
    A.n_dig = h->gen_ndig;
    A.scale = scale;
    const size_t lds = (size_t)A.S.P * (sizeof(cplx) + sizeof(int)) + (size_t)((A.S.n_rs + 3) & ~3) * sizeof(int) +
                       (size_t)A.S.n_sites * sizeof(GenSite) + 16;
    dim3 grid((unsigned)((h->dim + 255) / 256), h->B);
    hipLaunchKernelGGL(k_gen_apply_sites, grid, dim3(256), lds, st, A);
    HIPCHK(hipGetLastError());
    h->stats.n_launches++;
    h->stats.n_applications++;
    return RYD_OK;
  }
  GenArgs A;
  A.in = in;
  A.base = base;
  A.out = out;
  A.tcoef = h->gen_tcoef;
  A.terms = h->gen_terms_dev;
  A.dim = (long long)h->dim;
  A.n_terms = n;
  A.d = h->gen_d;
  A.n_dig = h->gen_ndig;
  A.scale = scale;
  dim3 grid((unsigned)((h->dim + 255) / 256), h->B);
  hipLaunchKernelGGL(k_gen_apply, grid, dim3(256), 0, st, A);
  HIPCHK(hipGetLastError());
  h->stats.n_launches++;
  h->stats.n_applications++;
  (void)m;
  return RYD_OK;
}

static int launch_eval_general(ryd_handle* h, const MixPoint& m, hipStream_t st) {
  const int n = (int)h->gen_host.size();
  if (!h->gen_sites_valid) {
    int rc = gen_build_sites(h);
    if (rc) return rc;
  }
  if (h->gen_fused_ok) {  // coefficients and the padded site matrices of this exponential in one launch
    hipLaunchKernelGGL(k_gen_coefs_fused, dim3(1), dim3(256), 0, st, h->pp_dev, h->n_knots - 1, h->gen_series_dev,
                       h->gen_conj_dev, h->gen_scale_dev, n, m.idx1, m.u1, m.w1, m.u2, m.w2, h->gen_tcoef, h->gen_fused);
    HIPCHK(hipGetLastError());
    return RYD_OK;
  }
  hipLaunchKernelGGL(k_gen_coefs, dim3((n + 63) / 64), dim3(64), 0, st, h->pp_dev, h->n_knots - 1,
                     h->gen_series_dev, h->gen_conj_dev, h->gen_scale_dev, n, m.idx1, m.u1, m.w1,
                     m.u2, m.w2, h->gen_tcoef);
  HIPCHK(hipGetLastError());
  if (h->gen_sites_ok) {  // the site matrices of this exponential
    hipLaunchKernelGGL(k_gen_sitevals, dim3((h->gen_sites.P + 127) / 128), dim3(128), 0, st, h->gen_sites, h->gen_tcoef);
    HIPCHK(hipGetLastError());
  }
  return RYD_OK;
}
