// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// State of the split-operator step-size controller for one kind of step (host_split.hpp: run_split)
struct SplitCtl {
  double tau = 1e300;     // working sub-step (us); 1e300 = never measured: whole steps until the first check
  double rate = 0.0;      // last measured local error per us ...
  double rate_tau = 0.0;  // ... at sub-steps of this length (us)
  double amp = 0.0;       // drive bound at the last check
  int since = 0;          // knot intervals of this kind since the last check
  int period = 0;         // knot intervals until the next periodic check (0: kSplitCheckFirst; doubles up to kSplitCheckEvery)
  double len_since = 0.0; // simulated time (us) this kind's sub-steps have covered since its last check
  bool known = false;
};

struct Pass {
  Segs tile, outer;
  int T = 0;
  int n_outer_bits = 0;
  std::vector<int> flip_q;
  std::vector<int> oflip;  // global bits flipped through partner tiles (single-pass plan)
  std::vector<std::pair<int, int>> dbl;  // (qb, qa)
  bool include_diag = false;
  bool use14 = false;  // pass 0 on 2^14 register tiles (k_apply14)
};

struct GenTermHost {
  GenTermDev dev{};
  int series = -1, conj = 0;
  std::complex<double> scale{1.0, 0.0};
  double row_norm = 0.0;
  double step_norm = 0.0;  // what sizes the CF4 steps: sum_g |w_g| x the largest row sum of |M| (>= row_norm, see compute_bounds_general)
  // host copy of a matrix-free local term (the site-fused application is assembled from these)
  std::vector<long long> h_strides;
  std::vector<double> h_weights;
  std::vector<int> h_rows, h_cols, h_shifts;
  std::vector<std::complex<double>> h_vals;
};

struct ryd_handle {
  ryd_config cfg{};
  int N = 0, nb = 0, B = 1, T = 12;
  size_t dim = 0;  // elements per state (2^nb)
  // tables
  int n_series = 0, n_knots = 0;
  std::vector<double> tknots;
  std::vector<std::complex<double>> pp_host;  // [series][int][4]
  std::vector<double> s_abs, s_pos, s_neg;    // per series, per interval bounds
  std::vector<double> s_curv;                 // |quadratic| dt^2 + |cubic| dt^3 per interval
  cplx* pp_dev = nullptr;
  std::vector<ryd_qdesc> desc_host;
  ryd_qdesc* desc_dev = nullptr;
  std::vector<ryd_dterm> dterms_host;  // extra detuning terms (ryd_qdesc.extra)
  ryd_dterm* dterms_dev = nullptr;
  std::vector<double> bd_drive, bd_pos, bd_neg;  // per interval, max over batch
  std::vector<double> bd_step;  // general path: the bound that sizes the steps (host_general.hpp)
  std::vector<double> bd_curv;                   // per interval: non-linearity of H(t)
  // multi-knot CF4 steps (host_sched.hpp): per-series slope bounds, per-knot "the next piece is the same
  // polynomial", per-interval per-ATOM maxima of |c|, |dc/dt|, |delta|, |d delta/dt| over the batch
  std::vector<double> s_der;
  std::vector<char> join_ok;  // [n_int - 1]: pieces i and i + 1 of EVERY series coincide as polynomials
  std::vector<char> lin_ok;   // [n_int]: piece i of EVERY series is linear in time (ramps, plateaus), to 1e-10 of the series' scale
  std::vector<double> bd_c1, bd_dc, bd_dl, bd_ddl;
  double u_rowsum = 0.0;      // max_i sum_j |U_ij|
  bool bounds_valid = false;
  double* e0_dev = nullptr;
  int e0_mats = 0;
  double e0_min = 0, e0_max = 0;
  double* coefs_dev = nullptr;
  cplx Sd[4]{}, J[4]{};
  double diss_norm = 0.0;
  bool has_dbl = false;
  // work vectors
  cplx *wA = nullptr, *wB = nullptr, *kbuf = nullptr;
  cplx* wC = nullptr;              // second checkpoint buffer of the split-operator controller (allocated on first use; round 6)
  double* split_err_pin = nullptr; // [B] pinned host copy of split_err (the one read-back of a check)
  // one-shot options of the next split_run on the register-resident kernel (run_split's check; consumed by the launch)
  cplx *fuse_dst = nullptr, *fuse_dst2 = nullptr;
  const cplx* fuse_cmp = nullptr;
  bool fuse_done = false;
  std::vector<Pass> passes;
  bool passes_valid = false;
  StepDesc* sched_dev = nullptr;
  size_t sched_cap = 0;
  KetStep* ksched_dev = nullptr;  // schedule of the register-resident ket kernel
  double* ftab_dev = nullptr;     // elementwise dissipator factor tables of the row path
  cplx* kry_V = nullptr;          // Krylov basis (m + 1 vectors), allocated on first use
  int kry_cap = 0;
  void* kry_pool = nullptr;
  KryScalars kry{};
  double* kry_sq = nullptr;       // [B][KRY_MAX_M + 1] inside kry_pool (KryScalars.sq when the iteration is fused)
  size_t ksched_cap = 0;
  // split-operator ket path (host_split.hpp)
  std::vector<Pass> split_tilings;
  double* split_coefs = nullptr;  // [stages][B][N][4]
  size_t split_cap = 0;
  double* split_err = nullptr;    // [B] local-error accumulators
  cplx* rows_chk = nullptr;       // [2][B][2^N]: the probed rows of the split-operator master equation (rows_split_probe)
  int* rows_idx_dev = nullptr;    // [B]
  bool no_split = false;          // test hook: keep the Taylor polynomial for 15+ atoms
  bool split_fixed = false;       // test hook: no step-size control (sub-step = schedule step)
  bool no_merge = false;          // test hook: CF4 steps never span more than one knot interval
  bool split_no_loop = false;     // test hook: 12-atom kets pass by pass instead of the one-launch loop
  bool split_small_tiles = false; // test / bench hook: keep 2^12 tiles for every register size
  bool split14_auto = false;      // this solve: a 14-atom batch goes to k_split14_loop instead of k_ket (host_step.hpp)
  bool no_split14 = false;        // test / bench hook: keep 14-atom batches on k_ket
  bool rows_ket = false;          // test / bench hook: row passes of the split-operator master equation on k_ket
  bool split_turns = false;       // test / bench hook: 14-atom one-launch runs on k_split14_loop (two LDS turns per stage,
                                  // round 3) instead of k_split_reg
  bool sched_for_split = false;   // the schedule being built is run by the split-operator path (host_sched.hpp: dev probe)
  bool snaps_outside = false;     // test / bench hook: every evaluation time closes a run of k_split_reg (round 4) instead of a
                                  // snapshot taken inside the run
  bool split_s10 = false;         // scheme of the current split-operator solve (host_step.hpp decides per call)
  bool split_s6_only = false;     // test / bench hook: the 4th-order scheme with one-knot sub-steps (round 2)
  void* many_args_dev = nullptr;  // ryd_general_solve_many: argument table of the batched launch (first handle)
  size_t many_cap = 0;
  bool split_known = false;       // controller state below is valid for the current tables
  SplitCtl split_ctl[2];          // the controller's state by kind of step (run_split: kind_of)
  double split_eps = 0.0;         // tolerance the state was measured for
  double split_since_len = 0.0;   // simulated time (us) covered since the last check
  const void* split_state_last = nullptr;  // ... and the state buffer it advanced
  double split_t_last = -1e300;   // end time (us) of the last split-operator solve: the state above belongs to its continuation
  // general path (explicit CSR terms)
  bool general = false;
  std::vector<GenTermHost> gen_host;
  int gen_d = 0, gen_ndig = 0;  // local dimension / digits of the vector index (matrix-free terms)
  cplx* gen_tcoef = nullptr;
  // site-fused application of matrix-free handles (k_gen_apply_sites)
  bool gen_sites_valid = false, gen_sites_ok = false;
  GenSitesDev gen_sites{};
  void* gen_sites_pool = nullptr;
  int* gen_diag_terms_dev = nullptr;
  int gen_n_diag = 0;
  bool gen_no_sites = false;  // test hook: keep the term-by-term kernel
  // padded site tables (k_gen_apply_fused, round 6)
  bool gen_fused_ok = false, gen_fused_xlds = false;
  bool gen_no_fused = false;  // test / A-B hook: keep k_gen_apply_sites (round 3)
  GenFusedDev gen_fused{};
  void* gen_fused_pool = nullptr;
  size_t gen_fused_lds = 0;
  std::vector<int> gen_diag_host;  // indices of the diagonal (kind 2) terms
  GenTermDev* gen_terms_dev = nullptr;
  int* gen_series_dev = nullptr;
  int* gen_conj_dev = nullptr;
  cplx* gen_scale_dev = nullptr;
  bool auto_tile = true;       // tile_bits == 0: tile size chosen by the library
  bool force_generic = false;
  bool no_tile14 = false;      // test hook: disable k_apply14 / the Hermitian mesolve path
  bool force_tile14 = false;   // test hook: use them even when too few tiles fill the GPU
  bool no_outer = false;       // test hook: disable the single-pass partner-tile plan
  bool force_outer = false;    // test hook: use it whatever the size of the state
  bool no_ket = false;         // test hook: disable the register-resident ket kernel / split-operator rows
  bool force_ket = false;      // test hook: use them from 10 atoms on (instead of 14 / 12)
  bool drive_real = false;     // every drive series is real-valued
  bool gauge_ok = false;       // complex drives can be gauged away inside k_ket (KET_GAUGE)
  double gauge_eps2 = 0.0;     // |c|^2 below which a drive has no direction of its own
  std::vector<double> bd_gpos, bd_gneg;  // per interval: sum over atoms of the bounds of +-theta' (batch maximum)
  std::vector<double> bd_gvar;           // per interval: largest change of theta' inside it (any series in use)
  bool gauge_active = false;             // this solve runs KET_GAUGE (set by ryd_solve)
  bool uniform_real_drive = false;  // persistent-kernel MODEL 1 applies
  // Monte-Carlo wavefunction mode (sesolve handles with ryd_set_collapse)
  bool mc = false;         // collapse operators set: H_eff carries -(i/2) sum C^dag C
  bool mc_active = false;  // inside ryd_mc_solve: jump bookkeeping after every step
  int mc_n_ops = 0;
  double mc_a = 0.0, mc_b = 0.0;  // real diagonal of G_eff: mc_a + mc_b * popc(index)
  void* mc_pool = nullptr;        // one allocation behind McState
  McState mcs{};
  unsigned long long* mc_seeds_dev = nullptr;
  cplx* mc_ops_dev = nullptr;
  int mc_slot = 0;
  ryd_stats stats{};
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
  double timing_ms = 0;
  int64_t timing_launches = 0;
};

static bool hermitian_path(const ryd_handle* h);

static Segs make_segs(std::vector<std::pair<int, int>> v) {
  Segs s;
  for (int i = 0; i < 3; ++i) {
    s.lo[i] = i < (int)v.size() ? v[i].first : 0;
    s.len[i] = i < (int)v.size() ? v[i].second : 0;
  }
  return s;
}

// Build a pass whose tile consists of the bit ranges in `tile` (ascending,
// disjoint); the outer segments are the complement within [0, nb).
static Pass make_pass(int nb, std::vector<std::pair<int, int>> tile) {
  Pass p;
  std::vector<std::pair<int, int>> t2, outer;
  for (auto& s : tile)
    if (s.second > 0) t2.push_back(s);
  // merge adjacent ranges
  std::vector<std::pair<int, int>> merged;
  for (auto& s : t2) {
    if (!merged.empty() && merged.back().first + merged.back().second == s.first)
      merged.back().second += s.second;
    else
      merged.push_back(s);
  }
  int pos = 0;
  for (auto& s : merged) {
    if (s.first > pos) outer.push_back({pos, s.first - pos});
    pos = s.first + s.second;
    p.T += s.second;
  }
  if (pos < nb) outer.push_back({pos, nb - pos});
  // at most 3 outer segments by construction (<= 3 tile segments, first at 0)
  p.tile = make_segs(merged);
  p.outer = make_segs(outer);
  p.n_outer_bits = nb - p.T;
  return p;
}

// tile-local index of global bit p (or -1)
static int local_of(const Segs& s, int p) {
  int off = 0;
  for (int i = 0; i < 3; ++i) {
    if (p >= s.lo[i] && p < s.lo[i] + s.len[i]) return off + p - s.lo[i];
    off += s.len[i];
  }
  return -1;
}

// A 2^14 register tile keeps one whole CU busy per 16384 amplitudes; it only
// pays when the launch has enough tiles for the 256 CUs (measured: 64 tiles of a
// 20-atom ket are 1.4x slower than 256 LDS tiles of 2^12).
static bool tile14_pays(const ryd_handle* h) {
  if (h->no_tile14 || h->nb < 14) return false;
  if (h->force_tile14) return true;
  const long long tiles = (long long)h->B << (h->nb - 14);
  if (tiles >= 512) return true;
  // fewer tiles: only when the bigger tile saves a whole pass (e.g. 14-atom kets)
  const int p12 = 1 + (std::max(h->nb - 12, 0) + 7) / 8, p14 = 1 + (h->nb - 14 + 7) / 8;
  // measured on 14-atom kets: 16 tiles lose to two tiled passes (44 vs 58 sim-us/s),
  // 64 tiles win (136 vs 103), 256 tiles win 2.2x
  return p14 < p12 && tiles >= 48;
}

// States up to 128 MiB whose high bits number at most MAXO: ONE launch per
// application.  Each workgroup stages its contiguous 2^T tile in LDS for the
// low-bit flips and reads the partner amplitude of every higher bit from the same
// offset of another tile (coalesced; the re-reads are served by L2 / the 256 MiB
// Infinity Cache): (1 + nb - T) reads + 1 write per tile instead of one read + one
// write (+ partial sums) per pass, and half or a third of the launches of a
// latency-bound regime (5-50 us per launch).  Measured against the best multi-pass
// plan: 1.5x (one 14-atom ket), 1.4x (one 20-atom ket), 1.3x (21 atoms, 16 x 17),
// 1.16x (64 x 16 atoms), 1.10x (256 x 15 atoms = 128 MiB), 1.06x (22 atoms).
static bool single_pass_pays(const ryd_handle* h, int T) {
  if (h->no_outer || (!h->auto_tile && !h->force_outer) || (h->cfg.mode == RYD_MESOLVE && h->has_dbl))
    return false;
  if (h->nb <= T || h->nb - T > MAXO) return false;
  const size_t bytes = (size_t)h->B * sizeof(cplx) << h->nb;
  return h->force_outer || bytes <= ((size_t)128 << 20);
}

static void plan_passes(ryd_handle* h) {
  h->passes.clear();
  const int nb = h->nb, N = h->N, T = std::min(h->T, nb);
  const int C = 4;  // run bits (256 B contiguous) kept in every tile
  // Tile size of the single-launch plan: about one tile per CU (256 tiles), between
  // 2^9 and 2^12 amplitudes - smaller tiles mean more partner reads but more
  // workgroups in flight (measured, tools/single_pass_bench2.py: 17 atoms 4.0 -> 5.6
  // sim-us/s at 2^9 instead of 2^11, 18 atoms 3.6 -> 4.3 at 2^10; 20 atoms is best
  // at 2^12).  States that fit one tile anyway are left alone.
  int Ts = T;
  if (h->auto_tile && nb > T) {
    int logB = 0;
    while ((1 << logB) < h->B) ++logB;
    Ts = std::min(12, std::max(9, nb + logB - 8));
    Ts = std::max(Ts, nb - MAXO);
  }
  // (the 2^14 register tiles keep 14-bit states, where they are a single pass too,
  // and whatever the test hook forces onto them)
  if (nb > T && single_pass_pays(h, Ts) &&
      !(nb >= 14 && tile14_pays(h) && (nb == 14 || h->force_tile14))) {
    Pass p = make_pass(nb, {{0, Ts}});
    for (int j = 0; j < Ts; ++j) p.flip_q.push_back(j);
    for (int j = Ts; j < nb; ++j) p.oflip.push_back(j);
    p.include_diag = true;
    h->passes.push_back(p);
    h->stats.passes = 1;
    h->passes_valid = true;
    return;
  }
  if (h->cfg.mode == RYD_MESOLVE && h->has_dbl) {
    // pair passes: both bits of each atom in the same tile
    int done = 0;  // atoms (counted from the low bit end) handled so far
    bool first = true;
    while (done < N) {
      int g;
      Pass p;
      if (first) {
        g = std::min(N, std::max(1, T / 2));
        p = make_pass(nb, {{0, g}, {N, g}});
      } else {
        const int c = std::max(0, std::min(std::min(C, done), T - 2));
        g = std::min(N - done, std::max(1, (T - c) / 2));
        p = make_pass(nb, {{0, c}, {done, g}, {N + done, g}});
      }
      for (int j = 0; j < g; ++j) {
        const int pb = done + j, pa = N + done + j;
        const int qb = local_of(p.tile, pb), qa = local_of(p.tile, pa);
        p.flip_q.push_back(qb);
        p.flip_q.push_back(qa);
        p.dbl.push_back({qb, qa});
      }
      p.include_diag = first;
      h->passes.push_back(p);
      done += std::max(g, 1);
      first = false;
    }
  } else {
    int done = 0;
    bool first = true;
    while (done < nb) {
      int g;
      Pass p;
      if (first) {
        g = T;
        if (h->auto_tile && nb >= 14 && tile14_pays(h)) {
          g = 14;
          p = make_pass(nb, {{0, g}});
          p.use14 = true;
        } else {
          p = make_pass(nb, {{0, g}});
        }
      } else {
        // the remaining bits are spread evenly over the passes they need, and
        // the rest of each tile is filled with contiguous run bits (longer
        // coalesced runs, full-size tiles) as long as >= 1024 workgroups remain
        const int cmin = std::max(0, std::min(std::min(C, done), T - 1));
        const int cap = std::max(1, T - cmin);
        const int left = nb - done;
        const int k = (left + cap - 1) / cap;
        g = std::max(1, (left + k - 1) / k);
        int c = cmin;
        int logB = 0;
        while ((1 << (logB + 1)) <= h->B) ++logB;
        while (c + g < T && c < done && (nb - (c + 1 + g)) + logB >= 10) ++c;
        p = make_pass(nb, {{0, c}, {done, g}});
      }
      for (int j = 0; j < g; ++j) p.flip_q.push_back(local_of(p.tile, done + j));
      p.include_diag = first;
      h->passes.push_back(p);
      done += std::max(g, 1);
      first = false;
    }
  }
  h->stats.passes = (int)h->passes.size();
  h->passes_valid = true;
}

extern "C" const char* ryd_last_error(void) { return g_err.c_str(); }
extern "C" int ryd_abi_version(void) { return RYD_ABI_VERSION; }

extern "C" int ryd_create(const ryd_config* cfg, ryd_handle** out) {
  if (!cfg || !out) return fail(RYD_ERR_INVALID, "null argument");
  if (cfg->abi_version != RYD_ABI_VERSION)
    return fail(RYD_ERR_INVALID, "ABI version mismatch: caller %d, library %d",
                cfg->abi_version, RYD_ABI_VERSION);
  if (cfg->mode != RYD_SESOLVE && cfg->mode != RYD_MESOLVE)
    return fail(RYD_ERR_INVALID, "unknown mode %d", cfg->mode);
  const int nb = cfg->mode == RYD_MESOLVE ? 2 * cfg->n_qubits : cfg->n_qubits;
  if (cfg->n_qubits < 1 || nb > RYD_MAX_QUBITS)
    return fail(RYD_ERR_INVALID, "n_qubits=%d out of range for mode %d (index bits %d > %d)",
                cfg->n_qubits, cfg->mode, nb, RYD_MAX_QUBITS);
  if (cfg->batch < 1 || cfg->batch > 65535)
    return fail(RYD_ERR_INVALID, "batch=%d out of range [1, 65535]", cfg->batch);
  int T = cfg->tile_bits ? cfg->tile_bits : 12;
  if (T < 2 || T > 13) return fail(RYD_ERR_INVALID, "tile_bits=%d out of range [2, 13]", T);
  HIPCHK(hipSetDevice(cfg->device));
  ryd_handle* h = new ryd_handle();
  h->cfg = *cfg;
  h->N = cfg->n_qubits;
  h->nb = nb;
  h->B = cfg->batch;
  h->T = T;
  h->auto_tile = cfg->tile_bits == 0;
  if (h->auto_tile) {
    // Small states: the fewest passes first, then enough tiles to occupy the
    // 256 CUs (a 2^12 tile keeps one CU busy for ~8 us; a 14-atom ket would
    // run on 4 CUs).  passes(T) = 1 + ceil((nb - T) / (T - 4)).
    auto passes = [&](int t) { return nb <= t ? 1 : 1 + (nb - t + (t - 5)) / (t - 4); };
    int best = 12;
    for (int t = 12; t >= 8; --t) {
      if (passes(t) > passes(12)) break;
      best = t;
      if (((long long)cfg->batch << std::max(nb - t, 0)) >= 256) break;
    }
    h->T = best;
    // A 2^13 tile (128 KiB of LDS, one workgroup per CU) when it saves a whole
    // pass and the launch still has >= 128 tiles (13-atom kets, batch 256: 465 vs
    // 273 sim-us/s; batch 16: 43 vs 81 - measured, tools/t13_bench.py).
    if (cfg->mode == RYD_SESOLVE && passes(13) < passes(12) &&
        ((long long)cfg->batch << std::max(nb - 13, 0)) >= 128)
      h->T = 13;
  }
  h->dim = (size_t)1 << nb;
  const size_t bytes = h->dim * (size_t)h->B * sizeof(cplx);
  hipError_t e;
  if ((e = hipMalloc((void**)&h->wA, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->wB, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->kbuf, bytes)) != hipSuccess ||
      (e = hipMalloc((void**)&h->coefs_dev, (size_t)h->B * h->N * 4 * sizeof(double))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc of work vectors (%zu B each) failed: %s", bytes,
                hipGetErrorString(e));
  }
  // default: no interaction, no dissipator
  h->e0_mats = 1;
  if ((e = hipMalloc((void**)&h->e0_dev, ((size_t)1 << h->N) * sizeof(double))) != hipSuccess ||
      (e = hipMemset(h->e0_dev, 0, ((size_t)1 << h->N) * sizeof(double))) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipMalloc e0 failed: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < 4; ++i) h->Sd[i] = h->J[i] = make_double2(0, 0);
  // the 2^12-amplitude tile needs 64 KiB + tables of dynamic LDS (CDNA4: 160 KiB/CU)
  if ((e = hipFuncSetAttribute((const void*)k_apply<RYD_SESOLVE, 512>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_MESOLVE, 512>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_SESOLVE, 1024>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply<RYD_MESOLVE, 1024>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, true, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_SESOLVE, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, false, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, true, false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
      (e = hipFuncSetAttribute((const void*)k_apply14<RYD_MESOLVE, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) {
    ryd_destroy(h);
    return fail(RYD_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
  }
  plan_passes(h);
  *out = h;
  return RYD_OK;
}

extern "C" void ryd_destroy(ryd_handle* h) {
  if (!h) return;
  hipFree(h->wA);
  hipFree(h->split_coefs);
  hipFree(h->split_err);
  hipFree(h->rows_chk);
  hipFree(h->rows_idx_dev);
  hipFree(h->wB);
  if (h->wC) hipFree(h->wC);
  if (h->split_err_pin) hipHostFree(h->split_err_pin);
  hipFree(h->kbuf);
  hipFree(h->coefs_dev);
  hipFree(h->e0_dev);
  hipFree(h->pp_dev);
  hipFree(h->desc_dev);
  hipFree(h->dterms_dev);
  hipFree(h->sched_dev);
  hipFree(h->many_args_dev);
  hipFree(h->ksched_dev);
  hipFree(h->kry_V);
  hipFree(h->ftab_dev);
  hipFree(h->kry_pool);
  hipFree(h->gen_tcoef);
  hipFree(h->gen_sites_pool);
  hipFree(h->gen_fused_pool);
  hipFree(h->gen_diag_terms_dev);
  hipFree(h->gen_terms_dev);
  hipFree(h->gen_series_dev);
  hipFree(h->gen_conj_dev);
  hipFree(h->gen_scale_dev);
  hipFree(h->mc_pool);
  for (auto& t : h->gen_host) {
    hipFree((void*)t.dev.row_ptr);
    hipFree((void*)t.dev.col);
    hipFree((void*)t.dev.val);
    hipFree((void*)t.dev.strides);
    hipFree((void*)t.dev.shifts);
    hipFree((void*)t.dev.weights);
    hipFree((void*)t.dev.rstart);
    hipFree((void*)t.dev.ecol);
  }
  for (auto& p : h->ev_used) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  for (auto& p : h->ev_free) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  delete h;
}

extern "C" int ryd_set_series(ryd_handle* h, int32_t n_series, int32_t n_knots,
                              const double* tknots, const double* pp) {
  if (!h || !tknots || !pp) return fail(RYD_ERR_INVALID, "null argument");
  if (n_series < 1 || n_knots < 2) return fail(RYD_ERR_INVALID, "need >= 1 series and >= 2 knots");
  for (int i = 1; i < n_knots; ++i)
    if (!(tknots[i] > tknots[i - 1]))
      return fail(RYD_ERR_INVALID, "tknots must be strictly increasing (index %d)", i);
  HIPCHK(hipSetDevice(h->cfg.device));
  const int n_int = n_knots - 1;
  h->n_series = n_series;
  h->n_knots = n_knots;
  h->tknots.assign(tknots, tknots + n_knots);
  const size_t cnt = (size_t)n_series * n_int * 4;
  h->pp_host.resize(cnt);
  for (size_t i = 0; i < cnt; ++i) h->pp_host[i] = {pp[2 * i], pp[2 * i + 1]};
  // per-interval bounds of each series: |S|, max(Re S, 0), max(-Re S, 0)
  h->s_abs.assign((size_t)n_series * n_int, 0.0);
  h->s_pos.assign((size_t)n_series * n_int, 0.0);
  h->s_neg.assign((size_t)n_series * n_int, 0.0);
  h->s_curv.assign((size_t)n_series * n_int, 0.0);
  h->s_der.assign((size_t)n_series * n_int, 0.0);
  for (int s = 0; s < n_series; ++s)
    for (int i = 0; i < n_int; ++i) {
      const std::complex<double>* p = &h->pp_host[((size_t)s * n_int + i) * 4];
      const double dt = tknots[i + 1] - tknots[i];
      // value at the left knot is p[3]; deviation bounded by the other terms
      const double dev = std::abs(p[2]) * dt + std::abs(p[1]) * dt * dt + std::abs(p[0]) * dt * dt * dt;
      h->s_abs[(size_t)s * n_int + i] = std::abs(p[3]) + dev;
      h->s_pos[(size_t)s * n_int + i] = std::max(p[3].real() + dev, 0.0);
      h->s_neg[(size_t)s * n_int + i] = std::max(-p[3].real() + dev, 0.0);
      h->s_curv[(size_t)s * n_int + i] = std::abs(p[1]) * dt * dt + std::abs(p[0]) * dt * dt * dt;
      h->s_der[(size_t)s * n_int + i] = std::abs(p[2]) + 2.0 * std::abs(p[1]) * dt + 3.0 * std::abs(p[0]) * dt * dt;
    }
  // knot i + 1 is removable when piece i, re-expanded about it, IS piece i + 1 (to rounding) for every
  // series: linear ramps, plateaus - not the ringing of the not-a-knot spline next to a kink
  h->join_ok.assign(std::max(n_int - 1, 0), 1);
  h->lin_ok.assign(n_int, 1);
  for (int s = 0; s < n_series; ++s) {
    double smax = 0.0;
    for (int i = 0; i < n_int; ++i) smax = std::max(smax, h->s_abs[(size_t)s * n_int + i]);
    const double thr = 1e-13 * std::max(smax, 1e-300);
    for (int i = 0; i < n_int; ++i)
      if (!(h->s_curv[(size_t)s * n_int + i] <= 1e-10 * std::max(smax, 1e-300))) h->lin_ok[i] = 0;
    for (int i = 0; i + 1 < n_int; ++i) {
      const std::complex<double>* p = &h->pp_host[((size_t)s * n_int + i) * 4];
      const std::complex<double>* q = p + 4;
      const double dt = tknots[i + 1] - tknots[i], dn = tknots[i + 2] - tknots[i + 1];
      const std::complex<double> e0 = p[0], e1 = 3.0 * p[0] * dt + p[1],
                                 e2 = 3.0 * p[0] * dt * dt + 2.0 * p[1] * dt + p[2],
                                 e3 = ((p[0] * dt + p[1]) * dt + p[2]) * dt + p[3];
      const double mis = std::abs(e3 - q[3]) + std::abs(e2 - q[2]) * dn + std::abs(e1 - q[1]) * dn * dn +
                         std::abs(e0 - q[0]) * dn * dn * dn;
      if (!(mis <= thr)) h->join_ok[i] = 0;
    }
  }
  if (h->pp_dev) hipFree(h->pp_dev);
  h->pp_dev = nullptr;
  HIPCHK(hipMalloc((void**)&h->pp_dev, cnt * sizeof(cplx)));
  HIPCHK(hipMemcpy(h->pp_dev, h->pp_host.data(), cnt * sizeof(cplx), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  return RYD_OK;
}

extern "C" int ryd_set_qubit_desc(ryd_handle* h, const ryd_qdesc* desc) {
  if (!h || !desc) return fail(RYD_ERR_INVALID, "null argument");
  if (h->n_series == 0) return fail(RYD_ERR_STATE, "ryd_set_series must be called first");
  const size_t cnt = (size_t)h->B * h->N;
  for (size_t i = 0; i < cnt; ++i) {
    const int idx[3] = {desc[i].drive_series, desc[i].det_series, desc[i].off_series};
    for (int j = 0; j < 3; ++j)
      if (idx[j] < -1 || idx[j] >= h->n_series)
        return fail(RYD_ERR_INVALID, "series index %d out of range at entry %zu", idx[j], i);
  }
  HIPCHK(hipSetDevice(h->cfg.device));
  h->desc_host.assign(desc, desc + cnt);
  if (!h->desc_dev) HIPCHK(hipMalloc((void**)&h->desc_dev, cnt * sizeof(ryd_qdesc)));
  HIPCHK(hipMemcpy(h->desc_dev, desc, cnt * sizeof(ryd_qdesc), hipMemcpyHostToDevice));
  h->bounds_valid = false;
  return RYD_OK;
}

extern "C" int ryd_set_detuning_terms(ryd_handle* h, int32_t n_terms, const ryd_dterm* terms) {
  if (!h || (n_terms > 0 && !terms)) return fail(RYD_ERR_INVALID, "null argument");
  if (h->general) return fail(RYD_ERR_INVALID, "not available on a general-path handle");
  if (n_terms < 0) return fail(RYD_ERR_INVALID, "n_terms < 0");
  if (n_terms > 0 && h->n_series == 0) return fail(RYD_ERR_STATE, "ryd_set_series must be called first");
  for (int i = 0; i < n_terms; ++i)
    if (terms[i].series < 0 || terms[i].series >= h->n_series)
      return fail(RYD_ERR_INVALID, "series index %d out of range at term %d", terms[i].series, i);
  for (int i = 0; i < n_terms; ++i)
    if (terms[i].remaining < 0 || i + terms[i].remaining >= n_terms ||
        (terms[i].remaining > 0 && terms[i + 1].remaining != terms[i].remaining - 1))
      return fail(RYD_ERR_INVALID, "detuning term %d: inconsistent `remaining` count", i);
  HIPCHK(hipSetDevice(h->cfg.device));
  if (h->dterms_dev) hipFree(h->dterms_dev);
  h->dterms_dev = nullptr;
  h->dterms_host.assign(terms, terms + n_terms);
  if (n_terms > 0) {
    HIPCHK(hipMalloc((void**)&h->dterms_dev, (size_t)n_terms * sizeof(ryd_dterm)));
    HIPCHK(hipMemcpy(h->dterms_dev, terms, (size_t)n_terms * sizeof(ryd_dterm), hipMemcpyHostToDevice));
  }
  h->bounds_valid = false;
  return RYD_OK;
}

// every ryd_qdesc.extra must point into the table of extra detuning terms
static int validate_extras(const ryd_handle* h) {
  const int n = (int)h->dterms_host.size();
  for (size_t i = 0; i < h->desc_host.size(); ++i) {
    const int e = h->desc_host[i].extra;
    if (e < 0 || e > n)
      return fail(RYD_ERR_INVALID, "descriptor %zu: extra = %d outside the %d detuning terms "
                  "(ryd_set_detuning_terms)", i, e, n);
  }
  return RYD_OK;
}

static void compute_bounds(ryd_handle* h) {
  const int n_int = h->n_knots - 1;
  h->bd_drive.assign(n_int, 0.0);
  h->bd_pos.assign(n_int, 0.0);
  h->bd_neg.assign(n_int, 0.0);
  h->bd_curv.assign(n_int, 0.0);
  h->bd_c1.assign(n_int, 0.0);
  h->bd_dc.assign(n_int, 0.0);
  h->bd_dl.assign(n_int, 0.0);
  h->bd_ddl.assign(n_int, 0.0);
  // batch entries are independent: a few host threads, each with its own
  // accumulators, merged by a maximum at the end
  const int n_thr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, h->B / 8}));
  std::vector<std::vector<double>> part(n_thr, std::vector<double>((size_t)n_int * 8, 0.0));
  auto work = [&](int tix) {
  std::vector<double> dr(n_int), po(n_int), ne(n_int), cu(n_int), q((size_t)n_int * 4);
  std::unordered_map<int, std::vector<double>> extra_cache;  // combined cubic of an extra-term list
  double* bd_drive = &part[tix][0];
  double* bd_pos = bd_drive + n_int;
  double* bd_neg = bd_pos + n_int;
  double* bd_curv = bd_neg + n_int;
  double* bd_c1 = bd_curv + n_int;  // per-atom maxima
  double* bd_dc = bd_c1 + n_int;
  double* bd_dl = bd_dc + n_int;
  double* bd_ddl = bd_dl + n_int;
  for (int b = tix; b < h->B; b += n_thr) {
    std::fill(dr.begin(), dr.end(), 0.0);
    std::fill(po.begin(), po.end(), 0.0);
    std::fill(ne.begin(), ne.end(), 0.0);
    std::fill(cu.begin(), cu.end(), 0.0);
    for (int k = 0; k < h->N; ++k) {
      const ryd_qdesc& d = h->desc_host[(size_t)b * h->N + k];
      if (d.drive_series >= 0) {
        const double* a = &h->s_abs[(size_t)d.drive_series * n_int];
        const double sc = std::fabs(d.drive_scale);
        for (int i = 0; i < n_int; ++i) dr[i] += sc * a[i];
        const double* cv = &h->s_curv[(size_t)d.drive_series * n_int];
        for (int i = 0; i < n_int; ++i) cu[i] += sc * cv[i];
        const double* dv = &h->s_der[(size_t)d.drive_series * n_int];
        for (int i = 0; i < n_int; ++i) {
          bd_c1[i] = std::max(bd_c1[i], sc * a[i]);
          bd_dc[i] = std::max(bd_dc[i], sc * dv[i]);
        }
      }
      // The detuning of the atom is ONE real cubic per interval: combine every
      // contribution (samples, doppler / constant offsets, the extra hf-noise
      // terms whose phases largely cancel) before bounding it - bounding the
      // terms one by one would over-estimate norm and curvature several times
      // and cost as many extra steps.
      std::fill(q.begin(), q.end(), 0.0);
      auto add_series = [&](std::vector<double>& dst, int sidx, double sc) {
        if (sidx < 0 || sc == 0.0) return;
        const std::complex<double>* p = &h->pp_host[(size_t)sidx * n_int * 4];
        for (size_t e = 0; e < (size_t)n_int * 4; ++e) dst[e] += sc * p[e].real();
      };
      add_series(q, d.det_series, d.det_scale);
      add_series(q, d.off_series, d.off_scale);
      if (d.extra > 0 && d.extra <= (int)h->dterms_host.size()) {
        auto it = extra_cache.find(d.extra);
        if (it == extra_cache.end()) {
          std::vector<double> acc((size_t)n_int * 4, 0.0);
          for (size_t e = (size_t)d.extra - 1; e < h->dterms_host.size(); ++e) {
            add_series(acc, h->dterms_host[e].series, h->dterms_host[e].scale);
            if (h->dterms_host[e].remaining == 0) break;
          }
          it = extra_cache.emplace(d.extra, std::move(acc)).first;
        }
        const std::vector<double>& acc = it->second;
        for (size_t e = 0; e < (size_t)n_int * 4; ++e) q[e] += acc[e];
      }
      for (int i = 0; i < n_int; ++i) {
        const double dt = h->tknots[i + 1] - h->tknots[i];
        const double* c = &q[(size_t)i * 4];  // c[0] u^3 + c[1] u^2 + c[2] u + c[3]
        const double curv = std::fabs(c[1]) * dt * dt + std::fabs(c[0]) * dt * dt * dt;
        const double dev = std::fabs(c[2]) * dt + curv;
        po[i] += std::max(c[3] + dev, 0.0);
        ne[i] += std::max(-c[3] + dev, 0.0);
        cu[i] += curv;
        bd_dl[i] = std::max(bd_dl[i], std::fabs(c[3]) + dev);
        bd_ddl[i] = std::max(bd_ddl[i], std::fabs(c[2]) + 2.0 * std::fabs(c[1]) * dt + 3.0 * std::fabs(c[0]) * dt * dt);
      }
    }
    for (int i = 0; i < n_int; ++i) {
      bd_drive[i] = std::max(bd_drive[i], dr[i]);
      bd_pos[i] = std::max(bd_pos[i], po[i]);
      bd_neg[i] = std::max(bd_neg[i], ne[i]);
      bd_curv[i] = std::max(bd_curv[i], cu[i]);
    }
  }
  };
  if (n_thr == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int tix = 0; tix < n_thr; ++tix) pool.emplace_back(work, tix);
    for (auto& th : pool) th.join();
  }
  for (int tix = 0; tix < n_thr; ++tix)
    for (int i = 0; i < n_int; ++i) {
      h->bd_drive[i] = std::max(h->bd_drive[i], part[tix][i]);
      h->bd_pos[i] = std::max(h->bd_pos[i], part[tix][(size_t)n_int + i]);
      h->bd_neg[i] = std::max(h->bd_neg[i], part[tix][(size_t)2 * n_int + i]);
      h->bd_curv[i] = std::max(h->bd_curv[i], part[tix][(size_t)3 * n_int + i]);
      h->bd_c1[i] = std::max(h->bd_c1[i], part[tix][(size_t)4 * n_int + i]);
      h->bd_dc[i] = std::max(h->bd_dc[i], part[tix][(size_t)5 * n_int + i]);
      h->bd_dl[i] = std::max(h->bd_dl[i], part[tix][(size_t)6 * n_int + i]);
      h->bd_ddl[i] = std::max(h->bd_ddl[i], part[tix][(size_t)7 * n_int + i]);
    }
  // MODEL 1 of the persistent kernel: inside every trajectory all driven atoms
  // share (series, scale) and that series is real-valued
  const int n_int2 = h->n_knots - 1;
  std::vector<char> series_real(h->n_series, 1);
  for (int sidx = 0; sidx < h->n_series; ++sidx)
    for (size_t i = 0; i < (size_t)n_int2 * 4; ++i)
      if (h->pp_host[(size_t)sidx * n_int2 * 4 + i].imag() != 0.0) { series_real[sidx] = 0; break; }
  bool uni = true;
  for (int b = 0; b < h->B && uni; ++b) {
    int ser = -2;
    double sc = 0.0;
    for (int k = 0; k < h->N; ++k) {
      const ryd_qdesc& d = h->desc_host[(size_t)b * h->N + k];
      if (d.drive_series < 0) continue;
      if (ser == -2) { ser = d.drive_series; sc = d.drive_scale; }
      else if (ser != d.drive_series || sc != d.drive_scale) { uni = false; break; }
      if (!series_real[d.drive_series]) { uni = false; break; }
    }
  }
  h->uniform_real_drive = uni;
  bool dreal = true;
  for (const ryd_qdesc& d : h->desc_host)
    if (d.drive_series >= 0 && !series_real[d.drive_series]) { dreal = false; break; }
  h->drive_real = dreal;
  // ---- KET_GAUGE: theta' = Im(c' conj c) / |c|^2 of every complex drive series, sampled per interval ----
  h->gauge_ok = false;
  h->bd_gpos.assign(n_int, 0.0);
  h->bd_gneg.assign(n_int, 0.0);
  h->bd_gvar.assign(n_int, 0.0);
  if (!dreal && h->cfg.mode == RYD_SESOLVE) {
    double smax = 0.0;
    for (int sidx = 0; sidx < h->n_series; ++sidx)
      if (!series_real[sidx])
        for (int i = 0; i < n_int; ++i) smax = std::max(smax, h->s_abs[(size_t)sidx * n_int + i]);
    const double eps = 1e-9 * smax;
    h->gauge_eps2 = eps * eps;
    // per series: the largest theta' of either sign at 9 points of every interval (+25 % margin: the bound
    // feeds the spectral shift and the choice of the in-place scheme, like the other detuning bounds)
    std::vector<double> tp((size_t)h->n_series * n_int, 0.0), tn((size_t)h->n_series * n_int, 0.0),
        tv((size_t)h->n_series * n_int, 0.0);
    bool ok = true;
    const double cap = 4000.0;  // rad/us: a phase that turns faster than this inside a step is not gauged
    for (int sidx = 0; sidx < h->n_series && ok; ++sidx) {
      if (series_real[sidx]) continue;
      // The kernel follows a drive's direction w = c / |c| through theta' alone.  While |c| is below the threshold it
      // HOLDS the last direction, and theta' is not sampled there - so a drive that falls to zero and comes back with
      // another phase (Ramsey / echo: pulse, delay, phase-shifted pulse) would lose the phase step: the frame of
      // psi~ is never turned by conj(w_old) w_new (round-3 ADVICE: 0.50 error in the final amplitudes for a pi / 2
      // step).  Such series are not gauged: the direction before a sub-threshold stretch must be the direction
      // after it, up to the sign that a zero crossing of the signed modulus r takes care of.
      std::complex<double> w_last(0.0, 0.0);  // direction at the last sample above the threshold
      bool in_gap = false;
      for (int i = 0; i < n_int && ok; ++i) {
        const std::complex<double>* pc = &h->pp_host[((size_t)sidx * n_int + i) * 4];
        const double dt = h->tknots[i + 1] - h->tknots[i];
        double hi = 0.0, lo = 0.0, vhi = -1e300, vlo = 1e300;
        for (int g = 0; g <= 8; ++g) {
          const double u = dt * g / 8.0;
          const std::complex<double> c = ((pc[0] * u + pc[1]) * u + pc[2]) * u + pc[3];
          const std::complex<double> dc = (3.0 * pc[0] * u + 2.0 * pc[1]) * u + pc[2];
          const double m2 = std::norm(c);
          if (m2 <= h->gauge_eps2) { in_gap = true; continue; }
          const std::complex<double> w_now = c / std::sqrt(m2);
          if (in_gap && std::norm(w_last) > 0.0 && std::fabs((std::conj(w_last) * w_now).imag()) > 1e-8) ok = false;
          in_gap = false;
          w_last = w_now;
          const double thd = (dc * std::conj(c)).imag() / m2;
          hi = std::max(hi, thd);
          lo = std::min(lo, thd);
          vhi = std::max(vhi, thd);
          vlo = std::min(vlo, thd);
        }
        if (hi > cap || -lo > cap) ok = false;
        tp[(size_t)sidx * n_int + i] = 1.25 * hi;
        tn[(size_t)sidx * n_int + i] = -1.25 * lo;
        tv[(size_t)sidx * n_int + i] = vhi > vlo ? vhi - vlo : 0.0;
      }
    }
    if (ok) {
      // detuning of atom k in the gauge: delta_k + theta_k' enters as -(...) n_k: a positive theta' lowers
      // the diagonal (bd_gpos), a negative one raises it (bd_gneg)
      for (int b = 0; b < h->B; ++b) {
        std::vector<double> gp(n_int, 0.0), gn(n_int, 0.0);
        for (int k = 0; k < h->N; ++k) {
          const ryd_qdesc& d = h->desc_host[(size_t)b * h->N + k];
          if (d.drive_series < 0 || d.drive_scale == 0.0 || series_real[d.drive_series]) continue;
          for (int i = 0; i < n_int; ++i) {
            gp[i] += tp[(size_t)d.drive_series * n_int + i];
            gn[i] += tn[(size_t)d.drive_series * n_int + i];
            h->bd_gvar[i] = std::max(h->bd_gvar[i], tv[(size_t)d.drive_series * n_int + i]);
          }
        }
        for (int i = 0; i < n_int; ++i) {
          h->bd_gpos[i] = std::max(h->bd_gpos[i], gp[i]);
          h->bd_gneg[i] = std::max(h->bd_gneg[i], gn[i]);
        }
      }
    }
    h->gauge_ok = ok;
  }
  h->bounds_valid = true;
  h->split_known = false;  // new tables: the split-operator controller starts over
}

extern "C" int ryd_set_interaction(ryd_handle* h, const double* U, int32_t n_mats) {
  if (!h || !U) return fail(RYD_ERR_INVALID, "null argument");
  if (n_mats != 1 && n_mats != h->B)
    return fail(RYD_ERR_INVALID, "n_mats must be 1 or batch (%d), got %d", h->B, n_mats);
  HIPCHK(hipSetDevice(h->cfg.device));
  const int N = h->N;
  const size_t D = (size_t)1 << N;
  double lo = 0, hi = 0;
  for (int m = 0; m < n_mats; ++m) {
    double l = 0, u = 0;
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j) {
        const double v = U[((size_t)m * N + i) * N + j];
        if (v != U[((size_t)m * N + j) * N + i])
          return fail(RYD_ERR_INVALID, "interaction matrix %d not symmetric at (%d,%d)", m, i, j);
        if (v > 0) u += v; else l += v;
      }
    lo = std::min(lo, l);
    hi = std::max(hi, u);
  }
  h->e0_min = lo;
  h->e0_max = hi;
  h->u_rowsum = 0.0;
  for (int m = 0; m < n_mats; ++m)
    for (int i = 0; i < N; ++i) {
      double r = 0.0;
      for (int j = 0; j < N; ++j)
        if (j != i) r += std::fabs(U[((size_t)m * N + i) * N + j]);
      h->u_rowsum = std::max(h->u_rowsum, r);
    }
  double* Udev = nullptr;
  HIPCHK(hipMalloc((void**)&Udev, (size_t)n_mats * N * N * sizeof(double)));
  hipError_t e = hipMemcpy(Udev, U, (size_t)n_mats * N * N * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    if (h->e0_dev) hipFree(h->e0_dev);
    h->e0_dev = nullptr;
    e = hipMalloc((void**)&h->e0_dev, (size_t)n_mats * D * sizeof(double));
  }
  if (e == hipSuccess) {
    dim3 grid((unsigned)((D + 255) / 256), n_mats);
    hipLaunchKernelGGL(k_build_e0, grid, dim3(256), 0, 0, Udev, N, h->e0_dev);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  hipFree(Udev);
  if (e != hipSuccess) return fail(RYD_ERR_HIP, "building E0 failed: %s", hipGetErrorString(e));
  h->e0_mats = n_mats;
  return RYD_OK;
}

extern "C" int ryd_set_dissipator(ryd_handle* h, const double* S) {
  if (!h || !S) return fail(RYD_ERR_INVALID, "null argument");
  if (h->cfg.mode != RYD_MESOLVE)
    return fail(RYD_ERR_INVALID, "dissipator only valid for a mesolve handle");
  bool dbl = false;
  double norm = 0.0;
  for (int r = 0; r < 4; ++r) {
    double row = 0.0;
    for (int c = 0; c < 4; ++c) {
      const double re = S[2 * (4 * r + c)], im = S[2 * (4 * r + c) + 1];
      const double a = std::hypot(re, im);
      row += a;
      if (a == 0.0) continue;
      if (c == r) continue;
      if (c == 3 - r) { dbl = true; continue; }
      return fail(RYD_ERR_UNSUPPORTED,
                  "dissipator entry S[%d][%d] (single flip with pair-dependent coefficient) "
                  "is not supported by this ABI version", r, c);
    }
    norm = std::max(norm, row);
  }
  for (int r = 0; r < 4; ++r) {
    h->Sd[r] = make_double2(S[2 * (4 * r + r)], S[2 * (4 * r + r) + 1]);
    h->J[r] = make_double2(S[2 * (4 * r + (3 - r))], S[2 * (4 * r + (3 - r)) + 1]);
  }
  h->diss_norm = norm * h->N;
  const bool replan = dbl != h->has_dbl;
  h->has_dbl = dbl;
  if (replan) plan_passes(h);
  return RYD_OK;
}
