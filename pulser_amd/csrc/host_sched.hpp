// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// CF4 schedule: steps, Taylor orders, spectral shifts (host)
// ---------------------------------------------------------------------------
static const double kS3 = 1.7320508075688772;
static const double kC1 = 0.5 - kS3 / 6.0, kC2 = 0.5 + kS3 / 6.0;  // Gauss nodes
static const double kA1 = 0.25 + kS3 / 6.0, kA2 = 0.25 - kS3 / 6.0;  // CF4 weights
// Default truncation bound per exponential.  The stated parity bar is 1e-7 on amplitudes after a
// whole sequence: 1e-10 per exponential ends 5e-9 .. 1.2e-8 from the tight oracle on the
// 8- and 12-atom anneal sequences (tests/probes/stepper_model.py, tools/order_probe.py); 1e-12 bought 2e-10
// for 14 % more generator applications.
static const double kDefaultTol = 1e-10;
// The explicit-term (general) path has no spectral shift and multi-level operators; its systems
// are tiny, so it keeps the tight bound (1e-10 ended 2.2e-7 from the oracle on a 3-level sequence).
static const double kDefaultTolGeneral = 1e-12;
// The per-exponential defaults (Taylor remainder, in-place schemes, Magnus-merge rate) were calibrated on
// 3.1-us sequences, where they end 5e-9 .. 2e-8 from the tight oracle against a bar of 1e-7.  Their
// budgets grow linearly with the number of exponentials, i.e. with the sequence duration, so for LONGER
// sequences they are derived from the same whole-sequence budget (as the split-operator controller does
// with t_total): scaled down by 3.1 us / duration.  Shorter sequences keep the calibrated values.
static const double kCalibratedDuration = 3.1;  // us
static double budget_scale(const ryd_handle* h) {
  if (h->tknots.size() < 2) return 1.0;
  const double T = h->tknots.back() - h->tknots.front();
  return T > kCalibratedDuration ? kCalibratedDuration / T : 1.0;
}
static double default_tol(const ryd_handle* h) {
  return (h->general ? kDefaultTolGeneral : kDefaultTol) * budget_scale(h);
}

// Smallest Taylor degree m with rho^(m+1)/(m+1)! <= tol (the remainder bound of the
// Horner polynomial for ||h G~|| <= rho), capped.
static int taylor_order_for(double rho, const ryd_opts& o, double dtol) {
  const int cap = std::min(o.max_order > 0 ? o.max_order : 32, 32);
  const double tol = o.tol > 0 ? o.tol : dtol;
  double term = rho;  // rho^(m+1)/(m+1)! for m = 0
  int order = 1;
  while (order < cap) {
    term *= rho / (order + 1);  // now rho^(order+1)/(order+1)!
    if (term <= tol) break;
    ++order;
  }
  return order;
}

// Taylor order and spectral shift of one exponential exp(h (w1 G(t1) + w2 G(t2)))
// with both Gauss points inside knot interval `idx`.
// maximum of a per-interval bound over the `span` intervals a (multi-knot) step covers
static double span_max(const std::vector<double>& v, int idx, int span) {
  double m = 0.0;
  for (int i = idx; i < idx + std::max(span, 1) && i < (int)v.size(); ++i) m = std::max(m, v[i]);
  return m;
}

static void plan_exp(ryd_handle* h, int idx, double hstep, double w1, double w2,
                     const ryd_opts& o, int* order_out, double* shift_out, int span = 1) {
  const double wmix = w1 + w2;
  const double drive = wmix * span_max(h->bd_drive, idx, span);
  const double dpos = wmix * span_max(h->bd_pos, idx, span), dneg = wmix * span_max(h->bd_neg, idx, span);
  const double lo = wmix * h->e0_min - dpos, hi = wmix * h->e0_max + dneg;
  double bound, shift = 0.0;
  if (h->general) {
    bound = drive;  // sum_t |coef_t| ||A_t||_inf; no spectral shift
  } else if (h->cfg.mode == RYD_SESOLVE) {
    shift = 0.5 * (lo + hi);  // H' = H - shift: halves the spectral radius
    bound = 0.5 * (hi - lo) + drive;
    if (h->mc) bound += wmix * std::fabs(h->mc_b) * 0.5 * h->N;  // centred decay diagonal
  } else {
    bound = 2.0 * (0.5 * (hi - lo) + drive) + wmix * h->diss_norm;
  }
  h->stats.norm_bound = bound / std::max(wmix, 1e-300);
  const double rho = std::fabs(hstep) * bound;
  int order = o.taylor_order;
  if (order <= 0) order = taylor_order_for(rho, o, default_tol(h));
  if (order < 2) order = 2;
  if (order > 32) order = 32;
  h->stats.last_order = order;
  *order_out = order;
  *shift_out = shift;
}

// CF4 steps covering [t0, t1]: never straddling a spline knot (inside a knot
// interval every coefficient is a single cubic), optionally capped by max_step.
// Multi-knot steps.  Where the waveforms are the SAME polynomial across knots (join_ok: linear ramps,
// plateaus - not the ringing next to a kink) a CF4 step may span several knot intervals: the kernels
// evaluate piece idx at offsets beyond its own interval, which is the same function.  Doubling the
// step halves the number of exponentials while their polynomial degree grows far less (degree ~ e rho
// / 2 + log(1 / tol) terms).  The 4th-order Magnus error of a step of length h over linear-in-time
// H(t) = A + B t is ~ h^5 / 720 ||[B, [B, A]]||; with B = c' X + delta' N the double commutators are
// delta'^2 |c| and c'^2 (|delta| + sum_j U_ij) per atom (tools/bigstep_probe.py: 8 and 12 atoms, steps of
// 2 / 3 / 4 / 6 knots end 1e-9 / 5e-9 / 2e-8 / 1e-7 from the converged solution - this estimate is ~3x above).
static const double kMergeRate = 4e-9;  // allowed Magnus-error ESTIMATE per us of merged steps (x magnus_tol / 1e-10)
static const int kMergeMax = 4;

static double merge_error(const ryd_handle* h, int idx, int span, double len) {
  const double dd = span_max(h->bd_ddl, idx, span), c1 = span_max(h->bd_c1, idx, span);
  const double dc = span_max(h->bd_dc, idx, span), dl = span_max(h->bd_dl, idx, span);
  const double h5 = len * len * len * len * len / 720.0;
  return (h->cfg.mode == RYD_MESOLVE ? 2.0 : 1.0) * h5 * (dd * dd * c1 + dc * dc * (dl + h->u_rowsum));
}

// RYD_DEV=1 RYD_SPLIT_NOCURV=1 (dev probe): no sub-steps from the CF4 curvature estimate under the split-operator path
static bool split_no_curv_env() {
  static const bool on = dev_env_flag("RYD_SPLIT_NOCURV", false);
  return on;
}

static void build_schedule(ryd_handle* h, double t0, double t1, const ryd_opts& o,
                           std::vector<StepDesc>& out, bool in_place_exp = false, int merge_cap = kMergeMax,
                           bool no_estimate = false) {
  const double eps = 1e-12;
  double t = t0;
  const bool merge = merge_cap > 1 && !h->general && !h->mc && !h->no_merge && o.taylor_order <= 0;
  while (t < t1 - eps) {
    const int idx = find_interval(h, t + eps);
    double tend = t1;  // the last interval extends to t1 (extrapolation, as scipy does)
    if (idx < h->n_knots - 2) tend = std::min(t1, h->tknots[idx + 1]);
    if (tend <= t + eps) tend = t1;
    int span = 1;
    if (merge && idx < h->n_knots - 2 && std::fabs(t - h->tknots[idx]) < eps &&
        std::fabs(tend - h->tknots[idx + 1]) < eps) {
      const double mtol = o.magnus_tol > 0 ? o.magnus_tol : 1e-10;
      int cap = no_estimate ? merge_cap : std::min(merge_cap, kMergeMax);
      if (no_estimate && cap > 1) {
        // Split-operator path (round 5): the smooth stretch ahead - up to t1, the next knot that cannot be removed, or
        // max_step - is cut into steps of EQUAL length rather than cap, cap, ..., remainder: with evaluation times at every
        // 10th knot the stretches are 10 knots long, and 5 + 5 costs 20 stages where the controller holds sub-steps of
        // 4.5 - 5 ns (one sub-step each) against 30 for 9 + 1 (two sub-steps + a whole composition over one knot).
        int avail = 1;
        while (idx + avail < h->n_knots - 2 && h->join_ok[idx + avail - 1] && h->tknots[idx + avail + 1] <= t1 + eps &&
               !(o.max_step > 0 && h->tknots[idx + avail + 1] - t > o.max_step * (1.0 + 1e-9)) && avail < 64 * cap)
          ++avail;
        const int n_steps = (avail + cap - 1) / cap;
        cap = std::min(cap, (avail + n_steps - 1) / n_steps);
      }
      while (span < cap && idx + span < h->n_knots - 2 &&
             h->join_ok[idx + span - 1] && h->tknots[idx + span + 1] <= t1 + eps) {
        const double cand = h->tknots[idx + span + 1] - t;
        if (o.max_step > 0 && cand > o.max_step * (1.0 + 1e-9)) break;
        if (!no_estimate) {
          if (merge_error(h, idx, span + 1, cand) > kMergeRate * budget_scale(h) * (mtol / 1e-10) * cand) break;
          // the spline's own non-linearity (same estimate as below, over the longer step)
          const double dtk = h->tknots[idx + 1] - h->tknots[idx];
          const double fr = cand / dtk;
          if (1e-5 * cand * span_max(h->bd_curv, idx, span + 1) * fr * fr > mtol) break;
        }
        ++span;
        tend = h->tknots[idx + span];
      }
    }
    const double len = tend - t;
    int nsub = 1;
    if (o.max_step > 0) nsub = std::max(1, (int)std::ceil(len / o.max_step - 1e-9));
    {
      // The local error of the 4th-order Magnus step is dominated by the
      // non-linear (quadratic + cubic) part of the spline inside the interval -
      // large only where it rings next to a kink of the waveform.  Calibrated
      // against converged references: err ~ 1e-5 * h * curvature,
      // and it falls as n^-4 with n equal sub-steps.
      const double dtk = h->tknots[idx + 1] - h->tknots[idx];
      const double frac = dtk > 0 ? std::min((double)span, len / dtk) : 1.0;
      const double est = 1e-5 * len * span_max(h->bd_curv, idx, span) * frac * frac;
      const double mtol = o.magnus_tol > 0 ? o.magnus_tol : 1e-10;
      if (est > mtol && !(h->sched_for_split && split_no_curv_env())) {
        const int nm = (int)std::ceil(std::pow(est / mtol, 0.25));
        nsub = std::max(nsub, std::min(nm, 256));
      }
    }
    static const bool int_rule = dev_env_flag("RYD_SCHED_INT", true);  // (dev A/B: RYD_DEV=1 RYD_SCHED_INT=0)
    if (int_rule && !h->sched_for_split && !h->general) {
      // Round 6 (the fuzz's worst cases pinned on the tight oracle, profiles/r06_fuzz_summary.md): the estimates above know the
      // waveforms' curvature and, for merged steps, [B, [B, A]] with the slopes B - not the 4th-order Magnus terms that carry
      // the DIAGONAL strength of A several times.  On registers at 4.5 - 5.4 um (300 - 650 rad/us between neighbours) one-knot
      // CF4 steps at tol = magnus_tol = 1e-12 ended 1e-8 .. 3.2e-8 from the oracle and fell 16 - 60 x per halving of the step
      // (tools/r06_taylor_probe.py).  Empirical per-step figure K h^5 R^3 |c'| with R = max|delta| + max_i sum_j U_ij and
      // K = 1e-7 (the five measured cases give 2e-8 .. 8e-8; products of norms overshoot the commutators by that much);
      // n equal sub-steps divide it by n^4.  R h = 0.1 on the registers of the benchmarks: no effect there.
      const double R = span_max(h->bd_dl, idx, span) + h->u_rowsum;
      const double sl = len / nsub;
      const double est = (h->cfg.mode == RYD_MESOLVE ? 2.0 : 1.0) * 1e-7 * sl * sl * sl * sl * sl * R * R * R * span_max(h->bd_dc, idx, span);
      const double mtol = o.magnus_tol > 0 ? o.magnus_tol : 1e-10;
      if (est > mtol) nsub *= std::min((int)std::ceil(std::pow(est / mtol, 0.25)), 64);
    }
    if (h->gauge_active) {
      // KET_GAUGE: the 4th-order Magnus step assumes a Hamiltonian that is smooth inside the step; the gauged
      // detuning theta'(t) = d/dt arg c(t) is not where the phase of a drive turns quickly (a phase jump
      // between two pulses is spread over ~2 knot intervals by the complex spline: theta' rises to
      // ~1500 rad/us and falls again).  Sub-steps keep the change of theta' across a step below 0.02 / h
      // (tests/probes/gauge_probe.py: a jump of 1.1 rad then ends 2e-8 from the converged solution, the
      // same drives without it 1e-8); only the few intervals around such a feature are affected.
      const double var = span_max(h->bd_gvar, idx, span);
      if (var * len > 0.02) nsub = std::max(nsub, std::min((int)std::ceil(var * len / 0.02), 512));
    }
    {
      // Sub-steps only where they pay: the Taylor degree grows like e*rho + log(1/tol), so the
      // number of generator applications per unit time FALLS with rho (about 16 / 11 / 8 per unit
      // rho at rho = 0.75 / 1.5 / 3 for tol = 1e-12).  Split only to stay under the degree cap and
      // below rho = 6, where the largest term (e^rho) starts to cost digits.
      int ord;
      double sh;
      plan_exp(h, idx, len / nsub, kA1, kA2, o, &ord, &sh, span);
      const double rho = (len / nsub) * h->stats.norm_bound * (kA1 + kA2);
      if (h->general) {
        // The argument of a general-path exponential stays near 1 AGAINST THE CRUDE BOUND bd_step (the sum of the terms'
        // own norms).  This is more than a Taylor-series matter: the general path has no estimate of the CF4 (4th-order
        // Magnus) error of a step with a time-dependent drive on top of a strong static interaction, and h ~ 1 / ||G|| is
        // what bounds it.  The (tighter) joint bound bd_drive only lowers the degree of the polynomial (plan_exp).  Round 6 tried the tuned path's search (minimise
        // sub-steps x degree, arguments up to 4: half the applications): the reference's XY sequence at 10-ns knots then
        // ends 1e-7 from the converged solution (tools/cf4_probe.py: error ~ h^4, 9e-9 at an argument of 1.2, 1.6e-7 at 2.4).
        const double rho_step = (len / nsub) * span_max(h->bd_step, idx, span) * (kA1 + kA2);
        if (rho_step > 1.5) nsub *= (int)std::ceil(rho_step / 1.0);
      } else if (in_place_exp) {
        // k_ket splits an exponential into equal sub-exponentials itself (pick_scheme)
      } else if (o.taylor_order <= 0) {
        const int cap = std::min(o.max_order > 0 ? o.max_order : 32, 32);
        const double r_max = 6.0;
        const int k_max = 64;
        int best_n = 0;
        double best_cost = 0.0;
        for (int k = 1; k <= k_max; ++k) {
          const double r = rho / k;
          if (r > r_max) continue;
          ryd_opts oo = o;
          const int m = taylor_order_for(r, oo, default_tol(h));
          double term = 1.0;
          for (int j = 1; j <= m + 1; ++j) term *= r / j;
          const double tol = o.tol > 0 ? o.tol : default_tol(h);
          if (m >= cap && term > tol) continue;  // capped before reaching the tolerance
          const double cost = (double)k * m;
          if (best_n == 0 || cost < best_cost) { best_n = k; best_cost = cost; }
          if (r < 0.25) break;
        }
        if (best_n > 1) nsub *= best_n;
      } else if (rho > 1.5) {
        nsub *= (int)std::ceil(rho / 1.0);  // fixed order: keep the argument near 1
      }
    }
    const double hs = len / nsub;
    for (int s = 0; s < nsub; ++s) {
      const double ta = t + s * hs;
      StepDesc d;
      std::memset(&d, 0, sizeof d);
      d.h = hs;
      d.idx = idx;
      d.u1 = ta + kC1 * hs - h->tknots[idx];
      d.u2 = ta + kC2 * hs - h->tknots[idx];
      plan_exp(h, idx, hs, kA1, kA2, o, &d.order_a, &d.shift_a, span);
      plan_exp(h, idx, hs, kA2, kA1, o, &d.order_b, &d.shift_b, span);
      d.snap = -1;
      d.pad = span;  // knot intervals this step's bounds must cover
      out.push_back(d);
    }
    t = tend;
  }
}

