// Part of librydemu (included by rydemu.hip, one translation unit).  HOST code only: no HIP call, no handle.
// ---------------------------------------------------------------------------
// ryd_replay_samples: the reference's sampling of a block of states, bit for bit
// ---------------------------------------------------------------------------
// What QutipEmulator.run does with every trajectory state at every evaluation time (simulation.py:853-861):
//   weights   = np.abs(state) ** 2 (kets) / np.abs(diag rho), reversed for the ground-rydberg measurement basis
//               (pulser_simulation/qutip_result.py:101-158)
//   multinomial: indices = np.searchsorted(np.cumsum(weights / cumsum(weights)[-1]), rnd)
//               (pulser-core/pulser/math/multinomial.py:32-36; Result.get_samples, pulser/result.py:103-115)
//   measurement flips with a pre-drawn uniform matrix (pulser_simulation/simresults.py:537-568): shots grouped by
//               outcome in order of first occurrence (collections.Counter), one matrix row per shot in that order.
// The random numbers are drawn on the host in the reference's order (pulser_amd/distributed.py: predraw) - here only
// the arithmetic runs, and it runs in the SAME floating-point operations in the SAME order as NumPy's (np.abs of a
// complex is C hypot; `** 2` is x * x; cumsum adds left to right; no contraction into FMAs), so the histograms equal
// the NumPy replay's exactly (tests/test_host_logic.py::test_c_replay_equals_the_numpy_replay).  Rows are independent:
// they are spread over host threads, each with its own histogram, merged at the end (integer sums: order-free).
// Round 6: the NumPy replay (20 ms of ufuncs + 12 ms of Python loop per block of 256 trajectories) was slower than the
// 25-ms solve of the block it was meant to hide behind.

#pragma clang fp contract(off)

struct ReplayJob {
  const double* kets;   // [n_rows][2 * D] interleaved complex (is_ket) or diagonal entries
  int64_t D;
  int n_qubits;
  int is_ket, reversed, matching;
  const int64_t* start;  // [n_rows] offset of the row's uniforms in rnd (and of its matrix rows in mat)
  const int64_t* count;  // [n_rows]
  const int32_t* slot;   // [n_rows] histogram the row adds to
  const double* rnd;
  const double* mat;     // [total][n_qubits] or null
  double eps, eps_p;
  int64_t* hist;         // [n_slots][D]
};

static int replay_rows(const ReplayJob& J, int64_t r0, int64_t r1, int64_t* hist_local /* [n_slots][D] */,
                       std::vector<double>& cum, std::vector<int64_t>& ind, std::vector<int64_t>& first_seen,
                       std::vector<int64_t>& cnt, std::vector<int64_t>& order) {
  const int64_t D = J.D;
  const int n = J.n_qubits;
  const bool flips = J.mat != nullptr && !(J.eps == 0.0 && J.eps_p == 0.0);
  for (int64_t r = r0; r < r1; ++r) {
    const double* x = J.kets + (size_t)r * 2 * D;
    // weights (qutip_result.py:101-122) -> w = weights / sum -> cumulative sums, all sequential
    if (!J.matching) {
      for (int64_t i = 0; i < D; ++i) cum[i] = 0.0;
      cum[0] = 1.0;
    } else {
      for (int64_t i = 0; i < D; ++i) {
        const int64_t src = J.reversed ? D - 1 - i : i;
        const double h = hypot(x[2 * src], x[2 * src + 1]);
        const double p = J.is_ket ? h * h : h;
        cum[i] = p;
      }
    }
    double total = 0.0;
    for (int64_t i = 0; i < D; ++i) { const double t = total + cum[i]; total = t; }
    double run = 0.0;
    for (int64_t i = 0; i < D; ++i) {
      const double w = cum[i] / total;
      const double t = run + w;
      run = t;
      cum[i] = run;
    }
    // multinomial: first index whose cumulative weight is >= the uniform (np.searchsorted, side = "left")
    const int64_t m = J.count[r], s0 = J.start[r];
    if ((int64_t)ind.size() < m) ind.resize(m);
    for (int64_t k = 0; k < m; ++k) {
      const double v = J.rnd[s0 + k];
      const int64_t pos = std::lower_bound(cum.begin(), cum.begin() + D, v) - cum.begin();
      if (pos >= D) return 1;  // beyond the last cumulative weight: NumPy would index past the histogram - the caller decides
      ind[k] = pos;
    }
    int64_t* hl = hist_local + (size_t)J.slot[r] * D;
    if (!flips) {
      for (int64_t k = 0; k < m; ++k) hl[ind[k]]++;
      continue;
    }
    // measurement flips (simresults.py:537-568): groups by outcome in order of first occurrence, matrix rows in that order
    order.clear();
    for (int64_t k = 0; k < m; ++k) {
      const int64_t o = ind[k];
      if (first_seen[o] < 0) { first_seen[o] = k; cnt[o] = 0; order.push_back(o); }
      cnt[o]++;
    }
    int64_t row = s0;
    for (int64_t o : order) {
      for (int64_t c = 0; c < cnt[o]; ++c, ++row) {
        const double* u = J.mat + (size_t)row * n;
        int64_t out = 0;
        for (int q = 0; q < n; ++q) {
          const int bit = (int)((o >> (n - 1 - q)) & 1);
          const double pflip = bit ? J.eps_p : J.eps;
          const int nb = bit ^ (u[q] < pflip ? 1 : 0);
          out |= (int64_t)nb << (n - 1 - q);
        }
        hl[out]++;
      }
      first_seen[o] = -1;
    }
  }
  return 0;
}

extern "C" int ryd_replay_samples(const void* states_host, int64_t n_rows, int64_t dim, int32_t n_qubits, int32_t is_ket,
                                  int32_t reversed, int32_t matching, const int64_t* start, const int64_t* count,
                                  const int32_t* slot, int32_t n_slots, const double* rnd, const double* flip_matrix,
                                  double eps, double eps_p, int64_t* hist, int32_t n_threads) {
  if (!states_host || !start || !count || !slot || !rnd || !hist) return fail(RYD_ERR_INVALID, "null argument");
  if (n_rows < 0 || dim < 1 || n_qubits < 1 || n_qubits > 30 || dim != ((int64_t)1 << n_qubits) || n_slots < 1)
    return fail(RYD_ERR_INVALID, "ryd_replay_samples: dim must be 2^n_qubits (two-level measurement outcomes)");
  for (int64_t r = 0; r < n_rows; ++r)
    if (slot[r] < 0 || slot[r] >= n_slots || count[r] < 0 || start[r] < 0) return fail(RYD_ERR_INVALID, "ryd_replay_samples: bad row %lld", (long long)r);
  ReplayJob J{(const double*)states_host, dim, n_qubits, is_ket, reversed, matching, start, count, slot, rnd, flip_matrix, eps, eps_p, hist};
  int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n_rows));
  std::vector<std::vector<int64_t>> local(nt);
  std::vector<int> rcs(nt, 0);
  auto work = [&](int t) {
    local[t].assign((size_t)n_slots * dim, 0);
    std::vector<double> cum(dim);
    std::vector<int64_t> ind, first_seen(dim, -1), cnt(dim, 0), order;
    const int64_t r0 = n_rows * t / nt, r1 = n_rows * (t + 1) / nt;
    rcs[t] = replay_rows(J, r0, r1, local[t].data(), cum, ind, first_seen, cnt, order);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  for (int t = 0; t < nt; ++t)
    if (rcs[t]) return fail(RYD_ERR_STATE, "ryd_replay_samples: a uniform lies beyond the last cumulative weight");
  for (int t = 0; t < nt; ++t)
    for (size_t i = 0; i < (size_t)n_slots * dim; ++i) hist[i] += local[t][i];
  return RYD_OK;
}
