// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Split-operator ket propagator for registers beyond one CU (15+ atoms): k_split
// ---------------------------------------------------------------------------
// Replaces the same seam as the Taylor / Lanczos exponentials (qutip.sesolve behind
// simulation.py:729-735) for two-level kets.  H(t) = D(t) + X(t):
//   D(t) = E0(s) - sum_k delta_k(t) n_k(s)      diagonal (hamiltonian.py:260-331 + detuning terms)
//   X(t) = sum_k c_k(t) |1><0|_k + h.c.         a SUM OF COMMUTING single-atom terms
// so exp(-i b X) = (x)_k R_k is an exact product of 2x2 rotations and exp(-i int D) an exact
// phase.  A symmetric composition  D(a_1) R(b_1) D(a_2) ... R(b_s) D(a_{s+1})  (4th order, 6 stages:
// Blanes & Moan 2002; time advances with the D flows, the drive is frozen at the current time) needs
// NO generator application: one stage = every amplitude read once and written once.
// One launch = one memory pass over the state, tile by tile: a workgroup holds 2^T amplitudes
// (16 per lane, 4 "register bits"), rotates the register bits in registers and turns the tile
// through LDS (XOR-swizzled 16-B slots) to make the next 4 bits the register bits.  Two tilings
// alternate (low bits | low 2T-N bits + high bits); a pass finishes the previous stage's rotation on the
// bits its tiling has and the previous one lacked, applies D, and starts the next rotation on all of its
// bits - so a stage costs ONE pass (32 B per amplitude + 8 B of E0) whatever N <= 2T - 3.
// Roofline: HBM / Infinity-Cache streaming, 40 B per amplitude and stage.

#include "split_types.hpp"

// out[stage][b][k] = (C, Re g, Im g, Delta): the rotation exp(-i beta (c |1><0| + conj(c) |0><1|)) =
// C + g |1><0| + g' |0><1| with the drive frozen at the stage's time, and the integral of the
// detuning over the stage's D intervals (2-point Gauss: exact on the cubic pieces).
__global__ __launch_bounds__(256) void k_split_coefs(const cplx* __restrict__ pp, int n_int,
                                                     const ryd_qdesc* __restrict__ desc,
                                                     const ryd_dterm* __restrict__ dterms, int total,
                                                     const SplitRun R, double* __restrict__ out) {
  // one wave per (trajectory, atom) when extra detuning terms have to be summed (high-frequency noise); one LANE per
  // (trajectory, atom) otherwise (dterms == nullptr: 256 items per workgroup - the wave-per-item launch of the 256 x
  // 14-atom batch took 205 us per closed run, 8 % of the bench step)
  const bool per_lane = dterms == nullptr;
  const int i = per_lane ? (int)(blockIdx.x * 256 + threadIdx.x) : (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int lane = per_lane ? 0 : (int)(threadIdx.x & 63);
  if (i >= total) return;
  const int ns = R.nsub;
  const int n_comp = splitrun_first(R, ns);  // stages of the compositions (mixed runs: not S ns)
  const bool pre_kick = R.kick_pre != 0.0 && blockIdx.y == 0;  // the extra stage 0: no D, rotation by kick_pre
  const int j = (int)blockIdx.y - (R.kick_pre != 0.0 ? 1 : 0);
  if (pre_kick) {
    if (lane == 0) {
      const ryd_qdesc d0 = desc[i];
      double C = 1.0, gi = 0.0;
      if (d0.drive_series >= 0) {
        const cplx* p = pp + ((size_t)d0.drive_series * n_int + R.kick_idx) * 4;
        const double cr = d0.drive_scale * fma(fma(fma(p[0].x, R.kick_u, p[1].x), R.kick_u, p[2].x), R.kick_u, p[3].x);
        double sn, cs;
        sincos(R.kick_pre * cr, &sn, &cs);
        C = cs;
        gi = -sn;
      }
      double* o = out + ((size_t)blockIdx.y * total + i) * 4;
      o[0] = C;
      o[1] = R.tan_form ? gi / C : 0.0;
      o[2] = gi;
      o[3] = 0.0;
    }
    return;
  }
  // records beyond the run's own S ns + 1 (launched only when the run takes evaluation-time snapshots inside the kernel):
  // record S ns + 1 + s closes sub-step s alone - the last D(a_{S+1}) of that sub-step, no rotation, the laboratory frame
  // (gauge) - for k_split_snap_close
  const bool snap_close = j > n_comp;
  const bool closing = j >= n_comp;
  int s, st;
  if (closing) {
    s = snap_close ? j - n_comp - 1 : ns - 1;
    st = splitrun_S(R, s);
  } else {
    splitrun_locate(R, j, s, st);
  }
  const int S = splitrun_S(R, s);
  // D intervals: (knot interval, start offset, length)
  int idx_d[2] = {R.idx[s], 0};
  double us_d[2], len_d[2] = {0.0, 0.0};
  double cum = 0.0;
  for (int l = 0; l < st; ++l) cum += splitrun_a(R, s, l);
  us_d[0] = R.u0[s] + cum * R.tau[s];
  len_d[0] = splitrun_a(R, s, st) * R.tau[s];
  us_d[1] = 0.0;
  if (!closing && st == 0 && s > 0) {
    const double a_last = splitrun_a(R, s - 1, splitrun_S(R, s - 1));  // the previous sub-step's own composition
    idx_d[1] = R.idx[s - 1];
    us_d[1] = R.u0[s - 1] + (1.0 - a_last) * R.tau[s - 1];
    len_d[1] = a_last * R.tau[s - 1];
  }
  (void)S;
  const int idx_c = closing ? R.kick_idx : R.idx[s];
  const double u_c = closing ? R.kick_u : us_d[0] + len_d[0];
  const double beta = snap_close ? 0.0 : closing ? R.kick_post : splitrun_b(R, s, st) * R.tau[s];

  const ryd_qdesc d = desc[i];
  auto val = [&](int sr, int idx, double u) -> cplx {
    const cplx* p = pp + ((size_t)sr * n_int + idx) * 4;
    cplx r = p[0];
    r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
    r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
    r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
    return r;
  };
  double dl = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (len_d[q] == 0.0) continue;
    const double ua = us_d[q] + len_d[q] * 0.21132486540518713, ub = us_d[q] + len_d[q] * 0.7886751345948129;
    double x = 0.0;
    if (lane == 0) {
      if (d.det_series >= 0)
        x += d.det_scale * (val(d.det_series, idx_d[q], ua).x + val(d.det_series, idx_d[q], ub).x);
      if (d.off_series >= 0)
        x += d.off_scale * (val(d.off_series, idx_d[q], ua).x + val(d.off_series, idx_d[q], ub).x);
    }
    if (d.extra > 0 && dterms) {
      const int count = dterms[d.extra - 1].remaining + 1;
      for (int e = lane; e < count; e += 64) {
        const ryd_dterm t = dterms[d.extra - 1 + e];
        x += t.scale * (val(t.series, idx_d[q], ua).x + val(t.series, idx_d[q], ub).x);
      }
      for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    }
    dl += 0.5 * len_d[q] * x;
  }
  if (lane == 0) {
    double C = 1.0, gr = 0.0, gi = 0.0;
    if (d.drive_series >= 0 && beta != 0.0) {
      const cplx a = val(d.drive_series, idx_c, u_c);
      double cr = d.drive_scale * a.x, ci = d.drive_scale * a.y;
      const double m = sqrt(cr * cr + ci * ci);
      if (R.gauge) { cr = m; ci = 0.0; }  // (the phase of c goes to the D factors, below)
      double sn, cs;
      sincos(beta * m, &sn, &cs);
      const double S = m > 1e-300 ? sn / m : beta;  // sin(beta |c|) / |c|
      C = cs;
      gr = S * ci;  // g = -i S c
      gi = -S * cr;
    }
    if (R.gauge && d.drive_series >= 0) {
      auto theta = [&](int idx, double u) {
        const cplx a = val(d.drive_series, idx, u);
        const double cr = d.drive_scale * a.x, ci = d.drive_scale * a.y;
        return (cr * cr + ci * ci > 1e-300) ? atan2(ci, cr) : 0.0;
      };
      const double th_cur = closing ? 0.0 : theta(idx_c, u_c);
      // the rotation before this stage's D: the previous stage of the sub-step, the last one of the previous sub-step,
      // or none (the run starts in the laboratory frame)
      const double th_prev = (st > 0) ? theta(R.idx[s], us_d[0]) : (s > 0 ? theta(idx_d[1], us_d[1]) : 0.0);
      dl += th_cur - th_prev;
    }
    if (R.tan_form == 1) gr = gi / C;  // (real drives: gr was 0; the host keeps |beta c| <= 1, so C >= 0.54)
    else if (R.tan_form == 2) { gr /= C; gi /= C; }  // complex drives on k_split_reg<.., CPLX>: both parts over C
    double* o = out + ((size_t)blockIdx.y * total + i) * 4;
    o[0] = C;
    o[1] = gr;
    o[2] = gi;
    o[3] = dl;
  }
}

__device__ static const double kSplitTrig[64][2] = {
    {1.0, 0.0}, {0.9951847266721969, 0.0980171403295606},
    {0.9807852804032304, 0.19509032201612825}, {0.9569403357322088, 0.29028467725446233},
    {0.9238795325112867, 0.3826834323650898}, {0.881921264348355, 0.47139673682599764},
    {0.8314696123025452, 0.5555702330196022}, {0.773010453362737, 0.6343932841636455},
    {0.7071067811865476, 0.7071067811865475}, {0.6343932841636455, 0.773010453362737},
    {0.5555702330196023, 0.8314696123025452}, {0.4713967368259978, 0.8819212643483549},
    {0.38268343236508984, 0.9238795325112867}, {0.29028467725446233, 0.9569403357322089},
    {0.19509032201612833, 0.9807852804032304}, {0.09801714032956077, 0.9951847266721968},
    {6.123233995736766e-17, 1.0}, {-0.09801714032956065, 0.9951847266721969},
    {-0.1950903220161282, 0.9807852804032304}, {-0.29028467725446216, 0.9569403357322089},
    {-0.3826834323650897, 0.9238795325112867}, {-0.4713967368259977, 0.881921264348355},
    {-0.555570233019602, 0.8314696123025455}, {-0.6343932841636454, 0.7730104533627371},
    {-0.7071067811865475, 0.7071067811865476}, {-0.773010453362737, 0.6343932841636455},
    {-0.8314696123025453, 0.5555702330196022}, {-0.8819212643483549, 0.47139673682599786},
    {-0.9238795325112867, 0.3826834323650899}, {-0.9569403357322088, 0.2902846772544624},
    {-0.9807852804032304, 0.1950903220161286}, {-0.9951847266721968, 0.09801714032956083},
    {-1.0, 1.2246467991473532e-16}, {-0.9951847266721969, -0.09801714032956059},
    {-0.9807852804032304, -0.19509032201612836}, {-0.9569403357322089, -0.2902846772544621},
    {-0.9238795325112868, -0.38268343236508967}, {-0.881921264348355, -0.47139673682599764},
    {-0.8314696123025455, -0.555570233019602}, {-0.7730104533627371, -0.6343932841636453},
    {-0.7071067811865477, -0.7071067811865475}, {-0.6343932841636459, -0.7730104533627367},
    {-0.5555702330196022, -0.8314696123025452}, {-0.47139673682599786, -0.8819212643483549},
    {-0.38268343236509034, -0.9238795325112865}, {-0.29028467725446244, -0.9569403357322088},
    {-0.19509032201612866, -0.9807852804032303}, {-0.09801714032956045, -0.9951847266721969},
    {-1.8369701987210297e-16, -1.0}, {0.09801714032956009, -0.9951847266721969},
    {0.1950903220161283, -0.9807852804032304}, {0.29028467725446205, -0.9569403357322089},
    {0.38268343236509, -0.9238795325112866}, {0.4713967368259976, -0.881921264348355},
    {0.5555702330196018, -0.8314696123025455}, {0.6343932841636456, -0.7730104533627369},
    {0.7071067811865474, -0.7071067811865477}, {0.7730104533627367, -0.6343932841636459},
    {0.8314696123025452, -0.5555702330196022}, {0.8819212643483548, -0.4713967368259979},
    {0.9238795325112865, -0.3826834323650904}, {0.9569403357322088, -0.2902846772544625},
    {0.9807852804032303, -0.19509032201612872}, {0.9951847266721969, -0.0980171403295605},
};

// exp(-i phi) by a 64-entry table of exp(i k pi/32) and a short series on |r| <= pi/64
__device__ __forceinline__ void split_sincos(double phi, const cplx* __restrict__ tab, double& c, double& s) {
  const double k = rint(phi * 10.185916357881302);  // 32 / pi
  double r = fma(-k, 0.09817477042468103, phi);
  r = fma(-k, 3.827021247335479e-18, r);  // pi/32 - double(pi/32)
  const int ki = ((int)k) & 63;
  const double r2 = r * r;
  const double cr = fma(r2, fma(r2, fma(r2, fma(r2, 2.48015873015873e-05, -1.388888888888889e-03),
                                        4.1666666666666664e-02), -0.5), 1.0);
  const double sr = r * fma(r2, fma(r2, fma(r2, fma(r2, 2.7557319223985893e-06, -1.984126984126984e-04),
                                            8.333333333333333e-03), -1.6666666666666666e-01), 1.0);
  const cplx t = tab[ki];
  c = fma(t.x, cr, -t.y * sr);
  s = fma(t.y, cr, t.x * sr);
}

// NT: workgroup size (>= 2^(T - 4): 16 amplitudes per active lane).  Tiles of 2^13 amplitudes (512 lanes,
// 128 KiB of LDS) give registers of 21 - 23 atoms ONE pass per stage (two tilings) where 2^12 tiles need three
// tilings = two passes: measured 0.62 -> 1.43 sim-us/s at 22 atoms although this runtime-indexed kernel
// takes 83 us per pass against 2 x 40 us of k_split12 (the wall clock also loses the second pass of every
// controller check).  A 2^14 tile (1024 lanes x 16 amplitudes, real and imaginary parts turned separately
// through 128 KiB) was built and measured too: at 128 registers per lane it spills (500 us per pass at 24
// atoms against 2 x 128 us) - 24+ atoms stay on 2^12 tiles.
template <int NT, bool DECAY>
__global__ __launch_bounds__(NT) void k_split_t(const SplitArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = A.T;
  const int G = (T + 3) >> 2;  // register-bit groups
  cplx* xs = reinterpret_cast<cplx*>(smem);                       // [2^T] swizzled slots
  cplx* trig = xs + ((size_t)1 << T);                             // [64] exp(i k pi/32)
  double* rot = reinterpret_cast<double*>(trig + 64);             // [2][SPLIT_TBIG][4]
  double* dlo = rot + 2 * SPLIT_TBIG * 4;                         // [128] detuning integrals of tile bits 0-6
  double* dhi = dlo + 128;                                        // [128] of tile bits 7-13
  double* cfs = dhi + 128;                                        // [2][SPLIT_NMAX][4] staged coefficients
  double* dlut = cfs + 2 * SPLIT_NMAX * 4;                        // [SPLIT_NMAX + 1] decay factors (quantum jumps)

  const int tid = threadIdx.x;
  const int N = A.N;
  const int b = blockIdx.y;
  const int nthr = 1 << (T - 4);
  const bool active = tid < nthr;
  const unsigned long long base = deposit((unsigned long long)blockIdx.x, A.outer);
  cplx* __restrict__ st = A.state + ((size_t)b << N);
  const double* __restrict__ cfin = A.cfin + (size_t)b * N * 4;
  const double* __restrict__ ccur = A.ccur + (size_t)b * N * 4;

  auto pos_of = [&](int g) { return min(4 * g, T - 4); };
  auto idx_of = [&](int r, int pos) -> unsigned {
    return ((unsigned)tid & ((1u << pos) - 1u)) | ((unsigned)r << pos) | (((unsigned)tid >> pos) << (pos + 4));
  };

  // bit q belongs to group q/4 below the top group, else to the top group
  auto group_mask = [&](int gg, unsigned mask) -> unsigned {
    unsigned m = 0;
    for (int q = 0; q < T; ++q) {
      const int owner = (q >> 2) < G - 1 ? (q >> 2) : G - 1;
      if (owner == gg && ((mask >> q) & 1u)) m |= 1u << q;
    }
    return m;
  };

  // ---- load the tile in the top layout (lanes run over the low tile bits: coalesced) ----
  int g = G - 1;
  cplx x[16];
  {
    const int pos = pos_of(g);
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = st[base | deposit((unsigned long long)idx_of(r, pos), A.tile)];
    }
  }
  // E0 of the amplitudes in the layout D will find them in (after the finishing rotations)
  int gD = G - 1;
  for (int gg = G - 1; gg >= 0; --gg)
    if (group_mask(gg, A.fin_mask)) gD = gg;
  // E0 is prefetched with the tile (its latency hides behind the finishing rotations)
  double ev[16];
  if (A.do_diag && active) {
    const int pos = pos_of(gD);
    const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) ev[r] = e0[base | deposit((unsigned long long)idx_of(r, pos), A.tile)];
  }

  // ---- per-pass tables: stage the coefficients with one coalesced read, then build from LDS ----
  if (tid < 4 * N) {
    cfs[tid] = cfin[tid];
    cfs[4 * SPLIT_NMAX + tid] = ccur[tid];
  }
  if (tid >= 128 && tid < 192) trig[tid - 128] = make_double2(kSplitTrig[tid - 128][0], kSplitTrig[tid - 128][1]);
  if constexpr (DECAY) {  // factor by number of excited atoms ne: popc(index) = N - ne
    if (tid >= 192 && tid < 192 + SPLIT_NMAX + 1) dlut[tid - 192] = exp(A.wE * (A.dec_a + A.dec_b * (double)(N - ((int)tid - 192))));
  }
  __syncthreads();
  if (tid < 2 * T) {
    const int set = tid / T, q = tid % T;
    const int k = N - 1 - tile_bit_pos(A.tile, q);
    const double* c = cfs + set * 4 * SPLIT_NMAX + 4 * k;
    double* o = rot + (set * SPLIT_TBIG + q) * 4;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
  }
  double d_outer = 0.0;
  int nexc_outer = 0;  // excited atoms among the bits outside the tile (fixed per workgroup)
  if (A.do_diag) {
    const double* cc = cfs + 4 * SPLIT_NMAX;
    // detuning integral part of the phase: sum_k Delta_k n_k, n_k = 1 - bit
    if (tid < 256) {
      const int e = tid;
      const bool hiHalf = e >= 128;
      const int v = e & 127;
      const int q0 = hiHalf ? 7 : 0;
      const int nq = hiHalf ? max(T - 7, 0) : min(T, 7);
      double s = 0.0;
      for (int q = 0; q < nq; ++q)
        if (!((v >> q) & 1)) s += cc[4 * (N - 1 - tile_bit_pos(A.tile, q0 + q)) + 3];
      (hiHalf ? dhi : dlo)[v] = s;
    }
    for (int p = 0; p < N; ++p) {
      // bits outside the tile are fixed per workgroup
      bool in_tile = false;
#pragma unroll
      for (int i = 0; i < 3; ++i) in_tile |= (p >= A.tile.lo[i] && p < A.tile.lo[i] + A.tile.len[i]);
      if (!in_tile && !((base >> p) & 1ull)) { d_outer += cc[4 * (N - 1 - p) + 3]; ++nexc_outer; }
    }
  }
  __syncthreads();

  auto rotate = [&](int gg, unsigned mask, int set) {
    const int pos = pos_of(gg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = pos + j;
      // bit q belongs to group q/4 below the top group, else to the top group
      const int owner = (q >> 2) < G - 1 ? (q >> 2) : G - 1;
      if (!((mask >> q) & 1u) || owner != gg) continue;
      const double* c = rot + (set * SPLIT_TBIG + q) * 4;
      const double C = c[0], gr = c[1], gi = c[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (r & (1 << j)) continue;
        const cplx a0 = x[r], a1 = x[r | (1 << j)];
        // y0 = C a0 + g' a1,  g' = (-gr, gi);   y1 = C a1 + g a0,  g = (gr, gi)
        x[r] = make_double2(fma(-gr, a1.x, fma(-gi, a1.y, C * a0.x)), fma(-gr, a1.y, fma(gi, a1.x, C * a0.y)));
        x[r | (1 << j)] = make_double2(fma(gr, a0.x, fma(-gi, a0.y, C * a1.x)), fma(gr, a0.y, fma(gi, a0.x, C * a1.y)));
      }
    }
  };
  auto turn = [&](int from, int to) {  // registers -> LDS -> registers in the layout of group `to`
    const int p0 = pos_of(from), p1 = pos_of(to);
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned i = idx_of(r, p0);
        xs[i ^ ((i >> 4) & 15u)] = x[r];
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned i = idx_of(r, p1);
        x[r] = xs[i ^ ((i >> 4) & 15u)];
      }
    }
    __syncthreads();
  };

  // ---- finish the previous stage's rotation ----
  for (int s = 0; s < G; ++s) {
    const int gg = G - 1 - s;
    if (!group_mask(gg, A.fin_mask)) continue;
    if (gg != g) { turn(g, gg); g = gg; }
    rotate(gg, A.fin_mask, 0);
  }

  // ---- D: exact phase of the diagonal ----
  if (A.do_diag && active) {
    const int pos = pos_of(g);  // == gD
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned i = idx_of(r, pos);
      const double phi = fma(A.wE, ev[r], -(d_outer + dlo[i & 127u] + dhi[i >> 7]));
      double c, s;
      split_sincos(phi, trig, c, s);
      if constexpr (DECAY) {
        const double f = dlut[nexc_outer + T - __popc(i)];
        c *= f;
        s *= f;
      }
      const cplx a = x[r];
      x[r] = make_double2(fma(a.x, c, a.y * s), fma(a.y, c, -a.x * s));  // a * (c - i s)
    }
  }

  // ---- start this stage's rotation on every bit of the tile ----
  {
    const int g0 = g;
    for (int s = 0; s < G; ++s) {
      const int gg = (g0 - s + G) % G;
      if (!group_mask(gg, A.cur_mask)) continue;
      if (gg != g) { turn(g, gg); g = gg; }
      rotate(gg, A.cur_mask, 1);
    }
  }
  if (g != G - 1) { turn(g, G - 1); g = G - 1; }
  {
    const int pos = pos_of(g);
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[base | deposit((unsigned long long)idx_of(r, pos), A.tile)] = x[r];
    }
  }
}

// The same pass with everything static, for tiles of 2^12 (N <= 20: one pass per stage) and 2^13 amplitudes (21 - 22
// atoms: one pass per stage): 256 lanes x R = 2^(T - 8) amplitudes.  Layouts (tile-local index of register r)
//   LA: tid | r << 8                                                    register bits 8 .. T-1 (the load / store layout)
//   LB: (tid & 15) | (r & 15) << 4 | (tid >> 4) << 8 | (r >> 4) << 12   register bits 4-7 (+ bit 12 along for the ride)
//   LC: (r & 15) | tid << 4 | (r >> 4) << 12                            register bits 0-3 (+ bit 12)
// in the fixed order
//   LA: finish bits 8.. | LB: finish bits 4-7, D, start bits 4-7 | LC: start bits 0-3 | LA: start bits 8..
// (the tilings keep >= 4 low bits, so a finishing rotation never touches bits 0-3).  LDS slots (XOR-swizzled: i ^ (i >> 4 &
// 15)) and global offsets fold into immediates; REAL: every drive coefficient is real (g = -i S c is imaginary).
// TAN (real drives, SplitRun.tan_form = 1): rotations in tan form  x' = x - T y_p, y' = y + T x_p  (T = Im g / C in the
// Re g slot: 2 FMAs per amplitude and bit instead of 4 instructions); the product of the N cosines of a stage's rotation,
// one number per sequence, rides on the phase factors of the NEXT D (A.pend: a rotation is pending; its coefficients = cfin).
// Round 4, measured at 20 atoms (tools/pass_variants.sh, profiles/r04_cfg5_pass_variants.md): 17.6 -> 12.8 us per pass -
// the per-pass tables behind ONE barrier (every entry from independent global loads), no branch per bit in the rotation
// code (bits outside the pass's masks rotate by the identity; the branches had left 843 v_mov_b64 copies at their joins),
// tan form, non-temporal stores.  What remains: ~3 us launch-to-launch floor + ~6 us of loads and store drain + ~3.8 us
// of vector instructions on one wave per SIMD.
#ifndef SPLITS_NT
#define SPLITS_NT 1  /* 1: non-temporal stores of the state (20 atoms: 13.4 -> 12.8 us per pass); 2: loads too (no gain) */
#endif
template <int T, bool REAL, bool DECAY, bool TAN = false>
__global__ __launch_bounds__(SPLIT_NT) void k_split_s(const SplitArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(T == 12 || T == 13, "tiles of 2^12 or 2^13 amplitudes");
  constexpr int RB = T - 8, R = 1 << RB;  // register bits of LA, amplitudes per lane
  constexpr int NHI = 1 << (T - 6);       // entries of the detuning table of tile bits 6 .. T-1
  cplx* xs = reinterpret_cast<cplx*>(smem);
  cplx* trig = xs + (1 << T);
  double* rot = reinterpret_cast<double*>(trig + 64);  // [2][SPLIT_TS][4]
  double* dlo = rot + 2 * SPLIT_TS * 4;                // [64]
  double* dhi = dlo + 64;                              // [128]
  double* cfs = dhi + 128;                             // [4] per-workgroup scalars
  double* dlut = cfs + 4;                              // [SPLIT_NMAX + 1]

  const unsigned tid = threadIdx.x;
  const int N = A.N;
  const int b = blockIdx.y;
  const unsigned long long base = deposit((unsigned long long)blockIdx.x, A.outer);
  cplx* __restrict__ st = A.state + ((size_t)b << N) + base;
  const double* __restrict__ cfin = A.cfin + (size_t)b * N * 4;
  const double* __restrict__ ccur = A.ccur + (size_t)b * N * 4;

  const unsigned long long gA = deposit((unsigned long long)tid, A.tile);
  const unsigned long long gB = deposit((unsigned long long)((tid & 15u) | ((tid >> 4) << 8)), A.tile);
  auto offA = [&](int r) { return gA | deposit((unsigned long long)(r << 8), A.tile); };
  auto offB = [&](int r) { return gB | deposit((unsigned long long)(((r & 15) << 4) | ((r >> 4) << 12)), A.tile); };
  cplx x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const cplx* src = st + offA(r);
    if (SPLITS_NT & 2) x[r] = make_double2(__builtin_nontemporal_load(&src->x), __builtin_nontemporal_load(&src->y));
    else x[r] = *src;
  }
  double ev[R];
  if (A.do_diag) {
    const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride + base;
#pragma unroll
    for (int r = 0; r < R; ++r) ev[r] = e0[offB(r)];
  }

  // Per-pass tables, ONE barrier: every entry comes from independent global loads of the coefficient rows (L2 hits after
  // the first workgroup of an XCD).
  if (tid < 64) trig[tid] = make_double2(kSplitTrig[tid][0], kSplitTrig[tid][1]);
  if (tid < 2 * T) {
    const int set = tid / T, q = tid % T;
    const int k = N - 1 - tile_bit_pos(A.tile, q);
    const double2* c = reinterpret_cast<const double2*>((set ? ccur : cfin) + 4 * k);
    double2 c01 = c[0], c23 = c[1];
    double2* o = reinterpret_cast<double2*>(rot + (set * SPLIT_TS + q) * 4);
    // every rotation runs, bits outside the pass's masks as the identity (C = 1, g = 0; TAN: T = 0)
    if (!(((set ? A.cur_mask : A.fin_mask) >> q) & 1u)) { c01 = make_double2(1.0, 0.0); c23.x = 0.0; }
    o[0] = c01;
    o[1] = c23;
  }
  if (A.do_diag) {
    if (tid >= 64 && tid < 128) {  // bits outside the tile (fixed per workgroup): lane = bit
      const int p = tid - 64;
      bool out_exc = p < N && !((base >> p) & 1ull);
#pragma unroll
      for (int i = 0; i < 3; ++i) out_exc &= !(p >= A.tile.lo[i] && p < A.tile.lo[i] + A.tile.len[i]);
      double v = out_exc ? ccur[4 * (N - 1 - p) + 3] : 0.0;
      const unsigned long long mk = __ballot(out_exc);
      double pc = 1.0;
      if constexpr (TAN) pc = (A.pend && p < N) ? cfin[4 * p] : 1.0;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        v += __shfl_xor(v, o, 64);
        if constexpr (TAN) pc *= __shfl_xor(pc, o, 64);
      }
      if (p == 0) { cfs[0] = v; cfs[1] = (double)__popcll(mk); cfs[2] = pc; }
      if constexpr (DECAY) {  // factor by number of excited atoms ne: popc(index) = N - ne
        if (p < SPLIT_NMAX + 1) dlut[p] = exp(A.wE * (A.dec_a + A.dec_b * (double)(N - p)));
      }
    }
    if (tid >= 128) {  // detuning integral part of the phase, sum_k Delta_k n_k (n_k = 1 - bit): tile bits 0-5 | 6 .. T-1
#pragma unroll
      for (int it = 0; it < (64 + NHI + 127) / 128; ++it) {
        const int e = (int)tid - 128 + 128 * it;
        if (e >= 64 + NHI) break;
        const bool hiHalf = e >= 64;
        const int v = hiHalf ? e - 64 : e;
        const int q0 = hiHalf ? 6 : 0;
        const int nq = hiHalf ? T - 6 : 6;
        double dq[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) dq[q] = q < nq ? ccur[4 * (N - 1 - tile_bit_pos(A.tile, q0 + q)) + 3] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q)
          if (!((v >> q) & 1)) s += dq[q];
        (hiHalf ? dhi : dlo)[v] = s;
      }
    }
  }
  __syncthreads();
  const double d_outer = A.do_diag ? cfs[0] : 0.0;
  const int nexc_outer = A.do_diag ? (int)cfs[1] : 0;
  const double pend = (TAN && A.do_diag) ? cfs[2] : 1.0;

  // rotations of register bits 0 .. nb-1 = tile-local bits [pos, pos + nb)
  auto rotate = [&](int pos, int nb, int set) {
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (j >= nb) continue;
      // (TAN: T by scalar loads from the coefficient rows instead of this LDS table: 15.0 -> 17.1 us per pass at 20 atoms)
      const double* c = rot + (set * SPLIT_TS + pos + j) * 4;
      const double C = c[0], gr = c[1], gi = c[2];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r & (1 << j)) continue;
        const cplx a0 = x[r], a1 = x[r | (1 << j)];
        if (TAN) {  // (gr holds T = gi / C)
          x[r] = make_double2(fma(-gr, a1.y, a0.x), fma(gr, a1.x, a0.y));
          x[r | (1 << j)] = make_double2(fma(-gr, a0.y, a1.x), fma(gr, a0.x, a1.y));
        } else if (REAL) {
          x[r] = make_double2(fma(-gi, a1.y, C * a0.x), fma(gi, a1.x, C * a0.y));
          x[r | (1 << j)] = make_double2(fma(-gi, a0.y, C * a1.x), fma(gi, a0.x, C * a1.y));
        } else {
          x[r] = make_double2(fma(-gr, a1.x, fma(-gi, a1.y, C * a0.x)), fma(-gr, a1.y, fma(gi, a1.x, C * a0.y)));
          x[r | (1 << j)] = make_double2(fma(gr, a0.x, fma(-gi, a0.y, C * a1.x)), fma(gr, a0.y, fma(gi, a0.x, C * a1.y)));
        }
      }
    }
  };
  // swizzled LDS slot of tile-local index i: i ^ ((i >> 4) & 15)
  const unsigned sA = tid ^ ((tid >> 4) & 15u);             // + (r << 8)
  const unsigned sB = (tid & 15u) | ((tid >> 4) << 8);      // ((tid & 15) ^ (r & 15)) | (r & 15) << 4 | (tid >> 4) << 8 | (r >> 4) << 12
  const unsigned sC = tid << 4;                             // ((r & 15) ^ (tid & 15)) | tid << 4 | (r >> 4) << 12
  auto slotA = [&](int r) { return sA + (unsigned)(r << 8); };
  auto slotB = [&](int r) { return (sB ^ (unsigned)(r & 15)) | (unsigned)((r & 15) << 4) | (unsigned)((r >> 4) << 12); };
  auto slotC = [&](int r) { return sC | ((unsigned)(r & 15) ^ (tid & 15u)) | (unsigned)((r >> 4) << 12); };

  rotate(8, RB, 0);
  // ---- LA -> LB ----
#pragma unroll
  for (int r = 0; r < R; ++r) xs[slotA(r)] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = xs[slotB(r)];
  rotate(4, 4, 0);
  if (A.do_diag) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const unsigned i = sB | (unsigned)((r & 15) << 4) | (unsigned)((r >> 4) << 12);
      const double phi = fma(A.wE, ev[r], -(d_outer + dlo[i & 63u] + dhi[i >> 6]));
      double c, s;
      split_sincos(phi, trig, c, s);
      if constexpr (DECAY) {
        const double f = dlut[nexc_outer + T - __popc(i)];
        c *= f;
        s *= f;
      }
      if constexpr (TAN) {
        c *= pend;
        s *= pend;
      }
      const cplx a = x[r];
      x[r] = make_double2(fma(a.x, c, a.y * s), fma(a.y, c, -a.x * s));
    }
  }
  rotate(4, 4, 1);
  __syncthreads();
  // ---- LB -> LC ----
#pragma unroll
  for (int r = 0; r < R; ++r) xs[slotB(r)] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = xs[slotC(r)];
  rotate(0, 4, 1);
  __syncthreads();
  // ---- LC -> LA ----
#pragma unroll
  for (int r = 0; r < R; ++r) xs[slotC(r)] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = xs[slotA(r)];
  rotate(8, RB, 1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    cplx* dst = st + offA(r);
    if (SPLITS_NT & 1) { __builtin_nontemporal_store(x[r].x, &dst->x); __builtin_nontemporal_store(x[r].y, &dst->y); }
    else *dst = x[r];
  }
}

// (k_split12_loop, the 12-atom one-launch loop of round 2 with its parity branch, is gone: real drives run on k_split_reg<12>
// (k_split_reg.hpp), complex drives pass by pass on k_split_s<12>.)

// Whole kets of exactly 14 atoms, one workgroup (512 lanes x 32 amplitudes = 5 register bits) per sequence: every stage
// of a closed run in ONE launch, the ket stays in registers (the headline batch: 256 sequences = one per CU).  A stage
// is D + the 14 rotations; three layouts
//   LA: i = t | r << 9 (register bits 9-13, the coalesced load / store layout)     LB: i = (t & 15) | r << 4 | (t >> 4) << 9
//   LC: i = r | t << 5 (register bits 0-4)                                         (register bits 4-8; rotates 5-8)
// alternate LA -> LB -> LC on even stages and LC -> LB -> LA on odd ones: two turns per stage.  A turn moves the real
// parts and then the imaginary parts through 132 KiB of LDS (8-B slots, padded per turn so that both sides of both
// turns are conflict-free per half-wave and every address is a lane base + an immediate).  The coefficients of a
// stage are uniform: scalar loads, SGPR operands.  E0 is never read inside the stage loop (pairwise additivity: 12 doubles per lane + two 32-entry tables, see below).
// Arithmetic per amplitude and stage (real drives): a rotation is  x' = x - T y_p,  y' = y + T x_p  with T = gi / C
// (SplitRun.tan_form: k_split_coefs stores it in the unused Re g slot; host_split.hpp keeps |beta c| <= 1) - 2 FMAs per
// amplitude and bit instead of 2 + 2 - and the product of the 14 cosines of a stage, one number, rides on the next
// stage's phase factor; exp(-i phi) from a 512-entry table of exp(2 pi i k / 512) (built once per launch) and a
// degree-5 series on |r| <= pi / 512.
#define SPLIT14_NT 512
#define SPLIT14_TRIG 512
#define SPLIT14_SLOTS (16384 + 512)
template <bool REAL>
__global__ __launch_bounds__(SPLIT14_NT) void k_split14_loop(const SplitArgs A, const SplitRun R, long long stage_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int N = 14;
  double* xs = reinterpret_cast<double*>(smem);             // (2^14 + 512 pads) x 8 B
  cplx* trig = reinterpret_cast<cplx*>(xs + SPLIT14_SLOTS);  // 512 x 16 B

  const unsigned t = threadIdx.x;
  const int b = blockIdx.y;
  const int n_stages = R.S * R.nsub + 1;
  cplx* __restrict__ st = A.state + ((size_t)b << N);
  const double* __restrict__ coefs = A.ccur + (size_t)b * N * 4;
  const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;

  double* eregA = reinterpret_cast<double*>(trig + SPLIT14_TRIG);  // 32: E0 of the register atoms alone, layout LA
  double* eregC = eregA + 32;                                      // ... layout LC
  {  // (before the state is loaded: the library routine wants registers)
    double sn, cs;
    sincospi((double)t * (2.0 / SPLIT14_TRIG), &sn, &cs);
    trig[t] = make_double2(cs, sn);
  }
  // E0 is pairwise additive, so for an amplitude (lane t, register r) it is
  //   E0 = Et(t) + Ereg(r) + sum over the excited register atoms j of V_j(t)
  // with Et = E0 of the lane's atoms alone, Ereg = E0 of the register atoms alone (uniform: an LDS table of 32), and
  // V_j = the interaction of register atom j with the lane's excited atoms - all read off the E0 table once per
  // launch (12 doubles per lane for the two layouts D meets), so no stage reads E0 from memory.
  double etA, vA[5], etC, vC[5];
  etA = e0[t | (31u << 9)];
  etC = e0[31u | (t << 5)];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    vA[j] = e0[t | ((31u ^ (1u << j)) << 9)] - etA;
    vC[j] = e0[(31u ^ (1u << j)) | (t << 5)] - etC;
  }
  if (t < 32) {
    eregA[t] = e0[(t << 9) | 511u];
    eregC[t] = e0[t | (511u << 5)];
  }
  __syncthreads();
  double xr[32], xi[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const cplx v = st[t | (unsigned)(r << 9)];
    xr[r] = v.x;
    xi[r] = v.y;
  }

  // tile-local index of register r in the three layouts
  auto iA = [](unsigned tt, int r) -> unsigned { return tt | (unsigned)(r << 9); };
  auto iB = [](unsigned tt, int r) -> unsigned { return (tt & 15u) | (unsigned)(r << 4) | ((tt >> 4) << 9); };
  auto iC = [](unsigned tt, int r) -> unsigned { return (unsigned)r | (tt << 5); };
  // LDS slots (8 B) of index i: padded, not XOR-swizzled, so that every address is (a per-lane base) + (a
  // compile-time offset of the register): T1 (LA <-> LB) pads 16 slots per 512, T2 (LB <-> LC) one slot per 32.
  //   LA side of T1: t + 528 r            LB side of T1: [a + 528 T] + 16 r'           (a = t & 15, T = t >> 4)
  //   LB side of T2: [a + 528 T] + 16 r + (r >> 1)                LC side of T2: 33 t + r
  // Banks (4 B, 64 of them; a ds_*_b64 serves 32 lanes per pass): LA: consecutive lanes; LB (either turn): 2a (+ 32
  // for odd T); LC: 2 (t + r) mod 64 - all conflict-free.
  auto slot1 = [](unsigned i) -> unsigned { return i + ((i >> 9) << 4); };
  auto slot2 = [](unsigned i) -> unsigned { return i + (i >> 5); };

  // the coefficients of the current stage, by index bit p (atom N - 1 - p): uniform values, read through the constant
  // address space so that they are scalar loads (all of a stage's issued back to back, one wait) into scalar
  // registers - free operands of the vector arithmetic, no vector registers held.  (As vector loads +
  // v_readfirstlane the compiler waited per atom; fetched a stage ahead through lanes + v_readlane: slower)
  typedef const __attribute__((address_space(4))) double* cptr_t;
  double cC[N], cT[N], cGi[N], cDl[N];
  double cprod = 1.0;  // product of the cosines of the previous stage's rotations (tan form): rides on this stage's phase
  double cnext = 1.0;
  auto load_coefs = [&](const double* cs) {
    cptr_t c4 = (cptr_t)(unsigned long long)cs;
    double prod = 1.0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      cptr_t c = c4 + 4 * (N - 1 - p);
      if (REAL) {
        prod *= c[0];
        cC[p] = 1.0;
        cGi[p] = 0.0;
      } else {
        cC[p] = c[0];
        cGi[p] = c[2];
      }
      cT[p] = c[1];  // REAL: gi / C;  else Re g
      cDl[p] = c[3];
    }
    cprod = cnext;
    cnext = REAL ? prod : 1.0;
  };
  // rotations of the register bits [lo, hi) at index bits pos + j
  auto rotate = [&](int pos, int lo, int hi) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < lo || j >= hi) continue;
      const double C = cC[pos + j], T = cT[pos + j], gi = cGi[pos + j];
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        if (r & (1 << j)) continue;
        const int q = r | (1 << j);
        const double a0x = xr[r], a0y = xi[r], a1x = xr[q], a1y = xi[q];
        if (REAL) {
          xr[r] = fma(-T, a1y, a0x);
          xi[r] = fma(T, a1x, a0y);
          xr[q] = fma(-T, a0y, a1x);
          xi[q] = fma(T, a0x, a1y);
        } else {
          xr[r] = fma(-T, a1x, fma(-gi, a1y, C * a0x));
          xi[r] = fma(-T, a1y, fma(gi, a1x, C * a0y));
          xr[q] = fma(T, a0x, fma(-gi, a0y, C * a1x));
          xi[q] = fma(T, a0y, fma(gi, a0x, C * a1y));
        }
      }
    }
  };
  // D in layout LA (register bits 9-13, lane bits 0-8) or LC (register bits 0-4, lane bits 5-13):
  //   phi(r) = wE E0 - sum over excited atoms of Delta = [wE Et - Delta(lane atoms)] + wE Ereg(r) + sum_{j excited in r} Q_j,
  //   Q_j = wE V_j - Delta_j  (a tree over r: one addition per amplitude)
  auto phase = [&](double wE, auto inA) {
    constexpr bool kA = decltype(inA)::value;
    constexpr int rpos = kA ? 9 : 0, tpos = kA ? 0 : 5;
    double dthr = 0.0;
#pragma unroll
    for (int p = 0; p < 9; ++p) dthr += ((t >> p) & 1u) ? 0.0 : cDl[tpos + p];
    const double base0 = fma(wE, kA ? etA : etC, -dthr);
    double Q[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) Q[j] = fma(wE, kA ? vA[j] : vC[j], -cDl[rpos + j]);
    const double* ereg = kA ? eregA : eregC;
    const double scale = cprod;
    double P[32];
#pragma unroll
    for (int r = 31; r >= 0; --r) {
      if (r == 31) {
        P[r] = base0;
      } else {
        int j = 0;
        while (r & (1 << j)) ++j;  // lowest clear bit: the parent has it set
        P[r] = P[r | (1 << j)] + Q[j];
      }
      const double phi = fma(wE, ereg[r], P[r]);
      // exp(-i phi) = (c, -s): table of exp(2 pi i k / 512), series on |rr| <= pi / 512
      const double kk = rint(phi * (SPLIT14_TRIG / 6.283185307179586));
      double rr = fma(-kk, 6.283185307179586 / SPLIT14_TRIG, phi);
      rr = fma(-kk, 2.4492935982947064e-16 / SPLIT14_TRIG, rr);  // 2 pi - double(2 pi)
      const cplx tb = trig[((int)kk) & (SPLIT14_TRIG - 1)];
      const double r2 = rr * rr;
      const double cr = fma(r2, fma(r2, 4.1666666666666664e-02, -0.5), 1.0) * scale;
      const double sr = rr * fma(r2, fma(r2, 8.333333333333333e-03, -1.6666666666666666e-01), 1.0) * scale;
      const double c = fma(tb.x, cr, -tb.y * sr), sn = fma(tb.y, cr, tb.x * sr);
      const double ax = xr[r], ay = xi[r];
      xr[r] = fma(ax, c, ay * sn);
      xi[r] = fma(ay, c, -ax * sn);
    }
  };
  auto turn = [&](auto from, auto to, auto slot) {
    const unsigned tt = t;  // (lane bases: 4 registers held across the loop; the 14 scratch instructions left in the
                            // loop are reloads of the 12 E0 pieces - no stores, no HBM traffic)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) xs[slot(from(tt, r))] = xr[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) xr[r] = xs[slot(to(tt, r))];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) xs[slot(from(tt, r))] = xi[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) xi[r] = xs[slot(to(tt, r))];
  };

  // weight of E0 in the D of stage j: a_i tau (+ the last a tau carried over from the previous sub-step)
  auto stage_w = [&](int j) -> double {
    const bool last = j == n_stages - 1;
    const int sub = last ? R.nsub - 1 : j / R.S, i = last ? R.S : j % R.S;
    double w = R.a[i] * R.tau[sub];
    if (!last && i == 0 && sub > 0) w += R.a[R.S] * R.tau[sub - 1];
    return w;
  };
  // Two stages per iteration, straight-line (n_stages is odd - S is even - so the closing stage is an even one after
  // the loop).  An if / else over the stage parity inside the loop made the two bodies use different registers for
  // the state: 64 v_mov_b64 at every merge and, with both copies live, the spills that put 5 GB per launch on HBM.
  for (int sgi = 0; sgi + 1 < n_stages; sgi += 2) {
    load_coefs(coefs + (size_t)sgi * stage_stride);
    phase(stage_w(sgi), std::true_type{});
    rotate(9, 0, 5);
    turn(iA, iB, slot1);
    rotate(4, 1, 5);
    turn(iB, iC, slot2);
    rotate(0, 0, 5);
    load_coefs(coefs + (size_t)(sgi + 1) * stage_stride);
    phase(stage_w(sgi + 1), std::false_type{});
    rotate(0, 0, 5);
    turn(iC, iB, slot2);
    rotate(4, 1, 5);
    turn(iB, iA, slot1);
    rotate(9, 0, 5);
  }
  load_coefs(coefs + (size_t)(n_stages - 1) * stage_stride);
  phase(stage_w(n_stages - 1), std::true_type{});  // the closing D
#pragma unroll
  for (int r = 0; r < 32; ++r) st[t | (unsigned)(r << 9)] = make_double2(xr[r], xi[r]);
}

// err[b] = max |x - y|^2 over the amplitudes, err[B + b] = their sum (local-error estimate of the step-size controller);
// non-negative doubles order like their bit patterns
__global__ __launch_bounds__(256) void k_split_diff(const cplx* __restrict__ x, const cplx* __restrict__ y, int nb,
                                                    double* err) {
  const size_t D = (size_t)1 << nb;
  const size_t boff = (size_t)blockIdx.y * D;
  double s = 0.0, q = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx a = x[boff + i], c = y[boff + i];
    const double dx = a.x - c.x, dy = a.y - c.y;
    const double d2 = fma(dx, dx, dy * dy);
    s = fmax(s, d2);
    q += d2;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s = fmax(s, __shfl_down(s, o, 64));
    q += __shfl_down(q, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned long long*>(err + blockIdx.y), (unsigned long long)__double_as_longlong(s));
    atomicAdd(err + gridDim.y + blockIdx.y, q);  // err[B + b] = sum |x - y|^2 (round 6: the controller's 2-norm)
  }
}

// Evaluation-time snapshots taken inside a closed run of k_split_reg<.., SNAP> hold OPEN states (SplitArgs.snaps): this
// kernel closes them in place, all of a run at once and across the chip (the run itself sits on one CU per sequence):
//   psi_closed[i] = cprod * exp(-i (wE E0[i] - sum_{k excited in i} Delta_k)) * psi_open[i]
// with wE = a_{S+1} tau of the sub-step, Delta_k from the sub-step's closing record (k_split_coefs: record S nsub + 1 + s)
// and cprod = the product of the cosines of the sub-step's last tan-form rotation (record S (s + 1) - 1 + n_pre, field 0; 1
// when the run is not in tan form).  grid (2^N / 256, B, marked sub-steps).
__global__ __launch_bounds__(256) void k_split_snap_close(cplx* __restrict__ snaps, long long snap_stride,
                                                          const double* __restrict__ e0, long long e0_stride,
                                                          const double* __restrict__ coefs, long long stage_stride, int N,
                                                          const SplitRun R, const SplitSnapList Ls) {
  __shared__ double dl[SPLIT_NMAX];
  __shared__ double cp;
  const int b = blockIdx.y, q = blockIdx.z;
  const int s = Ls.sub[q];
  const int n_pre = R.kick_pre != 0.0 ? 1 : 0;
  const double* cl = coefs + (size_t)(splitrun_first(R, R.nsub) + 1 + n_pre + s) * stage_stride + (size_t)b * N * 4;
  const double* lastrot = coefs + (size_t)(splitrun_first(R, s + 1) - 1 + n_pre) * stage_stride + (size_t)b * N * 4;
  if (threadIdx.x < (unsigned)N) dl[threadIdx.x] = cl[4 * (N - 1 - (int)threadIdx.x) + 3];  // by index bit
  if (threadIdx.x == 64) {
    double p = 1.0;
    if (R.tan_form)
      for (int k = 0; k < N; ++k) p *= lastrot[4 * k];
    cp = p;
  }
  __syncthreads();
  const double wE = splitrun_a(R, s, splitrun_S(R, s)) * R.tau[s];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ((size_t)1 << N)) return;
  double phi = wE * e0[(size_t)b * e0_stride + i];
  for (int k = 0; k < N; ++k) phi -= ((i >> k) & 1) ? 0.0 : dl[k];
  double sn, cs;
  sincos(phi, &sn, &cs);
  cplx* v = snaps + (size_t)Ls.slot[q] * snap_stride + ((size_t)b << N) + i;
  const cplx a = *v;
  const double c = cp;
  *v = make_double2(c * fma(a.x, cs, a.y * sn), c * fma(a.y, cs, -a.x * sn));
}
