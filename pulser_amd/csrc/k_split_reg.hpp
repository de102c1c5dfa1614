// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// k_split_reg<N, NR>: register-resident split-operator kets, 2^NR amplitudes per lane (round 4)
// ---------------------------------------------------------------------------
// Same seam, composition, coefficient tables and step-size controller as k_split14_loop (k_split.hpp; the call
// replaced is qutip.sesolve behind simulation.py:729-735): one workgroup per sequence holds the whole ket in registers
// over every stage of a closed run.  Positions of the N index bits: NR register bits, 6 lane bits, NW = N - 6 - NR
// wave bits (64 << NW lanes).  A stage = D (phase) + the N single-atom rotations; every rotation is done where its bit
// is cheapest to reach:
//   * register bits: plain FMAs (tan form: x' = x - T y_p, y' = y + T x_p, the same formula on both partners);
//   * the low ND = 6 - (NR - NW) lane bits: the partner comes over the DPP crossbar (quad permutes, row rotate by 8;
//     bit 2 takes two hops) - no layout change, no LDS;
//   * the upper NTB = NR - NW lane bits: an INTRA-WAVE transposition through LDS exchanges them with the register bits
//     NW .. NR-1 ("T bits").  A wave only reads what it wrote itself: no barrier, only the LDS queue's own order, so
//     the waves of a workgroup drift apart and the transposition of one overlaps the FMAs of another (and, inside a
//     wave, the next group's phase factors);
//   * the NW wave bits: one pass through LDS per stage exchanges them with register bits 0 .. NW-1, in four chunks
//     (the top two T bits fixed) through two buffers: 4 workgroup barriers per stage, the next chunk's rotations issue
//     while the LDS absorbs the current chunk's stores.  Lane bits do not take part: every ds_write_b128 /
//     ds_read_b128 of a wave covers 1 KiB of consecutive addresses.
// After a stage the T bits and the upper lane bits have changed places, and so have the wave bits and the low register
// bits: layouts alternate with period 2.  The code of a stage only knows POSITIONS; which atom's coefficient belongs
// to a position is decided when the stage's coefficients are loaded (scalar loads, parity of the stage), and the E0
// pieces of the two layouts are selected by parity - ONE stage body in the loop (half the instruction footprint of
// even / odd copies).  The closing stage of a run (D only: its rotation angles are zero, tan = 0) runs the same body, so a
// run may end in either layout; the store addresses follow.
// Phase factors: E0 is pairwise additive, so for lane t and register r
//   exp(-i phi) = B(t) G(r) prod_{j excited in r} F_j(t),
// B = the lane's atoms alone (+ their detunings, + the cosines of the previous stage's tan-form rotations), F_j =
// register atom j against the lane's excited atoms (+ its detuning), G = the register atoms among themselves (uniform:
// lane r of every wave computes G(r) into a wave-private LDS table, read back as broadcasts): NR + 2 table-and-series
// sin / cos per lane and stage, then one complex multiplication per amplitude down a product tree, the uniform factor,
// and the amplitude itself.
// Roofline: fp64 vector pipe (the ket never leaves the CU between the first load and the last store).

#ifndef SPLITR_TRIG
#define SPLITR_TRIG 256  /* phase table entries: exp(2 pi i k / SPLITR_TRIG) */
#endif
// dev probe (tools/lane_variants.sh): knock out one component of the stage (wrong results, timing only)
//   1 the intra-wave transposition, 2 lane-bit rotations (DPP), 4 the LDS pass and its barriers, 8 phase,
//   16 register rotations
#ifndef SPLITR_KO
#define SPLITR_KO 0
#endif
// A/B switches (dev builds, profiles/r04_ksplit14_variants.md):
//   SPLITR_TMODE 1: lane bits 4, 5 by v_permlane16_swap / v_permlane32_swap instead of the LDS transposition (NTB == 2);
//                2: the LDS transposition, D first and half of the DPP rotations as its filler;
//                3: no exchange: partners of lane bits 4, 5 over the LDS crossbar (ds_swizzle / ds_bpermute; NTB == 2)
//   SPLITR_GMODE 1: the uniform factor G(r) by v_readlane from the lane that computed it instead of the LDS table
//   SPLITR_WMODE 1: the stages' E0 weights from an LDS table filled at launch (no kernel-argument loads per stage)
#ifndef SPLITR_TMODE
#define SPLITR_TMODE 1
#endif
#ifndef SPLITR_GMODE
#define SPLITR_GMODE 0
#endif
#ifndef SPLITR_WMODE
#define SPLITR_WMODE 1
#endif
//   SPLITR_EMODE 1: the per-lane E0 pieces et / eg of both layouts in LDS (read once per stage) instead of 8 registers
//   SPLITR_KWARM 1: the next stage's coefficient record is touched (one scalar load per 64-B line) half a stage ahead
//   SPLITR_PREF  1: the detuning integrals and the E0 weight of the NEXT stage are loaded at the end of a stage
#ifndef SPLITR_EMODE
#define SPLITR_EMODE 0
#endif
#ifndef SPLITR_KWARM
#define SPLITR_KWARM 1
#endif
#ifndef SPLITR_PREF
#define SPLITR_PREF 0
#endif
//   SPLITR_LATE 1: complex drives: the rotation coefficients are loaded after the phase factors (scalar register pressure)
#ifndef SPLITR_LATE
#define SPLITR_LATE 1
#endif
//   SPLITR_P5 1 (round-6 variant, NTB == 2 shapes with TMODE 1): lane bit 5 changes places with T bit 1 INSIDE the LDS pass - the
//                reader fetches from lane (l & 31) | (t1 << 5) - instead of by 64 v_permlane32_swap per stage; the pass then moves
//                HALF the ket at a time (T bit 0 fixed: two super-chunks through the one 128-KiB buffer, 4 barriers as before)
#ifndef SPLITR_P5
#define SPLITR_P5 0
#endif
#ifndef SPLITR_LATE_REAL
#define SPLITR_LATE_REAL 0  /* the same for real drives (A/B) */
#endif

template <int N, int NR>
struct SplitRegLayout {
  static constexpr int NW = N - 6 - NR;   // wave bits
  static constexpr int NTB = NR - NW;     // register bits exchanged with lane bits inside a wave
  static constexpr int ND = 6 - NTB;      // lane bits served by the DPP crossbar
  static_assert(NW >= 0 && NW <= 3 && NTB >= 2 && ND >= 1 && ND <= 4, "unsupported shape");
  // index bit held by a position at the start of an even / odd stage
  // (SPLITR_TMODE 3: the T registers and the upper lane bits never change places - those lane bits get their rotation
  // through partner fetches over the LDS crossbar)
  static constexpr bool kXchg = SPLITR_TMODE != 3;
  __host__ __device__ static constexpr int regbit(bool odd, int j) {
    if (!odd) return 6 + NW + j;
    return j < NW ? 6 + j : (kXchg ? ND + (j - NW) : 6 + NW + j);
  }
  __host__ __device__ static constexpr int lanebit(bool odd, int j) {
    if (j < ND || !odd || !kXchg) return j;
    return 6 + 2 * NW + (j - ND);
  }
  __host__ __device__ static constexpr int wavebit(bool odd, int j) { return odd ? 6 + NW + j : 6 + j; }
  __host__ __device__ static constexpr unsigned index(bool odd, unsigned t, unsigned r) {
    unsigned i = 0;
    for (int j = 0; j < 6; ++j) i |= ((t >> j) & 1u) << lanebit(odd, j);
    for (int j = 0; j < NW; ++j) i |= ((t >> (6 + j)) & 1u) << wavebit(odd, j);
    for (int j = 0; j < NR; ++j) i |= ((r >> j) & 1u) << regbit(odd, j);
    return i;
  }
  // swizzle of the intra-wave transposition: slot of (low lane bits a, X, run Y) = a + 2^ND (X ^ (Y & FM)); chosen so
  // that the 16-lane groups of a ds_read_b128 and the 8-lane groups of a ds_write_b128 meet 16 different bank quads
  // (tools/lane_layout_check.py replays the bank rules of the guide for every instantiated shape)
  static constexpr int FM = ND == 4 ? 0 : (ND == 3 ? 2 : 3);
  __host__ __device__ static constexpr int swz(int y) { return ND == 3 ? ((y >> 1) & 1) : (y & FM); }
};

// compile-time loop: every register index below is a constant expression (runtime-trip loops with `continue` filters
// over 64 registers were left rolled by the unroller - 2 400 scratch instructions per stage - hence no loops at all)
template <int I, int E, class Fn>
__device__ __forceinline__ void splitr_for(Fn&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    splitr_for<I + 1, E>(f);
  }
}
template <int V>
using splitr_c = std::integral_constant<int, V>;
__host__ __device__ constexpr int splitr_low_clear(int v) {
  int j = 0;
  while (v & (1 << j)) ++j;
  return j;
}

template <int CTRL>
__device__ __forceinline__ double splitr_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
//   SPLITR_B2 1: the partner of lane bit 2 (xor 4: no single DPP pattern - two dependent v_mov_dpp per dword) over the LDS
//                crossbar instead (ds_swizzle, xor mask 4: no vector-pipe cycles beyond the issue) - round-5 variant
#ifndef SPLITR_B2
#define SPLITR_B2 0
#endif
//   SPLITR_HM 1 (round 6, default): the third DPP lane bit is the DIRECTION xor 7 of the lane number instead of xor 4 - one
//                row_half_mirror move per dword where xor 4 takes two dependent ones (640 -> 512 DPP moves per stage at
//                32 amplitudes per lane).  {1, 2, 7, 8} is a basis of the 4-bit lane numbers of a row like {1, 2, 4, 8}: a lane
//                holds the index bits (b0, b1, b2) = (l0 ^ l2, l1 ^ l2, l2) on its three low positions - splitr_lane() below;
//                everything that asks "which index bits does this lane hold" goes through it, the LDS passes and the swaps
//                only ever pair a lane with itself or with lanes of the same low bits and do not care.
#ifndef SPLITR_HM
#define SPLITR_HM 1
#endif
// the lane / thread number whose BITS are the index bits a lane holds (positions as in SplitRegLayout)
// (only where lane bit 2 is a DPP bit, ND >= 3: with ND = 2 it is a T bit, the LDS transposition moves it, and the two low
// bits must not depend on it)
template <int ND>
__device__ __forceinline__ unsigned splitr_lane(unsigned t) { return (SPLITR_HM && ND >= 3) ? t ^ ((t & 4u) ? 3u : 0u) : t; }
template <int F>
__device__ __forceinline__ double splitr_partner(double v) {
  if constexpr (F == 0) return splitr_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (F == 1) return splitr_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (F == 3) return splitr_dpp<0x128>(v);  // row_ror:8
  else if constexpr (SPLITR_HM == 1) return splitr_dpp<0x141>(v);  // row_half_mirror: xor 7 (F == 2 is only asked for with ND >= 3)
  else if constexpr (SPLITR_B2 == 1) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x101F);  // bit-mask mode: and 0x1F, or 0, xor 4
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x101F);
    return __hiloint2double(hi, lo);
  }
  else return splitr_dpp<0x1B>(splitr_dpp<0x141>(v));      // xor 4 = (xor 3) o (xor 7): row_half_mirror, quad reverse
}

// ROWS: the kets are the rows of a density matrix (the split-operator master equation of host_ket.hpp): blockIdx.z = the
// matrix, and the workgroups of a matrix are PERSISTENT - workgroup blockIdx.y takes the rows blockIdx.y, + gridDim.y, ...
// one after another (tables, E0 pieces and stage weights are set up once, and the store of a row overlaps the load of
// the next).  The coefficients and E0 of the matrix serve its 2^N rows, the propagator
// may be the complex conjugate one (SplitArgs.conj), the elementwise dissipator factors are applied at the load and /
// or the store (SplitArgs.ftab: exponentials of linear functions of the bit-pair counts, so the factor of an
// amplitude is (a factor of its lane) x (a factor of its register)), and the run may open with / close on the
// drive-only kick of the 4th-order splitting (SplitRun.kick_*).
// CPLX: complex drive coefficients (a pulse with a phase, local addressing with per-atom phases): the rotation of an atom
// is C (1 + u |1><0| - conj(u) |0><1|), u = (Re g + i Im g) / C (SplitRun.tan_form 2: both slots of the table) - 4 FMAs
// per amplitude and bit instead of 2; on the DPP bits the sign of the Re part depends on the lane's own bit.
// SNAP: evaluation-time snapshots inside the run (round 5): after the last stage of a sub-step s with R.snap[s] >= 0 the
// registers - the OPEN state, see SplitArgs.snaps - are stored to the snapshot slot; k_split_snap_close finishes them.  A
// front end that asks for the state at every knot (evaluation_times="Full", the reference's default, simulation.py:137)
// or at every 10th used to close a run - a launch, a load and a store of the ket, a closing stage - per evaluation time.
template <int N, int NR, bool DECAY, bool ROWS = false, bool CPLX = false, bool SNAP = false>
__global__ __launch_bounds__(64 << (N - 6 - NR)) __attribute__((amdgpu_waves_per_eu(1, NR >= 6 ? 1 : 2))) void k_split_reg(const SplitArgs A, const SplitRun R, long long stage_stride) {
  static_assert(!(SNAP && (ROWS || DECAY)), "snapshots inside a run: plain ket runs only");
  typedef SplitRegLayout<N, NR> L;
  constexpr int NW = L::NW, NTB = L::NTB, ND = L::ND;
  constexpr int NT = 64 << NW;
  constexpr int NA = 1 << NR;          // amplitudes per lane
  constexpr int PW = (1 << NW) - 1;    // register bits the pass exchanges
  constexpr int CH = NA / 4;           // registers per chunk of the pass (top two T bits fixed) = LDS runs per wave
  constexpr int NG = 1 << NW;          // groups of the transposition (pass bits fixed)
  constexpr int GR = 1 << NTB;         // registers per group
  constexpr int NBUF = CH / GR;        // groups that fit a wave's private slice of buffer 0 (1 or 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* pbuf = reinterpret_cast<cplx*>(smem);  // 2 buffers x (NT * CH) slots of 16 B
  cplx* trig = pbuf + 2 * NT * CH;             // SPLITR_TRIG
  cplx* gtab = trig + SPLITR_TRIG;             // [waves][NA]: G(r) of the current stage, wave-private
  double* wtab = reinterpret_cast<double*>(gtab + (NT / 64) * NA);  // [SPLIT_MAX_SUB * SPLIT_MAX_STAGES + 1] (WMODE 1)
  double* etab = wtab + (SPLIT_MAX_SUB * SPLIT_MAX_STAGES + 2);     // [4][NT]: et, eg of the even / odd layout (EMODE 1)

  const unsigned t = threadIdx.x;
  const unsigned l = t & 63u, w = t >> 6;
  const unsigned tq = splitr_lane<ND>(t);  // the thread number by the index bits it holds (SPLITR_HM)
  const int b = ROWS ? (int)blockIdx.z : (int)blockIdx.y;
  const int n_pre = R.kick_pre != 0.0 ? 1 : 0;  // the opening kick is an extra stage 0 (no D)
  const int n_stages = splitrun_first(R, R.nsub) + 1 + n_pre;
  // (SplitArgs.conj: conj(U) psi = conj(U conj(psi)) - the ket is conjugated at the load and at the store, the stages
  // in between are the ordinary ones: nothing in the stage loop knows about it)
  const double csgn = (ROWS && A.conj) ? -1.0 : 1.0;
  const double* __restrict__ coefs = A.ccur + (size_t)b * N * 4;
  const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;

  if (SPLITR_WMODE) {
    // weight of E0 in the D of stage j: a_i tau (+ the last a tau carried over from the previous sub-step)
    for (int jj = (int)t; jj < n_stages; jj += NT) {
      const int j = jj - n_pre;
      wtab[jj] = j < 0 ? 0.0 : splitrun_weight(R, j);
    }
  }
  for (unsigned k = t; k < SPLITR_TRIG; k += NT) {  // (before the state is loaded: the library routine wants registers)
    double sn, cs;
    sincospi((double)k * (2.0 / SPLITR_TRIG), &sn, &cs);
    trig[k] = make_double2(cs, sn);
  }
  // E0 pieces of the two layouts (bit = 1 is the ground state: all register bits set = the lane's atoms alone, all
  // lane / wave bits set = the register atoms alone)
  double et[2], ev[2][NR], eg[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    et[o] = e0[L::index(o, tq, NA - 1)];
#pragma unroll
    for (int j = 0; j < NR; ++j) ev[o][j] = e0[L::index(o, tq, (NA - 1) ^ (1u << j))] - et[o];
    eg[o] = e0[L::index(o, NT - 1, l & (NA - 1))];
    if (SPLITR_EMODE) {
      etab[(2 * o) * NT + t] = et[o];
      etab[(2 * o + 1) * NT + t] = eg[o];
    }
  }
  // ROWS: elementwise factor of amplitude (row a, column i) = prod_k tab_k[n_k], n = the counts of (row bit, column bit)
  // = (0,0), (0,1), (1,0), (1,1); every table is an exponential of a linear function, so the product splits into a
  // factor of the lane's column bits and one of the register's column bits (uniform: table of NA per layout use)
  double* ftl = etab + (SPLITR_EMODE ? 4 * NT : 0);  // [128] the host's tables, [NA] register factors (rewritten per use)
  unsigned a_row = 0;  // ROWS: the current row
  auto row_factor = [&](int which, unsigned col_bits, unsigned mask) -> double {
    const unsigned a = a_row;
    const int n11 = __popc(a & col_bits & mask), n10 = __popc(a & ~col_bits & mask), n01 = __popc(~a & col_bits & mask);
    const int n00 = __popc(mask) - n11 - n10 - n01;
    const double* tb = ftl + 64 * which;
    return tb[n00] * tb[16 + n01] * tb[32 + n10] * tb[48 + n11];
  };
  if (ROWS && (A.use_pre || A.use_post) && t < 128) ftl[t] = A.ftab[t];
  __syncthreads();
  typedef const __attribute__((address_space(4))) double* cptr_t;
  // exp(-i phi) = (cos, -sin): table of exp(2 pi i k / SPLITR_TRIG), series on |rr| <= pi / SPLITR_TRIG
  auto expmi = [&](double phi) -> cplx {
    const double kk = rint(phi * (SPLITR_TRIG / 6.283185307179586));
    double rr = fma(-kk, 6.283185307179586 / SPLITR_TRIG, phi);
    rr = fma(-kk, 2.4492935982947064e-16 / SPLITR_TRIG, rr);  // 2 pi - double(2 pi)
    const cplx tb = trig[((int)kk) & (SPLITR_TRIG - 1)];
    const double r2 = rr * rr;
    double cr, sr;
    if (SPLITR_TRIG >= 512) {  // |rr| <= pi / 512: the next terms are below 1e-19
      cr = fma(r2, fma(r2, 4.1666666666666664e-02, -0.5), 1.0);
      sr = rr * fma(r2, fma(r2, 8.333333333333333e-03, -1.6666666666666666e-01), 1.0);
    } else {
      cr = fma(r2, fma(r2, fma(r2, -1.3888888888888889e-03, 4.1666666666666664e-02), -0.5), 1.0);
      sr = rr * fma(r2, fma(r2, fma(r2, -1.984126984126984e-04, 8.333333333333333e-03), -1.6666666666666666e-01), 1.0);
    }
    return make_double2(fma(tb.x, cr, -tb.y * sr), -fma(tb.y, cr, tb.x * sr));
  };
  auto cm = [](cplx a, cplx c) -> cplx { return make_double2(fma(a.x, c.x, -a.y * c.y), fma(a.x, c.y, a.y * c.x)); };
  // private slice of buffer 0: the slots a wave's own pass stores go to (slot = l + 64 (w + NG q), q < CH)
  cplx* __restrict__ const tslice = pbuf + 64 * w;
  const unsigned la = l & ((1u << ND) - 1u), lT = l >> ND;

  for (unsigned row = ROWS ? blockIdx.y : 0u; row < (ROWS ? (1u << N) : 1u); row += ROWS ? gridDim.y : 1u) {
  a_row = row;
  cplx* __restrict__ st = A.state + ((((size_t)b << (ROWS ? N : 0)) + (ROWS ? row : 0u)) << N);
  if (ROWS) __syncthreads();  // the previous row's readers of the factor table are done
  if (ROWS && A.use_pre && t < NA) ftl[128 + t] = row_factor(0, L::index(false, 0u, t), L::index(false, 0u, NA - 1));
  if (ROWS && A.use_pre) __syncthreads();
  double xr[NA], xi[NA];
  {
    const double fl = (ROWS && A.use_pre) ? row_factor(0, L::index(false, tq, 0u), L::index(false, NT - 1, 0u)) : 1.0;
    splitr_for<0, NA>([&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      const cplx v = st[L::index(false, tq, r)];
      const double f = (ROWS && A.use_pre) ? fl * ftl[128 + r] : 1.0;
      xr[r] = v.x * f;
      xi[r] = v.y * (ROWS ? f * csgn : f);
    });
  }


  // rotation of register bit J (tan T) on the registers r with (r & MASK) == VAL
  auto rot_reg = [&](auto J, double T, double U, auto MASK, auto VAL) {  // T = Im g / C, U = Re g / C (CPLX only)
    if (SPLITR_KO & 16) return;
    splitr_for<0, NA>([&](auto Rc) {
      constexpr int r = decltype(Rc)::value, j = decltype(J)::value;
      if constexpr ((r & decltype(MASK)::value) == decltype(VAL)::value && !(r & (1 << j))) {
        constexpr int q = r | (1 << j);
        const double a0x = xr[r], a0y = xi[r], a1x = xr[q], a1y = xi[q];
        if constexpr (CPLX) {
          xr[r] = fma(-U, a1x, fma(-T, a1y, a0x));
          xi[r] = fma(-U, a1y, fma(T, a1x, a0y));
          xr[q] = fma(U, a0x, fma(-T, a0y, a1x));
          xi[q] = fma(U, a0y, fma(T, a0x, a1y));
        } else {
          xr[r] = fma(-T, a1y, a0x);
          xi[r] = fma(T, a1x, a0y);
          xr[q] = fma(-T, a0y, a1x);
          xi[q] = fma(T, a0x, a1y);
        }
      }
    });
  };

  double cprod = 1.0;  // product of the cosines of the previous stage's rotations (tan form): rides on B
  bool odd = false;
  // (SPLITR_PREF) the detuning integrals by position and the E0 weight of the next stage, loaded a stage ahead
  double Dn[6 + NW + NR], wEn = 0.0;
  auto load_next = [&](int sn, bool oddn) {
    cptr_t cn = (cptr_t)(unsigned long long)(coefs + (size_t)sn * stage_stride);
#pragma unroll
    for (int j = 0; j < 6; ++j) Dn[j] = cn[4 * (N - 1 - (oddn ? L::lanebit(true, j) : L::lanebit(false, j))) + 3];
#pragma unroll
    for (int j = 0; j < NW; ++j) Dn[6 + j] = cn[4 * (N - 1 - (oddn ? L::wavebit(true, j) : L::wavebit(false, j))) + 3];
#pragma unroll
    for (int j = 0; j < NR; ++j) Dn[6 + NW + j] = cn[4 * (N - 1 - (oddn ? L::regbit(true, j) : L::regbit(false, j))) + 3];
    if (SPLITR_WMODE) wEn = uniform_d(wtab[sn]);
  };
  if (SPLITR_PREF) load_next(0, false);
  unsigned warm = 0;
  // One stage.  PLAIN_CLOSE: the closing stage of a run without a closing kick - D only (its rotation angles are zero):
  // the phase factors and nothing else, the layout stays (an instantiation of the same body after the loop: the loop's
  // back edge sees one register assignment)
  auto stage = [&](const int sg, auto plain_close_t) {
    constexpr bool PLAIN_CLOSE = decltype(plain_close_t)::value;
    // ---- coefficients by POSITION (uniform: scalar loads; the index bit of a position depends on the parity) ----
    cptr_t c4 = (cptr_t)(unsigned long long)(coefs + (size_t)sg * stage_stride);
    auto coef = [&](int bit_even, int bit_odd, int field) -> double {
      const int p = odd ? bit_odd : bit_even;
      return c4[4 * (N - 1 - p) + field];
    };
    double Tl[6], Dl[6], Tw[NW > 0 ? NW : 1], Dw[NW > 0 ? NW : 1], Tr[NR], Dr[NR];
    double Ul[6], Uw[NW > 0 ? NW : 1], Ur[NR];  // CPLX: Re g / C by position
    constexpr int TF = CPLX ? 2 : 1;  // the field that holds Im g / C (SplitRun.tan_form)
    double cnext = 1.0;
    // the detuning integrals first (the phase factors need them at once); the rotation coefficients follow - for
    // complex drives only AFTER the phase factors (SPLITR_LATE: the pointer is laundered through an empty asm so that
    // the loads cannot be hoisted): 14 x (C, Re g / C, Im g / C) on top of the 14 integrals do not fit the scalar
    // registers (79 spilled, and a spilled scalar load is a serialised one: +1 us per stage, round-4 variants table)
#pragma unroll
    for (int j = 0; j < 6; ++j) Dl[j] = SPLITR_PREF ? Dn[j] : coef(L::lanebit(false, j), L::lanebit(true, j), 3);
#pragma unroll
    for (int j = 0; j < NW; ++j) Dw[j] = SPLITR_PREF ? Dn[6 + j] : coef(L::wavebit(false, j), L::wavebit(true, j), 3);
#pragma unroll
    for (int j = 0; j < NR; ++j) Dr[j] = SPLITR_PREF ? Dn[6 + NW + j] : coef(L::regbit(false, j), L::regbit(true, j), 3);
    auto load_rot = [&](cptr_t cl) {
      auto coefl = [&](int bit_even, int bit_odd, int field) -> double {
        const int p = odd ? bit_odd : bit_even;
        return cl[4 * (N - 1 - p) + field];
      };
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        Tl[j] = coefl(L::lanebit(false, j), L::lanebit(true, j), TF);
        Ul[j] = CPLX ? coefl(L::lanebit(false, j), L::lanebit(true, j), 1) : 0.0;
        cnext *= coefl(L::lanebit(false, j), L::lanebit(true, j), 0);
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        Tw[j] = coefl(L::wavebit(false, j), L::wavebit(true, j), TF);
        Uw[j] = CPLX ? coefl(L::wavebit(false, j), L::wavebit(true, j), 1) : 0.0;
        cnext *= coefl(L::wavebit(false, j), L::wavebit(true, j), 0);
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        Tr[j] = coefl(L::regbit(false, j), L::regbit(true, j), TF);
        Ur[j] = CPLX ? coefl(L::regbit(false, j), L::regbit(true, j), 1) : 0.0;
        cnext *= coefl(L::regbit(false, j), L::regbit(true, j), 0);
      }
    };
    constexpr bool kLate = (SPLITR_LATE && CPLX) || (SPLITR_LATE_REAL && !CPLX);
    if (!kLate && !PLAIN_CLOSE) load_rot(c4);
    // weight of E0 in this stage's D: a_i tau (+ the last a tau carried over from the previous sub-step)
    double wE;
    if (SPLITR_WMODE && SPLITR_PREF) {
      wE = wEn;
    } else if (SPLITR_WMODE) {
      wE = uniform_d(wtab[sg]);
    } else {
      const int sj = sg - n_pre;
      wE = sj < 0 ? 0.0 : splitrun_weight(R, sj);
    }
    double et_s, eg_s;
    if (SPLITR_EMODE) {
      et_s = etab[(odd ? 2 : 0) * NT + t];
      eg_s = etab[(odd ? 3 : 1) * NT + t];
    } else {
      et_s = odd ? et[1] : et[0];
      eg_s = odd ? eg[1] : eg[0];
    }

    // ---- the lane's phase factors ----
    cplx Bf, F[NR], Gl;
    if (!(SPLITR_KO & 8)) {
      double dthr = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) dthr += ((tq >> j) & 1u) ? 0.0 : Dl[j];
#pragma unroll
      for (int j = 0; j < NW; ++j) dthr += ((tq >> (6 + j)) & 1u) ? 0.0 : Dw[j];
      Bf = expmi(fma(wE, et_s, -dthr));
      double sc = cprod;
      if (DECAY)  // H_eff: the real factor exp(wE (dec_a + dec_b popc(index))): lane part here, register part in F
        sc *= exp(wE * (A.dec_a + A.dec_b * (double)(__popc(tq) + NR)));
      Bf = make_double2(Bf.x * sc, Bf.y * sc);
      const double dF = DECAY ? exp(-wE * A.dec_b) : 1.0;  // an excited register atom: one set bit fewer
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        F[j] = expmi(fma(wE, odd ? ev[1][j] : ev[0][j], -Dr[j]));
        if (DECAY) F[j] = make_double2(F[j].x * dF, F[j].y * dF);
      }
      Gl = expmi(wE * eg_s);
      if (!SPLITR_GMODE) {
        gtab[w * NA + (l & (NA - 1))] = Gl;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (kLate && !PLAIN_CLOSE) {
      unsigned long long cl = (unsigned long long)(coefs + (size_t)sg * stage_stride);
      asm volatile("" : "+s"(cl));
      load_rot((cptr_t)cl);
    }
    cprod = cnext;
    const cplx* __restrict__ gw = gtab + w * NA;
    auto gval = [&](auto Rc) -> cplx {  // G(r): uniform
      constexpr int r = decltype(Rc)::value;
      if (SPLITR_GMODE) {
        const int a = __builtin_amdgcn_readlane(__double2loint(Gl.x), r), b2 = __builtin_amdgcn_readlane(__double2hiint(Gl.x), r);
        const int c = __builtin_amdgcn_readlane(__double2loint(Gl.y), r), d = __builtin_amdgcn_readlane(__double2hiint(Gl.y), r);
        return make_double2(__hiloint2double(b2, a), __hiloint2double(d, c));
      }
      return gw[r];
    };

    // ---- per group (pass bits fixed): D, the T-bit rotations, the intra-wave transposition, the rotations of
    //      the lane bits that arrived.  Product tree: groups from NG-1 down (parent = lowest clear bit set), the T
    //      bits innermost. ----
    cplx Pg[NG];
    auto phase_group = [&](auto Gc) {
      constexpr int g = decltype(Gc)::value;
      if (SPLITR_KO & 8) return;
      if constexpr (g == NG - 1) Pg[g] = Bf;
      else Pg[g] = cm(Pg[g | (1 << splitr_low_clear(g))], F[splitr_low_clear(g)]);
      cplx Pt[GR];
      splitr_for<0, GR>([&](auto Ic) {
        constexpr int rho = GR - 1 - decltype(Ic)::value;
        if constexpr (rho == GR - 1) Pt[rho] = Pg[g];
        else Pt[rho] = cm(Pt[rho | (1 << splitr_low_clear(rho))], F[NW + splitr_low_clear(rho)]);
        constexpr int r = g | (rho << NW);
        const cplx q = cm(Pt[rho], gval(splitr_c<r>{}));
        const double ax = xr[r], ay = xi[r];
        xr[r] = fma(ax, q.x, -ay * q.y);
        xi[r] = fma(ax, q.y, ay * q.x);
      });
    };
    constexpr bool kP5 = SPLITR_P5 && SPLITR_TMODE == 1 && NTB == 2 && NW > 0 && !(SPLITR_KO & 4);
    auto swap_d = [](double& a, double& c, auto is32) {
      unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
      unsigned clo = (unsigned)__double2loint(c), chi = (unsigned)__double2hiint(c);
      if constexpr (decltype(is32)::value) {
        auto p = __builtin_amdgcn_permlane32_swap(alo, clo, false, false);
        auto q = __builtin_amdgcn_permlane32_swap(ahi, chi, false, false);
        alo = p[0]; clo = p[1]; ahi = q[0]; chi = q[1];
      } else {
        auto p = __builtin_amdgcn_permlane16_swap(alo, clo, false, false);
        auto q = __builtin_amdgcn_permlane16_swap(ahi, chi, false, false);
        alo = p[0]; clo = p[1]; ahi = q[0]; chi = q[1];
      }
      a = __hiloint2double((int)ahi, (int)alo);
      c = __hiloint2double((int)chi, (int)clo);
    };
    auto t_swap = [&](auto Gc) {  // SPLITR_TMODE 1 (NTB == 2): T bit 0 <-> lane bit 4, T bit 1 <-> lane bit 5
      constexpr int g = decltype(Gc)::value;
      if (SPLITR_KO & 1) return;
      constexpr int r0 = g, r1 = g | (1 << NW), r2 = g | (2 << NW), r3 = g | (3 << NW);
      swap_d(xr[r0], xr[r1], std::false_type{}); swap_d(xi[r0], xi[r1], std::false_type{});
      swap_d(xr[r2], xr[r3], std::false_type{}); swap_d(xi[r2], xi[r3], std::false_type{});
      if constexpr (!kP5) {  // (SPLITR_P5: lane bit 5 <-> T bit 1 rides on the LDS pass)
        swap_d(xr[r0], xr[r2], std::true_type{}); swap_d(xi[r0], xi[r2], std::true_type{});
        swap_d(xr[r1], xr[r3], std::true_type{}); swap_d(xi[r1], xi[r3], std::true_type{});
      }
    };
    auto t_write = [&](auto Gc) {
      constexpr int g = decltype(Gc)::value;
      if (SPLITR_KO & 1) return;
      splitr_for<0, GR>([&](auto Ic) {
        constexpr int rho = decltype(Ic)::value;
        constexpr int run = rho + (NBUF == 2 ? (g & 1) * GR : 0);
        tslice[64 * NG * run + la + ((lT ^ (unsigned)L::swz(rho)) << ND)] = make_double2(xr[g | (rho << NW)], xi[g | (rho << NW)]);
      });
    };
    auto t_read = [&](auto Gc) {
      constexpr int g = decltype(Gc)::value;
      if (SPLITR_KO & 1) return;
      __builtin_amdgcn_wave_barrier();
      const unsigned run = lT + (NBUF == 2 ? (g & 1) * GR : 0);
      splitr_for<0, GR>([&](auto Ic) {
        constexpr int rho = decltype(Ic)::value;
        const cplx v = tslice[64 * NG * run + la + (((unsigned)rho ^ (unsigned)L::swz((int)lT)) << ND)];
        xr[g | (rho << NW)] = v.x;
        xi[g | (rho << NW)] = v.y;
      });
    };
    auto rot_t_old = [&](auto Gc) {  // the T bits this layout holds
      splitr_for<0, NTB>([&](auto Mc) {
        constexpr int m = decltype(Mc)::value;
        rot_reg(splitr_c<NW + m>{}, Tr[NW + m], Ur[NW + m], splitr_c<PW>{}, Gc);
      });
    };
    auto rot_t_new = [&](auto Gc) {  // the upper lane bits, now in the T registers
      splitr_for<0, (kP5 ? 1 : NTB)>([&](auto Mc) {  // (SPLITR_P5: lane bit 5 arrives with the pass; its rotation follows the read)
        constexpr int m = decltype(Mc)::value;
        rot_reg(splitr_c<NW + m>{}, Tl[ND + m], Ul[ND + m], splitr_c<PW>{}, Gc);
      });
    };
    // CPLX: Re g / C of the DPP bits with the sign of the lane's own bit (+ on bit 1, - on bit 0)
    double Us[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if constexpr (CPLX) {
#pragma unroll
      for (int j = 0; j < 6; ++j) Us[j] = ((tq >> j) & 1u) ? Ul[j] : -Ul[j];
    }
    // SPLITR_TMODE 3: partner of lane bit 4 (ds_swizzle, xor 16 inside 32 lanes) / 5 (ds_bpermute) of a double - the LDS
    // crossbar, no LDS memory, no vector-pipe cycles beyond the issue
    auto xpartner = [&](double v, auto Jc) -> double {
      int lo = __double2loint(v), hi = __double2hiint(v);
      if constexpr (decltype(Jc)::value == 4) {
        lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
        hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
      } else {
        const int addr = (int)((l ^ 32u) << 2);
        lo = __builtin_amdgcn_ds_bpermute(addr, lo);
        hi = __builtin_amdgcn_ds_bpermute(addr, hi);
      }
      return __hiloint2double(hi, lo);
    };
    auto rot_lane_x = [&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      if (SPLITR_KO & 1) return;
      splitr_for<4, 6>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        const double T = Tl[j], px = xpartner(xr[r], Jc), py = xpartner(xi[r], Jc);
        if constexpr (CPLX) {
          xr[r] = fma(Us[j], px, fma(-T, py, xr[r]));
          xi[r] = fma(Us[j], py, fma(T, px, xi[r]));
        } else {
          xr[r] = fma(-T, py, xr[r]);
          xi[r] = fma(T, px, xi[r]);
        }
      });
    };
    // rotations of the DPP lane bits in MASK on register r (any time between this stage's D and the next one's: they
    // commute with every other rotation and exchange of the stage)
    auto rot_lane = [&](auto Rc, auto MASK) {
      constexpr int r = decltype(Rc)::value, mask = decltype(MASK)::value;
      if (SPLITR_KO & 2) return;
      if constexpr (ND > 0 && (mask & 1)) {
        const double T = Tl[0], px = splitr_partner<0>(xr[r]), py = splitr_partner<0>(xi[r]);
        if constexpr (CPLX) {
          xr[r] = fma(Us[0], px, fma(-T, py, xr[r]));
          xi[r] = fma(Us[0], py, fma(T, px, xi[r]));
        } else {
          xr[r] = fma(-T, py, xr[r]);
          xi[r] = fma(T, px, xi[r]);
        }
      }
      if constexpr (ND > 1 && (mask & 2)) {
        const double T = Tl[1], px = splitr_partner<1>(xr[r]), py = splitr_partner<1>(xi[r]);
        if constexpr (CPLX) {
          xr[r] = fma(Us[1], px, fma(-T, py, xr[r]));
          xi[r] = fma(Us[1], py, fma(T, px, xi[r]));
        } else {
          xr[r] = fma(-T, py, xr[r]);
          xi[r] = fma(T, px, xi[r]);
        }
      }
      if constexpr (ND > 3 && (mask & 8)) {
        const double T = Tl[3], px = splitr_partner<3>(xr[r]), py = splitr_partner<3>(xi[r]);
        if constexpr (CPLX) {
          xr[r] = fma(Us[3], px, fma(-T, py, xr[r]));
          xi[r] = fma(Us[3], py, fma(T, px, xi[r]));
        } else {
          xr[r] = fma(-T, py, xr[r]);
          xi[r] = fma(T, px, xi[r]);
        }
      }
      if constexpr (ND > 2 && (mask & 4)) {
        const double T = Tl[2], px = splitr_partner<2>(xr[r]), py = splitr_partner<2>(xi[r]);
        if constexpr (CPLX) {
          xr[r] = fma(Us[2], px, fma(-T, py, xr[r]));
          xi[r] = fma(Us[2], py, fma(T, px, xi[r]));
        } else {
          xr[r] = fma(-T, py, xr[r]);
          xi[r] = fma(T, px, xi[r]);
        }
      }
    };
    // SPLITR_TMODE 2: the transposition through LDS with the vector work arranged around it - D on every group first
    // (its G-table reads would otherwise sit in the LDS queue between a group's stores and loads and every wait for a
    // G value would wait for the transposition too), then per group the old T bits, the DPP bits TMASK (filler: the
    // LDS takes ~540 cycles per group for the 8 waves), the stores, the loads; the other DPP bits stay with the pass
    if constexpr (PLAIN_CLOSE) {
      splitr_for<0, NG>([&](auto Ic) { phase_group(splitr_c<(NG - 1 - decltype(Ic)::value)>{}); });
    } else {
    constexpr int TMASK = SPLITR_TMODE == 2 ? 3 : 0;           // DPP lane bits rotated inside the transposition loop
    constexpr int PMASK = ((1 << ND) - 1) & ~TMASK;            // ... and with the pass chunks
    if constexpr (SPLITR_TMODE == 2) {
      splitr_for<0, NG>([&](auto Ic) { phase_group(splitr_c<(NG - 1 - decltype(Ic)::value)>{}); });
      splitr_for<0, NG>([&](auto Ic) {
        constexpr int g = NG - 1 - decltype(Ic)::value;
        rot_t_old(splitr_c<g>{});
        splitr_for<0, GR>([&](auto Qc) { rot_lane(splitr_c<(g | (decltype(Qc)::value << NW))>{}, splitr_c<TMASK>{}); });
        t_write(splitr_c<g>{});
        t_read(splitr_c<g>{});
        if constexpr (g + 1 < NG) rot_t_new(splitr_c<g + 1>{});
      });
      rot_t_new(splitr_c<0>{});
    } else {
      splitr_for<0, NG>([&](auto Ic) {
        constexpr int g = NG - 1 - decltype(Ic)::value;
        phase_group(splitr_c<g>{});
        rot_t_old(splitr_c<g>{});
        if constexpr (SPLITR_TMODE == 3) {
          // nothing to exchange
        } else if constexpr (SPLITR_TMODE == 1 && NTB == 2) {
          t_swap(splitr_c<g>{});
          rot_t_new(splitr_c<g>{});
        } else {
          t_write(splitr_c<g>{});
          t_read(splitr_c<g>{});
          if constexpr (g + 1 < NG) rot_t_new(splitr_c<g + 1>{});
        }
      });
      if constexpr (!(SPLITR_TMODE == 1 && NTB == 2) && SPLITR_TMODE != 3) rot_t_new(splitr_c<0>{});
    }

    // ---- per chunk (top two T bits fixed): the pass bits and the DPP lane bits, then the pass ----
    constexpr int CM = 3 << (NR - 2);  // the chunk's bits of the register index
    auto pre = [&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      splitr_for<0, NW>([&](auto Jc) { rot_reg(Jc, Tr[decltype(Jc)::value], Ur[decltype(Jc)::value], splitr_c<CM>{}, splitr_c<(c << (NR - 2))>{}); });
      splitr_for<0, CH>([&](auto Kc) {
        rot_lane(splitr_c<(decltype(Kc)::value | (c << (NR - 2)))>{}, splitr_c<PMASK>{});
        if constexpr (SPLITR_TMODE == 3) rot_lane_x(splitr_c<(decltype(Kc)::value | (c << (NR - 2)))>{});
      });
    };
    auto post = [&](auto Cc) {  // the wave bits, now register bits 0 .. NW-1
      constexpr int c = decltype(Cc)::value;
      splitr_for<0, NW>([&](auto Jc) { rot_reg(Jc, Tw[decltype(Jc)::value], Uw[decltype(Jc)::value], splitr_c<CM>{}, splitr_c<(c << (NR - 2))>{}); });
    };
    // slot = l + 64 (A + NG (B + NG s)):  writer A = wave, B = its pass bits;  reader A = its pass bits, B = its wave;
    // s = the chunk's T bits below the top two
    auto pass_write = [&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      cplx* __restrict__ pb = pbuf + (c & 1) * (NT * CH) + t;
      splitr_for<0, CH>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, kp = k & PW, ks = k >> NW;
        pb[64 * NG * (kp + NG * ks)] = make_double2(xr[k | (c << (NR - 2))], xi[k | (c << (NR - 2))]);
      });
    };
    auto pass_read = [&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      const cplx* __restrict__ pb = pbuf + (c & 1) * (NT * CH) + l + 64 * NG * w;
      splitr_for<0, CH>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, kp = k & PW, ks = k >> NW;
        const cplx v = pb[64 * (kp + NG * NG * ks)];
        xr[k | (c << (NR - 2))] = v.x;
        xi[k | (c << (NR - 2))] = v.y;
      });
    };
    typedef splitr_c<0> C0; typedef splitr_c<1> C1; typedef splitr_c<2> C2; typedef splitr_c<3> C3;
    // SPLITR_P5: super-chunk S = the registers with T bit 0 (register bit NR - 2) = S, i.e. chunks S and S + 2;
    // slot = lane + 64 (A + NG (B + NG t1)): writer A = wave, B = pass bits, t1 = its T bit 1, lane = its own;
    //                                         reader A = its pass bits, B = its wave, t1 = ITS LANE BIT 5, lane = (l & 31) | (t1' << 5)
    auto pass_write5 = [&](auto Sc) {
      constexpr int S = decltype(Sc)::value;
      cplx* __restrict__ pb = pbuf + t;
      splitr_for<0, 2 * CH>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, kp = k & PW, t1 = k >> NW;
        constexpr int r = kp | (S << (NR - 2)) | (t1 << (NR - 1));
        pb[64 * NG * (kp + NG * t1)] = make_double2(xr[r], xi[r]);
      });
    };
    auto pass_read5 = [&](auto Sc) {
      constexpr int S = decltype(Sc)::value;
      const cplx* __restrict__ pb = pbuf + (l & 31u) + 64 * NG * (w + NG * (l >> 5));
      splitr_for<0, 2 * CH>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, kp = k & PW, t1 = k >> NW;
        constexpr int r = kp | (S << (NR - 2)) | (t1 << (NR - 1));
        const cplx v = pb[32 * t1 + 64 * kp];
        xr[r] = v.x;
        xi[r] = v.y;
      });
    };
    auto pre5 = [&](auto Sc) {
      constexpr int S = decltype(Sc)::value;
      pre(splitr_c<S>{}); pre(splitr_c<S + 2>{});
    };
    auto post5 = [&](auto Sc) {  // the wave bits (register bits 0 .. NW-1) and lane bit 5's atom (T bit 1) of the half that arrived
      constexpr int S = decltype(Sc)::value;
      post(splitr_c<S>{}); post(splitr_c<S + 2>{});
      rot_reg(splitr_c<NW + 1>{}, Tl[ND + 1], Ul[ND + 1], splitr_c<(1 << (NR - 2))>{}, splitr_c<(S << (NR - 2))>{});
    };
    if constexpr (kP5) {
      static_assert(CH == (1 << NW), "SPLITR_P5: chunks of 2^NW registers");
      pre5(C0{});
      __syncthreads();  // the previous stage's pass_read5(1) of every wave is done: the buffer may be overwritten
      pass_write5(C0{});
      pre5(C1{});
      __syncthreads();
      pass_read5(C0{});
      if (SPLITR_KWARM) {
        typedef const __attribute__((address_space(4))) unsigned* uptr_t;
        uptr_t nx = (uptr_t)(unsigned long long)(coefs + (size_t)(sg + 1 < n_stages ? sg + 1 : sg) * stage_stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) warm ^= nx[k < 7 ? 16 * k : 8 * N - 1];
      }
      __syncthreads();
      pass_write5(C1{});
      post5(C0{});
      __syncthreads();
      pass_read5(C1{});
      if (SPLITR_PREF) load_next(sg + 1 < n_stages ? sg + 1 : sg, !odd);
      post5(C1{});
    } else if constexpr (NW == 0 || (SPLITR_KO & 4)) {
      pre(C0{}); pre(C1{}); pre(C2{}); pre(C3{});
      post(C0{}); post(C1{}); post(C2{}); post(C3{});
    } else {
      // (no barrier here: until the first one below a wave only touches its own slice of buffer 0 - the
      // transposition above and pass_write(0) - and buffer 0's last cross-wave reads, pass_read(2) of the previous
      // stage, sit before that stage's last barrier)
      pre(C0{});
      pass_write(C0{});
      pre(C1{});
      __syncthreads();
      pass_read(C0{});
      pass_write(C1{});
      if (SPLITR_KWARM) {
        typedef const __attribute__((address_space(4))) unsigned* uptr_t;
        uptr_t nx = (uptr_t)(unsigned long long)(coefs + (size_t)(sg + 1 < n_stages ? sg + 1 : sg) * stage_stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) warm ^= nx[k < 7 ? 16 * k : 8 * N - 1];
      }
      pre(C2{});
      __syncthreads();
      pass_read(C1{});
      pass_write(C2{});
      pre(C3{});
      __syncthreads();
      pass_read(C2{});
      pass_write(C3{});
      post(C0{});
      post(C1{});
      __syncthreads();
      pass_read(C3{});
      if (SPLITR_PREF) load_next(sg + 1 < n_stages ? sg + 1 : sg, !odd);
      post(C2{});
      post(C3{});
    }
    odd = !odd;
    }
    if (SPLITR_KWARM) asm volatile("" ::"s"(warm));
  };
  const int n_full = R.kick_post == 0.0 ? n_stages - 1 : n_stages;
  if constexpr (SNAP) {
    int snap_at = splitrun_first(R, 1) + n_pre - 1, snap_sub = 0;  // the stage that ends sub-step snap_sub
    for (int sg = 0; sg < n_full; ++sg) {
      stage(sg, std::false_type{});
      if (sg == snap_at) {
        const int slot = R.snap[snap_sub];
        if (slot >= 0) {
          // address = uniform base + the lane's word + a constant per register.  The lane's word is laundered through an
          // empty asm: loop-invariant otherwise, and the 32 hoisted 64-bit addresses cost 20 spilled vector registers
          // in every stage of the loop (measured: 10.6 instead of 9.4 us per stage)
          unsigned lw = odd ? L::index(true, tq, 0u) : L::index(false, tq, 0u);
          asm volatile("" : "+v"(lw));
          cplx* __restrict__ sd = A.snaps + (size_t)slot * A.snap_stride + ((size_t)b << N);
          if (odd) {
            splitr_for<0, NA>([&](auto Rc) {
              constexpr int r = decltype(Rc)::value;
              __builtin_nontemporal_store(xr[r], &sd[lw + L::index(true, 0u, r)].x);
              __builtin_nontemporal_store(xi[r], &sd[lw + L::index(true, 0u, r)].y);
            });
          } else {
            splitr_for<0, NA>([&](auto Rc) {
              constexpr int r = decltype(Rc)::value;
              __builtin_nontemporal_store(xr[r], &sd[lw + L::index(false, 0u, r)].x);
              __builtin_nontemporal_store(xi[r], &sd[lw + L::index(false, 0u, r)].y);
            });
          }
        }
        ++snap_sub;
        snap_at = snap_sub < R.nsub ? splitrun_first(R, snap_sub + 1) + n_pre - 1 : (1 << 30);
      }
    }
  } else {
    for (int sg = 0; sg < n_full; ++sg) stage(sg, std::false_type{});
  }
  if (n_full < n_stages) stage(n_stages - 1, std::true_type{});
  // the layout is the odd one when an odd number of full stages ran
  // (tan form defers the product of a stage's cosines to the next stage's D; a run that closes on the kick has no next
  // stage, so the row is finished here: prod_k cos(kick c_k) = 1 - O(1e-18) for the default kick, 1e-13 per atom for
  // long blocks set by the caller - a trace drift that accumulated over the blocks)
  double fl = (ROWS && n_full == n_stages) ? cprod : 1.0;
  if (ROWS && n_full == n_stages && !A.use_post) {
    splitr_for<0, NA>([&](auto Rc) {
      constexpr int r = decltype(Rc)::value;
      xr[r] *= fl;
      xi[r] *= fl;
    });
  }
  if (ROWS && A.use_post) {
    __syncthreads();
    if (t < NA) ftl[128 + t] = row_factor(1, odd ? L::index(true, 0u, t) : L::index(false, 0u, t), odd ? L::index(true, 0u, NA - 1) : L::index(false, 0u, NA - 1));
    __syncthreads();
    fl *= row_factor(1, odd ? L::index(true, tq, 0u) : L::index(false, tq, 0u), odd ? L::index(true, NT - 1, 0u) : L::index(false, NT - 1, 0u));
  }
  bool stored = false;
  if constexpr (!ROWS) {
    if (A.dst || A.dst2 || A.cmp) {
      // the controller's check (SplitArgs.dst / dst2 / cmp): out-of-place store, second store, comparison on the way out
      stored = true;
      const size_t boff = (size_t)b << N;
      cplx* __restrict__ out = A.dst ? A.dst + boff : st;
      cplx* __restrict__ out2 = A.dst2 ? A.dst2 + boff : nullptr;
      const cplx* __restrict__ ref = A.cmp ? A.cmp + boff : nullptr;
      const unsigned lw = odd ? L::index(true, tq, 0u) : L::index(false, tq, 0u);
      double dmax = 0.0, dsum = 0.0;
      auto put = [&](unsigned ix, double vx, double vy) {
        const cplx v = make_double2(vx, vy);
        out[ix] = v;
        if (out2) out2[ix] = v;
        if (ref) {
          const cplx c = ref[ix];
          const double dx = vx - c.x, dy = vy - c.y;
          const double d2 = fma(dx, dx, dy * dy);
          dmax = fmax(dmax, d2);
          dsum += d2;
        }
      };
      if (odd) {
        splitr_for<0, NA>([&](auto Rc) { constexpr int r = decltype(Rc)::value; put(lw + L::index(true, 0u, r), xr[r], xi[r]); });
      } else {
        splitr_for<0, NA>([&](auto Rc) { constexpr int r = decltype(Rc)::value; put(lw + L::index(false, 0u, r), xr[r], xi[r]); });
      }
      if (ref) {
        for (int o = 32; o > 0; o >>= 1) {
          dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
          dsum += __shfl_xor(dsum, o, 64);
        }
        double* red = reinterpret_cast<double*>(gtab);  // (the G tables are dead after the last stage)
        __syncthreads();
        if (l == 0) { red[2 * w] = dmax; red[2 * w + 1] = dsum; }
        __syncthreads();
        if (t == 0) {
          double m = red[0], sm = red[1];
          for (int k = 1; k < NT / 64; ++k) { m = fmax(m, red[2 * k]); sm += red[2 * k + 1]; }
          A.cmp_err[b] = m;                 // max |x - cmp|^2
          A.cmp_err[gridDim.y + b] = sm;    // sum |x - cmp|^2 (the 2-norm of the difference, squared)
        }
      }
    }
  }
  if (!stored) {
  splitr_for<0, NA>([&](auto Rc) {
    constexpr int r = decltype(Rc)::value;
    const double f = (ROWS && A.use_post) ? fl * ftl[128 + r] : 1.0;
    st[odd ? L::index(true, tq, r) : L::index(false, tq, r)] = make_double2(xr[r] * f, xi[r] * (ROWS ? f * csgn : f));
  });
  }
  }  // rows
}
