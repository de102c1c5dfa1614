// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// k_split_lane<N>: register-resident split-operator kets of 11 - 14 atoms, lane bits rotated in place (round 4)
// ---------------------------------------------------------------------------
// Same seam, same composition, same coefficient tables and controller as k_split14_loop (k_split.hpp; the call
// replaced is qutip.sesolve behind simulation.py:729-735): one workgroup per sequence holds the whole ket in
// registers (32 amplitudes per lane = 5 "register bits") over every stage of a closed run.  What changed is how the
// index bits that are NOT register bits get their rotation.  k_split14_loop turned the whole ket through LDS twice
// per stage (8 workgroup barriers per stage, the real and imaginary halves separately: 132 KiB each); SQ counters
// showed the vector pipe issuing 51.5 % of the time because all 8 waves march through the same barrier-separated
// LDS phases.  Here (NW = N - 11 wave bits, 64 << NW lanes):
//   * lane bits 0 - 3: the partner amplitude comes over the DPP crossbar (quad permutes, row rotate by 8; bit 2 needs
//     two hops) - no layout change, no LDS, no barrier: x' = x - T y_p, y' = y + T x_p is the SAME formula on both
//     partners in tan form;
//   * lane bits 4 and 5: v_permlane16_swap / v_permlane32_swap (gfx950) exchange the bit with a register bit -
//     one instruction per pair of 32-bit registers, no LDS; the layout it leaves is undone by the next stage's swap;
//   * the NW wave bits: ONE pass through LDS per stage that exchanges them with register bits 0 .. NW-1.  The lane
//     bits do not take part, so every ds_write_b128 / ds_read_b128 of a wave covers 1 KiB of consecutive
//     addresses (conflict-free without padding or swizzle, lane base + immediate).  The pass runs in four chunks
//     (register bits 3, 4 fixed: 8 amplitudes per lane, real and imaginary parts together) through two 16 * lanes * 8
//     byte buffers: 4 barriers per stage, and the rotations of the next chunk issue while the LDS absorbs the
//     stores of the current one.
// The layout after a stage is the layout before it with (wave bits <-> register bits 0..NW-1) and (lane bits 4, 5
// <-> register bits 3, 4) exchanged, so stages come in even / odd pairs (as in k_split14_loop: the stage count of a
// closed run is odd, the closing D follows the loop in the even layout = the coalesced load / store layout).
// Phase factors (the D of a stage): E0 is pairwise additive, so for lane t and register r
//   exp(-i phi) = B(t) G(r) prod_{j excited in r} F_j(t):   B = the lane's atoms alone (and their detunings, and the
//   cosines of the previous stage's tan-form rotations), F_j = register atom j against the lane's excited atoms (and
//   its detuning), G = the register atoms among themselves (uniform: lane r of every wave computes G(r) and
//   publishes it in a wave-private LDS table, read back as a broadcast)
// - 7 table-and-series sin / cos per lane and stage instead of 32, then a product tree over r (one complex
// multiplication per amplitude), the uniform factor, and the multiplication of the amplitude: 12 flops-instructions
// per amplitude against 21 + the table lookups of the per-amplitude evaluation.
// Roofline: fp64 vector pipe (the ket never leaves the CU between the first load and the last store).

#define SPLITL_TRIG 512

// index bit held by each position at the START of an even / odd stage
template <int N>
struct SplitLaneLayout {
  static constexpr int NW = N - 11;
  __host__ __device__ static constexpr int regbit(bool odd, int j) {
    if (!odd) return 6 + NW + j;
    if (j < NW) return 6 + j;        // the even layout's wave bits
    if (j < 3) return 6 + NW + j;    // register bits that never move
    return j + 1;                    // 3 -> index bit 4, 4 -> index bit 5 (the even layout's lane bits 4, 5)
  }
  __host__ __device__ static constexpr int lanebit(bool odd, int j) {
    if (j < 4 || !odd) return j;
    return 6 + NW + 3 + (j - 4);     // lane bits 4, 5 hold the even layout's register bits 3, 4
  }
  __host__ __device__ static constexpr int wavebit(bool odd, int j) { return odd ? 6 + NW + j : 6 + j; }
  __host__ __device__ static constexpr unsigned index(bool odd, unsigned t, unsigned r) {
    unsigned i = 0;
    for (int j = 0; j < 6; ++j) i |= ((t >> j) & 1u) << lanebit(odd, j);
    for (int j = 0; j < NW; ++j) i |= ((t >> (6 + j)) & 1u) << wavebit(odd, j);
    for (int j = 0; j < 5; ++j) i |= ((r >> j) & 1u) << regbit(odd, j);
    return i;
  }
};

template <int CTRL>
__device__ __forceinline__ double splitl_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// partner of lane bit f (0 - 3) of a double
template <int F>
__device__ __forceinline__ double splitl_partner(double v) {
  if constexpr (F == 0) return splitl_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (F == 1) return splitl_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (F == 3) return splitl_dpp<0x128>(v);  // row_ror:8
  else return splitl_dpp<0x1B>(splitl_dpp<0x141>(v));      // xor 4 = (xor 3) o (xor 7): row_half_mirror, then quad reverse
}

template <int N, bool DECAY>
__global__ __launch_bounds__(64 << (N - 11)) void k_split_lane(const SplitArgs A, const SplitRun R, long long stage_stride) {
  typedef SplitLaneLayout<N> L;
  constexpr int NW = L::NW;
  constexpr int NT = 64 << NW;
  constexpr int PW = (1 << NW) - 1;  // mask of the register bits the pass exchanges
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* pbuf = reinterpret_cast<cplx*>(smem);            // 2 buffers x (NT * 8) slots of 16 B
  cplx* trig = pbuf + 2 * NT * 8;                        // SPLITL_TRIG
  cplx* gtab = trig + SPLITL_TRIG;                       // [waves][32]: G(r) of the current stage, wave-private

  const unsigned t = threadIdx.x;
  const unsigned l = t & 63u, w = t >> 6;
  const int b = blockIdx.y;
  const int n_stages = R.S * R.nsub + 1;
  cplx* __restrict__ st = A.state + ((size_t)b << N);
  const double* __restrict__ coefs = A.ccur + (size_t)b * N * 4;
  const double* __restrict__ e0 = A.e0 + (size_t)b * A.e0_stride;

  for (unsigned k = t; k < SPLITL_TRIG; k += NT) {  // (before the state is loaded: the library routine wants registers)
    double sn, cs;
    sincospi((double)k * (2.0 / SPLITL_TRIG), &sn, &cs);
    trig[k] = make_double2(cs, sn);
  }
  // E0 pieces of the two layouts D meets (bit = 1 is the ground state: index with all register bits set = the lane's
  // atoms alone; with all lane bits set = the register atoms alone)
  double et[2], ev[2][5], eg[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    et[o] = e0[L::index(o, t, 31u)];
#pragma unroll
    for (int j = 0; j < 5; ++j) ev[o][j] = e0[L::index(o, t, 31u ^ (1u << j))] - et[o];
    eg[o] = e0[L::index(o, NT - 1, l & 31u)];
  }
  __syncthreads();
  double xr[32], xi[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    const cplx v = st[L::index(false, t, r)];
    xr[r] = v.x;
    xi[r] = v.y;
  }

  // coefficients of the current stage by index bit p (atom N - 1 - p): uniform, scalar loads (see k_split14_loop)
  typedef const __attribute__((address_space(4))) double* cptr_t;
  double cT[N], cDl[N];
  double cprod = 1.0, cnext = 1.0;
  auto load_coefs = [&](const double* cs) {
    cptr_t c4 = (cptr_t)(unsigned long long)cs;
    double prod = 1.0;
#pragma unroll
    for (int p = 0; p < N; ++p) {
      cptr_t c = c4 + 4 * (N - 1 - p);
      prod *= c[0];
      cT[p] = c[1];  // gi / C (SplitRun.tan_form)
      cDl[p] = c[3];
    }
    cprod = cnext;
    cnext = prod;
  };
  // exp(-i phi) = (cos, -sin): table of exp(2 pi i k / 512), series on |rr| <= pi / 512
  auto expmi = [&](double phi) -> cplx {
    const double kk = rint(phi * (SPLITL_TRIG / 6.283185307179586));
    double rr = fma(-kk, 6.283185307179586 / SPLITL_TRIG, phi);
    rr = fma(-kk, 2.4492935982947064e-16 / SPLITL_TRIG, rr);  // 2 pi - double(2 pi)
    const cplx tb = trig[((int)kk) & (SPLITL_TRIG - 1)];
    const double r2 = rr * rr;
    const double cr = fma(r2, fma(r2, 4.1666666666666664e-02, -0.5), 1.0);
    const double sr = rr * fma(r2, fma(r2, 8.333333333333333e-03, -1.6666666666666666e-01), 1.0);
    return make_double2(fma(tb.x, cr, -tb.y * sr), -fma(tb.y, cr, tb.x * sr));
  };
  auto cm = [](cplx a, cplx c) -> cplx { return make_double2(fma(a.x, c.x, -a.y * c.y), fma(a.x, c.y, a.y * c.x)); };

  // D in the layout at the start of an even / odd stage
  auto phase = [&](double wE, auto odd_t) {
    constexpr bool odd = decltype(odd_t)::value;
    double dthr = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) dthr += ((t >> j) & 1u) ? 0.0 : cDl[L::lanebit(odd, j)];
#pragma unroll
    for (int j = 0; j < NW; ++j) dthr += ((t >> (6 + j)) & 1u) ? 0.0 : cDl[L::wavebit(odd, j)];
    cplx Bf = expmi(fma(wE, et[odd], -dthr));
    double sc = cprod;
    if (DECAY) {
      // H_eff: the real factor exp(wE (dec_a + dec_b popc(index))) (SplitArgs): lane part here, register part in F
      sc *= exp(wE * (A.dec_a + A.dec_b * (double)(__popc(t) + 5)));
    }
    Bf = make_double2(Bf.x * sc, Bf.y * sc);
    cplx F[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      F[j] = expmi(fma(wE, ev[odd][j], -cDl[L::regbit(odd, j)]));
      if (DECAY) {
        const double d = exp(-wE * A.dec_b);  // an excited register atom: one set bit fewer
        F[j] = make_double2(F[j].x * d, F[j].y * d);
      }
    }
    gtab[w * 32 + (l & 31u)] = expmi(wE * eg[odd]);  // (lanes l and l + 32 write the same value)
    __builtin_amdgcn_wave_barrier();
    const cplx* __restrict__ gw = gtab + w * 32;
    cplx P[32];
#pragma unroll
    for (int r = 31; r >= 0; --r) {
      if (r == 31) {
        P[r] = Bf;
      } else {
        int j = 0;
        while (r & (1 << j)) ++j;  // lowest clear bit: the parent has it set
        P[r] = cm(P[r | (1 << j)], F[j]);
      }
      const cplx q = cm(P[r], gw[r]);
      const double ax = xr[r], ay = xi[r];
      xr[r] = fma(ax, q.x, -ay * q.y);
      xi[r] = fma(ax, q.y, ay * q.x);
    }
    __builtin_amdgcn_wave_barrier();  // (the next stage's table is written after this stage's reads)
  };

  // rotation of register bit j on the registers [r_lo, r_hi) with the coefficient of index bit p
  auto rot_reg = [&](int j, int p, int r_lo, int r_hi) {
    const double T = cT[p];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (r < r_lo || r >= r_hi || (r & (1 << j))) continue;
      const int q = r | (1 << j);
      const double a0x = xr[r], a0y = xi[r], a1x = xr[q], a1y = xi[q];
      xr[r] = fma(-T, a1y, a0x);
      xi[r] = fma(T, a1x, a0y);
      xr[q] = fma(-T, a0y, a1x);
      xi[q] = fma(T, a0x, a1y);
    }
  };
  // rotations of lane bits 0 - 3 (index bits 0 - 3 in both layouts) on register r
  auto rot_lane = [&](int r) {
    {
      const double T = cT[0], px = splitl_partner<0>(xr[r]), py = splitl_partner<0>(xi[r]);
      xr[r] = fma(-T, py, xr[r]);
      xi[r] = fma(T, px, xi[r]);
    }
    {
      const double T = cT[1], px = splitl_partner<1>(xr[r]), py = splitl_partner<1>(xi[r]);
      xr[r] = fma(-T, py, xr[r]);
      xi[r] = fma(T, px, xi[r]);
    }
    {
      const double T = cT[3], px = splitl_partner<3>(xr[r]), py = splitl_partner<3>(xi[r]);
      xr[r] = fma(-T, py, xr[r]);
      xi[r] = fma(T, px, xi[r]);
    }
    {
      const double T = cT[2], px = splitl_partner<2>(xr[r]), py = splitl_partner<2>(xi[r]);
      xr[r] = fma(-T, py, xr[r]);
      xi[r] = fma(T, px, xi[r]);
    }
  };
  // register bit 3 <-> lane bit 4, register bit 4 <-> lane bit 5
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
  auto swap_lanes = [&]() {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (r & 8) continue;
      swap_d(xr[r], xr[r | 8], std::false_type{});
      swap_d(xi[r], xi[r | 8], std::false_type{});
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (r & 16) continue;
      swap_d(xr[r], xr[r | 16], std::true_type{});
      swap_d(xi[r], xi[r | 16], std::true_type{});
    }
  };
  // the pass: chunk c = register bits 3, 4; slot = l + 64 (A + 2^NW (B + 2^NW k_s)),  writer: A = wave, B = the
  // exchanged register bits;  reader: A = its register bits, B = its wave;  k_s = the chunk's register bits that stay
  auto pass_write = [&](int c) {
    cplx* __restrict__ pb = pbuf + (c & 1) * (NT * 8) + t;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kp = k & PW, ks = k >> NW;
      pb[64 * ((kp << NW) + (ks << (2 * NW)))] = make_double2(xr[c * 8 + k], xi[c * 8 + k]);
    }
  };
  auto pass_read = [&](int c) {
    const cplx* __restrict__ pb = pbuf + (c & 1) * (NT * 8) + l + 64 * (w << NW);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kp = k & PW, ks = k >> NW;
      const cplx v = pb[64 * (kp + (ks << (2 * NW)))];
      xr[c * 8 + k] = v.x;
      xi[c * 8 + k] = v.y;
    }
  };

  auto stage_w = [&](int j) -> double {
    const bool last = j == n_stages - 1;
    const int sub = last ? R.nsub - 1 : j / R.S, i = last ? R.S : j % R.S;
    double wgt = R.a[i] * R.tau[sub];
    if (!last && i == 0 && sub > 0) wgt += R.a[R.S] * R.tau[sub - 1];
    return wgt;
  };
  // one stage from the layout `odd` to the other one
  auto stage = [&](int sg, auto odd_t) {
    constexpr bool odd = decltype(odd_t)::value;
    load_coefs(coefs + (size_t)sg * stage_stride);
    phase(stage_w(sg), odd_t);
    // register bits 3, 4: this layout's, then (after the swap) the ones lane bits 4, 5 held
    rot_reg(3, L::regbit(odd, 3), 0, 32);
    rot_reg(4, L::regbit(odd, 4), 0, 32);
    swap_lanes();
    rot_reg(3, L::lanebit(odd, 4), 0, 32);
    rot_reg(4, L::lanebit(odd, 5), 0, 32);
    // per chunk: register bits 0 - 2 and lane bits 0 - 3, then into the pass
    auto pre = [&](int c) {
#pragma unroll
      for (int j = 0; j < 3; ++j) rot_reg(j, L::regbit(odd, j), c * 8, c * 8 + 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) rot_lane(c * 8 + k);
    };
    auto post = [&](int c) {  // the wave bits, now register bits 0 .. NW-1
#pragma unroll
      for (int j = 0; j < NW; ++j) rot_reg(j, L::wavebit(odd, j), c * 8, c * 8 + 8);
    };
    if constexpr (NW == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) pre(c);
    } else {
      pre(0);
      pass_write(0);
      pre(1);
      __syncthreads();
      pass_read(0);
      pass_write(1);
      pre(2);
      __syncthreads();
      pass_read(1);
      pass_write(2);
      pre(3);
      __syncthreads();
      pass_read(2);
      pass_write(3);
      post(0);
      post(1);
      __syncthreads();
      pass_read(3);
      post(2);
      post(3);
    }
  };

  for (int sg = 0; sg + 1 < n_stages; sg += 2) {
    stage(sg, std::false_type{});
    stage(sg + 1, std::true_type{});
  }
  load_coefs(coefs + (size_t)(n_stages - 1) * stage_stride);
  phase(stage_w(n_stages - 1), std::false_type{});  // the closing D
#pragma unroll
  for (int r = 0; r < 32; ++r) st[L::index(false, t, r)] = make_double2(xr[r], xi[r]);
}
