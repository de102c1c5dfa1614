// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Persistent trajectory kernel (sesolve, N <= 13)
// ---------------------------------------------------------------------------
// One workgroup evolves one state vector through a whole schedule of CF4 steps
// in a single launch: psi lives in registers (thread t owns amplitudes
// t + j*NT), the Horner iterate lives in LDS for the flip-partner reads, and
// HBM is touched only for the initial load, the snapshots and the final store.
// Flip partners of the high index bits (>= log2 NT) are register-to-register.
// 1/j for the Horner scale h/j (orders are capped at 32)
__constant__ double kInvInt[33] = {
    0.0, 1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6, 1.0 / 7, 1.0 / 8, 1.0 / 9, 1.0 / 10,
    1.0 / 11, 1.0 / 12, 1.0 / 13, 1.0 / 14, 1.0 / 15, 1.0 / 16, 1.0 / 17, 1.0 / 18, 1.0 / 19,
    1.0 / 20, 1.0 / 21, 1.0 / 22, 1.0 / 23, 1.0 / 24, 1.0 / 25, 1.0 / 26, 1.0 / 27, 1.0 / 28,
    1.0 / 29, 1.0 / 30, 1.0 / 31, 1.0 / 32};

struct StepDesc {
  double h, u1, u2;
  double shift_a, shift_b;
  int idx;
  int order_a, order_b;
  int snap;  // snapshot slot written after this step, or -1
  int pad;   // host: number of knot intervals the step spans (bounds); unused on the device
};

struct TrajArgs {
  cplx* state;         // [B][2^N] in/out
  cplx* snaps;         // [n_slots][B][2^N] or null
  const cplx* pp;      // [n_series][n_int][4]
  const ryd_qdesc* desc;
  const ryd_dterm* dterms;  // extra detuning terms (or null)
  const double* e0;
  long long e0_stride;
  const StepDesc* steps;
  int n_int, n_steps, B;
  double a1, a2;
  // Monte-Carlo wavefunction instantiations (MC = true)
  McState mc;
  double mc_a, mc_b;  // real diagonal of G_eff: mc_a + mc_b * popc(index)
  int mc_jumps;       // 0: no-jump evolution under H_eff only
};

// MODEL 0: per-atom complex drive coefficients (local addressing, noise).
// MODEL 1: one real drive coefficient shared by the driven atoms of the
//          trajectory (global channel with constant zero phase; bad atoms are
//          masked out) - the flip partners are summed first, 2 DADD each.
// MC: the generator is G_eff (adds the real decay diagonal); with A.mc_jumps the
//     norm threshold is tested after every step and collapses are applied in
//     place (same arithmetic and random stream as the k_mc_* kernels).
// partner amplitude of lane bit 0 / 1 / 3 over the DPP crossbar (quad permutes, row rotate by 8): the LDS read
// bandwidth is what binds this kernel at 12 atoms (40 ds_read_b128 per lane and stage x 16 waves = 5 120 LDS
// clocks against 3 200 vector-issue cycles), so three of the ten LDS-served bits move to the vector pipe
template <int CTRL>
__device__ __forceinline__ cplx traj_dpp(cplx v) {
  const int a = __builtin_amdgcn_mov_dpp(__double2loint(v.x), CTRL, 0xF, 0xF, true);
  const int b = __builtin_amdgcn_mov_dpp(__double2hiint(v.x), CTRL, 0xF, 0xF, true);
  const int c = __builtin_amdgcn_mov_dpp(__double2loint(v.y), CTRL, 0xF, 0xF, true);
  const int d = __builtin_amdgcn_mov_dpp(__double2hiint(v.y), CTRL, 0xF, 0xF, true);
  return make_double2(__hiloint2double(b, a), __hiloint2double(d, c));
}
#ifndef RYD_TRAJ_DPPM
#define RYD_TRAJ_DPPM 0xB
#endif

template <int N, int NTT, int MODEL, bool MC>
__global__ __launch_bounds__(NTT) void k_traj(const TrajArgs A) {
  constexpr int D = 1 << N;
  constexpr int R = D / NTT > 0 ? D / NTT : 1;
  constexpr int LOGNT = NTT == 1024 ? 10 : (NTT == 512 ? 9 : (NTT == 256 ? 8 : (NTT == 128 ? 7 : 6)));
  // two LDS buffers: the Horner iterate ping-pongs, one barrier per stage.  13 atoms
  // (128 KiB per copy) only fit once: one buffer, two barriers per stage; there every
  // partner (also of the register-index bits) is read from LDS, so a new amplitude
  // can replace the old one in its register at once (8 amplitudes per thread leave
  // no room for an old and a new copy).
  constexpr bool SINGLE = (size_t)2 * D * sizeof(cplx) > 144 * 1024;
  constexpr int NLDS = SINGLE ? N : (N < LOGNT ? N : LOGNT);  // bits whose partner is read from LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* ws0 = reinterpret_cast<cplx*>(smem);
  cplx* ws1 = SINGLE ? ws0 : ws0 + D;
  double* cfA = reinterpret_cast<double*>(ws0 + (SINGLE ? D : 2 * D));  // [16][4]: cr, ci, delta, 0 for exp A
  double* cfB = cfA + 64;                            // same for exp B
  double* mcred = cfB + 64;                          // [16][4] per-wave partial sums
  double* mcrho = mcred + 64;                        // [16][4] reduced density matrices
  double* hfx = mcrho + 64;                          // [16][2] extra detuning terms (exp A, exp B)

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const bool active = tid < D;
  cplx* st = A.state + (size_t)b * D;
  const double* e0g = A.e0 + (size_t)b * A.e0_stride;

  cplx psi[R];
  double e0r[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int l = tid + j * NTT;
    psi[j] = active ? st[l] : make_double2(0.0, 0.0);
    e0r[j] = active ? e0g[l] : 0.0;
  }
  // Monte-Carlo bookkeeping (block-uniform)
  const bool jumps = MC && A.mc_jumps != 0;
  double mc_target = 0.0, mc_ref = 1.0, mc_n2 = 1.0;
  int mc_count = 0;
  unsigned long long mc_seed = 0;
  if (jumps) {
    mc_target = A.mc.target[b];
    mc_ref = A.mc.refnorm[b];
    mc_n2 = A.mc.lastnorm[b];
    mc_count = A.mc.count[b];
    mc_seed = A.mc.seeds[b];
  }

  for (int s = 0; s < A.n_steps; ++s) {
    const StepDesc sd = A.steps[s];
    if (A.dterms) {
      // extra detuning terms (hf noise): wave w sums the list of atoms w, w + NW, ...
      constexpr int NW = NTT / 64;
      const int lane = tid & 63;
      for (int k = tid >> 6; k < N; k += NW) {
        const int ex = A.desc[(size_t)b * N + k].extra;
        double xa = 0.0, xb = 0.0;
        if (ex > 0) {
          const int count = A.dterms[ex - 1].remaining + 1;
          for (int e = lane; e < count; e += 64) {
            const ryd_dterm t = A.dterms[ex - 1 + e];
            const cplx* p = A.pp + ((size_t)t.series * A.n_int + sd.idx) * 4;
            const double o1 = ((p[0].x * sd.u1 + p[1].x) * sd.u1 + p[2].x) * sd.u1 + p[3].x;
            const double o2 = ((p[0].x * sd.u2 + p[1].x) * sd.u2 + p[2].x) * sd.u2 + p[3].x;
            xa += t.scale * (A.a1 * o1 + A.a2 * o2);
            xb += t.scale * (A.a2 * o1 + A.a1 * o2);
          }
          for (int o = 32; o > 0; o >>= 1) {
            xa += __shfl_down(xa, o, 64);
            xb += __shfl_down(xb, o, 64);
          }
        }
        if (lane == 0) { hfx[2 * k] = xa; hfx[2 * k + 1] = xb; }
      }
      __syncthreads();
    }
    if (tid < N) {
      const ryd_qdesc d = A.desc[(size_t)b * N + tid];
      auto val = [&](int ser, double u) -> cplx {
        const cplx* p = A.pp + ((size_t)ser * A.n_int + sd.idx) * 4;
        cplx r = p[0];
        r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
        r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
        r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
        return r;
      };
      // same arithmetic as k_eval_coefs (w1 * val(t1) + w2 * val(t2))
      double c1r = 0, c1i = 0, c2r = 0, c2i = 0, dlA = 0, dlB = 0;
      if (d.drive_series >= 0) {
        const cplx v1 = val(d.drive_series, sd.u1), v2 = val(d.drive_series, sd.u2);
        c1r = v1.x; c1i = v1.y; c2r = v2.x; c2i = v2.y;
      }
      if (d.det_series >= 0) {
        const double d1 = val(d.det_series, sd.u1).x, d2 = val(d.det_series, sd.u2).x;
        dlA += d.det_scale * (A.a1 * d1 + A.a2 * d2);
        dlB += d.det_scale * (A.a2 * d1 + A.a1 * d2);
      }
      if (d.off_series >= 0) {
        const double o1 = val(d.off_series, sd.u1).x, o2 = val(d.off_series, sd.u2).x;
        dlA += d.off_scale * (A.a1 * o1 + A.a2 * o2);
        dlB += d.off_scale * (A.a2 * o1 + A.a1 * o2);
      }
      if (d.extra > 0 && A.dterms) {  // summed by one wave per atom above
        dlA += hfx[2 * tid];
        dlB += hfx[2 * tid + 1];
      }
      cfA[4 * tid + 0] = d.drive_scale * (A.a1 * c1r + A.a2 * c2r);
      cfA[4 * tid + 1] = d.drive_scale * (A.a1 * c1i + A.a2 * c2i);
      cfA[4 * tid + 2] = dlA;
      cfA[4 * tid + 3] = d.drive_series >= 0 ? 1.0 : 0.0;
      cfB[4 * tid + 0] = d.drive_scale * (A.a2 * c1r + A.a1 * c2r);
      cfB[4 * tid + 1] = d.drive_scale * (A.a2 * c1i + A.a1 * c2i);
      cfB[4 * tid + 2] = dlB;
      cfB[4 * tid + 3] = d.drive_series >= 0 ? 1.0 : 0.0;
    }
    __syncthreads();

#pragma unroll 1
    for (int ex = 0; ex < 2; ++ex) {
      const double* cf = ex ? cfB : cfA;
      const int order = ex ? sd.order_b : sd.order_a;
      const double shift = ex ? sd.shift_b : sd.shift_a;
      const double wmix = A.a1 + A.a2;
      // all per-atom values of this exponential in one batch of LDS reads
      // (bit q <-> atom N-1-q), then wave-uniform ones -> scalar registers
      double craw[N], ciraw[N], draw[N], mraw[N];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const double2 a = *reinterpret_cast<const double2*>(cf + 4 * (N - 1 - q));
        const double2 c = *reinterpret_cast<const double2*>(cf + 4 * (N - 1 - q) + 2);
        craw[q] = a.x; ciraw[q] = a.y; draw[q] = c.x; mraw[q] = c.y;
      }
      double cr[MODEL == 0 ? N : 1], ci[MODEL == 0 ? N : 1];
      double mq[MODEL == 1 ? N : 1];  // 1.0 for driven atoms, 0.0 otherwise
      double cuni = 0.0;
      if (MODEL == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
          cr[q] = uniform_d(craw[q]);
          ci[q] = uniform_d(ciraw[q]);
        }
      } else {
        double cv = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q) cv = mraw[q] != 0.0 ? craw[q] : cv;
        cuni = uniform_d(cv);
#pragma unroll
        for (int q = 0; q < N; ++q) mq[q] = uniform_d(mraw[q]);
      }
      double eg[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int l = tid + j * NTT;
        double sdet = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q)
          if (!((l >> q) & 1)) sdet -= draw[q];
        eg[j] = sdet + (wmix * e0r[j] - shift);
      }
      double er[MC ? R : 1];  // centred real part of the G_eff diagonal
      if (MC) {
#pragma unroll
        for (int j = 0; j < R; ++j)
          er[j] = wmix * A.mc_b * ((double)__popc(tid + j * NTT) - 0.5 * N);
      }
      cplx w[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        w[j] = psi[j];
        if (active) ws0[tid + j * NTT] = w[j];
      }
      __syncthreads();
      const cplx* rd = ws0;
      cplx* wr = ws1;
      for (int jj = order; jj >= 1; --jj) {
        const double sc = sd.h * kInvInt[jj];  // a v_div_f64 costs ~15 VALU issue slots per stage
        cplx acc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int l = tid + j * NTT;
          // issue every LDS partner read of this element before using any
          cplx xv[NLDS > 0 ? NLDS : 1];
          if (SINGLE && MODEL == 1) {
            // 13 atoms: two half-batches of partner reads (7 + 6) keep psi and w in
            // registers; all partners share one real coefficient, so they are summed
            double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0;
#pragma unroll
            for (int q0 = 0; q0 < N; q0 += 7) {
#pragma unroll
              for (int q = q0; q < q0 + 7 && q < N; ++q) xv[q - q0] = rd[(l ^ (1 << q)) & (D - 1)];
#pragma unroll
              for (int q = q0; q < q0 + 7 && q < N; ++q) {
                const cplx x = xv[q - q0];
                if (q & 1) { s1x = fma(mq[q], x.x, s1x); s1y = fma(mq[q], x.y, s1y); }
                else { s0x = fma(mq[q], x.x, s0x); s0y = fma(mq[q], x.y, s0y); }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            const double sx = s0x + s1x, sy = s0y + s1y;
            acc[j] = make_double2(fma(cuni, sy, eg[j] * w[j].y), -fma(cuni, sx, eg[j] * w[j].x));
            if (MC) acc[j] = make_double2(fma(er[j], w[j].x, acc[j].x), fma(er[j], w[j].y, acc[j].y));
            w[j] = make_double2(fma(sc, acc[j].x, psi[j].x), fma(sc, acc[j].y, psi[j].y));
            __builtin_amdgcn_sched_barrier(0);
            continue;
          }
          // index bits 0, 1, 3 (lane bits; only the two-buffer layouts, where w[j] is the published iterate)
          constexpr unsigned DPPM = (!SINGLE && NTT >= 64) ? (unsigned)RYD_TRAJ_DPPM & ((1u << (N < 4 ? N : 4)) - 1u) : 0u;
#pragma unroll
          for (int q = 0; q < NLDS; ++q)
            if (!((DPPM >> q) & 1u)) xv[q] = rd[(l ^ (1 << q)) & (D - 1)];
          if constexpr (DPPM & 1u) xv[0] = traj_dpp<0xB1>(w[j]);
          if constexpr ((DPPM & 2u) != 0) xv[1 < NLDS ? 1 : 0] = traj_dpp<0x4E>(w[j]);
          if constexpr ((DPPM & 8u) != 0) xv[3 < NLDS ? 3 : 0] = traj_dpp<0x128>(w[j]);
          if (MODEL == 0) {
            cplx a = make_double2(eg[j] * w[j].y, -eg[j] * w[j].x);  // -i e x
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              // coefficient -i c (output bit 1) or -i conj(c) (output bit 0)
              const double sgi = ((l >> q) & 1) ? ci[q] : -ci[q];
              a = cfma(make_double2(sgi, -cr[q]), x, a);
            }
            acc[j] = a;
          } else {
            double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0;  // two chains
#pragma unroll
            for (int q = 0; q < N; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              if (q & 1) { s1x = fma(mq[q], x.x, s1x); s1y = fma(mq[q], x.y, s1y); }
              else { s0x = fma(mq[q], x.x, s0x); s0y = fma(mq[q], x.y, s0y); }
            }
            const double sx = s0x + s1x, sy = s0y + s1y;
            // -i (e w + c sum)
            acc[j] = make_double2(fma(cuni, sy, eg[j] * w[j].y), -fma(cuni, sx, eg[j] * w[j].x));
          }
          if (SINGLE) {  // in place: nothing else reads w[j] from the register
            if (MC) acc[j] = make_double2(fma(er[j], w[j].x, acc[j].x), fma(er[j], w[j].y, acc[j].y));
            w[j] = make_double2(fma(sc, acc[j].x, psi[j].x), fma(sc, acc[j].y, psi[j].y));
            __builtin_amdgcn_sched_barrier(0);  // keep the 13 partner reads of the next amplitude behind this one
          }
        }
        if (MC && !SINGLE) {
#pragma unroll
          for (int j = 0; j < R; ++j)
            acc[j] = make_double2(fma(er[j], w[j].x, acc[j].x), fma(er[j], w[j].y, acc[j].y));
        }
        if (!SINGLE) {
#pragma unroll
          for (int j = 0; j < R; ++j)
            w[j] = make_double2(fma(sc, acc[j].x, psi[j].x), fma(sc, acc[j].y, psi[j].y));
        }
        if (jj > 1) {
          if (SINGLE) __syncthreads();  // every partner read of the old iterate is done
#pragma unroll
          for (int j = 0; j < R; ++j)
            if (active) wr[tid + j * NTT] = w[j];
          __syncthreads();  // one barrier per stage: reads of `rd` done, `wr` visible
          const cplx* t = rd;
          rd = wr;
          wr = const_cast<cplx*>(t);
        }
      }
      const double mag = MC ? exp(sd.h * wmix * (A.mc_a + 0.5 * N * A.mc_b)) : 1.0;
      const cplx post = make_double2(mag * cos(sd.h * shift), -mag * sin(sd.h * shift));
#pragma unroll
      for (int j = 0; j < R; ++j) psi[j] = cmul(post, w[j]);
      __syncthreads();  // last-stage reads done before ws0 / cf are rewritten
    }
    if (jumps) {
      constexpr int NW = NTT / 64;
      const int lane = tid & 63, wave = tid >> 6;
      double s2 = 0.0;
#pragma unroll
      for (int j = 0; j < R; ++j) s2 = fma(psi[j].x, psi[j].x, fma(psi[j].y, psi[j].y, s2));
      if (!active) s2 = 0.0;  // lanes beyond a small state hold garbage
      for (int o = 32; o > 0; o >>= 1) s2 += __shfl_down(s2, o, 64);
      if (lane == 0) mcred[wave] = s2;
      __syncthreads();
      double n2 = 0.0;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) n2 += mcred[wv];
      mc_n2 = n2;
      __syncthreads();
      if (n2 <= mc_target * mc_ref) {  // block-uniform: this trajectory jumps now
#pragma unroll
        for (int j = 0; j < R; ++j)
          if (active) ws0[tid + j * NTT] = psi[j];
        __syncthreads();
        for (int a = 0; a < N; ++a) {
          const int bit = 1 << (N - 1 - a);
          double rr = 0.0, gg = 0.0, cr = 0.0, ci = 0.0;
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int l = tid + j * NTT;
            const cplx v = psi[j];
            const double m = v.x * v.x + v.y * v.y;
            if (l & bit) {
              gg += m;
            } else {
              const cplx wv = ws0[(l | bit) & (D - 1)];
              rr += m;
              cr += v.x * wv.x + v.y * wv.y;
              ci += v.y * wv.x - v.x * wv.y;
            }
          }
          if (!active) rr = gg = cr = ci = 0.0;
          for (int o = 32; o > 0; o >>= 1) {
            rr += __shfl_down(rr, o, 64);
            gg += __shfl_down(gg, o, 64);
            cr += __shfl_down(cr, o, 64);
            ci += __shfl_down(ci, o, 64);
          }
          if (lane == 0) {
            mcred[4 * wave + 0] = rr;
            mcred[4 * wave + 1] = gg;
            mcred[4 * wave + 2] = cr;
            mcred[4 * wave + 3] = ci;
          }
          __syncthreads();
          if (tid < 4) {
            double t = 0.0;
            for (int wv = 0; wv < NW; ++wv) t += mcred[4 * wv + tid];
            mcrho[4 * a + tid] = t;
          }
          __syncthreads();
        }
        // selection: every thread repeats the (uniform) arithmetic of k_mc_select
        double total = 0.0;
        for (int a = 0; a < N; ++a)
          for (int k = 0; k < A.mc.n_ops; ++k)
            total += fmax(mc_weight(A.mc.ops + 4 * k, mcrho + 4 * a), 0.0);
        if (total > 0.0) {
          double ut, us;
          mc_uniforms(mc_seed, (unsigned)mc_count, &ut, &us);
          const double x = us * total;
          double cum = 0.0, psel = 0.0, plast = 0.0;
          int sel = -1, lastpos = -1;
          for (int a = 0; a < N; ++a)
            for (int k = 0; k < A.mc.n_ops; ++k) {
              const double p = fmax(mc_weight(A.mc.ops + 4 * k, mcrho + 4 * a), 0.0);
              cum += p;
              if (p > 0.0) { lastpos = a * MC_MAX_OPS + k; plast = p; }
              if (sel < 0 && p > 0.0 && cum > x) { sel = a * MC_MAX_OPS + k; psel = p; }
            }
          if (sel < 0) { sel = lastpos; psel = plast; }
          const int pbit = N - 1 - sel / MC_MAX_OPS;
          const cplx* C = A.mc.ops + 4 * (sel % MC_MAX_OPS);
          const double sc2 = 1.0 / sqrt(psel);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const int l = tid + j * NTT;
            const int row = (l >> pbit) & 1;
            const cplx v0 = ws0[(l & ~(1 << pbit)) & (D - 1)], v1 = ws0[(l | (1 << pbit)) & (D - 1)];
            const cplx o = cfma(C[2 * row], v0, cmul(C[2 * row + 1], v1));
            if (active) psi[j] = make_double2(sc2 * o.x, sc2 * o.y);
          }
          ++mc_count;
          mc_uniforms(mc_seed, (unsigned)mc_count, &ut, &us);
          mc_target = ut;
          mc_ref = 1.0;
          mc_n2 = 1.0;
        }
        __syncthreads();  // partner reads of ws0 done before the next step rewrites it
      }
    }
    if (sd.snap >= 0 && A.snaps && active) {
      cplx* o = A.snaps + ((size_t)sd.snap * A.B + b) * D;
      const double ns = jumps ? rsqrt(mc_n2) : 1.0;  // stored kets are normalised
#pragma unroll
      for (int j = 0; j < R; ++j) o[tid + j * NTT] = make_double2(ns * psi[j].x, ns * psi[j].y);
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < R; ++j) st[tid + j * NTT] = psi[j];
  }
  if (jumps && tid == 0) {
    A.mc.target[b] = mc_target;
    A.mc.refnorm[b] = mc_ref;
    A.mc.lastnorm[b] = mc_n2;
    A.mc.count[b] = mc_count;
  }
}
