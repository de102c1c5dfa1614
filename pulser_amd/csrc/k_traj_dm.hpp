// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Persistent trajectory kernel for small density matrices (mesolve, N <= 6)
// ---------------------------------------------------------------------------
// The Lindbladian on rho as a 2N-bit vector (row bits [N, 2N), column bits
// [0, N)) has the structure of the ket generator - a diagonal, single-bit flips,
// and for C rho C^dag jumps double flips of the (row, column) bit pair of an atom -
// so a 6-atom density matrix is the same 4096-amplitude problem as a 12-atom ket:
// one workgroup integrates one trajectory's rho through the whole schedule in a
// single launch (registers + two LDS copies of the Horner iterate), instead of
// one launch per Taylor stage.  Same arithmetic as k_apply<RYD_MESOLVE>:
//   row bit:    -i c (bit 1) / -i conj(c) (bit 0);   column bit: +i conj(c) / +i c
//   diagonal:   wmix * sum_pairs Sd[2 a_k + b_k]  -  i (E(a) - E(b))
//   double flip on (a_k, b_k): wmix * J[2 a_k + b_k] (indexed by the OUTPUT pair).
// Small registers (the reference's own noisy workloads: 1-6 atoms with dephasing,
// relaxation, depolarizing, SPAM trajectories) are launch-latency-bound on the
// tiled kernels; this kernel removes the launches.
struct TrajDmArgs {
  cplx* state;   // [B][4^N] in/out
  cplx* snaps;   // [n_slots][B][4^N] or null
  const cplx* pp;
  const ryd_qdesc* desc;
  const ryd_dterm* dterms;
  const double* e0;  // [1 or B][2^N]
  long long e0_stride;
  const StepDesc* steps;
  int n_int, n_steps, B;
  double a1, a2;
  cplx Sd[4], J[4];
};

// REALU: every driven atom of the trajectory shares one real drive coefficient (global
// channel, phase 0; bad atoms masked out): the flip partners are summed first, row
// partners with +1 and column partners with -1, and multiplied once (-i c (S_row - S_col)).
template <int N, int NTT, bool DBL, bool REALU>
__global__ __launch_bounds__(NTT) void k_traj_dm(const TrajDmArgs A) {
  constexpr int NB = 2 * N;
  constexpr int D = 1 << NB;
  constexpr int DN = 1 << N;
  constexpr int R = D / NTT > 0 ? D / NTT : 1;
  constexpr int LOGNT = NTT == 1024 ? 10 : (NTT == 512 ? 9 : (NTT == 256 ? 8 : (NTT == 128 ? 7 : 6)));
  constexpr int NLDS = NB < LOGNT ? NB : LOGNT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx* ws0 = reinterpret_cast<cplx*>(smem);
  cplx* ws1 = ws0 + D;
  double* cfA = reinterpret_cast<double*>(ws1 + D);  // [8][4]: cr, ci, delta, driven
  double* cfB = cfA + 32;

  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const bool active = tid < D;
  cplx* st = A.state + (size_t)b * D;
  const double* e0g = A.e0 + (size_t)b * A.e0_stride;
  const double wmix = A.a1 + A.a2;  // both exponentials of a CF4 step carry weight 1/2

  // static part of the diagonal: dissipator popcounts and the interaction energies
  cplx psi[R];
  double ddr[R], dbase[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int l = tid + j * NTT;
    psi[j] = active ? st[l] : make_double2(0.0, 0.0);
    const unsigned a = (unsigned)(l >> N) & (DN - 1), bb = (unsigned)l & (DN - 1);
    const int n11 = __popc(a & bb), n10 = __popc(a & ~bb & (DN - 1)), n01 = __popc(~a & bb & (DN - 1)),
              n00 = N - n11 - n10 - n01;
    ddr[j] = wmix * (A.Sd[0].x * n00 + A.Sd[1].x * n01 + A.Sd[2].x * n10 + A.Sd[3].x * n11);
    const double dim = wmix * (A.Sd[0].y * n00 + A.Sd[1].y * n01 + A.Sd[2].y * n10 + A.Sd[3].y * n11);
    dbase[j] = dim - wmix * (active ? e0g[a] - e0g[bb] : 0.0);
  }

  for (int s = 0; s < A.n_steps; ++s) {
    const StepDesc sd = A.steps[s];
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
      double c1r = 0, c1i = 0, c2r = 0, c2i = 0, dlA = 0, dlB = 0;
      if (d.drive_series >= 0) {
        const cplx v1 = val(d.drive_series, sd.u1), v2 = val(d.drive_series, sd.u2);
        c1r = v1.x; c1i = v1.y; c2r = v2.x; c2i = v2.y;
      }
      auto add_det = [&](int ser, double scale) {
        const double d1 = val(ser, sd.u1).x, d2 = val(ser, sd.u2).x;
        dlA += scale * (A.a1 * d1 + A.a2 * d2);
        dlB += scale * (A.a2 * d1 + A.a1 * d2);
      };
      if (d.det_series >= 0) add_det(d.det_series, d.det_scale);
      if (d.off_series >= 0) add_det(d.off_series, d.off_scale);
      if (d.extra > 0 && A.dterms) {
        for (int e = d.extra - 1;; ++e) {  // few atoms, short lists: one lane per atom is enough
          const ryd_dterm t = A.dterms[e];
          add_det(t.series, t.scale);
          if (t.remaining == 0) break;
        }
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
      // per-bit coefficients (bit q <-> atom N-1-(q mod N)), wave-uniform -> SGPRs
      double cr[NB], ci[NB], dq[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int k = N - 1 - (q % N);
        // column bits carry +i (rho H), row bits -i (H rho): the flip coefficient is
        // (+-c_i, -c_r) on row bits and (+-c_i, +c_r) on column bits; the detuning enters
        // E(a) - E(b) with the opposite sign pattern (k_apply<RYD_MESOLVE>)
        const double s2 = q < N ? 1.0 : -1.0;
        cr[q] = uniform_d(-s2 * cf[4 * k + 0]);
        ci[q] = uniform_d(cf[4 * k + 1]);
        dq[q] = uniform_d(s2 * cf[4 * k + 2]);
      }
      double mq[REALU ? NB : 1];  // +1 row bit / -1 column bit of a driven atom, 0 otherwise
      double cuni = 0.0;
      if (REALU) {
        double cv = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) cv = cf[4 * k + 3] != 0.0 ? cf[4 * k + 0] : cv;
        cuni = uniform_d(cv);
#pragma unroll
        for (int q = 0; q < NB; ++q)
          mq[q] = uniform_d(cf[4 * (N - 1 - (q % N)) + 3] * (q < N ? -1.0 : 1.0));
      }
      double dgi[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int l = tid + j * NTT;
        double sdet = 0.0;  // detuning part of E(a) - E(b), signs as in k_apply<RYD_MESOLVE>
#pragma unroll
        for (int q = 0; q < NB; ++q)
          if (!((l >> q) & 1)) sdet += dq[q];
        dgi[j] = dbase[j] - sdet;
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
        const double sc = sd.h * kInvInt[jj];
        cplx acc[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int l = tid + j * NTT;
          cplx xv[NLDS > 0 ? NLDS : 1];
#pragma unroll
          for (int q = 0; q < NLDS; ++q) xv[q] = rd[(l ^ (1 << q)) & (D - 1)];
          cplx xd[DBL ? N : 1];
          if (DBL) {
#pragma unroll
            for (int k = 0; k < N; ++k) xd[k] = rd[(l ^ (1 << k) ^ (1 << (k + N))) & (D - 1)];
          }
          // diagonal: (dr + i di) x
          cplx a = make_double2(ddr[j] * w[j].x - dgi[j] * w[j].y, ddr[j] * w[j].y + dgi[j] * w[j].x);
          if (REALU) {
            double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0;  // two chains
#pragma unroll
            for (int q = 0; q < NB; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              if (q & 1) { s1x = fma(mq[q], x.x, s1x); s1y = fma(mq[q], x.y, s1y); }
              else { s0x = fma(mq[q], x.x, s0x); s0y = fma(mq[q], x.y, s0y); }
            }
            // -i c (S_row - S_col)
            a = make_double2(fma(cuni, s0y + s1y, a.x), fma(-cuni, s0x + s1x, a.y));
          } else {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
              const int rb = q >= LOGNT ? q - LOGNT : 0;
              const cplx x = q < NLDS ? xv[q < NLDS ? q : 0] : w[(j ^ (1 << rb)) & (R - 1)];
              // row bit: (sgi, -c_r); column bit: (sgi, +c_r); sgi = +-c_i by the output bit
              const double sgi = ((l >> q) & 1) ? ci[q] : -ci[q];
              a = cfma(make_double2(sgi, -cr[q]), x, a);
            }
          }
          if (DBL) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const int r = (((l >> (k + N)) & 1) << 1) | ((l >> k) & 1);
              const cplx jc = A.J[r];
              a = cfma(make_double2(jc.x * wmix, jc.y * wmix), xd[k], a);
            }
          }
          acc[j] = a;
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
          w[j] = make_double2(fma(sc, acc[j].x, psi[j].x), fma(sc, acc[j].y, psi[j].y));
        if (jj > 1) {
#pragma unroll
          for (int j = 0; j < R; ++j)
            if (active) wr[tid + j * NTT] = w[j];
          __syncthreads();
          const cplx* t = rd;
          rd = wr;
          wr = const_cast<cplx*>(t);
        }
      }
#pragma unroll
      for (int j = 0; j < R; ++j) psi[j] = w[j];
      __syncthreads();  // last-stage reads done before ws0 / cf are rewritten
    }
    if (sd.snap >= 0 && A.snaps && active) {
      cplx* o = A.snaps + ((size_t)sd.snap * A.B + b) * D;
#pragma unroll
      for (int j = 0; j < R; ++j) o[tid + j * NTT] = psi[j];
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < R; ++j) st[tid + j * NTT] = psi[j];
  }
}
