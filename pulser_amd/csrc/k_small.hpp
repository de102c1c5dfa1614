// Part of librydemu (included by rydemu.hip, one translation unit).
// coefs[b][k] = w1 * val(t1) + w2 * val(t2) for the drive (complex) and the
// detuning (real) of atom k of trajectory b.  pp: [n_series][n_int][4] complex.
// One wave per (b, k): lane 0 evaluates the three base series, all lanes share
// the list of extra detuning terms (hf noise), summed by a wave reduction.
__global__ __launch_bounds__(256) void k_eval_coefs(const cplx* __restrict__ pp, int n_int,
                                                    const ryd_qdesc* __restrict__ desc,
                                                    const ryd_dterm* __restrict__ dterms, int total,
                                                    int idx1, double u1, double w1, int idx2,
                                                    double u2, double w2,
                                                    double* __restrict__ coefs) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= total) return;
  const ryd_qdesc d = desc[i];
  auto val = [&](int s, int idx, double u) -> cplx {
    const cplx* p = pp + ((size_t)s * n_int + idx) * 4;
    cplx r = p[0];
    r = make_double2(fma(r.x, u, p[1].x), fma(r.y, u, p[1].y));
    r = make_double2(fma(r.x, u, p[2].x), fma(r.y, u, p[2].y));
    r = make_double2(fma(r.x, u, p[3].x), fma(r.y, u, p[3].y));
    return r;
  };
  double cr = 0, ci = 0, dl = 0;
  if (lane == 0) {
    if (d.drive_series >= 0) {
      const cplx a = val(d.drive_series, idx1, u1), b2 = val(d.drive_series, idx2, u2);
      cr = d.drive_scale * (w1 * a.x + w2 * b2.x);
      ci = d.drive_scale * (w1 * a.y + w2 * b2.y);
    }
    if (d.det_series >= 0)
      dl += d.det_scale * (w1 * val(d.det_series, idx1, u1).x + w2 * val(d.det_series, idx2, u2).x);
    if (d.off_series >= 0)
      dl += d.off_scale * (w1 * val(d.off_series, idx1, u1).x + w2 * val(d.off_series, idx2, u2).x);
  }
  if (d.extra > 0 && dterms) {  // high-frequency detuning noise on shared series
    const int count = dterms[d.extra - 1].remaining + 1;
    double x = 0.0;
    for (int e = lane; e < count; e += 64) {
      const ryd_dterm t = dterms[d.extra - 1 + e];
      x += t.scale * (w1 * val(t.series, idx1, u1).x + w2 * val(t.series, idx2, u2).x);
    }
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    dl += x;
  }
  if (lane == 0) {
    coefs[4 * (size_t)i + 0] = cr;
    coefs[4 * (size_t)i + 1] = ci;
    coefs[4 * (size_t)i + 2] = dl;
    coefs[4 * (size_t)i + 3] = 0.0;
  }
}

// E0[m][s] = sum_{i<j} U[m][i][j] n_i(s) n_j(s), n_k(s) = 1 - bit_{N-1-k}(s)
// (hamiltonian.py:260-274, 308-331; coefficient U/2 doubled by H + H^dagger).
__global__ void k_build_e0(const double* __restrict__ U, int N, double* __restrict__ e0) {
  const size_t D = (size_t)1 << N;
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= D) return;
  const int m = blockIdx.y;
  const double* u = U + (size_t)m * N * N;
  double e = 0.0;
  for (int i = 0; i < N; ++i) {
    if ((s >> (N - 1 - i)) & 1) continue;
    for (int j = i + 1; j < N; ++j)
      if (!((s >> (N - 1 - j)) & 1)) e += u[i * N + j];
  }
  e0[(size_t)m * D + s] = e;
}

// w[b][i'] = |psi_i|^2 (ket) or Re rho_ii (dm); i' = D-1-i when reverse.
__global__ void k_probabilities(const cplx* __restrict__ st, int N, int is_dm,
                                int reverse, double* __restrict__ w) {
  const size_t D = (size_t)1 << N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D) return;
  const int b = blockIdx.y;
  double p;
  if (is_dm) {
    p = st[((size_t)b << (2 * N)) + i * D + i].x;
  } else {
    const cplx v = st[((size_t)b << N) + i];
    p = v.x * v.x + v.y * v.y;
  }
  w[(size_t)b * D + (reverse ? D - 1 - i : i)] = p;
}

// out[b][k] += sum_i p_i n_k(i) (k < N), out[b][N] += sum_i p_i.
__global__ __launch_bounds__(256) void k_occupations(const cplx* __restrict__ st,
                                                     int N, int is_dm,
                                                     double* __restrict__ out) {
  const size_t D = (size_t)1 << N;
  const int b = blockIdx.y;
  double acc[RYD_MAX_QUBITS + 1];
#pragma unroll
  for (int k = 0; k <= RYD_MAX_QUBITS; ++k) acc[k] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < D;
       i += (size_t)gridDim.x * blockDim.x) {
    double p;
    if (is_dm) {
      p = st[((size_t)b << (2 * N)) + i * D + i].x;
    } else {
      const cplx v = st[((size_t)b << N) + i];
      p = v.x * v.x + v.y * v.y;
    }
#pragma unroll
    for (int k = 0; k < RYD_MAX_QUBITS; ++k)
      if (k < N && !((i >> (N - 1 - k)) & 1)) acc[k] += p;
    acc[RYD_MAX_QUBITS] += p;
  }
  __shared__ double red[4][RYD_MAX_QUBITS + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k <= RYD_MAX_QUBITS; ++k) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x <= N) {
    const int k = threadIdx.x == N ? RYD_MAX_QUBITS : threadIdx.x;
    const double v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    atomicAdd(&out[(size_t)b * (N + 1) + threadIdx.x], v);
  }
}

__global__ void k_ket_to_dm(const cplx* __restrict__ psi, int N, cplx* __restrict__ rho) {
  const size_t D = (size_t)1 << N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // a*D + b
  if (i >= D * D) return;
  const int bt = blockIdx.y;
  const cplx pa = psi[((size_t)bt << N) + (i >> N)];
  const cplx pb = psi[((size_t)bt << N) + (i & (D - 1))];
  rho[((size_t)bt << (2 * N)) + i] =
      make_double2(pa.x * pb.x + pa.y * pb.y, pa.y * pb.x - pa.x * pb.y);
}

// any state dimension D (3^N / 4^N for the multi-level bases): one thread per entry
__global__ void k_outer_acc(const cplx* __restrict__ psi, size_t D, int B,
                            const double* __restrict__ wts, cplx* __restrict__ acc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D * D) return;
  const size_t a = i / D, b = i - a * D;
  double sr = 0, si = 0;
  for (int t = 0; t < B; ++t) {
    const cplx pa = psi[(size_t)t * D + a], pb = psi[(size_t)t * D + b];
    const double w = wts ? wts[t] : 1.0;
    sr += w * (pa.x * pb.x + pa.y * pb.y);
    si += w * (pa.y * pb.x - pa.x * pb.y);
  }
  acc[i].x += sr;
  acc[i].y += si;
}

// ---------------------------------------------------------------------------
// k_outer_mfma: acc[a][b] += sum_t w_t psi_t[a] conj(psi_t[b]) on the fp64 matrix
// cores - the one GEMM-shaped operation of the path: the trajectory-averaged
// density matrix of density_matrix_aggregator (pulser_simulation/
// aggregators.py:20-37) and of qutip.mcsolve's averaged states.
//
// With X = Re psi, Y = Im psi (B x D):  Re = X^T X + Y^T Y,  Im = Y^T X - X^T Y,
// so two trajectories fill the K = 4 of one v_mfma_f64_16x16x4_f64:
//   A_re = (x, y, x', y'),  A_im = (y, -x, y', -x'),  B = (x, y, x', y').
// One workgroup (4 waves, 2 x 2) owns a 64 x 64 tile of the Hermitian result and
// only tiles on or above the diagonal are computed (the mirror is written from
// registers): half the flops of a ZGEMM.  KT trajectories of the two 64-wide
// column strips are staged per step in LDS (weights folded into the A strip);
// fragment reads are ds_read_b64, conflict-free (lanes 0-15 take the even
// doubles of a 256-B run, lanes 16-31 the odd ones).
// Operand / result lane maps of the f64 MFMA (MI355X guide): A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], C: col = l & 15, row = (l >> 4) + 4 * reg.
// ---------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

// Measured on MI355X (tools/ubench/mfma_f64.hip): one wave issues an f64 MFMA every
// ~59 ns whatever the number of independent accumulators, two waves per SIMD reach
// 47 TFLOP/s - the f64 matrix pipe needs >= 4 waves per SIMD to approach its
// 78.6 TFLOP/s, so the kernel is held to 128 registers (64 of them accumulators).
template <int KT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_outer_mfma(const cplx* __restrict__ psi, size_t D, int B,
                                                    const double* __restrict__ wts,
                                                    cplx* __restrict__ acc) {
  const int ta = blockIdx.y, tb = blockIdx.x;  // row / column tile
  if (tb < ta) return;                          // lower triangle: mirrored below
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sA = reinterpret_cast<double*>(smem);  // [KT][64][2] = w_t * psi_t[a0 + .]
  double* sB = sA + KT * 128;                    // [KT][64][2] =       psi_t[b0 + .]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;       // 32 x 32 sub-tile of this wave
  const int li = lane & 15, lk = lane >> 4;      // operand row/col, k slot
  const size_t a0 = (size_t)ta * 64, b0 = (size_t)tb * 64;

  mfma_d4 cre[2][2], cim[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      cre[r][c] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
      cim[r][c] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    }

  // software pipeline: the global loads of chunk c + 1 are in flight while chunk c
  // is multiplied; every thread stages PER = KT * 64 / 256 amplitudes of each strip
  constexpr int PER = KT * 64 / 256;
  cplx pa[PER], pb[PER];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int e = tid + it * 256, tt = e >> 6, col = e & 63, t = t0 + tt;
      pa[it] = make_double2(0.0, 0.0);
      pb[it] = pa[it];
      if (t < B) {
        const double w = wts ? wts[t] : 1.0;
        const cplx va = psi[(size_t)t * D + a0 + col];
        pb[it] = psi[(size_t)t * D + b0 + col];
        pa[it] = make_double2(w * va.x, w * va.y);
      }
    }
  };
  fetch(0);
  for (int t0 = 0; t0 < B; t0 += KT) {
    __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      const int e = tid + it * 256;
      *reinterpret_cast<cplx*>(sA + 2 * e) = pa[it];
      *reinterpret_cast<cplx*>(sB + 2 * e) = pb[it];
    }
    __syncthreads();
    if (t0 + KT < B) fetch(t0 + KT);
#pragma unroll 4
    for (int tt = 0; tt < KT; tt += 2) {
      // k slot lk: trajectory tt + (lk >> 1), component lk & 1
      const int trow = (tt + (lk >> 1)) * 128, comp = lk & 1;
      double are[2], aim[2], bb[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int idx = trow + 2 * (wr * 32 + r * 16 + li);
        are[r] = sA[idx + comp];
        const double o = sA[idx + (comp ^ 1)];
        aim[r] = comp ? -o : o;  // (y, -x)
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) bb[c] = sB[trow + 2 * (wc * 32 + c * 16 + li) + comp];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          cre[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(are[r], bb[c], cre[r][c], 0, 0, 0);
          cim[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[r], bb[c], cim[r][c], 0, 0, 0);
        }
    }
  }
  // accumulate into the result: this tile and, off the diagonal, its mirror
  // (conjugate transpose).  Eight old values are loaded before the eight stores of
  // a phase (tile and mirror never overlap), so the round trips overlap without
  // holding the whole tile twice in registers.
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int mirror = 0; mirror < 2; ++mirror) {
      if (mirror && ta == tb) continue;
      cplx old[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const size_t row = a0 + wr * 32 + r * 16 + (size_t)(lane >> 4) + 4 * g;
          const size_t col = b0 + wc * 32 + c * 16 + (size_t)(lane & 15);
          old[c][g] = mirror ? acc[col * D + row] : acc[row * D + col];
        }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const size_t row = a0 + wr * 32 + r * 16 + (size_t)(lane >> 4) + 4 * g;
          const size_t col = b0 + wc * 32 + c * 16 + (size_t)(lane & 15);
          const double vr = cre[r][c][g], vi = cim[r][c][g];
          if (mirror) acc[col * D + row] = make_double2(old[c][g].x + vr, old[c][g].y - vi);
          else acc[row * D + col] = make_double2(old[c][g].x + vr, old[c][g].y + vi);
        }
    }
}

// acc += w * x, elementwise (sum of the trajectories' density matrices, aggregators.py:20-37)
__global__ void k_axpy(const cplx* __restrict__ x, double w, size_t count, cplx* __restrict__ acc) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const cplx v = x[i];
    cplx a = acc[i];
    a.x += w * v.x;
    a.y += w * v.y;
    acc[i] = a;
  }
}

// Self-check of a solve (RYD_CHECK=1): out[b][0] += sum |x|^2 (kets) or Re trace (density matrices),
// out[b][1] += number of non-finite entries.  `row` = entries per state, `diag_stride` = D + 1 for a
// density matrix (0 for a ket).
__global__ __launch_bounds__(256) void k_selfcheck(const cplx* __restrict__ st, size_t row, size_t diag_stride,
                                                   double* __restrict__ out) {
  const int b = blockIdx.y;
  const cplx* x = st + (size_t)b * row;
  double s = 0.0, bad = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row; i += (size_t)gridDim.x * blockDim.x) {
    const cplx v = x[i];
    if (!(isfinite(v.x) && isfinite(v.y))) bad += 1.0;
    else if (diag_stride == 0) s += v.x * v.x + v.y * v.y;
    else if (i % diag_stride == 0) s += v.x;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_down(s, o, 64);
    bad += __shfl_down(bad, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out + 2 * b, s);
    if (bad != 0.0) atomicAdd(out + 2 * b + 1, bad);
  }
}
