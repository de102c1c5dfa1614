// Part of librydemu (included by rydemu.hip, one translation unit).
// ---------------------------------------------------------------------------
// Monte-Carlo wavefunction (quantum-jump) kernels - the work qutip.mcsolve does
// between and at the collapses (pulser-simulation/pulser_simulation/
// simulation.py:705-735: solver_fn = qutip.mcsolve, c_ops = one local operator
// per (spec, atom), hamiltonian.py:97-124).  The unnormalised ket evolves under
// H_eff = H - (i/2) sum C^dag C; when its squared norm has dropped below a
// uniform threshold, a collapse operator is drawn with weights ||C psi||^2,
// applied, and the ket renormalised.  Everything runs on the device without
// host synchronisation: one norm reduction per step, and - only for the
// trajectories that jump - the single-atom reduced density matrices, the
// selection and the 2x2 local update.  Random numbers: Philox4x32-10 keyed by a
// per-trajectory 64-bit seed, counter = jump index, so a trajectory's history
// does not depend on how the batch is split over launches or GPUs.
// ---------------------------------------------------------------------------
#define MC_MAX_OPS 16

__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// (threshold uniform, selection uniform) of jump number j: 53-bit doubles in [0, 1)
__device__ __forceinline__ void mc_uniforms(unsigned long long seed, unsigned j, double* ut,
                                            double* us) {
  unsigned c[4] = {j, 0u, 0u, 0u};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  *ut = ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
  *us = ((double)(c[2] >> 5) * 67108864.0 + (double)(c[3] >> 6)) * (1.0 / 9007199254740992.0);
}

struct McState {
  double* norm2;     // [2][B] squared norms (slot = step parity), zero between uses
  double* red;       // [B][N][4]: rho_rr, rho_gg, Re rho_rg, Im rho_rg of each atom
  double* target;    // [B] current threshold uniform
  double* refnorm;   // [B] squared norm right after the last jump (or at the start)
  double* lastnorm;  // [B] squared norm after the last completed step
  double* scale;     // [B] 1 / ||C psi|| of the selected collapse
  int* flag;         // [B] this step jumps
  int* sel;          // [B] atom * MC_MAX_OPS + op
  int* count;        // [B] jumps so far
  const unsigned long long* seeds;  // [B]
  const cplx* ops;   // [n_ops][4] local collapse operators, row-major (index 0 = r)
  int n_ops;
};

__device__ __forceinline__ double block_sum256(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void k_mc_norm(const cplx* __restrict__ st, int nb,
                                                 double* __restrict__ norm2) {
  __shared__ double sh[4];
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const cplx* __restrict__ x = st + ((size_t)b << nb);
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = x[i];
    s = fma(v.x, v.x, fma(v.y, v.y, s));
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) atomicAdd(&norm2[b], s);
}

// start of a Monte-Carlo solve: thresholds of jump 0, reference norms
__global__ void k_mc_init(McState M, int B, int N) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double ut, us;
  mc_uniforms(M.seeds[b], 0u, &ut, &us);
  M.target[b] = ut;
  M.refnorm[b] = M.norm2[b];  // slot 0 holds the initial squared norm
  M.lastnorm[b] = M.norm2[b];
  M.norm2[b] = 0.0;
  M.norm2[B + b] = 0.0;
  M.count[b] = 0;
  M.flag[b] = 0;
  for (int i = 0; i < 4 * N; ++i) M.red[(size_t)b * 4 * N + i] = 0.0;
}

// reduced single-atom density matrices of the trajectories that jump this step
__global__ __launch_bounds__(256) void k_mc_reduced(const cplx* __restrict__ st, int N,
                                                    McState M, int B, int slot) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  if (!(M.norm2[(size_t)slot * B + b] <= M.target[b] * M.refnorm[b])) return;  // block-uniform
  const size_t D = (size_t)1 << N;
  const cplx* __restrict__ x = st + ((size_t)b << N);
  for (int a = 0; a < N; ++a) {
    const size_t bit = (size_t)1 << (N - 1 - a);
    double rr = 0.0, gg = 0.0, cr = 0.0, ci = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
      const cplx v = x[i];
      const double m = v.x * v.x + v.y * v.y;
      if (i & bit) {
        gg += m;
      } else {
        const cplx w = x[i | bit];
        rr += m;
        cr += v.x * w.x + v.y * w.y;  // psi_r conj(psi_g)
        ci += v.y * w.x - v.x * w.y;
      }
    }
    rr = block_sum256(rr, sh);
    gg = block_sum256(gg, sh);
    cr = block_sum256(cr, sh);
    ci = block_sum256(ci, sh);
    if (threadIdx.x == 0) {
      double* r = M.red + ((size_t)b * N + a) * 4;
      atomicAdd(r + 0, rr);
      atomicAdd(r + 1, gg);
      atomicAdd(r + 2, cr);
      atomicAdd(r + 3, ci);
    }
  }
}

// ||C psi||^2 = Tr(C rho_atom C^dag) for a local 2x2 operator
__device__ __forceinline__ double mc_weight(const cplx* C, const double* r) {
  double p = 0.0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const cplx c0 = C[2 * i], c1 = C[2 * i + 1];
    // c0 conj(c1) rho_rg
    const double xr = c0.x * c1.x + c0.y * c1.y, xi = c0.y * c1.x - c0.x * c1.y;
    p += (c0.x * c0.x + c0.y * c0.y) * r[0] + (c1.x * c1.x + c1.y * c1.y) * r[1] +
         2.0 * (xr * r[2] - xi * r[3]);
  }
  return p;
}

__global__ void k_mc_select(McState M, int B, int N, int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double n2 = M.norm2[(size_t)slot * B + b];
  M.norm2[(size_t)(slot ^ 1) * B + b] = 0.0;  // the next step accumulates there
  M.norm2[(size_t)slot * B + b] = 0.0;
  int flag = 0;
  double last = n2;
  if (n2 <= M.target[b] * M.refnorm[b]) {
    double* red = M.red + (size_t)b * N * 4;
    double total = 0.0;
    for (int a = 0; a < N; ++a)
      for (int k = 0; k < M.n_ops; ++k) total += fmax(mc_weight(M.ops + 4 * k, red + 4 * a), 0.0);
    if (total > 0.0) {
      const unsigned j = (unsigned)M.count[b];
      double ut, us;
      mc_uniforms(M.seeds[b], j, &ut, &us);
      const double x = us * total;
      double cum = 0.0, psel = 0.0, plast = 0.0;
      int sel = -1, lastpos = -1;
      for (int a = 0; a < N; ++a)
        for (int k = 0; k < M.n_ops; ++k) {
          const double p = fmax(mc_weight(M.ops + 4 * k, red + 4 * a), 0.0);
          cum += p;
          if (p > 0.0) { lastpos = a * MC_MAX_OPS + k; plast = p; }
          if (sel < 0 && p > 0.0 && cum > x) { sel = a * MC_MAX_OPS + k; psel = p; }
        }
      if (sel < 0) { sel = lastpos; psel = plast; }  // rounding left x >= cum
      M.sel[b] = sel;
      M.scale[b] = 1.0 / sqrt(psel);
      M.count[b] = (int)j + 1;
      mc_uniforms(M.seeds[b], j + 1u, &ut, &us);
      M.target[b] = ut;
      M.refnorm[b] = 1.0;
      last = 1.0;
      flag = 1;
    }
    for (int i = 0; i < 4 * N; ++i) red[i] = 0.0;
  }
  M.flag[b] = flag;
  M.lastnorm[b] = last;
}

// psi <- C_k^(atom) psi / ||C psi|| for the flagged trajectories (in place, by pairs)
__global__ __launch_bounds__(256) void k_mc_jump(cplx* __restrict__ st, int N, McState M) {
  const int b = blockIdx.y;
  if (!M.flag[b]) return;
  const int sel = M.sel[b];
  const int p = N - 1 - sel / MC_MAX_OPS;
  const cplx* C = M.ops + 4 * (sel % MC_MAX_OPS);
  const cplx c00 = C[0], c01 = C[1], c10 = C[2], c11 = C[3];
  const double s = M.scale[b];
  const size_t half = (size_t)1 << (N - 1), bit = (size_t)1 << p;
  cplx* __restrict__ x = st + ((size_t)b << N);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (size_t)gridDim.x * 256) {
    const size_t l0 = ((i >> p) << (p + 1)) | (i & (bit - 1)), l1 = l0 | bit;
    const cplx v0 = x[l0], v1 = x[l1];
    const cplx o0 = cfma(c00, v0, cmul(c01, v1)), o1 = cfma(c10, v0, cmul(c11, v1));
    x[l0] = make_double2(s * o0.x, s * o0.y);
    x[l1] = make_double2(s * o1.x, s * o1.y);
  }
}

// dst = src / ||src|| using the norm recorded after the last step (dst may be src)
__global__ __launch_bounds__(256) void k_mc_normalize(const cplx* __restrict__ src,
                                                      cplx* __restrict__ dst, int nb,
                                                      const double* __restrict__ lastnorm) {
  const size_t D = (size_t)1 << nb;
  const int b = blockIdx.y;
  const double s = rsqrt(lastnorm[b]);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < D; i += (size_t)gridDim.x * 256) {
    const cplx v = src[((size_t)b << nb) + i];
    dst[((size_t)b << nb) + i] = make_double2(s * v.x, s * v.y);
  }
}
