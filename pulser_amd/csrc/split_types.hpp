// Part of librydemu: argument structures of the split-operator kernels (k_split.hpp, k_split_reg.hpp), shared by the
// translation units that compile them (rydemu.hip and the k_split_reg parts, rydemu_splitreg.hip).
#pragma once

#define SPLIT_TMAX 12
#define SPLIT_TS 13    /* largest tile of the static pass kernel k_split_s */
#define SPLIT_TBIG 14  /* table sizing of the generic pass kernel k_split_t (tiles of up to 2^14 amplitudes) */
#define SPLIT_NT 256
#define SPLIT_NMAX 32

struct SplitArgs {
  cplx* state;           // [B][2^N] in place
  const double* e0;      // [n_mats][2^N]
  long long e0_stride;   // 0 when shared by the batch
  const double* cfin;    // [B][N][4]: C, Re g, Im g, -    rotation to finish (previous stage)
  const double* ccur;    // [B][N][4]: C, Re g, Im g, Delta   this stage's rotation + detuning integral of D
  double wE;             // weight of E0 in D (us)
  Segs tile, outer;
  int N, T;
  unsigned fin_mask, cur_mask;  // tile-local bits to rotate before / after D
  int do_diag;
  // quantum-jump trajectories (H_eff = H - i/2 sum C^dag C, diagonal for every built-in channel): D also
  // carries the real factor exp(wE (dec_a + dec_b popc(index))) (template parameter DECAY of the pass kernels:
  // the plain passes carry neither the table nor the test)
  double dec_a, dec_b;
  // rows of a density matrix as kets (k_split_reg<.., ROWS>, the split-operator master equation of host_ket.hpp):
  // conj = 1 evolves with the complex-conjugate propagator (row <- row W^dagger); ftab = [2][4][16] elementwise factor
  // tables exp(f d(a, b)) by the counts n00, n01, n10, n11 of (row bit, column bit) pairs, [0] applied at the load
  // (use_pre), [1] at the store (use_post)
  int conj, use_pre, use_post;
  const double* ftab;
  int pend;  // k_split12<.., TAN>: a rotation precedes this pass's D (its cosine product is applied with D)
  // k_split_reg<.., SNAP>: evaluation-time snapshots taken INSIDE a closed run (round 5).  At the end of a sub-step s
  // with SplitRun.snap[s] >= 0 the kernel stores its registers - the OPEN state: the last D(a_{S+1}) of the sub-step is
  // fused into the next stage and the cosines of the last tan-form rotation ride on that stage too - to
  // snaps + snap[s] * snap_stride (+ the sequence's offset); k_split_snap_close then applies the closing phase and
  // the cosine product to every stored snapshot of the run at once, across the chip
  cplx* snaps;
  long long snap_stride;
  // k_split_reg (kets, not ROWS), round 6 - the step-size controller's check without copy / compare launches:
  //   dst      the final state is stored HERE instead of in place (the whole sub-step of a check runs from the state into
  //            the scratch buffer: no 2 x 16 B per amplitude copy before it);
  //   dst2     ... and ALSO here (the checkpoint a roll-back restores, written by the launch that produced the state);
  //   cmp      the final state is compared with this buffer on the way out: cmp_err[b] = max |x - cmp|^2 over the ket
  //            (what k_split_diff computed from two more reads of both buffers).
  cplx* dst;
  cplx* dst2;
  const cplx* cmp;
  double* cmp_err;
};

// One closed run of the composition: consecutive sub-steps (knot interval, start offset, length),
// by value in the kernel arguments.  Stage j = 6 s + i is D(a_i) R(b_i) of sub-step s (its D also
// carries the last D(a_7) of sub-step s - 1); stage 6 nsub only closes with D(a_7).
#define SPLIT_MAX_SUB 64
#define SPLIT_MAX_STAGES 10
struct SplitRun {
  int nsub;
  int S;  // stages of the composition: 6 (4th order, Blanes & Moan S6) or 10 (6th order, S10)
  int idx[SPLIT_MAX_SUB];
  double u0[SPLIT_MAX_SUB];
  double tau[SPLIT_MAX_SUB];
  double a[SPLIT_MAX_STAGES + 1];  // D(a_1) R(b_1) ... R(b_S) D(a_{S+1})
  double b[SPLIT_MAX_STAGES];
  int tan_form;  // 1: real drives (k_split14_loop, k_split_reg): the Re g slot (zero there) carries Im g / C;
                 // 2: complex drives on k_split_reg<.., CPLX>: Re g / C and Im g / C in their own slots
  // drive-only rotations exp(-i kick X(t_kick)) before the first stage (an extra stage 0 without D) / after the
  // last D (the closing stage's rotation): the commutator correction of the 4th-order operator splitting of the
  // master equation (host_ket.hpp); t_kick = knot interval kick_idx, offset kick_u
  double kick_pre, kick_post;
  int kick_idx;
  double kick_u;
  // complex drives on the REAL kernels (round 4): the rotation by c = |c| e^{i theta} is Z R(|c|) Z^+ with the diagonal
  // Z = exp(-i theta n), and Z commutes with every D - so stage j rotates by |c_j| and its D carries the extra per-atom
  // phase exp(i (theta_j - theta_{j-1}) n) (theta_0 = 0; the closing D returns to theta = 0): exact, for any per-atom,
  // time-dependent phase.  k_split_coefs adds theta_j - theta_{j-1} to the detuning integral.
  int gauge;
  // evaluation-time snapshot after sub-step s: slot (>= 0) or -1 (k_split_reg<.., SNAP> only; SplitArgs.snaps)
  int snap[SPLIT_MAX_SUB];
  // MIXED runs (round 5; k_split_coefs, k_split_reg, k_split_snap_close): sub-steps flagged alt[s] = 1 run the SECOND
  // composition (S2, a2, b2: the 4th-order 6-stage one for the one-knot sub-steps of a 6th-order run) inside the same
  // closed run - the one-knot steps next to a kink, or the knot an evaluation time cuts off a 9-knot step, used to close
  // the run on both sides (a launch, a load and a store of the ket, a closing stage each).  first[s] = index of the
  // first stage of sub-step s (first[nsub] = stages of the compositions; the closing D follows).
  int mixed;
  int S2;
  double a2[SPLIT_MAX_STAGES + 1];
  double b2[SPLIT_MAX_STAGES];
  unsigned char alt[SPLIT_MAX_SUB];
  short first[SPLIT_MAX_SUB + 1];
};

// the composition of sub-step s / the stage layout of a run (every kernel and the host go through these)
__host__ __device__ __forceinline__ int splitrun_S(const SplitRun& R, int s) { return (R.mixed && R.alt[s]) ? R.S2 : R.S; }
__host__ __device__ __forceinline__ double splitrun_a(const SplitRun& R, int s, int i) {
  return (R.mixed && R.alt[s]) ? R.a2[i] : R.a[i];
}
__host__ __device__ __forceinline__ double splitrun_b(const SplitRun& R, int s, int i) {
  return (R.mixed && R.alt[s]) ? R.b2[i] : R.b[i];
}
__host__ __device__ __forceinline__ int splitrun_first(const SplitRun& R, int s) { return R.mixed ? (int)R.first[s] : R.S * s; }
// stage j of the compositions (0 <= j < splitrun_first(R, nsub)) -> (sub-step, stage inside it)
__host__ __device__ __forceinline__ void splitrun_locate(const SplitRun& R, int j, int& s, int& st) {
  if (!R.mixed) {
    s = j / R.S;
    st = j % R.S;
    return;
  }
  int lo = 0, hi = R.nsub - 1;
  while (lo < hi) {  // the last sub-step whose first stage is <= j
    const int mid = (lo + hi + 1) >> 1;
    if ((int)R.first[mid] <= j) lo = mid; else hi = mid - 1;
  }
  s = lo;
  st = j - (int)R.first[lo];
}
// weight of E0 in the D of composition stage j (the last D of the previous sub-step rides on a sub-step's first stage);
// j == splitrun_first(R, nsub): the closing D
__host__ __device__ __forceinline__ double splitrun_weight(const SplitRun& R, int j) {
  const int total = splitrun_first(R, R.nsub);
  if (j >= total) return splitrun_a(R, R.nsub - 1, splitrun_S(R, R.nsub - 1)) * R.tau[R.nsub - 1];
  int s, st;
  splitrun_locate(R, j, s, st);
  double w = splitrun_a(R, s, st) * R.tau[s];
  if (st == 0 && s > 0) w += splitrun_a(R, s - 1, splitrun_S(R, s - 1)) * R.tau[s - 1];
  return w;
}

struct SplitSnapList {
  int n;
  int sub[SPLIT_MAX_SUB];
  int slot[SPLIT_MAX_SUB];
};
