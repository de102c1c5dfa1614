/*
 * rydemu.h - C ABI of librydemu.so, the MI355X (gfx950) emulation core.
 *
 * This is the native seam of the pulser_simulation classical-emulation path.
 * In the reference (pasqal-io/Pulser 1.10dev0, paths relative to the repo root)
 * the seam is the single call
 *
 *     solver_fn(hamiltonian._hamiltonian, self.initial_state,
 *               self._eval_times_array, c_ops=..., options=options)
 *     -- pulser-simulation/pulser_simulation/simulation.py:729-735
 *
 * whose inputs are built by Hamiltonian._construct_hamiltonian
 * (pulser-simulation/pulser_simulation/hamiltonian.py:246-439) and whose
 * outputs are `result.states` (simulation.py:739-748).  Every entry point
 * below names the reference interface it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *  - return 0 on success, negative ryd_status on error; ryd_last_error() gives
 *    a thread-local message.
 *  - `*_dev` pointers are DEVICE pointers owned by the caller (e.g.
 *    torch.Tensor.data_ptr()); host pointers are plain.  The library borrows
 *    them for the duration of the call.  `stream` is a hipStream_t (may be 0).
 *  - The handle owns its device tables (spline coefficients, interaction
 *    diagonal, per-stage coefficient buffers, Taylor work vectors).
 *  - One handle per device and stream; not thread-safe; re-entrant across
 *    handles.  Calls are asynchronous on `stream` unless documented.
 *  - State layout: complex128 (re, im interleaved).  ket: [batch][2^N], basis
 *    index = sum_k s_k 2^(N-1-k), s_k = 0 <=> atom k in |r>, 1 <=> |g>
 *    (docs/source/conventions.md:58-73).  Density matrix: row-major
 *    [batch][2^N][2^N], element (a, b) at a * 2^N + b.
 */
#ifndef RYDEMU_H
#define RYDEMU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RYD_ABI_VERSION 1
#define RYD_MAX_QUBITS 30

typedef enum ryd_status {
  RYD_OK = 0,
  RYD_ERR_INVALID = -1,     /* bad argument */
  RYD_ERR_HIP = -2,         /* HIP runtime error */
  RYD_ERR_UNSUPPORTED = -3, /* valid in the reference, not built yet */
  RYD_ERR_STATE = -4,       /* call order (tables not set) */
  RYD_ERR_NUMERIC = -5      /* self-check (environment RYD_CHECK=1): non-finite amplitudes, or the norm /
                               trace of a state moved by more than 1e-6 over a solve that conserves it */
} ryd_status;

typedef enum ryd_mode {
  RYD_SESOLVE = 0, /* i d/dt psi = H(t) psi          (qutip.sesolve) */
  RYD_MESOLVE = 1  /* d/dt rho = -i[H,rho] + D[rho]  (qutip.mesolve) */
} ryd_mode;

typedef struct ryd_handle ryd_handle;

typedef struct ryd_config {
  int32_t abi_version; /* RYD_ABI_VERSION */
  int32_t n_qubits;    /* N, 1..RYD_MAX_QUBITS (mesolve: 2N <= RYD_MAX_QUBITS) */
  int32_t batch;       /* B independent states evolved together (trajectories) */
  int32_t mode;        /* ryd_mode */
  int32_t device;      /* HIP device ordinal */
  int32_t tile_bits;   /* 0 = default (12); LDS tile = 2^tile_bits amplitudes */
  int32_t reserved[2];
} ryd_config;

/* Per (trajectory b, atom k) description of the time-dependent coefficients,
 * the factored form of the noisy samples of
 * HamiltonianData._sample_with_trajectory
 * (pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:408-534):
 *   c_k(t)     = drive_scale * S[drive_series](t)          (complex, = Omega/2 e^{-i phi})
 *   delta_k(t) = det_scale * Re S[det_series](t) + off_scale * Re S[off_series](t)
 *                + sum over the extra detuning terms of  scale * Re S[series](t)
 * series index -1 = absent (coefficient 0).  `extra` = 1-based index of the
 * first extra detuning term of this (trajectory, atom) in the table given to
 * ryd_set_detuning_terms (0 = none); its terms are contiguous and each carries
 * the number of terms that still follow it (`remaining`, 0 on the last), so the
 * first one gives the length of the list and a wave can evaluate it in parallel.  The extra terms carry the high-frequency detuning noise
 * sum_f A_f cos(w_f t + phi_f) of _generate_detuning_fluctuations
 * (hamiltonian_data.py:132-169) as  A_f cos(phi_f) * [m cos(w_f t)] -
 * A_f sin(phi_f) * [m sin(w_f t)]  on shared series (m = the slot mask). */
typedef struct ryd_qdesc {
  int32_t drive_series;
  int32_t det_series;
  int32_t off_series;
  int32_t extra;
  double drive_scale;
  double det_scale;
  double off_scale;
} ryd_qdesc;

typedef struct ryd_dterm {
  int32_t series;
  int32_t remaining; /* terms of this (trajectory, atom) after this one */
  double scale;
} ryd_dterm;

typedef struct ryd_opts {
  int32_t taylor_order; /* 0 = choose from norm bound and `tol` */
  int32_t max_order;    /* cap for the automatic choice (default and maximum 32) */
  double tol;           /* per-exponential truncation bound (default 1e-10: ends <= 2e-8 from the
                           tight oracle after a 3.1 us sequence, the stated bar being 1e-7) */
  double max_step;      /* us; 0 = no cap (steps never straddle a spline knot) */
  double magnus_tol;    /* per-interval Magnus error target driving the automatic
                           sub-stepping next to waveform kinks (default 1e-10); the allowed
                           Magnus-error estimate of steps that span several spline knots (taken where
                           every waveform is the same polynomial across them; 4e-9 per us of such
                           steps at the default) scales with it */
  int32_t split_steps;  /* mesolve split-operator path: CF4 steps per Strang block (0 = from the
                           dissipator rate: 4 / 2 / 1 for rates <= 0.1 / <= 0.5 / above) */
  int32_t method;       /* propagator of the multi-launch sesolve path: 0 = the library's choice (the
                           split-operator passes for two-level kets without collapse operators from 15
                           atoms on and for fewer than 8 sequences of 14 atoms; for batches of >= 8 real-drive
                           14-atom sequences the register-resident split-operator kernel when multi-knot
                           steps cover half of the call's schedule, else the register-resident polynomial
                           kernel; else the Taylor polynomial), 1 = Lanczos / Krylov subspace (batched inner products V^H w and
                           V c), 2 = split-operator (exact diagonal phases x exact single-atom rotations,
                           symmetric composition: 6th order / 10 stages over sub-steps of up to 8 knot
                           intervals where the call's schedule has such steps, else 4th order / 6 stages
                           inside one knot interval; `tol` x 500 = target of the accumulated
                           local-error estimate of a whole pulse sequence - the sum of the 2-NORMS of the
                           local errors, which bounds every amplitude error - default 8e-8),
                           3 = Taylor polynomial (Horner) */
  double reserved[2];
} ryd_opts;

typedef struct ryd_stats {
  int64_t n_applications; /* generator applications G.x (split-operator path: stages) since creation/reset */
  int64_t n_launches;     /* kernel launches of the apply kernel */
  int64_t n_steps;        /* CF4 steps */
  int32_t passes;         /* memory passes per application */
  int32_t last_order;     /* Taylor order used by the last step */
  double norm_bound;      /* last spectral-norm bound (rad/us) */
  double reserved[4];     /* split-operator path: [0] accumulated local-error estimate of the last
                             solve (sum of local 2-norms: a bound on every amplitude error), [1] last
                             measured local error, [2] its sub-step (us), [3] checkpoint restores */
} ryd_stats;

/* Replaces: construction of Hamiltonian/QobjEvo objects
 * (hamiltonian.py:45-81).  */
int ryd_create(const ryd_config* cfg, ryd_handle** out);
void ryd_destroy(ryd_handle* h);

/* Replaces: qutip.QobjEvo array coefficients with tlist
 * (hamiltonian.py:436; cubic not-a-knot spline).  `tknots` float64[n_knots]
 * (us, strictly increasing); `pp` float64[n_series][n_knots-1][4][2]: complex
 * polynomial coefficients, highest power first, in the local variable
 * (t - tknots[i]) (scipy PPoly convention).  Host pointers; copied. */
int ryd_set_series(ryd_handle* h, int32_t n_series, int32_t n_knots,
                   const double* tknots, const double* pp);

/* Replaces: the per-qubit / global [operator, coefficient] term list of
 * build_coeffs_ops (hamiltonian.py:333-389).  desc[batch][N], host pointer. */
int ryd_set_qubit_desc(ryd_handle* h, const ryd_qdesc* desc);

/* Replaces: the per-trajectory high-frequency detuning-noise synthesis of
 * HamiltonianData._generate_detuning_fluctuations
 * (pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:132-169, added to
 * the samples at :464-468).  terms[n_terms] (host pointer; copied) is the table
 * that ryd_qdesc.extra indexes (1-based); n_terms = 0 removes it.  May be called
 * before or after ryd_set_qubit_desc; every `extra` is checked against the table
 * when a solve / apply starts (RYD_ERR_INVALID if it points outside). */
int ryd_set_detuning_terms(ryd_handle* h, int32_t n_terms, const ryd_dterm* terms);

/* Replaces: make_vdw_term / make_interaction_term (hamiltonian.py:260-274,
 * 296-331).  U float64[n_mats][N][N] symmetric (rows/cols of bad atoms zeroed
 * by the caller); n_mats is 1 (shared by the batch) or `batch`.  Builds the
 * static diagonal E0[s] = sum_{i<j} U_ij n_i(s) n_j(s) on the device. */
int ryd_set_interaction(ryd_handle* h, const double* U, int32_t n_mats);

/* Replaces: Hamiltonian._build_collapse_operators (hamiltonian.py:97-124) and
 * the c_ops argument of qutip.mesolve (simulation.py:723-727).  The same local
 * 2x2 collapse operators act on every atom, so the dissipator is given as the
 * 4x4 local superoperator on the digit pair (a_k, b_k), row-major
 * S[(2a+b)][(2a'+b')], complex128 interleaved: float64[4][4][2] (host).  Only
 * the diagonal and the double-flip entries ((a,b)<-(1-a,1-b)) may be non-zero
 * in this ABI version (dephasing, relaxation, depolarizing, diagonal/flip
 * eff_noise); anything else returns RYD_ERR_UNSUPPORTED. */
int ryd_set_dissipator(ryd_handle* h, const double* S);

/* Replaces: one solver call qutip.sesolve/mesolve(H, state, [t0, t1])
 * (simulation.py:729-735): advances `state_dev` in place from t0 to t1 (us)
 * with the commutator-free 4th-order Magnus / Taylor stepper. */
int ryd_evolve(ryd_handle* h, void* state_dev, double t0, double t1,
               const ryd_opts* opts, void* stream);

/* Replaces: the whole solver call qutip.sesolve/mesolve(H, state, tlist)
 * (simulation.py:729-735, result.states at every evaluation time).  Advances
 * `state_dev` in place through times[0..n_times-1] (us, non-decreasing) and, if
 * `out_dev` is not NULL, stores the state reached at times[i] (i >= 1) in slot
 * i-1 of out_dev, complex128[n_times-1][batch][dim].  For sesolve with N <= 13
 * (and for mesolve with N <= 6: a density matrix of at most 4096 entries) the
 * whole call is ONE launch of the persistent LDS-resident trajectory kernel
 * (one workgroup per batch entry); batches of >= 8 kets of 13-14 atoms with real
 * drive coefficients run in ONE launch of the register-resident kernel (in-place
 * symplectic exponential); a master equation of 10-14 atoms with a dephasing-type
 * dissipator runs as 4th-order operator-splitting blocks (two row passes of that
 * kernel + one conjugate transposition per half block); otherwise the tiled
 * multi-pass kernels run once per Taylor stage. */
int ryd_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
              void* out_dev, const ryd_opts* opts, void* stream);

/* Replaces: the c_ops argument of qutip.mcsolve (simulation.py:705-727: with
 * collapse operators and stochastic noise Solver.DEFAULT picks qutip.mcsolve;
 * Solver.MCSOLVER forces it) - the local 2x2 collapse operators that
 * Hamiltonian._build_collapse_operators (hamiltonian.py:97-124) places on every
 * atom.  ops: complex128[n_ops][2][2] row-major, interleaved (host), local index
 * 0 = r (or h), 1 = g; n_ops <= 16.  Only for RYD_SESOLVE handles.  From then on
 * the handle evolves kets under H_eff = H - (i/2) sum_atoms sum_k C_k^dag C_k
 * (ryd_solve / ryd_evolve: the no-jump evolution, norm not conserved), and
 * ryd_mc_solve runs quantum-jump trajectories.  sum_k C_k^dag C_k must be
 * diagonal (true for dephasing, relaxation, depolarizing and diagonal / ladder
 * eff_noise operators); otherwise RYD_ERR_UNSUPPORTED.  n_ops = 0 removes them. */
int ryd_set_collapse(ryd_handle* h, int32_t n_ops, const double* ops);

/* Replaces: qutip.mcsolve(H, psi0, tlist, c_ops, ntraj=1) for every batch entry
 * (simulation.py:729-735 with solver_fn = qutip.mcsolve).  Same arguments as
 * ryd_solve plus one 64-bit seed per batch entry (host, uint64[batch]).  Each
 * batch entry is ONE Monte-Carlo wavefunction trajectory: the ket evolves under
 * H_eff until its squared norm falls below a uniform threshold; then one local
 * collapse (atom, operator) is drawn with weights ||C psi||^2, applied, and the
 * ket renormalised.  Jumps are taken at the end of the integrator step in which
 * the threshold is crossed (steps never exceed one sample interval, see
 * ryd_opts.max_step).  Stored states and the final state are normalised.
 * Random numbers: Philox4x32-10, key = seed, counter = jump index; a trajectory
 * is a function of (seed, Hamiltonian) only.  qutip.mcsolve seeds its own
 * generators, so parity with the reference is statistical: the trajectory
 * average converges to the qutip.mesolve state. */
int ryd_mc_solve(ryd_handle* h, void* state_dev, int32_t n_times, const double* times,
                 void* out_dev, const uint64_t* seeds, const ryd_opts* opts, void* stream);

/* Number of collapses of every batch entry in the last ryd_mc_solve
 * (McResult.col_times lengths); synchronises `stream`.  counts: int32[batch] (host). */
int ryd_mc_get_jumps(ryd_handle* h, int32_t* counts, void* stream);

/* General path for everything the tuned 2-level Ising kernels do not cover: the
 * 3-level "all" basis, leakage (d = 3/4), XY mode with its SLM-mask switching
 * terms, arbitrary eff_noise collapse operators.  The generator is given as an
 * explicit term list G(t) = sum_t coef_t(t) A_t with CSR matrices A_t over the
 * evolved vector (psi, or row-major vec(rho) for the Liouvillian):
 *   coef_t(t) = scale_t * S[series_t](t)   (conjugated when conj_t != 0), or the
 *   constant scale_t when series_t == -1.
 * Replaces the same reference interfaces as the tuned path (QobjEvo term list,
 * hamiltonian.py:246-439; liouvillian built by qutip.mesolve from c_ops).
 * Use ryd_set_series for the spline tables, then ryd_solve / ryd_evolve /
 * ryd_apply_generator as usual; states are complex128[batch][dim]. */
typedef struct ryd_general_config {
  int32_t abi_version;
  int32_t batch;
  int32_t device;
  int32_t reserved;
  int64_t dim;
} ryd_general_config;

int ryd_general_create(const ryd_general_config* cfg, ryd_handle** out);
int ryd_general_add_term(ryd_handle* h, int64_t nnz, const int32_t* row_ptr, const int32_t* col,
                         const double* val /* complex128[nnz] */, int32_t series, int32_t conj,
                         double scale_re, double scale_im, double row_norm);

/* Matrix-free terms of the general path (no operator is materialised): every operator of the
 * reference's term and collapse lists (hamiltonian.py:97-124, 246-439) is a sum of one- and two-site
 * operators or a diagonal.
 *   local:    A = sum_g weights[g] * embed(M on the digits with strides[g][0 .. n_per)),
 *             digit = (index / stride) % local_dim; M is (local_dim^n_per)^2 (row-major digit order for
 *             n_per = 2), given by its nnz non-zeros (rows, cols, complex vals).
 *   diagonal: A = diag(values), complex128[dim]  (Ising interaction, detuning projectors).
 * series / conj / scale / row_norm as for ryd_general_add_term (row_norm = a bound of the largest
 * absolute row sum of A). */
int ryd_general_add_local_term(ryd_handle* h, int32_t local_dim, int32_t n_per, int32_t n_groups,
                               const int64_t* strides, const double* weights, int32_t nnz,
                               const int32_t* rows, const int32_t* cols, const double* vals,
                               int32_t series, int32_t conj, double scale_re, double scale_im,
                               double row_norm);
int ryd_general_add_diag_term(ryd_handle* h, const double* values, int32_t series, int32_t conj,
                              double scale_re, double scale_im, double row_norm);

/* Replaces: the serial loop of QutipEmulator._noisy_runs (pulser_simulation/simulation.py:903-915) over the
 * noise trajectories of a MULTI-LEVEL / XY run (bases of pulser/_hamiltonian_data/hamiltonian_data.py:913-931):
 * every trajectory is its own general handle (bad atoms, detuning offsets and amplitude factors change the term
 * list), all of them advance through `times` in ONE launch - one workgroup per problem.  Handles: general,
 * batch 1, at most 4096 entries, same device, tables set; times strictly increasing.  states_dev[b]
 * complex128[dim_b] in place; outs_dev (or NULL) [b] -> complex128[n_times - 1][dim_b] or NULL. */
int ryd_general_solve_many(ryd_handle** hs, int32_t n, void* const* states_dev, int32_t n_times,
                           const double* times, void* const* outs_dev, const ryd_opts* opts,
                           void* stream);

/* Test/bench hook (bit mask): 1 = disable the persistent small-N kernel, 2 =
 * disable the single-launch plan of small states (partner tiles read through
 * L2; the multi-pass tiling is used instead), 4 = disable the 2^14
 * register-tile kernel and the Hermitian mesolve path, 8 = force them even when
 * the launch has too few tiles to fill the GPU, 16 = use the single-launch plan
 * whatever the size of the state, 32 = disable the register-resident ket kernel
 * (sesolve, 13-14 atoms, >= 8 sequences, real drives) and the split-operator master
 * equation built on it (mesolve, 10-14 atoms, dephasing-type dissipators), 64 = use
 * the ket kernel from 10 atoms and any batch size on, 128 = keep the Taylor
 * polynomial where ryd_opts.method 0 would choose the split-operator ket passes,
 * 256 = switch their step-size control off (one sub-step per schedule step),
 * 512 = 12- to 14-atom kets pass by pass instead of the one-launch loops over the stages
 * (k_split_reg; k_split14_loop),
 * 1024 = keep every CF4 step inside one knot interval (no multi-knot steps),
 * 2048 = 2^12-amplitude tiles of the split-operator passes where 2^13 ones are the default (21 - 23 atoms),
 * 4096 = general path: term-by-term kernel instead of the site-fused one,
 * 8192 = split-operator passes: the 4th-order 6-stage scheme with sub-steps that end at every knot,
 * 16384 = sequences of 12 - 14 atoms stay on the polynomial kernels (k_traj, k_ket) where ryd_solve would choose the
 * register-resident split-operator kernel (k_split_reg),
 * 32768 = one-launch split-operator runs of 14-atom kets on the round-3 kernel (k_split14_loop: two LDS turns per
 * stage) instead of k_split_reg,
 * 65536 = split-operator master equation: the row passes on the polynomial kernel (k_ket) instead of k_split_reg,
 * 131072 = every evaluation time closes a run of k_split_reg (round 4) instead of a snapshot stored from the registers
 * inside the run (k_split_reg<.., SNAP> + k_split_snap_close),
 * 262144 = general path: the round-3 site kernel (k_gen_apply_sites) instead of the padded site tables of
 * k_gen_apply_fused (round 6).
 * Never needed for results. */
int ryd_set_path(ryd_handle* h, int32_t force_generic);

/* Replaces: QobjEvo.__call__(t) applied to a state (used by
 * qutip_backend.py:259-264 and QutipOperator.apply_to, qutip_op.py:85-100).
 * out = G(t) in, G = -iH (sesolve layout) or the Lindbladian (mesolve). */
int ryd_apply_generator(ryd_handle* h, const void* in_dev, void* out_dev,
                        double t, void* stream);

/* Replaces: QutipResult._weights before normalisation
 * (pulser_simulation/qutip_result.py:101-118): w[b][i] = |psi_i|^2 (or
 * Re rho_ii), index-reversed when `reverse` != 0.  w_dev float64[batch][2^N]. */
int ryd_probabilities(ryd_handle* h, const void* state_dev, double* w_dev,
                      int32_t reverse, void* stream);

/* Replaces: expectation of the number operators n_k = |r><r|_k
 * (default_observables.py Occupation; qutip.expect, simresults.py:132).
 * out_dev float64[batch][N + 1]: <n_k> for k < N, then the squared norm /
 * trace in slot N. */
int ryd_occupations(ryd_handle* h, const void* state_dev, double* out_dev,
                    void* stream);

/* Replaces: the expectation values behind the default observables of the V2 backend for ONE
 * state - Occupation / CorrelationMatrix through the number operators n_k = |r><r|_k
 * (pulser-core/pulser/backend/default_observables.py:291-435) and Energy / EnergySecondMoment /
 * EnergyVariance through H(t) (:437-580; qutip_backend.py:259-264 materialises H(t) for them) -
 * in one call, on the device.  `what` = RYD_OBS_* bits.  out_dev float64[batch][N*N + N + 3]:
 *   [0, N)            <n_k>
 *   [N]               squared norm / trace  (the caller normalises, like state.unit())
 *   [N+1, N+1+N*N)    <n_k n_l>, row-major
 *   [N*N+N+1]         <H(t)>      (RYD_OBS_ENERGY; kets: one generator application + one dot;
 *   [N*N+N+2]         <H(t)^2>     density matrices: a gather of the elements of rho within two
 *                                  bit flips of the diagonal - H(t) is never materialised)
 * RYD_OBS_DENSITY: state_dev is a density matrix complex128[batch][2^N][2^N] even though the
 * handle is a ket (sesolve) handle - the V2 backend's noiseless-Hamiltonian engine observing the
 * states of a master-equation run.  Entries that were not requested are 0.  No host synchronisation. */
#define RYD_OBS_OCCUPATION 1
#define RYD_OBS_CORRELATION 2
#define RYD_OBS_ENERGY 4
#define RYD_OBS_DENSITY 8
int ryd_observe(ryd_handle* h, const void* state_dev, double t, int32_t what,
                double* out_dev, void* stream);

/* Replaces: building rho0 = |psi><psi| inside qutip.mesolve for a ket input.
 * psi_dev complex128[batch][2^N] -> rho_dev complex128[batch][2^N][2^N]. */
int ryd_ket_to_dm(ryd_handle* h, const void* psi_dev, void* rho_dev,
                  void* stream);

/* Replaces: density_matrix_aggregator's sum of |psi><psi| over trajectories
 * (pulser_simulation/aggregators.py:20-37): acc += sum_b w_b |psi_b><psi_b|.
 * acc_dev complex128[2^N][2^N]; weights host float64[batch] or NULL (=1). */
int ryd_outer_accumulate(ryd_handle* h, const void* psi_dev,
                         const double* weights, void* acc_dev, void* stream);

/* The same without a handle and for any state dimension `dim` (d^N for the 3- / 4-level bases,
 * pulser/_hamiltonian_data/hamiltonian_data.py:913-931): what density_matrix_aggregator
 * (pulser_simulation/aggregators.py:20-37) and the mean over trajectory Results
 * (pulser_simulation/qutip_backend.py:322-325) need - they hold states, not a solver.
 * psi_dev complex128[batch][dim], acc_dev complex128[dim][dim], weights host float64[batch] or NULL. */
int ryd_outer_accumulate_dim(const void* psi_dev, int64_t batch, int64_t dim,
                             const double* weights, void* acc_dev, int32_t device,
                             void* stream);

/* Replaces: the running sum of density matrices in density_matrix_aggregator when the trajectory
 * states are already density matrices (aggregators.py:29-35): acc += weight * x, complex128[count]. */
int ryd_accumulate(const void* x_dev, double weight, int64_t count, void* acc_dev,
                   int32_t device, void* stream);

int ryd_get_stats(const ryd_handle* h, ryd_stats* out);
int ryd_reset_stats(ryd_handle* h);

/* Live timing of the dominant kernel with HIP events on the launch stream
 * (bench.py roofline leg): enable, run, then read accumulated milliseconds and
 * launch count of the apply kernel.  Enabling inserts two events per launch. */
int ryd_set_kernel_timing(ryd_handle* h, int32_t enable);
int ryd_get_kernel_timing(ryd_handle* h, double* total_ms, int64_t* launches);

const char* ryd_last_error(void);
/* Replaces (HOST arithmetic, no device work): the sampling of every trajectory state at every evaluation time inside
 * QutipEmulator.run (pulser_simulation/simulation.py:853-861) - QutipResult._weights (pulser_simulation/qutip_result.py:
 * 101-158: |psi|^2 or |diag rho|, reversed for the ground-rydberg measurement basis; `matching` = 0: weights = delta_0,
 * :119-122), multinomial's searchsorted over the cumulative sums (pulser-core/pulser/math/multinomial.py:32-36) and the
 * SPAM measurement flips (pulser_simulation/simresults.py:537-568) - for a block of rows at once, with the uniforms the
 * caller drew in the reference's order.  Same IEEE operations in the same order as NumPy's: the histograms are
 * identical to the Python replay's.
 *   states_host  complex128[n_rows][dim] (kets, is_ket = 1) or the diagonals of density matrices (is_ket = 0), host memory
 *   start/count  int64[n_rows]: where row r's uniforms sit in `rnd` (and its rows in flip_matrix[total][n_qubits], or NULL)
 *   slot         int32[n_rows]: the histogram (evaluation time) row r adds to; hist int64[n_slots][dim], accumulated
 *   n_threads    host threads (0 = min(16, half the cores)); rows are independent, integer histograms merge exactly */
int ryd_replay_samples(const void* states_host, int64_t n_rows, int64_t dim, int32_t n_qubits, int32_t is_ket,
                       int32_t reversed, int32_t matching, const int64_t* start, const int64_t* count,
                       const int32_t* slot, int32_t n_slots, const double* rnd, const double* flip_matrix,
                       double eps, double eps_p, int64_t* hist, int32_t n_threads);

int ryd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RYDEMU_H */
