/*
 * mjhmc_hip.h -- C ABI of libmjhmc_hip.so, the MI355X (gfx950) engine for the particle-parallel
 * Markov-Jump-HMC hot path of rueberger/MJHMC.
 *
 * The reference has no FFI: its hot path is NumPy called from Python classes.  Each entry point
 * below replaces the reference interface cited next to it (paths relative to the reference
 * root); a reference maintainer binds them with ctypes.
 *
 * Conventions
 *   - C linkage, plain pointers and sizes.  Return 0 on success, a negative mjhmc_status
 *     otherwise; the message is available from mjhmc_last_error().  Nothing throws across the ABI.
 *   - Every pointer argument is caller-owned HOST memory, valid for the duration of the call.
 *     Matrices are C-ordered (ndims, nparticles) float64 exactly like the reference's arrays
 *     (mjhmc/samplers/markov_jump_hmc.py:29); the engine re-tiles them on the device.
 *   - Device memory is owned by the handles.  One HIP stream per sampler.  A handle is not
 *     thread-safe; distinct handles are independent.  No process-global mutable state except
 *     the last-error string of failed *_create calls (thread-local).
 */
#ifndef MJHMC_HIP_H
#define MJHMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJHMC_ABI_VERSION 2

typedef struct mjhmc_ctx mjhmc_ctx;         /* one per process x device                          */
typedef struct mjhmc_energy mjhmc_energy;   /* an energy model, parameters resident in HBM       */
typedef struct mjhmc_sampler mjhmc_sampler; /* particle state (HMCState) + jump-process machinery */
typedef struct mjhmc_comm mjhmc_comm;       /* this rank's end of an RCCL communicator (one process per GPU)     */

typedef enum {
  MJHMC_OK = 0,
  MJHMC_ERR_INVALID = -1,      /* bad argument                                                   */
  MJHMC_ERR_HIP = -2,          /* a HIP runtime call failed                                      */
  MJHMC_ERR_UNSUPPORTED = -3,  /* a dtype that names no arithmetic of the energy (bf16 outside   */
                               /* SPARSE_CODE), a SPARSE_CODE dictionary shape the reference     */
                               /* rejects too; every ndims of the other energies runs            */
  MJHMC_ERR_NO_DEVICE = -4,
  MJHMC_ERR_NONFINITE = -5,    /* informational: see mjhmc_iterate                               */
  MJHMC_ERR_COMM = -6          /* librccl could not be loaded, or an RCCL call failed            */
} mjhmc_status;

/* Energy models.  params are float64; layout per kind:
 *   ISO_GAUSS   {sigma}                         TestGaussian  mjhmc/misc/distributions.py:348-362, README.md:18-24
 *   DIAG_GAUSS  {j_0..j_{D-1}} diagonal of J    Gaussian      mjhmc/misc/distributions.py:256-273
 *   ROUGH_WELL  {scale1, scale2}                RoughWell     mjhmc/misc/distributions.py:283-304
 *   MM_GAUSS    {separation}                    MultimodalGaussian mjhmc/misc/distributions.py:314-335
 *   FUNNEL_NEAL {scale}                         Funnel as documented, mjhmc/misc/tf_distributions.py:142-147
 *   FUNNEL_REF  {scale}                         Funnel as coded,      mjhmc/misc/tf_distributions.py:157-165
 *   PRODUCT_OF_T {nbasis, W[D*nbasis] row-major (D,nbasis), nu[nbasis], b[nbasis]}
 *                                               ProductOfT    mjhmc/misc/distributions.py:373-433
 *                                               (nbasis == ndims, any size: the tile kernels up to 512, 512 x 512
 *                                               blocks on the multi-pass path beyond; dtype F64 = the reference's
 *                                               arithmetic, float64 state around the float32 force; F32 = float32 state)
 *   SPARSE_CODE {n_patches, img, n_coeffs, lambda, cauchy, B[img*n_coeffs], Y[n_patches*img]}
 *                                               SparseImageCode mjhmc/misc/tf_distributions.py:204-272
 *                                               (img = 256; n_coeffs = 1024 or 512, :219; 1 <= n_patches <= 32)
 */
typedef enum {
  MJHMC_E_ISO_GAUSS = 0,
  MJHMC_E_DIAG_GAUSS = 1,
  MJHMC_E_ROUGH_WELL = 2,
  MJHMC_E_MM_GAUSS = 3,
  MJHMC_E_FUNNEL_NEAL = 4,
  MJHMC_E_FUNNEL_REF = 5,
  MJHMC_E_PRODUCT_OF_T = 6,
  MJHMC_E_SPARSE_CODE = 7,
  MJHMC_E_USER_EXPR = 8,     /* created by mjhmc_energy_create_expr only */
  MJHMC_E_HOST = 9           /* no parameters: E and dE/dX are evaluated by the CALLER (opaque Python callables of
                                LambdaDistribution, README.md:27-36); samplers of it are driven by mjhmc_traj_* below */
} mjhmc_energy_kind;

/* arithmetic type of state and force.  BF16: bfloat16 state in HBM and as MFMA operands, float32
 * accumulation and integrator registers (SPARSE_CODE only). */
typedef enum { MJHMC_F64 = 0, MJHMC_F32 = 1, MJHMC_BF16 = 2 } mjhmc_dtype;

/* Sampler families (mjhmc/samplers/markov_jump_hmc.py). */
typedef enum {
  MJHMC_MODE_MJHMC = 0,   /* MarkovJumpHMC.sampling_iteration      :355-415 */
  MJHMC_MODE_CONTROL = 1, /* HMCBase / HMC / ControlHMC            :116-148 */
  MJHMC_MODE_CTHMC = 2    /* ContinuousTimeHMC.sampling_iteration  :251-290 */
} mjhmc_mode;

/* State fields (mjhmc/samplers/hmc_state.py:13-44) readable / writable through mjhmc_read/write.
 * Matrix fields are (ndims, nparticles) float64 C-order on the host side. */
typedef enum {
  MJHMC_F_X = 0,      /* HMCState.X                                  float64 (D,N)  rw */
  MJHMC_F_V = 1,      /* HMCState.V                                  float64 (D,N)  rw */
  MJHMC_F_EX = 2,     /* HMCState.EX                                 float64 (N)    r  */
  MJHMC_F_EV = 3,     /* HMCState.EV                                 float64 (N)    r  */
  MJHMC_F_DEDX = 4,   /* HMCState.dEdX (recomputed from X on demand) float64 (D,N)  r  */
  MJHMC_F_HFLF = 5,   /* H() of HMCState.cached_flf_state, NaN where the cache is cold  float64 (N) rw */
  MJHMC_F_CACHE = 6,  /* HMCState.cache_active (== H_flf is not NaN)  uint8   (N)    r  */
  MJHMC_F_DWELL = 7,  /* ContinuousTimeHMC.dwelling_times            float64 (N)    r  */
  MJHMC_F_TRANS = 8   /* argmin row of min_idx (0=L,1=F,2=R; CONTROL: bit0=FL accepted, bit1=flipped) uint8 (N) r */
} mjhmc_field;

/* Integer bookkeeping of one sampling_iteration attempt (bit-exact with the reference):
 * l/f/r/fl follow the sampler's l_count/f_count/r_count/fl_count increments
 * (markov_jump_hmc.py:143-148,288-290,413-415); E_evals/dEdX_evals are the increments of
 * Distribution.E_count / dEdX_count (mjhmc/misc/distributions.py:62-75). */
typedef struct {
  int64_t l, f, r, fl;
  int64_t n_cold;      /* particles with a cold FLF cache: the reference integrates F L F for each of them
                          (hmc_state.py:109-119) and counts the evaluations below                      */
  int64_t E_evals;
  int64_t dEdX_evals;
  int32_t nonfinite;   /* 1: some particle produced a non-finite rate (utils.py:41-48); the attempt
                          was NOT committed, state is as before the attempt                        */
  int32_t L_used;
  double eps_used;
  int64_t n_flf_run;   /* inverse-L trajectories the device actually integrated (<= n_cold): a particle that has
                          just moved by F has F L F (X, -V) = F L (X, V), the L proposal of the iteration in which
                          it flipped -- the same operations on the same numbers (the reference's authors left the
                          shortcut commented out, markov_jump_hmc.py:401); the dense kernels hand its H() on instead
                          of integrating it again.  Results and the reference's counters are unaffected           */
} mjhmc_iter_stats;

const char* mjhmc_last_error(void);
int mjhmc_abi_version(void);

int mjhmc_ctx_create(int device, mjhmc_ctx** out);
int mjhmc_ctx_destroy(mjhmc_ctx* ctx);
/* name ("<marketing name> (<gcn arch>) [<pci domain:bus:device.0>]"), CU count, HBM bytes of the bound device */
int mjhmc_ctx_info(mjhmc_ctx* ctx, char* name, size_t name_cap, int* n_cu, uint64_t* hbm_bytes);

/* Replaces constructing a Distribution subclass (mjhmc/misc/distributions.py:20-59). */
int mjhmc_energy_create(mjhmc_ctx* ctx, int kind, int ndims, const double* params, size_t nparams,
                        mjhmc_energy** out);
/* LambdaDistribution(energy_func, energy_grad_func, ...) with arbitrary callables (README.md:27-36,
 * mjhmc/misc/distributions.py:198-251).  A Python callable cannot run on the device; the caller states the same two
 * functions as C expressions of ONE coordinate for a separable energy
 *     E(x) = sum_d energy_expr(x_d)      dE/dx_d = grad_expr(x_d)
 * with `x` the coordinate (double), `d` its index (int) and `p[k]` the float64 parameters, e.g.
 * ("0.5*x*x/(p[0]*p[0])", "x/(p[0]*p[0])") or ("log(1.0 + x*x/p[d])", "2.0*x/(p[d] + x*x)").  The engine's kernel
 * templates are compiled around them with hipRTC for gfx950 (include_dir: the directory holding elementwise.hpp and
 * philox.hpp, mjhmc_amd/csrc of this package); float64 state; every sampler family and the replay mode work as for the
 * built-in elementwise energies.  A compile error in the expressions is returned as MJHMC_ERR_INVALID with the
 * compiler's log in mjhmc_last_error(). */
int mjhmc_energy_create_expr(mjhmc_ctx* ctx, int ndims, const char* energy_expr, const char* grad_expr,
                             const double* params, size_t nparams, const char* include_dir, mjhmc_energy** out);
/* compile-only check of a pair of expressions (needs no device): 0, or MJHMC_ERR_INVALID with the compiler's log */
int mjhmc_expr_check(int ndims, const char* energy_expr, const char* grad_expr, const char* include_dir);
/* The same for energies that are separable GIVEN a few per-particle statistics (coupled coordinates):
 *     S[k] = sum_d stat_k(x_d, d; p)                 stat_exprs: the stat_k separated by ';' (NULL / "": none)
 *     E(x) = energy0_expr(S; p) + sum_d energy_expr(x_d, d, S; p)          (energy0_expr may be NULL: 0)
 *     dE/dx_d = grad_expr(x_d, d, S; p)              with the chain-rule terms through S written out by the caller
 * e.g. Neal's funnel (mjhmc/misc/tf_distributions.py:143-147), scale p[0], D = ndims:
 *     stats  "d == 0 ? x : 0.0; d == 0 ? 0.0 : x*x"
 *     energy "0.0"      energy0 "S[0]*S[0]/(2*p[0]*p[0]) + 0.5*exp(-S[0])*S[1] + 0.5*(p[1]-1)*S[0]"      (p[1] = D)
 *     grad   "d == 0 ? x/(p[0]*p[0]) - 0.5*exp(-x)*S[1] + 0.5*(p[1]-1) : x*exp(-S[0])"
 * The statistics are recomputed at every force evaluation (one group reduction each). */
int mjhmc_energy_create_expr_coupled(mjhmc_ctx* ctx, int ndims, const char* stat_exprs, const char* energy_expr,
                                     const char* energy0_expr, const char* grad_expr, const double* params, size_t nparams,
                                     const char* include_dir, mjhmc_energy** out);
int mjhmc_expr_check_coupled(int ndims, const char* stat_exprs, const char* energy_expr, const char* energy0_expr,
                             const char* grad_expr, const char* include_dir);
int mjhmc_energy_destroy(mjhmc_energy* e);

/* One evaluation of E_val / dEdX_val (mjhmc/misc/distributions.py:66-81) on n columns.
 * E_out (n) and dEdX_out (D,n) may each be NULL. */
int mjhmc_eval(mjhmc_energy* e, int dtype, const double* X, int64_t n, double* E_out, double* dEdX_out);

/* Replaces HMCState.__init__ (mjhmc/samplers/hmc_state.py:13-44) as called from
 * ContinuousTimeHMC.__init__ / HMCBase.__init__ (markov_jump_hmc.py:46-65,225-234).
 *   Xinit (D,N) float64.  Vinit (D,N) float64 or NULL -> standard normals from the counter RNG
 *   (tick 0).  first_particle_id: global id of column 0 (shard offset); the RNG is keyed by
 *   global id so results do not depend on how columns are sharded over GPUs. */
int mjhmc_sampler_create(mjhmc_ctx* ctx, mjhmc_energy* e, int64_t nparticles, int64_t first_particle_id,
                         int dtype, const double* Xinit, const double* Vinit, uint64_t seed, int mode,
                         mjhmc_sampler** out);
int mjhmc_sampler_destroy(mjhmc_sampler* s);

/* epsilon, num_leapfrog_steps, p_r, beta as used by HMCState.R, p_flip
 * (markov_jump_hmc.py:67-80,189,197-200,221-223). */
int mjhmc_set_hparams(mjhmc_sampler* s, double epsilon, int num_leapfrog_steps, double p_r, double beta,
                      double p_flip);

/* Runs up to n_iter sampling_iteration()s back to back on the sampler's stream with no host
 * round trip in between.  Stops early when an attempt hits a non-finite rate: that attempt is
 * rolled back (nothing committed), *n_done is the number of committed iterations, and the
 * caller performs the reference's halve-epsilon / double-L / reset_flf_cache / retry / restore
 * sequence (markov_jump_hmc.py:376-389) through mjhmc_set_hparams + mjhmc_reset_flf_cache +
 * mjhmc_iterate(1).  Return value stays 0 in that case.
 *
 * Replay inputs (all nullable; NULL -> counter RNG):
 *   replay_normal (n_iter, D, N) the randn blocks of HMCState.R (hmc_state.py:125)
 *   replay_exp    (n_iter, 3, N) unit exponentials behind draw_from (utils.py:42), rows L,F,R
 *                 (MJHMC) or FL,F,R (CTHMC)
 *   replay_unif   (n_iter, 2N+1) CONTROL mode: rand(N) accept, rand(N) flip, random() R gate
 * per_iter: n_iter entries (nullable); entry i describes attempt i (valid for i <= *n_done).
 * ring_slot0 >= 0: iteration i additionally records its post-jump X in ring slot ring_slot0+i
 * and its dwelling times in dwell slot ring_slot0+i (see mjhmc_ring_alloc). */
int mjhmc_iterate(mjhmc_sampler* s, int n_iter, const double* replay_normal, const double* replay_exp,
                  const double* replay_unif, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done);

/* The sample loop of HMCBase.sample / ContinuousTimeHMC.sample (markov_jump_hmc.py:150-173, 293-338: an iteration, then
 * samples.append(state.copy().X)) with the copies crossing PCIe WHILE the following iterations run: mjhmc_iterate(n_iter,
 * counter RNG, ring_slot0) plus, for every iteration i as soon as its kernels are done, ring slot ring_slot0 + i re-tiled
 * and brought to host_out[:, (k0 + i) * N : (k0 + i + 1) * N] of a C-order float64 array (ndims, n_total * N) -- the layout
 * of np.concatenate(samples, axis=1) -- by a second stream and a worker thread of the call.  Returns when the slots of the
 * committed iterations are in host memory.  A ring smaller than the run (n_slots < n_total: n_total states exceed the
 * device) is walked in several calls with growing k0.  per_iter / n_done / non-finite rates: as mjhmc_iterate. */
int mjhmc_iterate_download(mjhmc_sampler* s, int n_iter, int ring_slot0, double* host_out, int64_t n_total, int64_t k0,
                           mjhmc_iter_stats* per_iter, int* n_done);

/* ---- energies only the caller can evaluate (MJHMC_E_HOST) ----------------------------------------------------------
 * LambdaDistribution(energy_func, energy_grad_func, init) takes two arbitrary Python callables (README.md:27-36,
 * mjhmc/misc/distributions.py:198-251).  When they match no built-in functor and are not stated as C expressions,
 * the sampler keeps everything on the device EXCEPT the two evaluations: the particle state (X, V, dE/dX, EX, EV, the
 * inverse-L cache), the leapfrog updates in the reference's literal order (hmc_state.py:86-100), the jump decision,
 * successor selection, momentum refresh, counters, dwelling times and the sample ring.  One sampling_iteration
 * (markov_jump_hmc.py:355-415; :116-148 and :251-290 for the other sampler families) is
 *     mjhmc_traj_begin(s, &n)                  n = N proposal columns (the L proposal of every particle) + n_cold columns
 *                                              (the inverse-L proposal F L F of the cold-cache particles, MJHMC only)
 *     X = mjhmc_traj_step(s, NULL, 0, X)       first step: uses the stored dE/dX
 *     for every further leapfrog step:  g = energy_grad_func(X);  mjhmc_traj_step(s, g, 0, X)
 *     g = energy_grad_func(X);                 mjhmc_traj_step(s, g, 1, NULL)       closing half kick
 *     mjhmc_traj_finish(s, energy_func(X), ..., &stats)      decide + commit; stats->nonfinite as mjhmc_iterate reports it
 * All matrices are (ndims, n) float64 C order, columns in the order above.  mjhmc_iterate refuses such samplers.
 * mjhmc_host_set_energy: E (N) and dE/dX (D,N) of the CURRENT state -- after mjhmc_sampler_create and after every
 * mjhmc_write of X (HMCState.__init__ evaluates both, hmc_state.py:30-38). */
int mjhmc_host_set_energy(mjhmc_sampler* s, const double* E, const double* dEdX);
int mjhmc_traj_begin(mjhmc_sampler* s, int64_t* n_cols);
int mjhmc_traj_step(mjhmc_sampler* s, const double* grad, int last, double* X_out);
/* replay_* as mjhmc_iterate's, for ONE iteration; ring_slot >= 0 records X and the dwelling times there */
int mjhmc_traj_finish(mjhmc_sampler* s, const double* E, const double* replay_normal, const double* replay_exp,
                      const double* replay_unif, int ring_slot, mjhmc_iter_stats* stats);

/* Transaction support for multi-GPU runs: the reference aborts and retries the WHOLE batch when any
 * particle hits a non-finite rate (markov_jump_hmc.py:376-389).  With columns sharded over ranks a
 * rank may have run past the iteration that failed elsewhere; it restores the checkpoint taken at the
 * start of the batch and replays (the counter RNG makes the replay bit-identical).
 * checkpoint: device copy of X, V, EX, EV, H_flf, dwell + the RNG tick.  restore: put it back. */
int mjhmc_checkpoint(mjhmc_sampler* s);
int mjhmc_restore(mjhmc_sampler* s);
/* Undo the last call when it was mjhmc_iterate(1) (or one mjhmc_traj_finish) and committed: the iteration's inputs are
 * the untouched other halves of the ping-pong buffers -- or, for the samplers that commit in place (rows wider than the
 * register kernels, host-evaluated energies), the pre-move state the commit left in its workspace -- so no copy is taken
 * beforehand (the RNG tick stays consumed). */
int mjhmc_rollback(mjhmc_sampler* s);
/* The position of the counter RNG (one tick per sampling_iteration attempt): together with X, V and H_flf
 * (mjhmc_read / mjhmc_write) it is the whole resumable state of a sampler -- mjhmc_amd's save_state / load_state write it
 * to an .npz file, and a restored sampler continues bit for bit. */
int mjhmc_get_tick(mjhmc_sampler* s, uint64_t* tick);
int mjhmc_set_tick(mjhmc_sampler* s, uint64_t tick);
/* skip n RNG ticks (a rank that did not execute a failed attempt must still consume its tick) */
int mjhmc_advance_tick(mjhmc_sampler* s, int64_t n);

/* HMCState.reset_flf_cache (hmc_state.py:145-148). */
int mjhmc_reset_flf_cache(mjhmc_sampler* s);

/* Field access for sampler.state.X / .V / .EX / ... and HMCState assignment (figures/poe_fig.py:59).
 * Writing X or V re-derives EX/EV on the device and clears the FLF cache. */
int mjhmc_read(mjhmc_sampler* s, int field, void* host_dst, size_t nbytes);
int mjhmc_write(mjhmc_sampler* s, int field, const void* host_src, size_t nbytes);

/* Sample ring for ContinuousTimeHMC.sample / HMCBase.sample (markov_jump_hmc.py:150-173,293-338):
 * n_slots snapshots of X (device resident) and of dwelling_times. */
int mjhmc_ring_alloc(mjhmc_sampler* s, int n_slots);   /* (never shrinks; a request the device cannot hold fails with the
                                                           sizes in mjhmc_last_error() and keeps the ring there was) */
/* device bytes of one ring slot (a padded state matrix + its dwelling times); free / total device memory in bytes:
 * what a caller sizes a sample ring by */
int mjhmc_ring_slot_bytes(mjhmc_sampler* s, uint64_t* bytes);
int mjhmc_mem_info(mjhmc_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
/* dwell of slots [slot0, slot0+n) -> (n, N) float64 */
int mjhmc_ring_read_dwell(mjhmc_sampler* s, int slot0, int n, double* host_dst);
/* out[:, k] = X_slot[idx[k] / N][:, idx[k] % N] for k < n, i.e. the column gather of
 * `samples[:, sample_idx]` (markov_jump_hmc.py:322-328) with idx into the time-major pool.
 * out is (D, n) float64. */
int mjhmc_ring_gather(mjhmc_sampler* s, const int64_t* idx, int64_t n, double* host_out);
/* whole slots [slot0, slot0+n): (D, n*N) time-major if stacked==0 (np.concatenate(axis=1)),
 * (D, N, n) if stacked==1 (np.stack(axis=-1)). */
int mjhmc_ring_read(mjhmc_sampler* s, int slot0, int n, int stacked, double* host_out);

/* sum_k (x_k - shift) and sum_k (x_k - shift)^2 over every state element (d < ndims, particle < N) of ring
 * slots [slot0, slot0 + n): the device half of the reference's online variance estimate
 * (mjhmc/misc/gen_mj_init.py:76-98), which walks sampler.sample(1).ravel() value by value. */
int mjhmc_ring_moments(mjhmc_sampler* s, int slot0, int n, double shift, double* sum, double* sumsq);

/* The leapfrog operator on caller-supplied states: HMCState.leapfrog (n_steps = 1) and HMCState.L
 * (n_steps = num_leapfrog_steps) of mjhmc/samplers/hmc_state.py:86-100, in the reference's literal operation order
 * (half kicks not merged, every product rounded before its sum).  X, V and the outputs are (ndims, n) float64 C order
 * (EX_out / EV_out: n values); EX_out, EV_out, dEdX_out may be NULL.  X_out / V_out may alias X / V.
 * PRODUCT_OF_T / SPARSE_CODE run their tile kernels' integrator (float32 / bfloat16 state, half kicks between drifts
 * merged; figures/poe_fig.py:58-76 integrates snapshots of a ProductOfT sampler this way). */
int mjhmc_leapfrog(mjhmc_energy* e, int dtype, const double* X, const double* V, int64_t n, double eps, int n_steps,
                   double* X_out, double* V_out, double* EX_out, double* EV_out, double* dEdX_out);

/* The two helpers of the jump process as operators on caller arrays (mjhmc/misc/utils.py; the samplers' kernels use
 * the same device functions, `wait_time` and `first_min3` of csrc/elementwise.hpp):
 *   draw_from (utils.py:31-50): waiting times of exponential clocks.  out[i] = inf where rates[i] == 0, otherwise
 *     (1 / rates[i]) * unit_exp[i] -- numpy's exponential(scale = 1/rate) is scale * standard_exponential(), so with the
 *     standard exponentials the caller drew the result is bit-identical.  A non-finite rate is the reference's
 *     ValueError: out[i] = NaN there, *first_bad = the lowest such index, return MJHMC_ERR_NONFINITE (no bad rate:
 *     *first_bad = -1, return 0).
 *   min_idx (utils.py:15-28): which[j] = argmin over the k rows of draws[k][n] (row-major) in column j, the FIRST
 *     minimum on ties and a NaN counting as the minimum, as np.argmin does; the caller lists the columns per row. */
int mjhmc_draw_from(mjhmc_ctx* ctx, const double* rates, const double* unit_exp, int64_t n, double* out,
                    int64_t* first_bad);
int mjhmc_min_idx(mjhmc_ctx* ctx, const double* draws, int k, int64_t n, int32_t* which);

/* Autocorrelation along the time axis of ring slots [slot0, slot0 + n):
 *   out[k] = sum_{d < ndims, particle < N} sum_t x_t * x_{t+k},   k = 0 .. n-1   (n float64 to the host)
 * linear == 0: t + k wraps modulo n.  out / out[0] is fft_autocor(samples) of mjhmc/misc/autocor.py:37-49
 *              (fftn along time, |.|^2, ifftn, mean over dims and particles, normalise by lag 0).
 * linear != 0: the sum stops at t + k < n.  out[k] / (ndims * N * (n - k)) is the lag product mean
 *              np.mean(samples[:, :, :-k] * samples[:, :, k:]) of slow_autocorrelation (:177-211) and of the
 *              brute-force branch of autocorrelation (:52-117).
 * The sums are returned unnormalised so that ranks holding column shards can add theirs before dividing. */
int mjhmc_ring_autocor(mjhmc_sampler* s, int slot0, int n, int linear, double* host_out);
/* The same for a host array laid out like the reference's samples, [n_dims, n_batch, n_samples] C order,
 * i.e. n_series = n_dims * n_batch contiguous series of n_samples float64 (no sampler needed). */
int mjhmc_autocor(mjhmc_ctx* ctx, const double* samples, int64_t n_series, int n_samples, int linear,
                  double* host_out);

/* ---- several GPUs: one process per GPU, particle COLUMNS sharded over the ranks ------------------
 * The reference is single-process; its particles are independent chains (mjhmc/samplers/hmc_state.py works column-wise
 * everywhere), so nothing is exchanged on the data path until sample() returns (markov_jump_hmc.py:150-173,293-338).
 * The communicator is RCCL (xGMI between the GPUs of a node), loaded with dlopen on first use.  Every rank calls every
 * collective in the same order.  Pointers are host memory unless stated. */
#define MJHMC_COMM_ID_BYTES 128
typedef enum { MJHMC_OP_SUM = 0, MJHMC_OP_MIN = 1, MJHMC_OP_MAX = 2 } mjhmc_reduce_op;
/* rank 0 draws the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any host channel (file, socket) */
int mjhmc_comm_unique_id(void* id);
int mjhmc_comm_create(mjhmc_ctx* ctx, int rank, int world, const void* id, mjhmc_comm** out);
int mjhmc_comm_destroy(mjhmc_comm* c);
/* 0 when librccl can be loaded and has every entry point the communicator uses, MJHMC_ERR_COMM otherwise (needs no
 * device and enters no collective: the ranks of a job can agree on a fallback BEFORE any of them calls mjhmc_comm_create) */
int mjhmc_comm_available(void);
/* the number of ranks of the communicator as RCCL reports it (ncclCommCount) */
int mjhmc_comm_count(mjhmc_comm* c, int* count);
/* host values, in place: integer bookkeeping (l/f/r counts, E_count / dEdX_count increments: sums), the number of
 * iterations every rank committed before a non-finite rate (min: the reference retries the WHOLE batch,
 * markov_jump_hmc.py:376-389), lag sums of the autocorrelation, elapsed times (max) */
int mjhmc_comm_allreduce_i64(mjhmc_comm* c, int64_t* inout, int64_t n, int op);
int mjhmc_comm_allreduce_f64(mjhmc_comm* c, double* inout, int64_t n, int op);
/* e.g. the np.random.random draws of the resampling step (markov_jump_hmc.py:325), drawn on rank 0 */
int mjhmc_comm_bcast(mjhmc_comm* c, void* inout, size_t nbytes, int root);
/* recv = rank 0's block, rank 1's block, ... (nbytes_per_rank[r] bytes each; send: this rank's block) */
int mjhmc_comm_allgatherv(mjhmc_comm* c, const void* send, const int64_t* nbytes_per_rank, void* recv);
/* THE data-path collective, device to device: ring slots [slot0, slot0 + n) of every rank's sampler are all-gathered
 * and re-tiled on the receiving GPU; host_out is the sample block of the UNSHARDED run, columns in global particle
 * order: (D, n * N_total) time-major if stacked == 0 (np.concatenate(axis=1), markov_jump_hmc.py:170-173,336-338),
 * (D, N_total, n) if stacked != 0 (np.stack(axis=-1)).  particles_per_rank[world]: the column shard sizes.
 * host_out == NULL: this rank takes part in the collective and keeps nothing -- no re-tile, no download (a caller that
 * wants the block on ONE rank passes NULL on the others: at C4 / n = 10 every rank's copy is 2.56 GB over its PCIe link);
 * the same holds for mjhmc_comm_allgather_columns. */
int mjhmc_comm_allgather_ring(mjhmc_comm* c, mjhmc_sampler* s, int slot0, int n, int stacked,
                              const int64_t* particles_per_rank, double* host_out);
/* resampled columns (markov_jump_hmc.py:322-328): every rank gathers the n_local ring columns IT owns (local pool indices
 * slot * N_local + column, as mjhmc_ring_gather takes them) on the device, the blocks are all-gathered, host_out is
 * (D, sum columns_per_rank) with rank 0's columns first; the caller puts them in sample order. */
int mjhmc_comm_allgather_columns(mjhmc_comm* c, mjhmc_sampler* s, const int64_t* local_idx, int64_t n_local,
                                 const int64_t* columns_per_rank, double* host_out);

/* Device time of the last mjhmc_iterate call in milliseconds: ONE HIP-event pair on the sampler's stream
 * brackets its whole launch sequence (first to last jump kernel); jump_kernel_ms == total_ms and
 * n_jump_launches is the number of sampling_iteration attempts it covers. */
int mjhmc_last_timing(mjhmc_sampler* s, double* total_ms, double* jump_kernel_ms, int* n_jump_launches);

/* The event pair is two marker packets on the stream (~8 us of a call: 3 % of a one-iteration call at C4's size).
 * on = 0: calls record nothing and mjhmc_last_timing keeps reporting the last recorded call; on by default.  (The
 * reference has no counterpart: its callers time `sampling_iteration()` with the wall clock; the drop-in classes of
 * mjhmc_amd.samplers switch it off.) */
int mjhmc_set_timing(mjhmc_sampler* s, int on);

/* Stream synchronisation (bench harness). */
int mjhmc_sync(mjhmc_sampler* s);

#ifdef __cplusplus
}
#endif
#endif /* MJHMC_HIP_H */
