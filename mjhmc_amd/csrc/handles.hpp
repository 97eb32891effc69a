// Handle structs of the C ABI (include/mjhmc_hip.h) and the few helpers api.hip shares with comm.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mjhmc_hip.h"
#include "dense_pot.hpp"
#include "dense_sic.hpp"
#include "elementwise.hpp"
#include "user_expr.hpp"

using namespace mjhmc;

int mjhmc_fail(int code, const std::string& msg);  // records the message for mjhmc_last_error(), returns code

#define HIPCHK(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess)                                                                                    \
      return mjhmc_fail(MJHMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                           ":" + std::to_string(__LINE__) + ")");                            \
  } while (0)

#define TRY(expr)             \
  do {                        \
    int r_ = (expr);          \
    if (r_ != 0) return r_;   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct mjhmc_ctx {
  int device;
  hipDeviceProp_t prop;
};

struct mjhmc_energy {
  mjhmc_ctx* ctx;
  EnergyParams ep;
  std::vector<double> params;
  void* dev64 = nullptr;
  void* dev32 = nullptr;
  float* pot[4] = {nullptr, nullptr, nullptr, nullptr};  // ProductOfT: W1, W2T, cb, alpha (float32, padded to 512)
  int pot_dim = kPotDim;  // rows padded to 128, 256 or 512 -- or, beyond the tile kernels, to a multiple of 512 (pot_big())
  PotModel pot_model() const { return PotModel{pot[0], pot[1], pot[2], pot[3], pot_dim, ep.ndims}; }
  bool pot_big() const { return is_pot() && pot_dim > kPotDim; }   // matrices stored as 512 x 512 blocks, multi-pass path only
  PotBigModel pot_big_model() const { return PotBigModel{pot[0], pot[1], pot[2], pot[3], pot_dim, ep.ndims}; }
  bool is_pot() const { return ep.kind == MJHMC_E_PRODUCT_OF_T; }
  void* sic[3] = {nullptr, nullptr, nullptr};  // SparseImageCode: A1, A2 (bf16, fragment order), y (float32 [P][256])
  float sic_lambda = 0.f;
  int sic_cauchy = 1;
  int sic_P = 1;  // n_patches
  int sic_copies = 1;
  int sic_nc = 1024;  // n_coeffs
  SicModel sic_model() const {
    return SicModel{sic[0], sic[1], (const float*)sic[2], sic_lambda, sic_cauchy, sic_P, 1.0f / (float)sic_P, sic_copies,
                    sic_nc};
  }
  bool is_sic() const { return ep.kind == MJHMC_E_SPARSE_CODE; }
  bool is_dense() const { return is_pot() || is_sic(); }
  UserEnergy* user = nullptr;  // MJHMC_E_USER_EXPR: the hipRTC-built kernels of this energy
  bool is_user() const { return ep.kind == MJHMC_E_USER_EXPR; }
  // MJHMC_E_HOST: E and dE/dX are evaluated by the caller (opaque Python callables), host_energy.hip
  bool is_host() const { return ep.kind == MJHMC_E_HOST; }
};

struct Shape {
  int E, logG, pitch, CH, esize;
  // ndims beyond the register-resident elementwise kernels (> 64 lanes x 16 elements): the sampler runs the multi-pass
  // path of host_energy.hip, state in HBM between the substeps (float64, built-in elementwise energies)
  bool wide = false;
  // float32 state was asked for, for rows only the multi-pass path handles: the sampler runs as a float64 one whose state
  // is rounded to float32 at the end of every operation that writes it (the caller sees float32 values throughout)
  bool round32 = false;
};


constexpr int kMaxDenseParts = 4;   // free-running parts of a dense batch (iterate_t)

struct DlSession;  // overlapped sample download of one mjhmc_iterate_download call (api.hip)
struct HostTraj;  // proposal workspace of a host-energy sampler (host_energy.hip)

struct mjhmc_sampler {
  mjhmc_ctx* ctx;
  mjhmc_energy* en;
  int64_t N, Npad, first_pid;  // Npad: rows allocated (N rounded up to 64; padding rows stay zero)
  int D, dtype, mode;
  Shape sh;
  hipStream_t stream = nullptr;
  char* h_pin = nullptr;          // pinned host staging of the per-call read-back (failure flag + tallies)
  size_t h_pin_cap = 0;
  std::vector<hipStream_t> part_streams;   // further parts of a split launch: fused elementwise launches (iterate_fused_t), dense batches (iterate_t)
  std::vector<hipEvent_t> part_events;
  hipEvent_t ev_join = nullptr;
  void* ick[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // the split schedule's own checkpoint (as ck[])
  hipEvent_t ev_fork = nullptr;                // "everything the first stream has been given so far"
  void* Xbuf[2] = {nullptr, nullptr};
  void* Vbuf[2] = {nullptr, nullptr};
  void* Xcur = nullptr;
  void* Gbuf[2] = {nullptr, nullptr};  // dEdX (dense energies keep it, like HMCState.dEdX); follows vcur
  float* Hwork = nullptr;              // dense energies: H of the inverse-L proposals an iteration integrates
  int* cold_list = nullptr;            // + the compacted lists of those particles (two of Npad entries, iterations alternate; then
                                       //   kMaxDenseParts x 3 rotating counters)
  double* pot64_scratch = nullptr;     // ProductOfT, float64 state: working rows, one set per concurrent launch (kMaxDenseParts)
  void* Hspec[2] = {nullptr, nullptr}; // dense energies, MJHMC: H(L proposal) of the particles that then moved by F (NaN elsewhere):
                                       //   next iteration's H of their inverse-L proposal, bit for bit (dense_pot.hip); follows scur
  void* Hspec_dump = nullptr;          // (test build, MJHMC_NO_FSPEC: where the hand-over goes instead, so that nothing is ever handed on)
  // elementwise energies, several particles per wave: inverse-L pass over the compacted cold particles
  int* flf_list = nullptr;     // [Npad]
  // The jump process of an iteration hands its movers on as the next iteration's list -- the last iteration of a call too:
  // a following call starts from it instead of scanning H_flf (sampling_iteration() callers: 9 us of C4's 256).  Valid
  // from a committed call of the compacted passes until anything else touches the state (every other iterate path, a
  // state / cache write, restore, rollback, reset_flf_cache).
  bool list_valid = false;
  int list_par = 0;            // which of the two lists
  int list_count = 0;          // its length
  int* flf_counts = nullptr;   // [stats_cap + 1] one counter per attempt of the current mjhmc_iterate call (inside call_block)
  void* Hpre = nullptr;        // [2][Npad] H_flf with the cold entries filled in (iterations alternate between the halves)
  int vcur = 0, scur = 0;
  void* EX[2] = {nullptr, nullptr};
  void* EV[2] = {nullptr, nullptr};
  void* Hflf[2] = {nullptr, nullptr};
  double* dwell = nullptr;
  double* dwell_scratch = nullptr;  // dwell_ring target when no ring slot is recorded
  void* ck[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // checkpoint: X, V, EX, EV, Hflf, dwell, dEdX (ProductOfT)
  uint64_t ck_tick = 0;
  bool ck_valid = false;
  void* undo_X = nullptr;   // pre-move X of the last single-iteration call (its ping-pong input, still intact)
  bool undo_valid = false;
  bool undo_multipass = false;   // the last committed iteration ran on the multi-pass path: its undo is a copy back (host_energy.hip)
  uint8_t* trans = nullptr;
  // What a call zeroes at its start and reads back at its end lives in ONE device block -- [Control, 64 bytes][flf_counts:
  // stats_cap + 1 counters, padded to 64 bytes][stats: stats_cap x 4] -- so that a call is one fill and one copy
  // (api.hip: ensure_call_block, zero_call, read_back_call).  ctl, flf_counts and stats point into it.
  char* call_block = nullptr;
  Control* ctl = nullptr;
  long long* stats = nullptr;  // [stats_cap][4]
  int stats_cap = 0;
  void* ring = nullptr;
  double* dwell_ring = nullptr;
  int ring_slots = 0;
  double* stage = nullptr;  // device staging, float64 host layout
  size_t stage_elems = 0;
  void* noise = nullptr;    // replay normals, particle-major
  double* rexp = nullptr;   // [3][N]
  double* runif = nullptr;  // [2N+1]
  void* scratch = nullptr;  // [N][pitch] scratch (dEdX reads)
  bool download_f32 = false;  // bf16 state: the next download_cols source is a float32 matrix (dEdX)
  double eps = 1e-4, p_r = 0, beta = 1, p_flip = 0.5;
  int L = 5;
  uint64_t seed = 0, tick = 1;
  hipEvent_t ev_total[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_k;
  double last_total_ms = 0, last_jump_ms = 0;
  int last_jump_launches = 0;
  bool timing_pending = false;
  bool timing_on = true;         // mjhmc_set_timing: the HIP-event pair around a call's launches (two marker packets, ~8 us of a call)
  void* pipe_pin[2] = {nullptr, nullptr};       // pinned double buffer of the pipelined device -> host copies (copy_to_host)
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};
  // mjhmc_iterate_download: the ring slot of an iteration goes to the host while the next iterations run -- its own stream,
  // staging matrix and pinned double buffer, driven by a worker thread of the call
  DlSession* dl = nullptr;
  hipStream_t dl_stream = nullptr;
  double* dl_stage = nullptr;
  size_t dl_stage_elems = 0;
  void* dl_pin[2] = {nullptr, nullptr};
  hipEvent_t dl_ev[2] = {nullptr, nullptr};
  HostTraj* ht = nullptr;         // MJHMC_E_HOST: trajectory workspace (mjhmc_traj_*)
  bool host_energy_set = false;   // MJHMC_E_HOST: EX and dE/dX of the current state are the caller's (mjhmc_host_set_energy)
};

inline size_t row_bytes(const mjhmc_sampler* s) { return (size_t)s->sh.pitch * s->sh.esize; }
// per-particle scalars (EX, EV, H_flf): float64 for float64 state, float32 otherwise (bf16 state included)
inline size_t ssize(const mjhmc_sampler* s) { return s->dtype == MJHMC_F64 ? 8 : 4; }
inline size_t mat_bytes(const mjhmc_sampler* s) { return (size_t)s->Npad * row_bytes(s); }


// host_energy.hip
void host_traj_free(mjhmc_sampler* s);
int host_run_eval(mjhmc_sampler* s, const void* V, void* Vgen, void* EVout);
int wide_run_eval(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen, void* EVout);
int multipass_iterate(mjhmc_sampler* s, int n_iter, const double* replay_normal, const double* replay_exp,
                      const double* replay_unif, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done);
int multipass_rollback(mjhmc_sampler* s);
int round_rows32(mjhmc_sampler* s, void* rows);   // float64 rows -> float32 values (Shape::round32), api.hip
int wide_leapfrog(mjhmc_sampler* w, const double* X, const double* V, double* Xo, double* Vo, double* G, double* EX,
                  double* EV, double eps, int n_steps);

// lanes-per-particle / elements-per-lane selection of the elementwise kernels for ndims = D
int pick_shape(int D, int dtype, Shape* out);
// device staging buffer of at least `elems` float64 (host layout side of every re-tiling)
int ensure_stage(mjhmc_sampler* s, size_t elems);
// device particle-major rows -> s->stage in the reference's (ndims, columns) layout: stage[d*rs + k*cs + off] = row(k)[d],
// row(k) = dev_idx ? dev_idx[k] : k; copy_out: then copy host_elems doubles of the stage to `host` and synchronise
int download_cols(mjhmc_sampler* s, const void* src, const int64_t* dev_idx, int64_t ncols, double* host,
                  size_t host_elems, int64_t rs, int64_t cs, int64_t off, bool copy_out);
// device -> pageable host memory on the sampler's stream, synchronised; big blocks go through a pinned double buffer
// with the host-side copy spread over several threads
int copy_to_host(mjhmc_sampler* s, const void* dev_src, void* host_dst, size_t bytes);
