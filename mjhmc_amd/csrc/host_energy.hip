// LambdaDistribution(energy_func, energy_grad_func, init) with OPAQUE Python callables (README.md:27-36,
// mjhmc/misc/distributions.py:198-251): energies the engine has no device form of -- neither a built-in functor nor
// C expressions -- for example a dense quadratic form.  A Python callable cannot run in a kernel, so for these energies
// E and dE/dX are evaluated by the CALLER, once per leapfrog step, and everything else stays on the device: the particle
// state (X, V, dE/dX, EX, EV, the inverse-L cache), the leapfrog arithmetic in the reference's literal operation order
// (hmc_state.py:86-100), the jump decision (rates, waiting times, first minimum: the same device functions the
// elementwise kernels call), the successor selection, momentum refresh, counters, dwelling times and the sample ring.
//
// Protocol of one sampling iteration (include/mjhmc_hip.h: mjhmc_traj_*):
//   traj_begin   proposal columns [0, N): (X, V, dE/dX) of every particle -- the L proposal; MJHMC: columns [N, N + n_cold):
//                (X, -V, dE/dX) of the particles whose inverse-L cache is cold, in ascending particle order -- F L F is
//                read only through H() (markov_jump_hmc.py:360,367)
//   traj_step    [V += (-eps/2) g(caller's gradient at the X handed out last)]; unless last: V += (-eps/2) g; X += eps V; X -> caller
//   traj_finish  caller's E at the end points -> H of the proposals -> decide -> commit (or, on a non-finite rate,
//                nothing: the attempt is not committed, markov_jump_hmc.py:376-389)
// It is slow by construction (a host round trip per leapfrog step); kernels here are plain one-thread-per-element loops.
//
// The same multi-pass machinery, with the state in HBM between the substeps and the gradient from a device kernel
// instead of a call-back, serves the built-in elementwise energies when ndims exceeds what the register-resident jump
// kernel holds (> 1024 float64 dims: 64 lanes x 16 elements): `wide` samplers (Shape::wide), driven by
// multipass_iterate from mjhmc_iterate.  The reference accepts any ndims (hmc_state.py:86-100 are plain array
// operations); so does the engine, on this slower path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "handles.hpp"

struct HostTraj {
  double* X = nullptr;  // [2 Npad][pitch] proposal columns
  double* V = nullptr;
  double* G = nullptr;
  double* E = nullptr;   // [2 Npad] caller's energies at the end points
  double* EVw = nullptr; // [2 Npad] kinetic energies of the end points
  double* noise = nullptr;  // [Npad][pitch] replay normals
  int* cold = nullptr;      // [Npad] cold particles, ascending
  int* coldpos = nullptr;   // [Npad] position in `cold`, -1 when the cache is hot
  int* kmove = nullptr;     // [Npad] decision of the attempt
  double* dtmp = nullptr;   // [Npad] dwelling time of the attempt
  double* hnew = nullptr;   // [3 Npad] EX, EV, H_flf of the successor
  float* pot32[4] = {nullptr, nullptr, nullptr, nullptr};  // ProductOfT with float64 state: float32 X, dE/dX rows and E of the force
                                                           // evaluation; [3]: the u / phi(u) rows of the blocked evaluation (ndims > 512)
  int n_cold = 0;
  int64_t n_cols = 0;
  int phase = 0;  // 0 idle, 1 begun (stepping), 2 last kick done
  int steps = 0;
};

int traj_begin_impl(mjhmc_sampler* s, int64_t* n_cols);
int traj_finish_impl(mjhmc_sampler* s, const double* replay_normal, const double* replay_exp, const double* replay_unif,
                     int ring_slot, mjhmc_iter_stats* st);

namespace {

__global__ void hk_copy_rows(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ idx,
                             int64_t nrows, int pitch, double sign) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * pitch) return;
  const int64_t r = i / pitch;
  const int d = (int)(i - r * pitch);
  const int64_t sr = idx ? idx[r] : r;
  dst[i] = sign * src[sr * pitch + d];
}

// y += a * x, product rounded before the sum (the library is built with -ffp-contract=off): NumPy's V += c * g
__global__ void hk_axpy(double* __restrict__ y, const double* __restrict__ x, double a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + a * x[i];
}

// (D, n) float64 row-major in `stage`  ->  rows [n][pitch]
__global__ void hk_to_rows(const double* __restrict__ stage, double* __restrict__ dst, int D, int64_t n, int pitch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * D) return;
  const int d = (int)(i / n);
  const int64_t r = i - (int64_t)d * n;
  dst[r * pitch + d] = stage[i];
}

// out[r] = sum_d V[r][d]^2 / 2 : one wavefront per row (np.sum(V * V, axis=0) / 2., hmc_state.py:74-78)
__global__ void hk_kinetic(const double* __restrict__ V, double* __restrict__ out, int64_t nrows, int D, int pitch) {
  const int64_t r = blockIdx.x;
  if (r >= nrows) return;
  double s = 0.0;
  for (int d = threadIdx.x; d < D; d += 64) {
    const double v = V[r * pitch + d];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) out[r] = s / 2.0;
}

// tick-0 normals: the initial momentum (hmc_state.py:24-26), as mjhmc_eval_kernel draws it
__global__ void hk_gen_v(double* __restrict__ V, RngKey key, int64_t first_pid, int64_t N, int D, int pitch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int pairs = (D + 1) / 2;
  if (i >= N * pairs) return;
  const int64_t p = i / pairs;
  const int pr = (int)(i - p * pairs);
  double z0, z1;
  normal_pair(key, (uint32_t)(first_pid + p), (uint32_t)pr, z0, z1);
  V[p * pitch + 2 * pr] = z0;
  if (2 * pr + 1 < D) V[p * pitch + 2 * pr + 1] = z1;
}

// ---- built-in elementwise energies on rows of any length: one wavefront per row, E and dE/dX in one pass ------------
struct WideEnergy {
  int kind;
  double a, b, c;        // the functor constants of elementwise.hpp (IsoGaussF, RoughWellF, MMGaussF, FunnelNealF, FunnelRefF)
  const double* jdiag;   // DIAG_GAUSS
};

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void hk_energy_grad(const double* __restrict__ X, double* __restrict__ G, double* __restrict__ E, int64_t nrows,
                               int D, int pitch, const WideEnergy w) {
  const int64_t r = blockIdx.x;
  if (r >= nrows) return;
  const int lane = threadIdx.x;
  const double* x = X + r * pitch;
  double* g = G ? G + r * pitch : nullptr;
  const double pi = 3.141592653589793;
  double e = 0.0;
  switch (w.kind) {
    case MJHMC_E_ISO_GAUSS: {  // a = 1/sigma^2, b = 1/(2 sigma^2)
      double s = 0.0;
      for (int d = lane; d < D; d += 64) {
        const double xv = x[d];
        s = __builtin_fma(xv, xv, s);
        if (g) g[d] = xv * w.a;
      }
      e = wave_sum(s) * w.b;
      break;
    }
    case MJHMC_E_DIAG_GAUSS: {
      double s = 0.0;
      for (int d = lane; d < D; d += 64) {
        const double xv = x[d], j = w.jdiag[d];
        s = __builtin_fma(xv, j * xv, s);
        if (g) g[d] = j * xv;
      }
      e = wave_sum(s) / 2.0;
      break;
    }
    case MJHMC_E_ROUGH_WELL: {  // a = s1^2, b = 2 s1^2, c = s2 (distributions.py:295-304, operation order as written there)
      double s = 0.0;
      for (int d = lane; d < D; d += 64) {
        const double xv = x[d];
        s += (xv * xv) / w.b + cos(xv * 2.0 * pi / w.c);
        if (g) g[d] = xv / w.a + -sin(xv * 2.0 * pi / w.c) * 2.0 * pi / w.c;
      }
      e = wave_sum(s);
      break;
    }
    case MJHMC_E_MM_GAUSS: {  // a = 2 * separation (distributions.py:323-335)
      const double common = exp(4.0 * w.a * x[0]);
      double sa = 0.0, sb = 0.0;
      for (int d = lane; d < D; d += 64) {
        const double xv = x[d], sp = d == 0 ? w.a : 0.0;
        sa += (xv + sp) * (xv + sp);
        sb += (xv - sp) * (xv - sp);
        if (g) g[d] = (2.0 * ((xv - sp) * common + sp + xv)) / (common + 1.0);
      }
      e = -log(exp(-wave_sum(sa)) + exp(-wave_sum(sb)));
      break;
    }
    case MJHMC_E_FUNNEL_NEAL:    // a = 1/scale^2, b = (D-1)/2 (tf_distributions.py:143-147)
    case MJHMC_E_FUNNEL_REF: {   // a = 1/scale^2, b = D-1     (tf_distributions.py:157-165, as coded)
      const double x0 = x[0], ex = exp(-x0);
      double s = 0.0;
      for (int d = lane; d < D; d += 64) {
        const double xv = x[d];
        if (d > 0) s = __builtin_fma(xv, xv, s);
      }
      const double S = wave_sum(s);
      const bool neal = w.kind == MJHMC_E_FUNNEL_NEAL;
      if (g)
        for (int d = lane; d < D; d += 64) {
          if (d == 0) g[0] = neal ? (x0 * w.a - 0.5 * ex * S + w.b) : (-2.0 * w.b * x0 * w.a + ex * S);
          else g[d] = neal ? x[d] * ex : -2.0 * x[d] * ex;
        }
      e = neal ? (x0 * x0 * (0.5 * w.a) + 0.5 * ex * S + w.b * x0) : (-(w.b * x0 * x0 * w.a) - ex * S);
      break;
    }
    default: break;
  }
  if (E && lane == 0) E[r] = e;
}

struct DecideArgs {
  const double* EX;
  const double* EV;
  const double* Hflf;
  const double* E;    // proposals
  const double* EVw;
  const int* coldpos;
  int* kmove;
  double* dtmp;
  double* hnew;
  const double* rexp;
  const double* runif;
  Control* ctl;
  int64_t N, Npad, first_pid;
  double p_r, p_flip;
  int mode;
  RngKey key;
};

// one thread per particle: the jump decision, with the device functions of the elementwise kernels (one lane per
// particle: their serial forms)
template <bool REPLAY>
__global__ void hk_decide(const DecideArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.N) return;
  const uint32_t pid = (uint32_t)(a.first_pid + p);
  const double EX0 = a.EX[p], EV0 = a.EV[p];
  const double H0 = EX0 + EV0;
  const double EXL = a.E[p], EVL = a.EVw[p];
  const double HL = EXL + EVL;
  JumpArgs<double> ja;
  ja.p_r = a.p_r;
  ja.p_flip = a.p_flip;
  ja.rexp = a.rexp;
  ja.runif = a.runif;
  ja.N = a.N;
  LaneMap m;
  m.j = 0;
  m.G = 1;
  m.D = 1;
  m.CH = 1;
  m.lane0 = 0;
  m.wpp = 0;
  int k = 0;
  double dwell = 0.0;
  bool bad = false;
  double EXn = EX0, EVn = EV0, Hc = __builtin_nan("");
  if (a.mode == kModeMJHMC) {
    const double hc = a.Hflf[p];
    double Hf = hc;
    if (!(hc == hc)) {
      const int j = a.coldpos[p];
      Hf = a.E[a.N + j] + a.EVw[a.N + j];
    }
    decide<double, REPLAY>(ja, a.key, m, H0, HL, Hf, p, pid, k, dwell, bad);
    if (k == 0) {  // L accepted: the pre-move state becomes the cached inverse-L state (markov_jump_hmc.py:399-404)
      EXn = EXL;
      EVn = EVL;
      Hc = H0;
    }
  } else if (a.mode == kModeCT) {
    decide_ct<double, REPLAY>(ja, a.key, m, H0, HL, p, pid, k, dwell, bad);
    if (k == 0) {
      EXn = EXL;
      EVn = EVL;
    }
  } else {  // markov_jump_hmc.py:116-148
    double uacc, uflip, ugate;
    if constexpr (REPLAY) {
      uacc = a.runif[p];
      uflip = a.runif[a.N + p];
      ugate = a.runif[2 * a.N];
    } else {
      const u32x4 q = philox4x32_10(pid, a.key.tick_lo, a.key.tick_hi, kSlotExpR, a.key.k0, a.key.k1);
      const u32x4 f = philox4x32_10(pid, a.key.tick_lo, a.key.tick_hi, kSlotFlip, a.key.k0, a.key.k1);
      const u32x4 g = philox4x32_10(0xFFFFFFFFu, a.key.tick_lo, a.key.tick_hi, kSlotFlip, a.key.k0, a.key.k1);
      uacc = u53(q.w2, q.w3);
      uflip = u53(f.w0, f.w1);
      ugate = u53(g.w2, g.w3);
    }
    const double dH = H0 - HL;
    const bool accept = !(dH < 0.0) || (uacc < exp(dH));
    const bool flip = uflip < a.p_flip;
    const bool gate = ugate < a.p_r;
    k = (accept ? 1 : 0) | (flip ? 2 : 0) | (gate ? 4 : 0);
    if (accept) {
      EXn = EXL;
      EVn = EVL;
    }
  }
  if (bad) a.ctl->failed = 1;
  a.kmove[p] = k;
  a.dtmp[p] = dwell;
  a.hnew[p] = EXn;
  a.hnew[a.Npad + p] = EVn;
  a.hnew[2 * a.Npad + p] = Hc;
}

struct CommitArgs {
  double* X;
  double* V;
  double* G;
  double* EXs;
  double* EVs;
  double* Hflf;
  double* TX;     // in: the proposal's end point; out: the pre-move rows (what mjhmc_rollback puts back)
  double* TV;
  double* TG;
  const int* kmove;
  const double* dtmp;
  double* hnew;   // in: EX, EV, H_flf of the successor; out: the pre-move values
  const double* noise;  // replay normals (rows) or nullptr
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  double* ring_slot;  // X snapshot target or nullptr
  unsigned long long* stats;
  int64_t N, Npad, first_pid;
  int D, pitch, mode;
  int round32;    // the state is float32-valued (Shape::round32): successors are rounded as they are written
  double r_keep, r_mix;
  RngKey key;
};

// one wavefront per particle: successor state (markov_jump_hmc.py:399-410 / 138-141 / 277-286), momentum refresh
// (hmc_state.py:121-129), bookkeeping
template <bool REPLAY>
__global__ void hk_commit(const CommitArgs a) {
  const int64_t p = blockIdx.x;
  if (p >= a.N) return;
  const int lane = threadIdx.x;
  const int km = a.kmove[p];
  const uint32_t pid = (uint32_t)(a.first_pid + p);
  double* x = a.X + p * a.pitch;
  double* v = a.V + p * a.pitch;
  double* g = a.G + p * a.pitch;
  bool take = false, flip_old = false, flip_new = false, refresh = false;
  int n0 = 0, n1 = 0, n2 = 0, n3 = 0, k = km;
  if (a.mode == kModeMJHMC) {
    take = km == 0;
    flip_old = km == 1;
    refresh = km == 2;
    n0 = km == 0;
    n1 = km == 1;
    n2 = km == 2;
  } else if (a.mode == kModeCT) {  // FL: leap, then flip (:258,278)
    take = km == 0;
    flip_new = km == 0;
    flip_old = km == 1;
    refresh = km == 2;
    n0 = km == 0;
    n1 = km == 1;
    n2 = km == 2;
  } else {
    const bool accept = km & 1, flip = km & 2, gate = km & 4;
    take = accept;
    flip_new = accept != flip;   // accepted L F, then possibly F again
    flip_old = !accept && flip;
    refresh = gate;
    k = km & 3;
    n0 = k == 3;
    n1 = k == 2;
    n2 = gate;
    n3 = k == 1;
  }
  double* tx = a.TX + p * a.pitch;
  double* tv = a.TV + p * a.pitch;
  double* tg = a.TG + p * a.pitch;
  double ev = 0.0;
  for (int d = lane; d < a.D; d += 64) {
    const double x0 = x[d], v0 = v[d], g0 = g[d];
    double xv = x0, vv = v0;
    if (take) {
      xv = tx[d];
      vv = tv[d];
      g[d] = tg[d];
      x[d] = xv;
      if (flip_new) vv = -vv;
    } else if (flip_old) {
      vv = -vv;
    }
    if (refresh) {
      double z;
      if constexpr (REPLAY) {
        z = a.noise[p * a.pitch + d];
      } else {
        double z0, z1;
        normal_pair(a.key, pid, (uint32_t)(d >> 1), z0, z1);
        z = (d & 1) ? z1 : z0;
      }
      vv = vv * a.r_keep + z * a.r_mix;
      if (a.round32) vv = (double)(float)vv;
      ev += vv * vv;
    }
    if (a.round32) {
      xv = (double)(float)xv;
      vv = (double)(float)vv;
      if (take) x[d] = xv;
    }
    v[d] = vv;
    if (a.ring_slot) a.ring_slot[p * a.pitch + d] = xv;
    // the proposal rows are spent: they now keep the pre-move state, so that a committed single iteration can be undone
    // without a copy having been taken (mjhmc_rollback; sharded runs whose failure happened on another rank)
    tx[d] = x0;
    tv[d] = v0;
    tg[d] = g0;
  }
  if (refresh) {
    for (int o = 32; o > 0; o >>= 1) ev += __shfl_xor(ev, o);
  }
  if (lane == 0) {
    const double ex_old = a.EXs[p], ev_old = a.EVs[p], hf_old = a.Hflf[p];
    a.EXs[p] = a.hnew[p];
    a.EVs[p] = refresh ? ev / 2.0 : a.hnew[a.Npad + p];
    a.Hflf[p] = a.hnew[2 * a.Npad + p];
    a.hnew[p] = ex_old;
    a.hnew[a.Npad + p] = ev_old;
    a.hnew[2 * a.Npad + p] = hf_old;
    a.dwell[p] = a.dtmp[p];
    if (a.dwell_ring) a.dwell_ring[p] = a.dtmp[p];
    a.trans[p] = (uint8_t)k;
    if (n0) atomicAdd(&a.stats[0], 1ull);
    if (n1) atomicAdd(&a.stats[1], 1ull);
    if (n2) atomicAdd(&a.stats[2], 1ull);
    if (n3) atomicAdd(&a.stats[3], 1ull);
  }
}

// mjhmc_rollback of a multi-pass sampler: the pre-move rows and scalars hk_commit left in the proposal workspace go back
__global__ void hk_undo(double* __restrict__ X, double* __restrict__ V, double* __restrict__ G, double* __restrict__ EX,
                        double* __restrict__ EV, double* __restrict__ Hflf, const double* __restrict__ TX,
                        const double* __restrict__ TV, const double* __restrict__ TG, const double* __restrict__ hold,
                        int64_t N, int64_t Npad, int D, int pitch) {
  const int64_t p = blockIdx.x;
  if (p >= N) return;
  for (int d = threadIdx.x; d < D; d += 64) {
    X[p * pitch + d] = TX[p * pitch + d];
    V[p * pitch + d] = TV[p * pitch + d];
    G[p * pitch + d] = TG[p * pitch + d];
  }
  if (threadIdx.x == 0) {
    EX[p] = hold[p];
    EV[p] = hold[Npad + p];
    Hflf[p] = hold[2 * Npad + p];
  }
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

int upload_rows(mjhmc_sampler* s, const double* host, int64_t n, double* dst) {
  TRY(ensure_stage(s, (size_t)s->D * n));
  HIPCHK(hipMemcpyAsync(s->stage, host, (size_t)s->D * n * sizeof(double), hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(hk_to_rows, grid1((int64_t)s->D * n), dim3(256), 0, s->stream, s->stage, dst, s->D, n, s->sh.pitch);
  HIPCHK(hipGetLastError());
  return 0;
}

int wide_energy_of(const mjhmc_energy* en, WideEnergy* w) {
  const EnergyParams& ep = en->ep;
  w->kind = ep.kind;
  w->a = w->b = w->c = 0.0;
  w->jdiag = (const double*)ep.dev_f64;
  switch (ep.kind) {
    case MJHMC_E_ISO_GAUSS: w->a = 1.0 / (ep.p[0] * ep.p[0]); w->b = 1.0 / (2.0 * (ep.p[0] * ep.p[0])); return 0;
    case MJHMC_E_DIAG_GAUSS: return 0;
    case MJHMC_E_ROUGH_WELL: w->a = ep.p[0] * ep.p[0]; w->b = 2.0 * (ep.p[0] * ep.p[0]); w->c = ep.p[1]; return 0;
    case MJHMC_E_MM_GAUSS: w->a = 2.0 * ep.p[0]; return 0;
    case MJHMC_E_FUNNEL_NEAL: w->a = 1.0 / (ep.p[0] * ep.p[0]); w->b = 0.5 * (ep.ndims - 1); return 0;
    case MJHMC_E_FUNNEL_REF: w->a = 1.0 / (ep.p[0] * ep.p[0]); w->b = (double)(ep.ndims - 1); return 0;
    default: return mjhmc_fail(MJHMC_ERR_UNSUPPORTED, "ndims too large for this energy's register-resident kernels, and it has no multi-pass form");
  }
}

__global__ void hk_narrow(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
__global__ void hk_widen(const float* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}

// One leapfrog step of ProductOfT's float64-state path in ONE pass over the rows: [closing half kick of the previous
// step] + opening half kick + drift + the float32 copy the force kernel reads.  Same operations in the same order as the
// separate passes (every product rounded before its sum; two half kicks stay two roundings), 2 GB instead of 4.5 GB of
// HBM traffic per step at 512 x 100 000.
template <bool G32>
__global__ void hk_pot_kick_drift(double* __restrict__ X, double* __restrict__ V, const void* __restrict__ Gsrc,
                                  float* __restrict__ X32, double c, double eps, int closing, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double g = G32 ? (double)((const float*)Gsrc)[i] : ((const double*)Gsrc)[i];
  double v = V[i];
  if (closing) v = v + c * g;
  v = v + c * g;
  const double x = X[i] + eps * v;
  V[i] = v;
  X[i] = x;
  X32[i] = (float)x;
}
// closing half kick from the float32 force; the stored dE/dX in float64
__global__ void hk_pot_close(double* __restrict__ V, const float* __restrict__ G32, double* __restrict__ G64, double c, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double g = (double)G32[i];
  V[i] = V[i] + c * g;
  G64[i] = g;
}

// the float32 force on rows buf[0] ([rows_pad][pitch] float32) -> dE/dX rows buf[1] (or none) and E buf[2] (or none): the
// tile kernel's evaluation up to 512 dims, the blocked evaluation (buf[3]: its u rows) beyond
static void pot_force32(mjhmc_sampler* s, float* const* buf, bool want_G, bool want_E, int64_t rows_pad, int64_t nrows) {
  if (s->en->pot_big()) {
    pot_big_eval(s->en->pot_big_model(), buf[0], want_G ? buf[1] : nullptr, want_E ? buf[2] : nullptr, buf[3], rows_pad, s->stream);
    return;
  }
  PotEvalArgs a;
  a.X = buf[0];
  a.G = want_G ? buf[1] : nullptr;
  a.E = want_E ? buf[2] : nullptr;
  a.EV = nullptr;
  a.V = nullptr;
  a.V_gen = nullptr;
  a.N = nrows;
  a.ntiles = rows_pad / 32;
  a.first_pid = 0;
  a.D = s->D;
  a.key = RngKey{0u, 0u, 0u, 0u};
  pot_launch_eval(a, s->en->pot_model(), s->stream);
}

// ProductOfT as the reference runs it (distributions.py:408-415 with hmc_state.py:29-38): float64 HMCState arrays around
// a float32 force -- the inputs are downcast (allow_input_downcast=True), E and dE/dX come back as float32.  The force is
// the float32 matrix-core evaluation kernel of dense_pot.hip on a float32 copy of the rows.
int pot_eval_rows_f64(mjhmc_sampler* s, const double* X, double* G, double* E, int64_t nrows) {
  const int pitch = s->sh.pitch;
  const int64_t rows_pad = (nrows + 63) / 64 * 64, ne = rows_pad * pitch;
  float* tmp[4] = {nullptr, nullptr, nullptr, nullptr};
  float** buf = tmp;
  const bool big = s->en->pot_big();
  const bool own = !(s->ht && rows_pad <= 2 * s->Npad);
  if (!own) {
    buf = s->ht->pot32;
    if (!buf[0]) {
      const size_t cap = (size_t)2 * s->Npad;
      HIPCHK(hipMalloc((void**)&buf[0], cap * pitch * sizeof(float)));
      HIPCHK(hipMalloc((void**)&buf[1], cap * pitch * sizeof(float)));
      HIPCHK(hipMalloc((void**)&buf[2], cap * sizeof(float)));
      if (big) HIPCHK(hipMalloc((void**)&buf[3], cap * pitch * sizeof(float)));
    }
  } else {
    HIPCHK(hipMalloc((void**)&buf[0], (size_t)ne * sizeof(float)));
    HIPCHK(hipMalloc((void**)&buf[1], (size_t)ne * sizeof(float)));
    HIPCHK(hipMalloc((void**)&buf[2], (size_t)rows_pad * sizeof(float)));
    if (big) HIPCHK(hipMalloc((void**)&buf[3], (size_t)ne * sizeof(float)));
  }
  auto body = [&]() -> int {
    HIPCHK(hipMemsetAsync(buf[0], 0, (size_t)ne * sizeof(float), s->stream));  // rows beyond nrows: zeros, never NaN garbage
    hipLaunchKernelGGL(hk_narrow, grid1(nrows * pitch), dim3(256), 0, s->stream, X, buf[0], nrows * pitch);
    pot_force32(s, buf, G != nullptr, E != nullptr, rows_pad, nrows);
    if (G) hipLaunchKernelGGL(hk_widen, grid1(nrows * pitch), dim3(256), 0, s->stream, (const float*)buf[1], G, nrows * pitch);
    if (E) hipLaunchKernelGGL(hk_widen, grid1(nrows), dim3(256), 0, s->stream, (const float*)buf[2], E, nrows);
    HIPCHK(hipGetLastError());
    if (own) HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
  };
  const int rc = body();
  if (own)
    for (float* b : tmp)
      if (b) (void)hipFree(b);
  return rc;
}

// E (nrows) and dE/dX (rows) of rows X on the device; either output may be NULL
int wide_eval_rows(mjhmc_sampler* s, const double* X, double* G, double* E, int64_t nrows) {
  if (s->en->is_pot()) return pot_eval_rows_f64(s, X, G, E, nrows);
  WideEnergy w;
  TRY(wide_energy_of(s->en, &w));
  hipLaunchKernelGGL(hk_energy_grad, dim3((unsigned)nrows), dim3(64), 0, s->stream, X, G, E, nrows, s->D, s->sh.pitch, w);
  HIPCHK(hipGetLastError());
  return 0;
}

int need_traj(mjhmc_sampler* s) {
  if (s->ht) return 0;
  HostTraj* t = new HostTraj();
  s->ht = t;
  const size_t rows = (size_t)2 * s->Npad, mb = rows * s->sh.pitch * sizeof(double);
  HIPCHK(hipMalloc((void**)&t->X, mb));
  HIPCHK(hipMalloc((void**)&t->V, mb));
  HIPCHK(hipMalloc((void**)&t->G, mb));
  HIPCHK(hipMemsetAsync(t->X, 0, mb, s->stream));
  HIPCHK(hipMemsetAsync(t->V, 0, mb, s->stream));
  HIPCHK(hipMemsetAsync(t->G, 0, mb, s->stream));
  HIPCHK(hipMalloc((void**)&t->E, rows * sizeof(double)));
  HIPCHK(hipMalloc((void**)&t->EVw, rows * sizeof(double)));
  HIPCHK(hipMalloc((void**)&t->cold, s->Npad * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t->coldpos, s->Npad * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t->kmove, s->Npad * sizeof(int)));
  HIPCHK(hipMalloc((void**)&t->dtmp, s->Npad * sizeof(double)));
  HIPCHK(hipMalloc((void**)&t->hnew, (size_t)3 * s->Npad * sizeof(double)));
  return 0;
}

int check_host(mjhmc_sampler* s) {
  if (!s) return mjhmc_fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (!s->en->is_host()) return mjhmc_fail(MJHMC_ERR_INVALID, "the sampler's energy is not MJHMC_E_HOST");
  return 0;
}

}  // namespace

void host_traj_free(mjhmc_sampler* s) {
  HostTraj* t = s->ht;
  if (!t) return;
  void* bufs[] = {t->X, t->V, t->G, t->E, t->EVw, t->noise, t->cold, t->coldpos, t->kmove, t->dtmp, t->hnew,
                  t->pot32[0], t->pot32[1], t->pot32[2], t->pot32[3]};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete t;
  s->ht = nullptr;
}

// run_eval of api.hip for a host energy: the kinetic energy of V (or the tick-0 momentum) is device work, E and dE/dX
// are the caller's (mjhmc_host_set_energy)
int host_run_eval(mjhmc_sampler* s, const void* V, void* Vgen, void* EVout) {
  if (Vgen) {
    const RngKey key{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), 0u, 0u};
    const int64_t n = s->N * ((s->D + 1) / 2);
    hipLaunchKernelGGL(hk_gen_v, grid1(n), dim3(256), 0, s->stream, (double*)Vgen, key, s->first_pid, s->N, s->D, s->sh.pitch);
    HIPCHK(hipGetLastError());
    V = Vgen;
  }
  if (V && EVout) {
    hipLaunchKernelGGL(hk_kinetic, dim3((unsigned)s->N), dim3(64), 0, s->stream, (const double*)V, (double*)EVout, s->N, s->D,
                       s->sh.pitch);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

extern "C" {

int mjhmc_host_set_energy(mjhmc_sampler* s, const double* E, const double* dEdX) {
  TRY(check_host(s));
  if (!E || !dEdX) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemcpyAsync(s->EX[s->scur], E, (size_t)s->N * sizeof(double), hipMemcpyHostToDevice, s->stream));
  TRY(upload_rows(s, dEdX, s->N, (double*)s->Gbuf[s->vcur]));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->host_energy_set = true;
  return 0;
}

int mjhmc_traj_begin(mjhmc_sampler* s, int64_t* n_cols) {
  TRY(check_host(s));
  if (!n_cols) return mjhmc_fail(MJHMC_ERR_INVALID, "n_cols is NULL");
  if (!s->host_energy_set)
    return mjhmc_fail(MJHMC_ERR_INVALID, "E and dE/dX of the current state are unknown: call mjhmc_host_set_energy first");
  HIPCHK(hipSetDevice(s->ctx->device));
  return traj_begin_impl(s, n_cols);
}

}  // extern "C"

int traj_begin_impl(mjhmc_sampler* s, int64_t* n_cols) {
  TRY(need_traj(s));
  s->undo_valid = false;   // the workspace is about to be overwritten
  HostTraj* t = s->ht;
  const int pitch = s->sh.pitch;
  t->n_cold = 0;
  if (s->mode == MJHMC_MODE_MJHMC) {  // the cold list on the host, ascending (the order of NumPy's boolean mask)
    std::vector<double> h((size_t)s->N);
    HIPCHK(hipMemcpyAsync(h.data(), s->Hflf[s->scur], (size_t)s->N * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<int> cold, pos((size_t)s->N, -1);
    for (int64_t p = 0; p < s->N; ++p)
      if (!(h[(size_t)p] == h[(size_t)p])) {
        pos[(size_t)p] = (int)cold.size();
        cold.push_back((int)p);
      }
    t->n_cold = (int)cold.size();
    if (t->n_cold) HIPCHK(hipMemcpyAsync(t->cold, cold.data(), cold.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(t->coldpos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));  // the vectors go out of scope
  }
  const double* X = (const double*)s->Xcur;
  const double* V = (const double*)s->Vbuf[s->vcur];
  const double* G = (const double*)s->Gbuf[s->vcur];
  const int64_t n1 = s->N * pitch;
  hipLaunchKernelGGL(hk_copy_rows, grid1(n1), dim3(256), 0, s->stream, t->X, X, (const int*)nullptr, s->N, pitch, 1.0);
  hipLaunchKernelGGL(hk_copy_rows, grid1(n1), dim3(256), 0, s->stream, t->V, V, (const int*)nullptr, s->N, pitch, 1.0);
  hipLaunchKernelGGL(hk_copy_rows, grid1(n1), dim3(256), 0, s->stream, t->G, G, (const int*)nullptr, s->N, pitch, 1.0);
  if (t->n_cold) {
    const int64_t n2 = (int64_t)t->n_cold * pitch;
    const size_t off = (size_t)s->N * pitch;
    hipLaunchKernelGGL(hk_copy_rows, grid1(n2), dim3(256), 0, s->stream, t->X + off, X, (const int*)t->cold, (int64_t)t->n_cold, pitch, 1.0);
    hipLaunchKernelGGL(hk_copy_rows, grid1(n2), dim3(256), 0, s->stream, t->V + off, V, (const int*)t->cold, (int64_t)t->n_cold, pitch, -1.0);
    hipLaunchKernelGGL(hk_copy_rows, grid1(n2), dim3(256), 0, s->stream, t->G + off, G, (const int*)t->cold, (int64_t)t->n_cold, pitch, 1.0);
  }
  HIPCHK(hipGetLastError());
  t->n_cols = s->N + t->n_cold;
  t->phase = 1;
  t->steps = 0;
  if (n_cols) *n_cols = t->n_cols;
  return 0;
}

extern "C" {

int mjhmc_traj_step(mjhmc_sampler* s, const double* grad, int last, double* X_out) {
  TRY(check_host(s));
  HostTraj* t = s->ht;
  if (!t || t->phase != 1) return mjhmc_fail(MJHMC_ERR_INVALID, "no trajectory in progress (mjhmc_traj_begin)");
  if (!last && !X_out) return mjhmc_fail(MJHMC_ERR_INVALID, "X_out is NULL");
  if (t->steps > 0 && !grad) return mjhmc_fail(MJHMC_ERR_INVALID, "the gradient at the positions handed out last is missing");
  HIPCHK(hipSetDevice(s->ctx->device));
  const int64_t n = t->n_cols, ne = n * s->sh.pitch;
  const double c = -s->eps / 2.;  // hmc_state.py:88
  if (grad) {  // closes the step the caller evaluated the gradient for
    if (t->steps == 0) return mjhmc_fail(MJHMC_ERR_INVALID, "the first step uses the stored dE/dX: pass NULL");
    TRY(upload_rows(s, grad, n, t->G));
    hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->V, (const double*)t->G, c, ne);
  }
  if (!last) {
    hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->V, (const double*)t->G, c, ne);
    hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->X, (const double*)t->V, s->eps, ne);
    HIPCHK(hipGetLastError());
    t->steps += 1;
    TRY(ensure_stage(s, (size_t)s->D * n));
    return download_cols(s, t->X, nullptr, n, X_out, (size_t)s->D * n, n, 1, 0, true);
  }
  HIPCHK(hipGetLastError());
  t->phase = 2;
  return 0;
}

int mjhmc_traj_finish(mjhmc_sampler* s, const double* E, const double* replay_normal, const double* replay_exp,
                      const double* replay_unif, int ring_slot, mjhmc_iter_stats* st) {
  TRY(check_host(s));
  HostTraj* t = s->ht;
  if (!t || t->phase != 2) return mjhmc_fail(MJHMC_ERR_INVALID, "the trajectory is not complete (mjhmc_traj_step with last != 0)");
  if (!E) return mjhmc_fail(MJHMC_ERR_INVALID, "E is NULL");
  if (ring_slot >= s->ring_slots) return mjhmc_fail(MJHMC_ERR_INVALID, "ring slot out of range");
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemcpyAsync(t->E, E, (size_t)t->n_cols * sizeof(double), hipMemcpyHostToDevice, s->stream));
  return traj_finish_impl(s, replay_normal, replay_exp, replay_unif, ring_slot, st);
}

}  // extern "C"

// t->E holds the energies of the end points (device); decide + commit
int traj_finish_impl(mjhmc_sampler* s, const double* replay_normal, const double* replay_exp, const double* replay_unif,
                     int ring_slot, mjhmc_iter_stats* st) {
  HostTraj* t = s->ht;
  const int64_t n = t->n_cols;
  const int pitch = s->sh.pitch;
  const bool replay = s->mode == MJHMC_MODE_CONTROL ? (replay_unif != nullptr) : (replay_exp != nullptr);
  if (replay && !replay_normal) return mjhmc_fail(MJHMC_ERR_INVALID, "replay needs the normals as well");
  hipLaunchKernelGGL(hk_kinetic, dim3((unsigned)n), dim3(64), 0, s->stream, (const double*)t->V, t->EVw, n, s->D, pitch);
  if (replay) {
    if (!t->noise) HIPCHK(hipMalloc((void**)&t->noise, (size_t)s->Npad * pitch * sizeof(double)));
    TRY(upload_rows(s, replay_normal, s->N, t->noise));
    if (replay_exp) {
      if (!s->rexp) HIPCHK(hipMalloc((void**)&s->rexp, 3 * s->N * sizeof(double)));
      HIPCHK(hipMemcpyAsync(s->rexp, replay_exp, 3 * s->N * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    if (replay_unif) {
      if (!s->runif) HIPCHK(hipMalloc((void**)&s->runif, (2 * s->N + 1) * sizeof(double)));
      HIPCHK(hipMemcpyAsync(s->runif, replay_unif, (2 * s->N + 1) * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
  }
  HIPCHK(hipMemsetAsync(s->ctl, 0, sizeof(Control), s->stream));
  HIPCHK(hipMemsetAsync(s->stats, 0, 4 * sizeof(long long), s->stream));
  const uint64_t tick = s->tick;
  const RngKey key{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), (uint32_t)(tick & 0xFFFFFFFFu),
                   (uint32_t)(tick >> 32)};
  DecideArgs d;
  d.EX = (const double*)s->EX[s->scur];
  d.EV = (const double*)s->EV[s->scur];
  d.Hflf = (const double*)s->Hflf[s->scur];
  d.E = t->E;
  d.EVw = t->EVw;
  d.coldpos = t->coldpos;
  d.kmove = t->kmove;
  d.dtmp = t->dtmp;
  d.hnew = t->hnew;
  d.rexp = replay ? s->rexp : nullptr;
  d.runif = replay ? s->runif : nullptr;
  d.ctl = s->ctl;
  d.N = s->N;
  d.Npad = s->Npad;
  d.first_pid = s->first_pid;
  d.p_r = s->p_r;
  d.p_flip = s->p_flip;
  d.mode = s->mode;
  d.key = key;
  if (replay) hipLaunchKernelGGL(hk_decide<true>, grid1(s->N), dim3(256), 0, s->stream, d);
  else hipLaunchKernelGGL(hk_decide<false>, grid1(s->N), dim3(256), 0, s->stream, d);
  HIPCHK(hipGetLastError());
  Control hc;
  HIPCHK(hipMemcpyAsync(&hc, s->ctl, sizeof(Control), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->tick += 1;  // one tick per attempt, committed or not (a retry redraws)
  t->phase = 0;
  long long hs[4] = {0, 0, 0, 0};
  if (!hc.failed) {
    CommitArgs c;
    c.X = (double*)s->Xcur;
    c.V = (double*)s->Vbuf[s->vcur];
    c.G = (double*)s->Gbuf[s->vcur];
    c.EXs = (double*)s->EX[s->scur];
    c.EVs = (double*)s->EV[s->scur];
    c.Hflf = (double*)s->Hflf[s->scur];
    c.TX = t->X;
    c.TV = t->V;
    c.TG = t->G;
    c.kmove = t->kmove;
    c.dtmp = t->dtmp;
    c.hnew = t->hnew;
    c.noise = replay ? t->noise : nullptr;
    c.dwell = s->dwell;
    c.dwell_ring = ring_slot >= 0 ? s->dwell_ring + (size_t)ring_slot * s->Npad : nullptr;
    c.trans = s->trans;
    c.ring_slot = ring_slot >= 0 ? (double*)((char*)s->ring + (size_t)ring_slot * mat_bytes(s)) : nullptr;
    c.stats = (unsigned long long*)s->stats;
    c.N = s->N;
    c.Npad = s->Npad;
    c.first_pid = s->first_pid;
    c.D = s->D;
    c.pitch = pitch;
    c.mode = s->mode;
    c.round32 = s->sh.round32 ? 1 : 0;
    c.r_keep = std::sqrt(1.0 - s->beta);  // hmc_state.py:125-126
    c.r_mix = std::sqrt(s->beta);
    c.key = key;
    if (replay) hipLaunchKernelGGL(hk_commit<true>, dim3((unsigned)s->N), dim3(64), 0, s->stream, c);
    else hipLaunchKernelGGL(hk_commit<false>, dim3((unsigned)s->N), dim3(64), 0, s->stream, c);
    HIPCHK(hipGetLastError());
    if (s->sh.round32 && !s->en->is_host()) {
      // float32-valued state: E, dE/dX and the kinetic energy are those of what is stored (as the float32 register
      // kernels report them); one more evaluation per iteration on this fallback path, not a counted one
      TRY(wide_eval_rows(s, (const double*)s->Xcur, (double*)s->Gbuf[s->vcur], (double*)s->EX[s->scur], s->N));
      hipLaunchKernelGGL(hk_kinetic, dim3((unsigned)s->N), dim3(64), 0, s->stream, (const double*)s->Vbuf[s->vcur],
                         (double*)s->EV[s->scur], s->N, s->D, pitch);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(hs, s->stats, sizeof(hs), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
  }
  // a committed attempt can be undone until the next one begins: hk_commit left the pre-move state in the workspace
  s->undo_valid = !hc.failed;
  s->undo_multipass = true;
  if (st) {
    std::memset(st, 0, sizeof(*st));
    if (s->mode == MJHMC_MODE_MJHMC) {
      st->l = hs[0];
      st->f = hs[1];
      st->r = hs[2];
      st->n_cold = t->n_cold;
    } else if (s->mode == MJHMC_MODE_CTHMC) {
      st->fl = hs[0];
      st->f = hs[1];
      st->r = hs[2];
    } else {
      st->l = hs[0];
      st->f = hs[1];
      st->r = hs[2];
      st->fl = hs[3];
    }
    st->E_evals = n;
    st->dEdX_evals = (int64_t)t->steps * n;
    st->nonfinite = hc.failed ? 1 : 0;
    st->L_used = t->steps;
    st->eps_used = s->eps;
    st->n_flf_run = st->n_cold;
  }
  return 0;
}

int multipass_rollback(mjhmc_sampler* s) {
  HostTraj* t = s->ht;
  if (!t) return mjhmc_fail(MJHMC_ERR_INVALID, "nothing to roll back");
  HIPCHK(hipSetDevice(s->ctx->device));   // (a process may hold contexts on several devices)
  hipLaunchKernelGGL(hk_undo, dim3((unsigned)s->N), dim3(64), 0, s->stream, (double*)s->Xcur, (double*)s->Vbuf[s->vcur],
                     (double*)s->Gbuf[s->vcur], (double*)s->EX[s->scur], (double*)s->EV[s->scur], (double*)s->Hflf[s->scur],
                     (const double*)t->X, (const double*)t->V, (const double*)t->G, (const double*)t->hnew, s->N, s->Npad, s->D,
                     s->sh.pitch);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

// ---- wide samplers: built-in elementwise energies with ndims beyond the register-resident kernels -------------------

// run_eval of api.hip for a host energy or a wide sampler: the kinetic energy of V (or the tick-0 momentum) always; E and
// dE/dX from the device kernel for a wide sampler, the caller's (mjhmc_host_set_energy) for a host energy
int wide_run_eval(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen, void* EVout) {
  if (!s->en->is_host() && X && (Gout || Eout)) TRY(wide_eval_rows(s, (const double*)X, (double*)Gout, (double*)Eout, s->N));
  return host_run_eval(s, V, Vgen, EVout);
}

// one kick / drift pass of the proposal columns: [V += c g_new]; unless last: V += c g; X += eps V
static int traj_advance(mjhmc_sampler* s, bool closing_kick, bool last) {
  HostTraj* t = s->ht;
  const int64_t ne = t->n_cols * s->sh.pitch;
  const double c = -s->eps / 2.;
  if (closing_kick) hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->V, (const double*)t->G, c, ne);
  if (!last) {
    hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->V, (const double*)t->G, c, ne);
    hipLaunchKernelGGL(hk_axpy, grid1(ne), dim3(256), 0, s->stream, t->X, (const double*)t->V, s->eps, ne);
    t->steps += 1;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// the L leapfrog steps of ProductOfT's float64-state path on the proposal columns: per step one fused row pass
// (hk_pot_kick_drift) and one float32 matrix-core force evaluation (pot_launch_eval); the first step's force is the stored
// float64 dE/dX, the later ones the float32 output of the previous evaluation, exactly as the separate passes use them
static int pot_trajectory_f64(mjhmc_sampler* s) {
  HostTraj* t = s->ht;
  const int pitch = s->sh.pitch;
  const int64_t n = t->n_cols, rows_pad = (n + 63) / 64 * 64, ne = n * pitch;
  if (!t->pot32[0]) {
    const size_t cap = (size_t)2 * s->Npad;
    HIPCHK(hipMalloc((void**)&t->pot32[0], cap * pitch * sizeof(float)));
    HIPCHK(hipMalloc((void**)&t->pot32[1], cap * pitch * sizeof(float)));
    HIPCHK(hipMalloc((void**)&t->pot32[2], cap * sizeof(float)));
    if (s->en->pot_big()) HIPCHK(hipMalloc((void**)&t->pot32[3], cap * pitch * sizeof(float)));
  }
  HIPCHK(hipMemsetAsync(t->pot32[0], 0, (size_t)rows_pad * pitch * sizeof(float), s->stream));  // rows beyond n: zeros
  const double c = -s->eps / 2.;
  for (int l = 0; l < s->L; ++l) {
    if (l == 0)
      hipLaunchKernelGGL(hk_pot_kick_drift<false>, grid1(ne), dim3(256), 0, s->stream, t->X, t->V, (const void*)t->G, t->pot32[0], c,
                         s->eps, 0, ne);
    else
      hipLaunchKernelGGL(hk_pot_kick_drift<true>, grid1(ne), dim3(256), 0, s->stream, t->X, t->V, (const void*)t->pot32[1],
                         t->pot32[0], c, s->eps, 1, ne);
    pot_force32(s, t->pot32, true, l == s->L - 1, rows_pad, n);
    t->steps += 1;
  }
  hipLaunchKernelGGL(hk_pot_close, grid1(ne), dim3(256), 0, s->stream, t->V, (const float*)t->pot32[1], t->G, c, ne);
  hipLaunchKernelGGL(hk_widen, grid1(n), dim3(256), 0, s->stream, (const float*)t->pot32[2], t->E, n);
  HIPCHK(hipGetLastError());
  return 0;
}

// mjhmc_iterate for a wide sampler: n_iter sampling iterations, each L x (kick, drift, device gradient) over the proposal
// columns in HBM, then decide + commit -- the contract of iterate_t (stops at the first attempt with a non-finite rate)
int multipass_iterate(mjhmc_sampler* s, int n_iter, const double* replay_normal, const double* replay_exp,
                      const double* replay_unif, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done) {
  const size_t DN = (size_t)s->D * s->N;
  int done = 0;
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[0], s->stream));
  for (int i = 0; i < n_iter; ++i) {
    TRY(traj_begin_impl(s, nullptr));
    HostTraj* t = s->ht;
    if (s->en->is_pot() && s->L > 0) {
      TRY(pot_trajectory_f64(s));   // fused kick / drift / downcast passes around the float32 force kernel
    } else {
      for (int l = 0; l < s->L; ++l) {
        TRY(traj_advance(s, l > 0, false));
        TRY(wide_eval_rows(s, t->X, t->G, l == s->L - 1 ? t->E : nullptr, t->n_cols));
      }
      TRY(traj_advance(s, s->L > 0, true));
      if (s->L == 0) TRY(wide_eval_rows(s, t->X, nullptr, t->E, t->n_cols));
    }
    t->phase = 2;
    mjhmc_iter_stats st;
    TRY(traj_finish_impl(s, replay_normal ? replay_normal + (size_t)i * DN : nullptr,
                         replay_exp ? replay_exp + (size_t)i * 3 * s->N : nullptr,
                         replay_unif ? replay_unif + (size_t)i * (2 * s->N + 1) : nullptr,
                         ring_slot0 >= 0 ? ring_slot0 + i : -1, &st));
    if (per_iter) per_iter[i] = st;
    if (st.nonfinite) break;
    done += 1;
  }
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[1], s->stream));
  s->undo_valid = (n_iter == 1 && done == 1);   // mjhmc_rollback's contract: the last call was ONE committed iteration
  s->last_jump_launches = std::min(done + 1, n_iter);
  s->timing_pending = s->timing_on;
  if (n_done) *n_done = done;
  return 0;
}

// mjhmc_leapfrog for a wide shape: n_steps of hmc_state.py:86-100 on rows X, V (device), literal operation order
int wide_leapfrog(mjhmc_sampler* w, const double* X, const double* V, double* Xo, double* Vo, double* G, double* EX,
                  double* EV, double eps, int n_steps) {
  const int64_t ne = w->Npad * w->sh.pitch, nv = w->N * w->sh.pitch;
  double* g = G;
  if (!g) HIPCHK(hipMalloc((void**)&g, (size_t)ne * sizeof(double)));
  auto body = [&]() -> int {
    HIPCHK(hipMemcpyAsync(Xo, X, (size_t)ne * sizeof(double), hipMemcpyDeviceToDevice, w->stream));
    HIPCHK(hipMemcpyAsync(Vo, V, (size_t)ne * sizeof(double), hipMemcpyDeviceToDevice, w->stream));
    TRY(wide_eval_rows(w, Xo, g, EX, w->N));
    const double c = -eps / 2.;
    for (int l = 0; l < n_steps; ++l) {
      hipLaunchKernelGGL(hk_axpy, grid1(nv), dim3(256), 0, w->stream, Vo, (const double*)g, c, nv);
      hipLaunchKernelGGL(hk_axpy, grid1(nv), dim3(256), 0, w->stream, Xo, (const double*)Vo, eps, nv);
      TRY(wide_eval_rows(w, Xo, g, EX, w->N));
      hipLaunchKernelGGL(hk_axpy, grid1(nv), dim3(256), 0, w->stream, Vo, (const double*)g, c, nv);
    }
    if (w->sh.round32) {   // float32-valued state: the end point is rounded, its energies and dE/dX are those of what is handed out
      TRY(round_rows32(w, Xo));
      TRY(round_rows32(w, Vo));
      TRY(wide_eval_rows(w, Xo, g, EX, w->N));
    }
    if (EV) hipLaunchKernelGGL(hk_kinetic, dim3((unsigned)w->N), dim3(64), 0, w->stream, (const double*)Vo, EV, w->N, w->D, w->sh.pitch);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(w->stream));
    return 0;
  };
  const int rc = body();
  if (!G && g) (void)hipFree(g);
  return rc;
}
