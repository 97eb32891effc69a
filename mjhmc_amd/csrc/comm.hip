// Multi-GPU entry points of the C ABI (include/mjhmc_hip.h, "one process per GPU"): a communicator over RCCL --
// loaded with dlopen, so single-GPU use never touches it and the library has no link-time dependency on it -- the
// small host-value collectives the sampler needs between ranks (integer tallies, the non-finite flag, uniforms of the
// resampling step), and THE data-path collective of the whole path: the all-gather of sample columns at the end of
// sample() (markov_jump_hmc.py:150-173, 293-338), device ring to device ring over xGMI, re-tiled to the reference's
// (ndims, columns) layout once, on the receiving GPU.
//
// Every all-gather pads the per-rank blocks to the largest one and runs ONE ncclAllGather: column shards differ by at
// most one column (mjhmc_amd/parallel.py: ShardPlan), so the padding is noise, and there is a single code path whatever
// the rank count.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "handles.hpp"

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  if (r.so || !r.err.empty()) return r;
  // MJHMC_RCCL_LIB names the library to load and is then the ONLY name tried; otherwise the sonames of the ROCm install
  const char* named = std::getenv("MJHMC_RCCL_LIB");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  if (named && *named) {
    r.so = dlopen(named, RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char* n : names) {
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
  }
  if (!r.so) {
    const char* why = dlerror();  // ONE call: dlerror() clears the message it returns
    r.err = std::string("librccl.so could not be loaded: ") + (why ? why : "?");
    return r;
  }
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(r.so, name);
    if (!p && r.err.empty()) r.err = std::string("librccl.so lacks ") + name;
    return p;
  };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  return r;
}

}  // namespace

struct mjhmc_comm {
  mjhmc_ctx* ctx;
  ncclComm_t nccl = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;
  void* buf[2] = {nullptr, nullptr};  // grow-only device scratch: [0] send side, [1] receive side
  size_t cap[2] = {0, 0};
};

#define NCCLCHK(expr)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess)                                                                              \
      return mjhmc_fail(MJHMC_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (" +  \
                                            __FILE__ + ":" + std::to_string(__LINE__) + ")");           \
  } while (0)

static int need(mjhmc_comm* c, int which, size_t bytes) {
  if (c->cap[which] >= bytes) return 0;
  if (c->buf[which]) HIPCHK(hipFree(c->buf[which]));
  c->buf[which] = nullptr;
  c->cap[which] = 0;
  const size_t want = std::max<size_t>(bytes, 4096);
  HIPCHK(hipMalloc(&c->buf[which], want));
  c->cap[which] = want;
  return 0;
}

// rows [r][pitch] of the sample ring picked by index (16-byte chunks; pitch * sizeof(T) is a multiple of 16)
__global__ void gather_rows_kernel(const uint4* __restrict__ src, const int64_t* __restrict__ rows, uint4* __restrict__ dst,
                                   int64_t n, int chunks_per_row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = i / chunks_per_row;
  if (k >= n) return;
  const int c = (int)(i - k * chunks_per_row);
  dst[k * chunks_per_row + c] = src[rows[k] * chunks_per_row + c];
}

extern "C" {

int mjhmc_comm_unique_id(void* id) {
  if (!id) return mjhmc_fail(MJHMC_ERR_INVALID, "id is NULL");
  Rccl& r = rccl();
  if (!r.err.empty()) return mjhmc_fail(MJHMC_ERR_COMM, r.err);
  static_assert(sizeof(ncclUniqueId) == MJHMC_COMM_ID_BYTES, "unique id size");
  ncclUniqueId u;
  NCCLCHK(r.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int mjhmc_comm_create(mjhmc_ctx* ctx, int rank, int world, const void* id, mjhmc_comm** out) {
  if (!ctx || !id || !out) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return mjhmc_fail(MJHMC_ERR_INVALID, "rank / world out of range");
  Rccl& r = rccl();
  if (!r.err.empty()) return mjhmc_fail(MJHMC_ERR_COMM, r.err);
  HIPCHK(hipSetDevice(ctx->device));
  mjhmc_comm* c = new mjhmc_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  auto body = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    NCCLCHK(r.CommInitRank(&c->nccl, world, u, rank));
    return 0;
  };
  const int rc = body();
  if (rc) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

int mjhmc_comm_destroy(mjhmc_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->ctx->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  for (void* b : c->buf)
    if (b) (void)hipFree(b);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

static int allreduce_host(mjhmc_comm* c, void* inout, int64_t n, ncclDataType_t dt, int op) {
  if (!c || (!inout && n) || n < 0) return mjhmc_fail(MJHMC_ERR_INVALID, "bad argument");
  if (op < MJHMC_OP_SUM || op > MJHMC_OP_MAX) return mjhmc_fail(MJHMC_ERR_INVALID, "unknown reduction");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->ctx->device));
  const size_t bytes = (size_t)n * 8;
  TRY(need(c, 0, bytes));
  HIPCHK(hipMemcpyAsync(c->buf[0], inout, bytes, hipMemcpyHostToDevice, c->stream));
  const ncclRedOp_t rop = op == MJHMC_OP_SUM ? ncclSum : (op == MJHMC_OP_MIN ? ncclMin : ncclMax);
  NCCLCHK(rccl().AllReduce(c->buf[0], c->buf[0], (size_t)n, dt, rop, c->nccl, c->stream));
  HIPCHK(hipMemcpyAsync(inout, c->buf[0], bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mjhmc_comm_allreduce_i64(mjhmc_comm* c, int64_t* inout, int64_t n, int op) {
  return allreduce_host(c, inout, n, ncclInt64, op);
}

int mjhmc_comm_allreduce_f64(mjhmc_comm* c, double* inout, int64_t n, int op) {
  return allreduce_host(c, inout, n, ncclFloat64, op);
}

int mjhmc_comm_bcast(mjhmc_comm* c, void* inout, size_t nbytes, int root) {
  if (!c || (!inout && nbytes)) return mjhmc_fail(MJHMC_ERR_INVALID, "bad argument");
  if (root < 0 || root >= c->world) return mjhmc_fail(MJHMC_ERR_INVALID, "root out of range");
  if (nbytes == 0) return 0;
  HIPCHK(hipSetDevice(c->ctx->device));
  TRY(need(c, 0, nbytes));
  if (c->rank == root) HIPCHK(hipMemcpyAsync(c->buf[0], inout, nbytes, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(rccl().Broadcast(c->buf[0], c->buf[0], nbytes, ncclInt8, root, c->nccl, c->stream));
  if (c->rank != root) HIPCHK(hipMemcpyAsync(inout, c->buf[0], nbytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mjhmc_comm_allgatherv(mjhmc_comm* c, const void* send, const int64_t* nbytes_per_rank, void* recv) {
  if (!c || !nbytes_per_rank || !recv) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(c->ctx->device));
  size_t mx = 0;
  for (int r = 0; r < c->world; ++r) {
    if (nbytes_per_rank[r] < 0) return mjhmc_fail(MJHMC_ERR_INVALID, "negative block size");
    mx = std::max(mx, (size_t)nbytes_per_rank[r]);
  }
  if (mx == 0) return 0;
  const size_t mine = (size_t)nbytes_per_rank[c->rank];
  if (mine && !send) return mjhmc_fail(MJHMC_ERR_INVALID, "send is NULL");
  const size_t blk = (mx + 15) / 16 * 16;
  TRY(need(c, 0, blk));
  TRY(need(c, 1, blk * c->world));
  if (mine) HIPCHK(hipMemcpyAsync(c->buf[0], send, mine, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(rccl().AllGather(c->buf[0], c->buf[1], blk, ncclInt8, c->nccl, c->stream));
  char* dst = (char*)recv;
  for (int r = 0; r < c->world; ++r) {
    const size_t nb = (size_t)nbytes_per_rank[r];
    if (nb) HIPCHK(hipMemcpyAsync(dst, (char*)c->buf[1] + (size_t)r * blk, nb, hipMemcpyDeviceToHost, c->stream));
    dst += nb;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- the sample all-gather in three steps: pack | collective | unpack ------------------------------------------------
// A rank's block of the ring gather is [n][mx][pitch] (slot-major, mx = the largest shard; the rows beyond the rank's
// own count stay unread); of the column gather [mx][pitch].  The receive side holds `world` such blocks, rank-major.

// pack: ring slots [slot0, slot0 + n) of one sampler, padding rows dropped (each slot holds its N valid rows first)
static int ring_pack(mjhmc_sampler* s, int slot0, int n, int64_t mx, void* dst) {
  const size_t rb = row_bytes(s), mb = mat_bytes(s);
  HIPCHK(hipMemcpy2DAsync(dst, (size_t)mx * rb, (const char*)s->ring + (size_t)slot0 * mb, mb, (size_t)s->N * rb, (size_t)n,
                          hipMemcpyDeviceToDevice, s->stream));
  return 0;
}

// unpack: `world` received blocks -> the unsharded sample block in the reference's layout, on the host.
//   stacked == 0: (D, n * total) time-major like np.concatenate(axis=1) of the per-iteration states
//   stacked != 0: (D, total, n) like np.stack(axis=-1)
// ONE launch for all world x n (rank, slot) blocks (until round 5: a re-tile launch per block -- 80 launches of a few
// microseconds each for the eight shards of a ten-sample gather); the ranks' column counts and offsets ride in the kernel
// argument (up to kMaxUnpackRanks ranks; beyond that the launch-per-block form).
}  // extern "C"
constexpr int kMaxUnpackRanks = 64;
struct UnpackShards {
  int64_t cnt[kMaxUnpackRanks], off[kMaxUnpackRanks];
};
template <typename T>
__global__ void ring_unpack_kernel(const char* __restrict__ recv, size_t blk, size_t slot_bytes, UnpackShards sh, int n,
                                   double* __restrict__ dst, int D, int pitch, int64_t total, int stacked) {
  __shared__ double tile[32][33];
  const int r = blockIdx.z / n, t = blockIdx.z % n;
  const int64_t cnt = sh.cnt[r], k0 = (int64_t)blockIdx.x * 32;
  if (k0 >= cnt) return;   // (the grid is sized for the largest shard)
  const T* src = reinterpret_cast<const T*>(recv + (size_t)r * blk + (size_t)t * slot_bytes);
  // element (d, column k of rank r, slot t) -> time-major: d * (n total) + t total + off + k; stacked: d * (total n) + (off + k) n + t
  const int64_t rs = (int64_t)n * total, cs = stacked ? n : 1, off = stacked ? sh.off[r] * n + t : (int64_t)t * total + sh.off[r];
  const int d0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kk = threadIdx.y + 8 * i;
    const int64_t k = k0 + kk;
    const int d = d0 + threadIdx.x;
    if (d < D && k < cnt) tile[kk][threadIdx.x] = (double)src[(size_t)k * pitch + d];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dd = threadIdx.y + 8 * i;
    const int d = d0 + dd;
    const int64_t k = k0 + threadIdx.x;
    if (d < D && k < cnt) dst[(size_t)d * rs + (size_t)k * cs + off] = tile[threadIdx.x][dd];
  }
}

extern "C" {

static int ring_unpack(mjhmc_sampler* s, const char* recv, size_t blk, int world, const int64_t* counts, int64_t total,
                       int64_t mx, int n, int stacked, double* host_out) {
  const size_t rb = row_bytes(s);
  const size_t elems = (size_t)s->D * total * n;
  TRY(ensure_stage(s, elems));
  if (world <= kMaxUnpackRanks && (int64_t)world * n <= 65535) {
    UnpackShards sh;
    int64_t off = 0;
    for (int r = 0; r < world; ++r) {
      sh.cnt[r] = counts[r];
      sh.off[r] = off;
      off += counts[r];
    }
    const dim3 grid((unsigned)((mx + 31) / 32), (unsigned)((s->D + 31) / 32), (unsigned)(world * n)), block(32, 8);
    if (s->dtype == MJHMC_F64)
      hipLaunchKernelGGL(ring_unpack_kernel<double>, grid, block, 0, s->stream, recv, blk, (size_t)mx * rb, sh, n, s->stage, s->D,
                         s->sh.pitch, total, stacked);
    else if (s->dtype == MJHMC_BF16)
      hipLaunchKernelGGL(ring_unpack_kernel<__bf16>, grid, block, 0, s->stream, recv, blk, (size_t)mx * rb, sh, n, s->stage, s->D,
                         s->sh.pitch, total, stacked);
    else
      hipLaunchKernelGGL(ring_unpack_kernel<float>, grid, block, 0, s->stream, recv, blk, (size_t)mx * rb, sh, n, s->stage, s->D,
                         s->sh.pitch, total, stacked);
    HIPCHK(hipGetLastError());
    return copy_to_host(s, s->stage, host_out, elems * sizeof(double));
  }
  int64_t off = 0;
  for (int r = 0; r < world; ++r) {
    const int64_t cnt = counts[r];
    for (int t = 0; t < n; ++t) {
      const char* src = recv + (size_t)r * blk + (size_t)t * mx * rb;
      if (!stacked)
        TRY(download_cols(s, src, nullptr, cnt, host_out, elems, (int64_t)n * total, 1, (int64_t)t * total + off, false));
      else
        TRY(download_cols(s, src, nullptr, cnt, host_out, elems, total * n, n, off * n + t, false));
    }
    off += cnt;
  }
  return copy_to_host(s, s->stage, host_out, elems * sizeof(double));
}

// pack of the column gather: the n_local ring columns this rank owns (local pool indices slot * N + column)
static int columns_pack(mjhmc_sampler* s, const int64_t* local_idx, int64_t n_local, void* dst, int64_t* dev_rows) {
  if (!n_local) return 0;
  const int64_t pool = (int64_t)s->ring_slots * s->N;
  std::vector<int64_t> rows((size_t)n_local);  // pool index (slot * N + p) -> padded ring row (slot * Npad + p)
  for (int64_t k = 0; k < n_local; ++k) {
    if (local_idx[k] < 0 || local_idx[k] >= pool) return mjhmc_fail(MJHMC_ERR_INVALID, "gather index outside the sample ring");
    rows[(size_t)k] = (local_idx[k] / s->N) * s->Npad + local_idx[k] % s->N;
  }
  // the row list is pageable host memory: the copy is done with it when hipMemcpyAsync returns
  HIPCHK(hipMemcpyAsync(dev_rows, rows.data(), (size_t)n_local * sizeof(int64_t), hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  const int cpr = (int)(row_bytes(s) / 16);
  const int64_t threads = n_local * cpr;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s->stream,
                     (const uint4*)s->ring, dev_rows, (uint4*)dst, n_local, cpr);
  HIPCHK(hipGetLastError());
  return 0;
}

// unpack of the column gather: rank-major (rank 0's columns, then rank 1's, ...), (D, total) on the host
static int columns_unpack(mjhmc_sampler* s, const char* recv, size_t blk, int world, const int64_t* counts, int64_t total,
                          double* host_out) {
  const size_t elems = (size_t)s->D * total;
  TRY(ensure_stage(s, elems));
  int64_t off = 0;
  for (int r = 0; r < world; ++r) {
    const int64_t cnt = counts[r];
    if (cnt) TRY(download_cols(s, recv + (size_t)r * blk, nullptr, cnt, host_out, elems, total, 1, off, false));
    off += cnt;
  }
  return copy_to_host(s, s->stage, host_out, elems * sizeof(double));
}

static int shard_totals(const int64_t* counts, int world, int64_t least, int64_t* total, int64_t* mx) {
  *total = 0;
  *mx = 0;
  for (int r = 0; r < world; ++r) {
    if (counts[r] < least)
      return mjhmc_fail(MJHMC_ERR_INVALID, least ? "every rank must own at least one particle" : "negative column count");
    *total += counts[r];
    *mx = std::max(*mx, counts[r]);
  }
  return 0;
}

int mjhmc_comm_allgather_ring(mjhmc_comm* c, mjhmc_sampler* s, int slot0, int n, int stacked,
                              const int64_t* particles_per_rank, double* host_out) {
  if (!c || !s || !particles_per_rank) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return mjhmc_fail(MJHMC_ERR_INVALID, "slots out of range");
  if (particles_per_rank[c->rank] != s->N) return mjhmc_fail(MJHMC_ERR_INVALID, "particles_per_rank[rank] != nparticles");
  HIPCHK(hipSetDevice(c->ctx->device));
  int64_t total, mx;
  TRY(shard_totals(particles_per_rank, c->world, 1, &total, &mx));
  const size_t blk = (size_t)n * mx * row_bytes(s);
  TRY(need(c, 0, blk));
  TRY(need(c, 1, blk * c->world));
  TRY(ring_pack(s, slot0, n, mx, c->buf[0]));
  NCCLCHK(rccl().AllGather(c->buf[0], c->buf[1], blk, ncclInt8, c->nccl, s->stream));
  if (!host_out) {   // this rank takes part in the collective and keeps nothing (the caller wants the block on one rank)
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
  }
  return ring_unpack(s, (const char*)c->buf[1], blk, c->world, particles_per_rank, total, mx, n, stacked, host_out);
}

int mjhmc_comm_allgather_columns(mjhmc_comm* c, mjhmc_sampler* s, const int64_t* local_idx, int64_t n_local,
                                 const int64_t* columns_per_rank, double* host_out) {
  if (!c || !s || !columns_per_rank) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  if (n_local < 0 || (n_local && !local_idx)) return mjhmc_fail(MJHMC_ERR_INVALID, "bad local index list");
  if (columns_per_rank[c->rank] != n_local) return mjhmc_fail(MJHMC_ERR_INVALID, "columns_per_rank[rank] != n_local");
  HIPCHK(hipSetDevice(c->ctx->device));
  int64_t total, mx;
  TRY(shard_totals(columns_per_rank, c->world, 0, &total, &mx));
  if (total == 0) return 0;
  const size_t blk = (size_t)mx * row_bytes(s);
  TRY(need(c, 0, blk + (size_t)std::max<int64_t>(n_local, 1) * sizeof(int64_t)));
  TRY(need(c, 1, blk * c->world));
  TRY(columns_pack(s, local_idx, n_local, c->buf[0], (int64_t*)((char*)c->buf[0] + blk)));
  NCCLCHK(rccl().AllGather(c->buf[0], c->buf[1], blk, ncclInt8, c->nccl, s->stream));
  if (!host_out) {   // as mjhmc_comm_allgather_ring
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
  }
  return columns_unpack(s, (const char*)c->buf[1], blk, c->world, columns_per_rank, total, host_out);
}

int mjhmc_comm_available(void) {
  Rccl& r = rccl();
  if (!r.err.empty()) return mjhmc_fail(MJHMC_ERR_COMM, r.err);
  return 0;
}

int mjhmc_comm_count(mjhmc_comm* c, int* count) {
  if (!c || !count) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  NCCLCHK(rccl().CommCount(c->nccl, count));
  return 0;
}

#ifdef MJHMC_TEST_HOOKS
// TEST HOOKS (libmjhmc_hip_test.so only): the pack and unpack steps of the two sample gathers with the collective
// replaced by "the world's samplers all live on THIS GPU": sampler r plays rank r, its packed block is written where
// ncclAllGather would have put it, and the REAL unpack runs on samplers[0]'s stream.  This executes the rank > 0
// offsets, the padding to the largest shard and both output layouts on a single-GPU box.
int mjhmc_test_gather_ring_local(mjhmc_sampler** samplers, int world, int slot0, int n, int stacked, double* host_out) {
  if (!samplers || world < 1 || !host_out) return mjhmc_fail(MJHMC_ERR_INVALID, "bad argument");
  mjhmc_sampler* s0 = samplers[0];
  std::vector<int64_t> counts((size_t)world);
  for (int r = 0; r < world; ++r) {
    mjhmc_sampler* s = samplers[r];
    if (!s || s->D != s0->D || s->dtype != s0->dtype || s->sh.pitch != s0->sh.pitch)
      return mjhmc_fail(MJHMC_ERR_INVALID, "samplers of one gather must share ndims and dtype");
    if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return mjhmc_fail(MJHMC_ERR_INVALID, "slots out of range");
    counts[(size_t)r] = s->N;
  }
  HIPCHK(hipSetDevice(s0->ctx->device));
  int64_t total, mx;
  TRY(shard_totals(counts.data(), world, 1, &total, &mx));
  const size_t blk = (size_t)n * mx * row_bytes(s0);
  char* recv = nullptr;
  HIPCHK(hipMalloc(&recv, blk * world));
  // padding rows must never be read: NaN bytes.  (hipMemset runs on the null stream, the samplers' streams do not wait
  // for it: synchronise before the packs.)
  int rc = (hipMemset(recv, 0xff, blk * world) == hipSuccess && hipDeviceSynchronize() == hipSuccess) ? 0 : mjhmc_fail(MJHMC_ERR_HIP, "hipMemset");
  for (int r = 0; r < world && !rc; ++r) {
    rc = ring_pack(samplers[r], slot0, n, mx, recv + (size_t)r * blk);
    if (!rc && hipStreamSynchronize(samplers[r]->stream) != hipSuccess) rc = mjhmc_fail(MJHMC_ERR_HIP, "sync");
  }
  if (!rc) rc = ring_unpack(s0, recv, blk, world, counts.data(), total, mx, n, stacked, host_out);
  (void)hipFree(recv);
  return rc;
}

// local_idx: the ranks' index lists back to back (n_local[r] entries for rank r)
int mjhmc_test_gather_columns_local(mjhmc_sampler** samplers, int world, const int64_t* local_idx, const int64_t* n_local,
                                    double* host_out) {
  if (!samplers || world < 1 || !n_local || !host_out) return mjhmc_fail(MJHMC_ERR_INVALID, "bad argument");
  mjhmc_sampler* s0 = samplers[0];
  HIPCHK(hipSetDevice(s0->ctx->device));
  int64_t total, mx;
  TRY(shard_totals(n_local, world, 0, &total, &mx));
  if (total == 0) return 0;
  const size_t blk = (size_t)mx * row_bytes(s0);
  char* recv = nullptr;
  int64_t* rows = nullptr;
  HIPCHK(hipMalloc(&recv, blk * world));
  if (hipMalloc(&rows, (size_t)mx * sizeof(int64_t)) != hipSuccess) {
    (void)hipFree(recv);
    return mjhmc_fail(MJHMC_ERR_HIP, "hipMalloc");
  }
  int rc = (hipMemset(recv, 0xff, blk * world) == hipSuccess && hipDeviceSynchronize() == hipSuccess) ? 0 : mjhmc_fail(MJHMC_ERR_HIP, "hipMemset");
  const int64_t* idx = local_idx;
  for (int r = 0; r < world && !rc; ++r) {
    mjhmc_sampler* s = samplers[r];
    if (!s || s->D != s0->D || s->dtype != s0->dtype) rc = mjhmc_fail(MJHMC_ERR_INVALID, "samplers of one gather must share ndims and dtype");
    if (!rc) rc = columns_pack(s, idx, n_local[r], recv + (size_t)r * blk, rows);
    if (!rc && hipStreamSynchronize(s->stream) != hipSuccess) rc = mjhmc_fail(MJHMC_ERR_HIP, "sync");
    idx += n_local[r];
  }
  if (!rc) rc = columns_unpack(s0, recv, blk, world, n_local, total, host_out);
  (void)hipFree(rows);
  (void)hipFree(recv);
  return rc;
}
#endif  // MJHMC_TEST_HOOKS

}  // extern "C"
