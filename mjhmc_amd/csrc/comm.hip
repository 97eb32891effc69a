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
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  if (r.so || !r.err.empty()) return r;
  const char* names[] = {std::getenv("MJHMC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
  }
  if (!r.so) {
    r.err = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?");
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

int mjhmc_comm_allgather_ring(mjhmc_comm* c, mjhmc_sampler* s, int slot0, int n, int stacked,
                              const int64_t* particles_per_rank, double* host_out) {
  if (!c || !s || !particles_per_rank || !host_out) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return mjhmc_fail(MJHMC_ERR_INVALID, "slots out of range");
  if (particles_per_rank[c->rank] != s->N) return mjhmc_fail(MJHMC_ERR_INVALID, "particles_per_rank[rank] != nparticles");
  HIPCHK(hipSetDevice(c->ctx->device));
  int64_t total = 0, mx = 0;
  for (int r = 0; r < c->world; ++r) {
    if (particles_per_rank[r] < 1) return mjhmc_fail(MJHMC_ERR_INVALID, "every rank must own at least one particle");
    total += particles_per_rank[r];
    mx = std::max(mx, particles_per_rank[r]);
  }
  const size_t rb = row_bytes(s), mb = mat_bytes(s);
  const size_t blk = (size_t)n * mx * rb;  // [n][mx][pitch], rank r fills [t][0 .. cnt_r)
  TRY(need(c, 0, blk));
  TRY(need(c, 1, blk * c->world));
  // this rank's n slots, padding rows dropped (each slot holds its N valid rows first)
  HIPCHK(hipMemcpy2DAsync(c->buf[0], (size_t)mx * rb, (const char*)s->ring + (size_t)slot0 * mb, mb, (size_t)s->N * rb,
                          (size_t)n, hipMemcpyDeviceToDevice, s->stream));
  NCCLCHK(rccl().AllGather(c->buf[0], c->buf[1], blk, ncclInt8, c->nccl, s->stream));
  const size_t elems = (size_t)s->D * total * n;
  TRY(ensure_stage(s, elems));
  int64_t off = 0;
  for (int r = 0; r < c->world; ++r) {
    const int64_t cnt = particles_per_rank[r];
    for (int t = 0; t < n; ++t) {
      const char* src = (const char*)c->buf[1] + (size_t)r * blk + (size_t)t * mx * rb;
      if (!stacked)  // (D, n * total), time-major like np.concatenate(axis=1) of the per-iteration states
        TRY(download_cols(s, src, nullptr, cnt, host_out, elems, (int64_t)n * total, 1, (int64_t)t * total + off, false));
      else           // (D, total, n) like np.stack(axis=-1)
        TRY(download_cols(s, src, nullptr, cnt, host_out, elems, total * n, n, off * n + t, false));
    }
    off += cnt;
  }
  HIPCHK(hipMemcpyAsync(host_out, s->stage, elems * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

int mjhmc_comm_allgather_columns(mjhmc_comm* c, mjhmc_sampler* s, const int64_t* local_idx, int64_t n_local,
                                 const int64_t* columns_per_rank, double* host_out) {
  if (!c || !s || !columns_per_rank || !host_out) return mjhmc_fail(MJHMC_ERR_INVALID, "NULL argument");
  if (n_local < 0 || (n_local && !local_idx)) return mjhmc_fail(MJHMC_ERR_INVALID, "bad local index list");
  if (columns_per_rank[c->rank] != n_local) return mjhmc_fail(MJHMC_ERR_INVALID, "columns_per_rank[rank] != n_local");
  HIPCHK(hipSetDevice(c->ctx->device));
  int64_t total = 0, mx = 0;
  for (int r = 0; r < c->world; ++r) {
    if (columns_per_rank[r] < 0) return mjhmc_fail(MJHMC_ERR_INVALID, "negative column count");
    total += columns_per_rank[r];
    mx = std::max(mx, columns_per_rank[r]);
  }
  if (total == 0) return 0;
  const int64_t pool = (int64_t)s->ring_slots * s->N;
  std::vector<int64_t> rows((size_t)n_local);  // pool index (slot * N + p) -> padded ring row (slot * Npad + p)
  for (int64_t k = 0; k < n_local; ++k) {
    if (local_idx[k] < 0 || local_idx[k] >= pool) return mjhmc_fail(MJHMC_ERR_INVALID, "gather index outside the sample ring");
    rows[(size_t)k] = (local_idx[k] / s->N) * s->Npad + local_idx[k] % s->N;
  }
  const size_t rb = row_bytes(s);
  const size_t blk = (size_t)mx * rb;
  TRY(need(c, 0, blk + (size_t)std::max<int64_t>(n_local, 1) * sizeof(int64_t)));
  TRY(need(c, 1, blk * c->world));
  if (n_local) {
    int64_t* drows = (int64_t*)((char*)c->buf[0] + blk);
    HIPCHK(hipMemcpyAsync(drows, rows.data(), (size_t)n_local * sizeof(int64_t), hipMemcpyHostToDevice, s->stream));
    const int cpr = (int)(rb / 16);
    const int64_t threads = n_local * cpr;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s->stream,
                       (const uint4*)s->ring, drows, (uint4*)c->buf[0], n_local, cpr);
    HIPCHK(hipGetLastError());
  }
  NCCLCHK(rccl().AllGather(c->buf[0], c->buf[1], blk, ncclInt8, c->nccl, s->stream));
  const size_t elems = (size_t)s->D * total;
  TRY(ensure_stage(s, elems));
  int64_t off = 0;
  for (int r = 0; r < c->world; ++r) {  // rank-major: rank 0's columns, then rank 1's, ...
    const int64_t cnt = columns_per_rank[r];
    if (cnt) TRY(download_cols(s, (const char*)c->buf[1] + (size_t)r * blk, nullptr, cnt, host_out, elems, total, 1, off, false));
    off += cnt;
  }
  HIPCHK(hipMemcpyAsync(host_out, s->stage, elems * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

}  // extern "C"
