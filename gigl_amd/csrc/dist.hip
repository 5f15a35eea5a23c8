// dist.hip — the hash-partitioned (multi-GPU) step inside the library: communicators + the sharded batch plan.
//
// One process per GPU.  owner(v) = v % world (python/gigl/distributed/dist_link_prediction_data_partitioner.py:
// 692-695): rank r holds the CSC rows (in-edges) and the feature rows of the nodes it owns (row v / world).  Where
// the reference's distributed loader issues one RPC per batch and partition from Python worker processes
// (python/gigl/distributed/distributed_neighborloader.py:162-192), a step here is a fixed schedule of device work
// and equal-split all-to-alls issued from C++ on the ctx stream — no host read anywhere in a step:
//
//   per hop k:   bucket the frontier by owner (fixed-capacity buckets, counts stay on the requester)
//                -> all_to_all (node ids, path sums) -> owners expand on their shard (same selection rule and hash
//                table as the single-GPU sampler) -> all_to_all (f ids per request) -> scatter into the tree layout
//   union graph: built locally (gigl_union_build_groups: node / edge dedup per batch, level-ordered numbering)
//   feature pull: the UNIQUE node ids of the union graph go to their owners in buckets -> all_to_all -> the owners
//                gather the rows STRAIGHT INTO THE SEND BUFFER (raw rows, or rows already projected by the first
//                layer's weights so that dims[1] floats travel instead of the raw row: lin_l(mean x) == mean(lin_l x))
//                -> all_to_all -> the first layer reads the rows where they arrived, through an index
//   forward:     the remaining layers are local (gigl_gather_mean + gigl_linear), one row per root comes out
//
// Transports (gigl_comm): RCCL (ncclSend/ncclRecv groups on the ctx stream — librccl is opened at run time, the
// one torch already mapped when there is one), an in-process group (every rank of a world is a ctx of this process
// on one device: exchanges are device copies, ranks advance phase by phase — tests and single-process drivers), and
// a host callback (the caller moves the bytes: MPI, gloo, ... — synchronises the stream around every exchange).
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

// ------------------------------------------------------------------------------------------ communicators
namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  // the copy another library of this process (torch) already mapped wins: two RCCL instances must not share a GPU
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* n : names)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!api.handle) return api;
#define GIGL_RCCL_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name)
  GIGL_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
  GIGL_RCCL_SYM(CommInitRank, "ncclCommInitRank");
  GIGL_RCCL_SYM(CommDestroy, "ncclCommDestroy");
  GIGL_RCCL_SYM(GroupStart, "ncclGroupStart");
  GIGL_RCCL_SYM(GroupEnd, "ncclGroupEnd");
  GIGL_RCCL_SYM(Send, "ncclSend");
  GIGL_RCCL_SYM(Recv, "ncclRecv");
  GIGL_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef GIGL_RCCL_SYM
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send &&
           api.Recv && api.GetErrorString;
  return api;
}

struct Pending {
  const void* send;
  void* recv;
  int64_t bytes;
  bool self_in_place;
  // count-sized blocks (comm_exchange_rows): only the first send_units[p] units of block p are moved
  const int32_t* send_units = nullptr;  // device [world]
  int64_t unit = 0;
};

struct LocalGroup {
  int world = 0;
  std::vector<gigl_comm*> members;
  std::vector<int32_t> h_units;  // [world][world] unit counts of the count-sized exchange being flushed
};

}  // namespace

struct gigl_comm {
  int32_t kind = 0;  // GIGL_COMM_*
  gigl_ctx* ctx = nullptr;
  int32_t rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  LocalGroup* group = nullptr;      // in-process group (shared by its members, freed with the last one)
  std::vector<Pending> pending;     // in-process group: exchanges registered since the last flush
  gigl_exchange_fn fn = nullptr;    // host callback
  void* user = nullptr;
  int32_t* h_units = nullptr;       // pinned host [2 * world]: the unit counts of a count-sized exchange
  int64_t moved_bytes = 0, block_bytes = 0;  // sent to OTHER ranks since creation: as moved / had every block been full
  bool fixed_blocks = false;  // gigl_comm_set_fixed_blocks: count-sized exchanges move whole blocks (no host read: capturable)
};

namespace {

// self_in_place: the producer already wrote this rank's own block into `recv` (comm_self_in_place() said it may):
// nothing moves for it.  Otherwise the own block is a device copy — it never goes through the transport.
bool comm_self_in_place(const gigl_comm* c) { return c->kind != GIGL_COMM_CALLBACK; }

int32_t comm_exchange(gigl_comm* c, const void* send, void* recv, int64_t bytes, bool self_in_place = false) {
  gigl_ctx* ctx = c->ctx;
  if (bytes == 0) return GIGL_OK;
  if (c->kind != GIGL_COMM_LOCAL) {  // (an in-process group counts when it performs the copies)
    c->moved_bytes += bytes * (c->world - 1);
    c->block_bytes += bytes * (c->world - 1);
  }
  if (c->kind == GIGL_COMM_RCCL) {
    if (!self_in_place)
      GIGL_HIP_CHECK(ctx, hipMemcpyAsync((char*)recv + (int64_t)c->rank * bytes,
                                         (const char*)send + (int64_t)c->rank * bytes, (size_t)bytes,
                                         hipMemcpyDeviceToDevice, ctx->stream));
    if (c->world == 1) return GIGL_OK;
    RcclApi& api = rccl();
    ncclResult_t r = api.GroupStart();
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
      if (p == c->rank) continue;
      r = api.Send((const char*)send + (int64_t)p * bytes, (size_t)bytes, ncclInt8, p, c->nccl, ctx->stream);
      if (r == ncclSuccess)
        r = api.Recv((char*)recv + (int64_t)p * bytes, (size_t)bytes, ncclInt8, p, c->nccl, ctx->stream);
    }
    ncclResult_t r2 = api.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return gigl_fail(ctx, GIGL_E_HIP, "RCCL all-to-all failed: %s", api.GetErrorString(r));
    return GIGL_OK;
  }
  if (c->kind == GIGL_COMM_LOCAL) {
    // performed by gigl_comm_flush_local once every rank is here
    c->pending.push_back(Pending{send, recv, bytes, self_in_place});
    return GIGL_OK;
  }
  // host callback: the bytes leave through the caller's transport
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t rc = c->fn(c->user, send, recv, bytes);
  if (rc != 0) return gigl_fail(ctx, GIGL_E_HIP, "the exchange callback failed (%d)", rc);
  return GIGL_OK;
}

// The all-to-all of `bytes`-sized blocks, moving only the head of each block: the first send_units[p] units (of `unit`
// bytes) of the block for rank p, the first recv_units[r] units of the block from rank r — both arrays on the device,
// [world], values above the block's capacity are cut to it.  The receiver's tail keeps what it held (the rows behind a
// count are never read).  RCCL needs the sizes on the host: one small copy + a stream synchronisation per exchange,
// paid once per call of a plan (many batches); the host-callback transport has a fixed-size contract and moves full
// blocks.  GIGL_DIST_FIXED_BLOCKS=1: full blocks everywhere (A/B).
int32_t comm_exchange_rows(gigl_comm* c, const void* send, void* recv, int64_t bytes, const int32_t* send_units,
                           const int32_t* recv_units, int64_t unit, bool self_in_place) {
  static const bool fixed = getenv("GIGL_DIST_FIXED_BLOCKS") != nullptr;
  gigl_ctx* ctx = c->ctx;
  if (bytes == 0) return GIGL_OK;
  if (fixed || c->fixed_blocks || c->kind == GIGL_COMM_CALLBACK || c->world == 1 || !send_units || !recv_units || unit <= 0)
    return comm_exchange(c, send, recv, bytes, self_in_place);  // (a single rank: nothing leaves the device)
  const int64_t cap_units = bytes / unit;
  if (c->kind == GIGL_COMM_LOCAL) {
    Pending pd{send, recv, bytes, self_in_place};
    pd.send_units = send_units;
    pd.unit = unit;
    c->pending.push_back(pd);
    return GIGL_OK;
  }
  // RCCL
  const int W = c->world;
  if (!c->h_units) GIGL_HIP_CHECK(ctx, hipHostMalloc((void**)&c->h_units, (size_t)2 * W * 4, hipHostMallocDefault));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(c->h_units, send_units, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(c->h_units + W, recv_units, (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  auto cut = [&](int32_t v) { return (int64_t)(v < 0 ? 0 : (v > cap_units ? cap_units : v)) * unit; };
  if (!self_in_place && cut(c->h_units[c->rank]) > 0)
    GIGL_HIP_CHECK(ctx, hipMemcpyAsync((char*)recv + (int64_t)c->rank * bytes, (const char*)send + (int64_t)c->rank * bytes,
                                       (size_t)cut(c->h_units[c->rank]), hipMemcpyDeviceToDevice, ctx->stream));
  if (W == 1) return GIGL_OK;
  RcclApi& api = rccl();
  ncclResult_t r = api.GroupStart();
  for (int p = 0; p < W && r == ncclSuccess; ++p) {
    if (p == c->rank) continue;
    const int64_t sb = cut(c->h_units[p]), rb = cut(c->h_units[W + p]);
    c->moved_bytes += sb;
    c->block_bytes += bytes;
    if (sb > 0) r = api.Send((const char*)send + (int64_t)p * bytes, (size_t)sb, ncclInt8, p, c->nccl, ctx->stream);
    if (r == ncclSuccess && rb > 0)
      r = api.Recv((char*)recv + (int64_t)p * bytes, (size_t)rb, ncclInt8, p, c->nccl, ctx->stream);
  }
  ncclResult_t r2 = api.GroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) return gigl_fail(ctx, GIGL_E_HIP, "RCCL count-sized all-to-all failed: %s", api.GetErrorString(r));
  return GIGL_OK;
}

}  // namespace

extern "C" {

int32_t gigl_comm_unique_id(void* id) {
  if (!id) return GIGL_E_INVALID_ARG;
  RcclApi& api = rccl();
  if (!api.ok) return GIGL_E_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == GIGL_COMM_ID_BYTES, "unique id size");
  return api.GetUniqueId((ncclUniqueId*)id) == ncclSuccess ? GIGL_OK : GIGL_E_HIP;
}

int32_t gigl_dist_init(gigl_ctx* ctx, int32_t rank, int32_t world, const void* rccl_unique_id, gigl_comm** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world && rccl_unique_id, "bad rank / world / id");
  RcclApi& api = rccl();
  if (!api.ok) return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror());
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_comm* c = new (std::nothrow) gigl_comm();
  if (!c) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  c->kind = GIGL_COMM_RCCL;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, rccl_unique_id, sizeof(id));
  ncclResult_t r = api.CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return gigl_fail(ctx, GIGL_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                     api.GetErrorString(r));
  }
  *out = c;
  return GIGL_OK;
}

int32_t gigl_dist_init_local(gigl_ctx* const* ctxs, int32_t world, gigl_comm** out) {
  if (!ctxs || !out || world < 1) return GIGL_E_INVALID_ARG;
  for (int r = 0; r < world; ++r) {
    if (!ctxs[r]) return GIGL_E_INVALID_ARG;
    GIGL_REQUIRE(ctxs[r], ctxs[r]->device == ctxs[0]->device && ctxs[r]->stream == ctxs[0]->stream,
                 "the ctxs of an in-process group must share one device and one stream (gigl_ctx_set_stream)");
  }
  LocalGroup* g = new (std::nothrow) LocalGroup();
  if (!g) return GIGL_E_OOM;
  g->world = world;
  for (int r = 0; r < world; ++r) {
    gigl_comm* c = new (std::nothrow) gigl_comm();
    if (!c) return GIGL_E_OOM;
    c->kind = GIGL_COMM_LOCAL;
    c->ctx = ctxs[r];
    c->rank = r;
    c->world = world;
    c->group = g;
    g->members.push_back(c);
    out[r] = c;
  }
  return GIGL_OK;
}

int32_t gigl_dist_init_callback(gigl_ctx* ctx, int32_t rank, int32_t world, gigl_exchange_fn fn, void* user,
                                gigl_comm** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world && fn, "bad rank / world / callback");
  gigl_comm* c = new (std::nothrow) gigl_comm();
  if (!c) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  c->kind = GIGL_COMM_CALLBACK;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->fn = fn;
  c->user = user;
  *out = c;
  return GIGL_OK;
}

int32_t gigl_comm_info(gigl_comm* c, int32_t* rank, int32_t* world, int32_t* kind) {
  if (!c) return GIGL_E_INVALID_ARG;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (kind) *kind = c->kind;
  return GIGL_OK;
}

int32_t gigl_comm_all_to_all(gigl_comm* c, const void* send, void* recv, int64_t bytes_per_peer) {
  if (!c) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(c->ctx, bytes_per_peer >= 0 && (bytes_per_peer == 0 || (send && recv)), "bad all-to-all arguments");
  GIGL_HIP_CHECK(c->ctx, hipSetDevice(c->ctx->device));
  return comm_exchange(c, send, recv, bytes_per_peer);
}

int32_t gigl_comm_set_fixed_blocks(gigl_comm* c, int32_t on) {
  if (!c) return GIGL_E_INVALID_ARG;
  c->fixed_blocks = on != 0;
  return GIGL_OK;
}

int32_t gigl_comm_traffic(gigl_comm* c, int64_t* moved_bytes, int64_t* full_block_bytes) {
  if (!c) return GIGL_E_INVALID_ARG;
  if (moved_bytes) *moved_bytes = c->moved_bytes;
  if (full_block_bytes) *full_block_bytes = c->block_bytes;
  return GIGL_OK;
}

int32_t gigl_comm_flush_local(gigl_comm* any) {
  if (!any) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = any->ctx;
  GIGL_REQUIRE(ctx, any->kind == GIGL_COMM_LOCAL && any->group, "not an in-process group");
  LocalGroup* g = any->group;
  const size_t n = g->members[0]->pending.size();
  for (gigl_comm* m : g->members)
    GIGL_REQUIRE(ctx, m->pending.size() == n, "ranks of the in-process group registered different exchange counts");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int W = g->world;
  for (size_t x = 0; x < n; ++x) {
    const int64_t bytes = g->members[0]->pending[x].bytes;
    for (gigl_comm* m : g->members)
      GIGL_REQUIRE(ctx, m->pending[x].bytes == bytes, "ranks disagree on the exchange size");
    // count-sized blocks: every sender's unit counts come to the host first (one synchronisation per such exchange)
    const bool sized = g->members[0]->pending[x].send_units != nullptr;
    if (sized) {
      g->h_units.resize((size_t)W * W);
      for (int r = 0; r < W; ++r) {
        GIGL_REQUIRE(ctx, g->members[r]->pending[x].send_units && g->members[r]->pending[x].unit == g->members[0]->pending[x].unit,
                     "ranks disagree on the kind of exchange");
        GIGL_HIP_CHECK(ctx, hipMemcpyAsync(g->h_units.data() + (size_t)r * W, g->members[r]->pending[x].send_units,
                                           (size_t)W * 4, hipMemcpyDeviceToHost, ctx->stream));
      }
      GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int64_t unit = g->members[0]->pending[x].unit, cap_units = sized ? bytes / unit : 0;
    for (int r = 0; r < W; ++r)    // sender
      for (int p = 0; p < W; ++p) {  // receiver: block r of p's receive buffer <- block p of r's send buffer
        int64_t nb = bytes;
        if (sized) {
          const int32_t u = g->h_units[(size_t)r * W + p];
          nb = (int64_t)(u < 0 ? 0 : (u > cap_units ? cap_units : u)) * unit;
        }
        if (r != p) {
          g->members[r]->moved_bytes += nb;
          g->members[r]->block_bytes += bytes;
        }
        if ((r == p && g->members[r]->pending[x].self_in_place) || nb == 0) continue;
        GIGL_HIP_CHECK(ctx, hipMemcpyAsync((char*)g->members[p]->pending[x].recv + (int64_t)r * bytes,
                                           (const char*)g->members[r]->pending[x].send + (int64_t)p * bytes,
                                           (size_t)nb, hipMemcpyDeviceToDevice, ctx->stream));
      }
  }
  for (gigl_comm* m : g->members) m->pending.clear();
  return GIGL_OK;
}

int32_t gigl_comm_destroy(gigl_comm* c) {
  if (!c) return GIGL_OK;
  if (c->ctx) {
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
  }
  if (c->kind == GIGL_COMM_RCCL && c->nccl) rccl().CommDestroy(c->nccl);
  if (c->h_units) hipHostFree(c->h_units);
  if (c->group) {
    LocalGroup* g = c->group;
    for (auto& m : g->members)
      if (m == c) m = nullptr;
    bool any = false;
    for (auto m : g->members) any = any || m != nullptr;
    if (!any) delete g;
  }
  delete c;
  return GIGL_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ kernels
namespace {

// Everything a call of the plan must find cleared — the request buckets of every hop (0xFFFFFFFF = no entry), the id
// buckets of the feature pull, every bucket counter — in ONE launch at the start of the call (round 5: a dozen
// hipMemsetAsync launches per call were 9 % of the world-1 step's GPU time and a quarter of its host time).  The
// buffers belong to this plan and are next touched by later phases of the same call on the same stream.
struct ClearSegs {
  uint32_t* p[20];
  int64_t end[20];  // cumulative length in 4-word units (a segment's length is rounded up to a multiple of 4 words)
  int64_t words[20];
  uint32_t val[20];
  int n;
};
__global__ __launch_bounds__(256) void dist_clear_kernel(ClearSegs s) {
  const int64_t total = s.end[s.n - 1];
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
    int k = 0;
    while (u >= s.end[k]) ++k;
    const int64_t w0 = (u - (k ? s.end[k - 1] : 0)) * 4;
    uint32_t* q = s.p[k] + w0;
    const uint32_t v = s.val[k];
    // (a segment that starts inside a buffer — a rank's own block of a receive buffer at rank * cap words — is only 4-byte
    // aligned when cap is not a multiple of 4: scalar stores there)
    if (w0 + 4 <= s.words[k] && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
      *reinterpret_cast<uint4*>(q) = make_uint4(v, v, v, v);
    } else {
      for (int64_t t = w0; t < s.words[k]; ++t) s.p[k][t] = v;
    }
  }
}

// frontier slot i (node v, path sum K) -> bucket owner(v) = v % world, at the next free position p (fixed capacity):
// nodes_out[r*cap + p] = v, ksum_out[r*cap + p] = K, slot_idx[r*cap + p] = i, pos[i] = r*cap + p (optional).
// One global atomic per (workgroup, owner): slots take a rank inside the workgroup from LDS counters (same-address
// atomics serialise; with a small world every slot hits one of `world` counters).  n_valid (optional, device): only
// the first *n_valid slots exist.
__global__ __launch_bounds__(256) void bucket_kernel(const uint32_t* __restrict__ nodes,
                                                     const uint32_t* __restrict__ ksums, int64_t m,
                                                     const int32_t* __restrict__ n_valid, uint32_t world, int64_t cap,
                                                     uint32_t* __restrict__ nodes_out, uint32_t* __restrict__ ksum_out,
                                                     int32_t* __restrict__ slot_idx, int32_t* __restrict__ pos,
                                                     int32_t* __restrict__ counts, uint32_t self_rank,
                                                     uint32_t* __restrict__ self_nodes, uint32_t* __restrict__ self_ksum,
                                                     int32_t own_skip_rank = -1) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t lim = n_valid ? (int64_t)*n_valid : m;
  uint32_t v = (i < m && i < lim) ? nodes[i] : GIGL_INVALID;
  // own_skip_rank >= 0: a node of this rank is not bucketed at all — pos[i] = -1 - (2^30 + its row of the rank's own
  // table), read in place by the consumer
  bool own = false;
  if (own_skip_rank >= 0 && v != GIGL_INVALID && v % world == (uint32_t)own_skip_rank) {
    if (pos) pos[i] = -1 - ((1 << 30) + (int32_t)(v / world));
    own = true;
    v = GIGL_INVALID;
  }
  const uint32_t r = v == GIGL_INVALID ? 0xFFFFFFFFu : v % world;
  const int lane = threadIdx.x & 63;
  int32_t p = 0;
  if (world <= 64) {
    __shared__ int32_t s_cnt[64], s_base[64];
    if (threadIdx.x < world) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int32_t rank = 0;
    if (r != 0xFFFFFFFFu) rank = atomicAdd(&s_cnt[r], 1);
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x] > 0)
      s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (r != 0xFFFFFFFFu) p = s_base[r] + rank;
  } else {
    unsigned long long todo = __ballot(r != 0xFFFFFFFFu);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const uint32_t r_lead = __shfl(r, lead, 64);
      const unsigned long long same = __ballot(r == r_lead);
      int32_t base = 0;
      if (lane == lead) base = atomicAdd(&counts[r_lead], (int32_t)__popcll(same));
      base = __shfl(base, lead, 64);
      if (r == r_lead) p = base + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
      todo &= ~same;
    }
  }
  if (r == 0xFFFFFFFFu) {
    if (pos && i < m && !own) pos[i] = -1;
    return;
  }
  if (p >= cap) {
    atomicOr(&counts[world], 1);
    if (pos) pos[i] = -1;
    return;
  }
  const int64_t e = (int64_t)r * cap + p;
  if (self_nodes && r == self_rank) {  // this rank's own requests need no exchange: written where they are read
    nodes_out = self_nodes;
    ksum_out = self_ksum;
  }
  nodes_out[e] = v;
  if (ksum_out) ksum_out[e] = ksums ? ksums[i] : v;
  if (slot_idx) slot_idx[e] = (int32_t)i;
  if (pos) pos[i] = (int32_t)e;
}

constexpr uint32_t DIST_HOT_STAMP = 0xFFFFFFFFu;  // stamp of a replicated (hot) id: above every tag

// Dense pull bookkeeping (two-hop plans): a node id is requested ONCE per call — the first thread to stamp it with the
// call's tag wins, takes the next entry of its owner's bucket and records that entry in slot_map[id]; everybody else
// (other occurrences, other batches of the call) finds it there.  Replaces hashing the leaves into the union's node
// table: one atomic exchange per occurrence on an array indexed by the id.  Workgroup-aggregated bucket counters as in
// bucket_kernel.
__global__ __launch_bounds__(256) void claim_bucket_kernel(const uint32_t* __restrict__ nodes, int64_t m,
                                                           const int32_t* __restrict__ n_valid, uint32_t world,
                                                           int64_t cap, uint32_t* __restrict__ ids_out,
                                                           int32_t* __restrict__ counts, uint32_t* __restrict__ stamp,
                                                           uint32_t tag, int32_t* __restrict__ slot_map,
                                                           int64_t n_global, int32_t own_rank) {
  __shared__ int32_t s_cnt[64], s_base[64];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t lim = n_valid ? (int64_t)*n_valid : m;
  uint32_t v = (i < m && i < lim) ? nodes[i] : GIGL_INVALID;
  if (v != GIGL_INVALID && (int64_t)v >= n_global) v = GIGL_INVALID;
  // this rank's own row: read in place — the consumers find it by arithmetic (id % world == rank -> row id / world of the
  // rank's table: gather_mean_kernel's own_world), so it is neither stamped nor entered in slot_map
  if (v != GIGL_INVALID && own_rank >= 0 && v % world == (uint32_t)own_rank) v = GIGL_INVALID;
  // tags only grow, so atomicMax returns the call that last requested the id: this one -> somebody holds its entry
  // already; DIST_HOT_STAMP (never overwritten by a max) -> a replicated row, read locally through its permanent
  // slot_map entry -1-h (gigl_dist_plan_set_hot_rows) and never requested.  One random access per occurrence.
  if (v != GIGL_INVALID) {
    const uint32_t was = atomicMax(&stamp[v], tag);
    if (was == tag || was == DIST_HOT_STAMP) v = GIGL_INVALID;
  }
  const uint32_t r = v == GIGL_INVALID ? 0xFFFFFFFFu : v % world;
  const int lane = threadIdx.x & 63;
  int32_t p = 0;
  if (world <= 64) {
    if (threadIdx.x < world) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int32_t rank = 0;
    if (r != 0xFFFFFFFFu) rank = atomicAdd(&s_cnt[r], 1);
    __syncthreads();
    if (threadIdx.x < world && s_cnt[threadIdx.x] > 0)
      s_base[threadIdx.x] = atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (r != 0xFFFFFFFFu) p = s_base[r] + rank;
  } else {
    unsigned long long todo = __ballot(r != 0xFFFFFFFFu);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const uint32_t r_lead = __shfl(r, lead, 64);
      const unsigned long long same = __ballot(r == r_lead);
      int32_t base = 0;
      if (lane == lead) base = atomicAdd(&counts[r_lead], (int32_t)__popcll(same));
      base = __shfl(base, lead, 64);
      if (r == r_lead) p = base + (int32_t)__popcll(same & ((1ull << lane) - 1ull));
      todo &= ~same;
    }
  }
  if (r == 0xFFFFFFFFu) return;
  if (p >= cap) {
    atomicOr(&counts[world], 1);
    slot_map[v] = 0;  // (the step is reported failed; keep the map inside the receive buffer)
    return;
  }
  const int64_t e = (int64_t)r * cap + p;
  ids_out[e] = v;
  slot_map[v] = (int32_t)e;
}

// stamps back to 0: all == 1 the hot marks (a new hot set is coming), all == 0 everything BUT the hot marks (tag wrap)
__global__ __launch_bounds__(256) void hot_unmark_kernel(uint32_t* __restrict__ stamp, int64_t n, int all_hot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool hot = stamp[i] == DIST_HOT_STAMP;
  if (all_hot ? hot : !hot) stamp[i] = 0u;
}

// pos[i] = receive-buffer row of union node i (its own feature row)
__global__ __launch_bounds__(256) void pos_from_map_kernel(const uint32_t* __restrict__ nodes,
                                                           const int32_t* __restrict__ n_valid, int64_t m,
                                                           const int32_t* __restrict__ slot_map, int64_t n_global,
                                                           int32_t* __restrict__ pos, uint32_t world, int32_t own_rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m || i >= (int64_t)*n_valid) return;
  const uint32_t v = nodes[i];
  if (own_rank >= 0 && v % world == (uint32_t)own_rank) {  // (own rows are not in slot_map: claim_bucket_kernel)
    pos[i] = -1 - ((1 << 30) + (int32_t)(v / world));
    return;
  }
  pos[i] = (int64_t)v < n_global ? slot_map[v] : 0;
}

// owner side of the feature pull: entry e of the received id buckets -> its feature row, written at row e of the send
// buffer.  Raw copy (any element type): the rows are flattened into 16-byte chunks, one per lane, so every lane of
// every wave moves data whatever the row length; this rank's own requests land straight in the receive buffer.
__global__ __launch_bounds__(256) void serve_rows_copy_kernel(const uint32_t* __restrict__ ids, int64_t n_entries,
                                                              uint32_t world, const char* __restrict__ rows,
                                                              int64_t n_rows, uint32_t row_bytes, uint32_t unit,
                                                              uint32_t units_per_row, char* __restrict__ out,
                                                              int64_t self_lo, int64_t self_hi,
                                                              char* __restrict__ self_out, int64_t src_stride = 0) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t / units_per_row;
  const uint32_t c = (uint32_t)(t - e * units_per_row);
  if (e >= n_entries) return;
  const uint32_t v = ids[e];
  if (v == GIGL_INVALID) return;
  const int64_t row = (int64_t)(v / world);
  if (row >= n_rows) return;
  char* o = (self_out && e >= self_lo && e < self_hi) ? self_out : out;
  // (src_stride: bytes between table rows when a row is a slice of a wider one — the projected table [W_l x | W_r x])
  const char* sp = rows + row * (src_stride ? src_stride : (int64_t)row_bytes) + (int64_t)c * unit;
  char* op = o + e * (int64_t)row_bytes + (int64_t)c * unit;
  if (unit == 16) *(uint4*)op = *(const uint4*)sp;
  else if (unit == 4) *(uint32_t*)op = *(const uint32_t*)sp;
  else *(uint16_t*)op = *(const uint16_t*)sp;
}

// requester side, staged plans: union node i's pulled row (row pos[i] of the receive buffer) widened to fp32 at row i of
// the batch's dense feature matrix — what a training batch hands to the encoder as x.  One wave per node; a node whose
// request did not fit its bucket (pos < 0: the step is flagged as overflowed) gets NaN.
__global__ __launch_bounds__(256) void batch_features_kernel(const char* __restrict__ rows, int32_t dtype, int32_t d,
                                                             const int32_t* __restrict__ pos, int64_t cap,
                                                             const int32_t* __restrict__ n_nodes,
                                                             float* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= cap || i >= (int64_t)*n_nodes) return;
  const int32_t at = pos[i];
  float* o = x + i * (int64_t)d;
  if (at < 0) {
    for (int q = lane; q < d; q += 64) o[q] = __builtin_nanf("");
    return;
  }
  if (dtype == GIGL_DTYPE_F32) {
    const float* sp = (const float*)rows + (int64_t)at * d;
    for (int q = lane; q < d; q += 64) o[q] = sp[q];
  } else {
    const __half* sp = (const __half*)rows + (int64_t)at * d;
    for (int q = lane; q < d; q += 64) o[q] = __half2float(sp[q]);
  }
}

// the same rows widened to fp32 (operand of the owner-side projection).  One wave per entry.
__global__ __launch_bounds__(256) void serve_rows_f32_kernel(const uint32_t* __restrict__ ids, int64_t n_entries,
                                                             uint32_t world, const void* __restrict__ rows,
                                                             int64_t n_rows, int32_t d, int32_t dtype,
                                                             float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (e >= n_entries) return;
  const uint32_t v = ids[e];
  if (v == GIGL_INVALID) return;
  const int64_t row = (int64_t)(v / world);
  if (row >= n_rows) return;
  float* o = out + e * (int64_t)d;
  if (dtype == GIGL_DTYPE_F32) {
    const float* sp = (const float*)rows + row * (int64_t)d;
    for (int q = lane; q < d; q += 64) o[q] = sp[q];
  } else {
    const __half* sp = (const __half*)rows + row * (int64_t)d;
    for (int q = lane; q < d; q += 64) o[q] = __half2float(sp[q]);
  }
}

// answers of the owners -> tree layout, one thread per (frontier slot, j): slot i's answer sits at bucket entry
// pos[i] (-1: the slot was not sent — an empty parent — and has no children).  Writes every output word, so
// nothing has to be cleared first.
__global__ __launch_bounds__(256) void scatter_slots_kernel(const uint32_t* __restrict__ resp,
                                                            const int32_t* __restrict__ pos,
                                                            const uint32_t* __restrict__ parent_ksums, int64_t m,
                                                            int f, uint32_t* __restrict__ out_nbr,
                                                            int32_t* __restrict__ out_cnt,
                                                            uint32_t* __restrict__ child_ksums) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / f;
  const int j = (int)(t - i * f);
  if (i >= m) return;
  const int32_t e = pos[i];
  const uint32_t v = e >= 0 ? resp[(int64_t)e * f + j] : GIGL_INVALID;
  out_nbr[t] = v;
  if (child_ksums) child_ksums[t] = v == GIGL_INVALID ? 0u : parent_ksums[i] + v;  // (NULL: the last hop has no children)
  if (j == 0) {
    int n = 0;
    if (e >= 0)
      for (int q = 0; q < f; ++q) n += resp[(int64_t)e * f + q] != GIGL_INVALID ? 1 : 0;
    out_cnt[i] = n;
  }
}

// first layer on owner-projected rows: h[i][c] = act( mean[i][c] + self[posb[i]][c] + bias[c] ) for i < *n_rows;
// mean = left half of the gather operand [rows][2*dout] (mean of the neighbours' W_l x), self rows = W_r x
__global__ __launch_bounds__(256) void projected_layer_kernel(const float* __restrict__ a2, const float* __restrict__ self,
                                                              const int32_t* __restrict__ posb,
                                                              const float* __restrict__ bias, int dout, int act,
                                                              const int32_t* __restrict__ n_rows, int64_t rows_cap,
                                                              float* __restrict__ h) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / dout;
  const int c = (int)(t - i * dout);
  if (i >= rows_cap || i >= *n_rows) return;
  float v = a2[i * 2 * dout + c] + self[(int64_t)posb[i] * dout + c] + (bias ? bias[c] : 0.f);
  h[i * dout + c] = act ? fmaxf(v, 0.f) : v;
}

// fold the bucket-overflow flags of the step into meta[GIGL_META_OVERFLOW]; a failed step computes nothing
__global__ void fold_overflow_kernel(int32_t* meta, const int32_t* const* flags, int n_flags, int hops,
                                     int32_t act_rows) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int32_t over = 0;
  for (int q = 0; q < n_flags; ++q) over |= *flags[q];
  // the activation buffers hold b*(1 + f0 + ...) rows: roots that are each other's sampled neighbours can push the
  // inner levels past that (pipeline.hip guard_levels_kernel) — such a batch is reported failed too
  for (int l = 0; l < hops; ++l) over |= meta[GIGL_META_LEVEL0 + l] > act_rows ? 1 : 0;
  if (over) {
    for (int l = 0; l <= hops; ++l) meta[GIGL_META_LEVEL0 + l] = 0;
    atomicAdd(&meta[GIGL_META_OVERFLOW], 1);
  }
}

struct DistStatsArgs {
  const int32_t* cnt[GIGL_MAX_HOPS];
  int64_t parents[GIGL_MAX_HOPS];
  int32_t hops;
  const int32_t* meta;
  const int32_t* rowptr;
  const int32_t* rowend;
  const int32_t* pull_counts;  // [world]
  int32_t world;
  // peer-mapped plans: PULLED_ROWS = the rows the first layer reads from OTHER ranks' tables (per occurrence: nothing is
  // deduplicated on this route) — the sources of its rows that are neither this rank's nor replicated, plus, over
  // pre-projected tables, the W_r x half rows of the inner nodes other ranks own
  int32_t peer, rank, self_rows;
  const uint32_t* nodes;
  const int32_t* col;
  const int32_t* hot_map;
};

__global__ __launch_bounds__(256) void dist_stats_kernel(DistStatsArgs a, unsigned long long* acc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = a.hops;
  long long sampled = 0;
#pragma unroll
  for (int k = 0; k < GIGL_MAX_HOPS; ++k)
    if (k < L)
      for (int64_t p = t0; p < a.parents[k]; p += stride) sampled += a.cnt[k][p];
  long long agg[GIGL_MAX_HOPS] = {0, 0, 0, 0};
  const int32_t n_rows = a.meta[GIGL_META_LEVEL0 + L - 1];
  for (int64_t i = t0; i < n_rows; i += stride) {
    const long long len = a.rowend[i] - a.rowptr[i];
#pragma unroll
    for (int l = 0; l < GIGL_MAX_HOPS; ++l)
      if (l < L && i < a.meta[GIGL_META_LEVEL0 + (L - 1 - l)]) agg[l] += len;
  }
  auto add = [&](int slot, long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[slot], (unsigned long long)v);
  };
  long long agg_all = 0;
#pragma unroll
  for (int l = 0; l < GIGL_MAX_HOPS; ++l) agg_all += agg[l];
  add(GIGL_STATS_SAMPLED, sampled);
  add(GIGL_STATS_AGGREGATED, agg_all);
#pragma unroll
  for (int l = 0; l < GIGL_MAX_HOPS; ++l) add(GIGL_STATS_AGG_LAYER0 + l, agg[l]);
  if (a.peer) {
    long long remote = 0;
    const int32_t n_local = L >= 2 ? a.meta[GIGL_META_LEVEL0 + L - 2] : 0;
    for (int64_t i = t0; i < n_rows; i += stride) {
      for (int32_t e = a.rowptr[i]; e < a.rowend[i]; ++e) {
        const uint32_t v = i < n_local ? a.nodes[a.col[e]] : (uint32_t)a.col[e];
        if (v % (uint32_t)a.world != (uint32_t)a.rank && !(a.hot_map && a.hot_map[v] < 0)) ++remote;
      }
      if (a.self_rows && a.nodes[i] % (uint32_t)a.world != (uint32_t)a.rank) ++remote;
    }
    add(GIGL_STATS_PULLED_ROWS, remote);
  }
  if (t0 == 0) {
    atomicAdd(&acc[GIGL_STATS_UNION_EDGES], (unsigned long long)a.meta[GIGL_META_N_EDGES]);
    atomicAdd(&acc[GIGL_STATS_UNION_NODES], (unsigned long long)a.meta[GIGL_META_N_NODES]);
    atomicAdd(&acc[GIGL_STATS_OVERFLOW], (unsigned long long)a.meta[GIGL_META_OVERFLOW]);
    for (int l = 0; l < L; ++l)
      atomicAdd(&acc[GIGL_STATS_ROWS_LAYER0 + l], (unsigned long long)a.meta[GIGL_META_LEVEL0 + (L - 1 - l)]);
    long long pulled = 0, most = 0;
    for (int r = 0; r < a.world; ++r) {
      pulled += a.pull_counts[r];
      most = a.pull_counts[r] > most ? a.pull_counts[r] : most;
    }
    atomicAdd(&acc[GIGL_STATS_PULLED_ROWS], (unsigned long long)pulled);
    atomicMax(&acc[GIGL_STATS_PULL_BUCKET_MAX], (unsigned long long)most);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------ the sharded plan
struct gigl_dist_plan {
  gigl_comm* comm = nullptr;
  gigl_ctx* ctx = nullptr;
  gigl_graph* shard = nullptr;
  gigl_feat* feat = nullptr;
  int32_t world = 1, rank = 0;
  int32_t b = 0, group_roots = 0, hops = 0;
  int32_t fan[GIGL_MAX_HOPS] = {0};
  int32_t dims[GIGL_MAX_HOPS + 1] = {0};
  const float* w[GIGL_MAX_HOPS] = {nullptr};
  const float* bias[GIGL_MAX_HOPS] = {nullptr};
  int32_t act_last = 0;
  int32_t aggr = GIGL_AGGR_MEAN;  // the SAGE layers' reduction (gigl_dist_plan_set_aggr)
  bool project = false;
  // pre-projected rows (opts->projected): this rank's [W_l x | W_r x] table, [shard rows][2*dims[1]] fp32, borrowed.  The
  // dense pull then moves W_l x rows (dims[1] fp32 instead of the raw row), a second small pull brings the W_r x rows of
  // the nodes of level < hops, and the first layer is one reduction (gigl_gather_project_mixed) — no projection per step
  const float* preproj = nullptr;
  int64_t mwe = -1;
  // kind 1 (gigl_dist_gat_plan_create): GAT layers over the pulled rows — w[l] = lin weight [heads*channels][dims[l]],
  // first layer from the input side (gigl_gat_input_layer_fused over the receive buffer through pos[]), layers >= 1
  // projection + attention (the layer stages of gigl_gat_plan_create); generic union + raw rows
  int32_t kind = 0;
  int32_t heads[GIGL_MAX_HOPS] = {0}, channels[GIGL_MAX_HOPS] = {0};
  const float* att_src[GIGL_MAX_HOPS] = {nullptr};
  const float* att_dst[GIGL_MAX_HOPS] = {nullptr};
  float slope = 0.2f;
  float *gat_scratch = nullptr, *alpha_scratch = nullptr, *hw = nullptr;
  // sampling
  gigl_tree tree{};
  uint32_t* child_ksum[GIGL_MAX_HOPS] = {nullptr};  // [slots of hop k]: K of the path root..slot
  int64_t m[GIGL_MAX_HOPS] = {0}, cap[GIGL_MAX_HOPS] = {0};
  uint32_t *rq_nodes_s[GIGL_MAX_HOPS] = {nullptr}, *rq_ksum_s[GIGL_MAX_HOPS] = {nullptr};
  uint32_t *rq_nodes_r[GIGL_MAX_HOPS] = {nullptr}, *rq_ksum_r[GIGL_MAX_HOPS] = {nullptr};
  uint32_t *resp_s[GIGL_MAX_HOPS] = {nullptr}, *resp_r[GIGL_MAX_HOPS] = {nullptr};
  int32_t* hop_pos[GIGL_MAX_HOPS] = {nullptr};  // [m[k]]: bucket entry of frontier slot i, -1 = not sent
  int32_t* counts[GIGL_MAX_HOPS] = {nullptr};  // [world + 1], last = overflow flag
  int32_t* own_cnt = nullptr;                  // owner-side out_cnt scratch
  // in-step overlap (world > 1): this rank's OWN block of a hop's requests never travels — it is expanded on a side
  // stream while the requests of the peers are still on the links (local expansion under the remote fetch), straight
  // into the receive buffer of the answers; the main stream expands the peers' blocks when they have arrived
  gigl_ctx* side = nullptr;
  int32_t* side_cnt = nullptr;                 // the side expansion's out_cnt scratch [largest cap]
  hipEvent_t ev_bucket[GIGL_MAX_HOPS] = {nullptr}, ev_own[GIGL_MAX_HOPS] = {nullptr};
  bool overlap = false;
  // union
  gigl_union un{};
  // feature pull: A = every union node (rows the first layer aggregates), B = nodes of level < hops (their own row;
  // projected mode only — raw rows serve both)
  int64_t pull_cap = 0, pull_cap_b = 0;
  uint32_t *ids_s = nullptr, *ids_r = nullptr, *idsb_s = nullptr, *idsb_r = nullptr;
  int32_t *pos = nullptr, *posb = nullptr;
  int32_t *pull_counts = nullptr, *pullb_counts = nullptr;
  int32_t* req_counts = nullptr;  // [world] rows rank r asked of this rank (its pull_counts[this rank]): the row blocks' sizes
  void *rows_s = nullptr, *rows_r = nullptr, *rowsb_s = nullptr, *rowsb_r = nullptr;
  int64_t row_bytes = 0;
  float* stage = nullptr;             // projected mode: fp32 operand of the owner-side projection
  float *wl0 = nullptr, *wr0 = nullptr;  // projected mode: contiguous [dims[1]][dims[0]] halves of w[0]
  int32_t* n_entries_dev = nullptr;   // device constants: world*pull_cap, world*pull_cap_b
  const int32_t** flag_ptrs = nullptr;  // device array of the overflow flags
  int n_flags = 0;
  // dense pull bookkeeping (two hops, raw rows): the union graph is the leaf-global build (leaves stay global ids in
  // their parents' rows), ids are claimed through stamp[] and located through slot_map[] (claim_bucket_kernel)
  bool dense = false;
  // dense mode: a rank's OWN rows are never requested, served or copied — the aggregation reads them from the feature
  // table (GIGL_DIST_COPY_OWN_ROWS=1: through the receive buffer like everybody else's, for measurements)
  bool own_in_place = false;
  // replicated hot rows (gigl_dist_plan_set_hot_rows)
  bool has_hot = false;  // stamp[] carries DIST_HOT_STAMP marks (and slot_map[] the rows' entries -1-h)
  const void* hot_rows = nullptr;
  uint32_t* stamp = nullptr;   // [n_global] tag of the call that last requested the id
  int32_t* slot_map = nullptr; // [n_global] receive-buffer row of the id in that call
  uint32_t tag = 0;
  int64_t n_global = 0, last_slots = 0;
  // peer-mapped route (opts->peer_direct; dense plans): the first layer reads every source row WHERE IT LIVES — row v / world
  // of rank (v % world)'s table, peers_dev[v % world] (hipIpc-mapped between processes, plain pointers inside one) — so the
  // feature pull has no claim, no id exchange, no owner-side gather and no receive buffer: phases 2L and 2L + 1 end after
  // the union graph.  Replicated hot rows keep their marks in slot_map (the only thing it holds then).
  bool peer = false;
  const void** peers_dev = nullptr;  // device [world]
  bool peers_ready = false;
  // ... and the peer-SAMPLED route (opts->peer_sample, with peer_direct): the ranks' CSC shards are mapped too, and a rank
  // expands its OWN frontier, reading the owners' adjacency rows where they live (gigl_sample_khop_peer) — no request / answer
  // buckets, no hop exchange, no scatter: the step has no collective at all
  bool peer_sample = false, peer_graphs_ready = false;
  const int64_t** peer_rowptr_dev = nullptr;  // device [world]
  const uint32_t** peer_col_dev = nullptr;    // device [world]
  // activations
  bool tiled_layers = false;  // layers >= 1: tiled gather operand + two-source projection (every dims[l], l >= 1, % 4 == 0)
  float* abuf = nullptr;
  float* hbuf[2] = {nullptr, nullptr};
  int64_t act_rows = 0;
  std::vector<void*> owned;
};

namespace {

int n_phases(const gigl_dist_plan* p) { return 2 * p->hops + 3; }

int64_t grid256(int64_t n) { return (n + 255) / 256; }

int32_t split_w0(gigl_dist_plan* p) {  // w[0] = [W_l | W_r] row-interleaved -> two contiguous matrices
  const size_t in = (size_t)p->dims[0], out = (size_t)p->dims[1];
  GIGL_HIP_CHECK(p->ctx, hipMemcpy2DAsync(p->wl0, in * 4, p->w[0], 2 * in * 4, in * 4, out, hipMemcpyDeviceToDevice,
                                          p->ctx->stream));
  GIGL_HIP_CHECK(p->ctx, hipMemcpy2DAsync(p->wr0, in * 4, p->w[0] + in, 2 * in * 4, in * 4, out,
                                          hipMemcpyDeviceToDevice, p->ctx->stream));
  return GIGL_OK;
}

// the rows this rank owns are read in place and found by arithmetic: never claimed, requested or served
bool own_rows_in_place(const gigl_dist_plan* p) { return p->dense && p->own_in_place; }
// ... and then nothing of a rank's own ever sits in the id / row buckets: their self blocks stay empty (cleared at
// creation) and no exchange touches them
bool pull_self_in_place(const gigl_dist_plan* p) { return own_rows_in_place(p) && comm_self_in_place(p->comm); }

// one launch at the start of a call: every bucket and counter the call's phases expect cleared (dist_clear_kernel)
int32_t clear_call(gigl_dist_plan* p) {
  ClearSegs sg{};
  int n = 0;
  int64_t at = 0;
  auto add = [&](void* q, int64_t words, uint32_t val) {
    if (!q || words <= 0) return;
    sg.p[n] = (uint32_t*)q;
    sg.words[n] = words;
    sg.val[n] = val;
    at += (words + 3) / 4;
    sg.end[n] = at;
    ++n;
  };
  const int64_t W = p->world;
  const bool in_place = comm_self_in_place(p->comm);
  for (int k = 0; k < p->hops; ++k) {
    if (!(W == 1 && in_place)) add(p->rq_nodes_s[k], W * p->cap[k], 0xFFFFFFFFu);  // (a lone rank buckets into the receive side)
    if (in_place) add(p->rq_nodes_r[k] + (int64_t)p->rank * p->cap[k], p->cap[k], 0xFFFFFFFFu);
    add(p->counts[k], W + 1, 0u);
  }
  if (!(W == 1 && pull_self_in_place(p))) add(p->ids_s, W * p->pull_cap, 0xFFFFFFFFu);  // (... and requests no row)
  add(p->pull_counts, W + 1, 0u);
  if (p->pullb_counts) {
    if (!(W == 1 && pull_self_in_place(p))) add(p->idsb_s, W * p->pull_cap_b, 0xFFFFFFFFu);
    add(p->pullb_counts, W + 1, 0u);
  }
  if (n == 0) return GIGL_OK;
  sg.n = n;
  int64_t blocks = (at + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  gigl_prof_scope ps(p->ctx, GIGL_K_DIST_PREP);
  hipLaunchKernelGGL(dist_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, p->ctx->stream, sg);
  GIGL_HIP_CHECK(p->ctx, hipGetLastError());
  return GIGL_OK;
}

// owner side of a row pull: entries -> their rows in the send buffer (the self block straight into the receive buffer)
int32_t serve_rows(gigl_dist_plan* p, const uint32_t* ids, int64_t n_entries, const char* table, int64_t stride, char* out,
                   int64_t self_lo, int64_t self_hi, char* self_out) {
  gigl_prof_scope ps(p->ctx, GIGL_K_DIST_SERVE);
  hipStream_t st = p->ctx->stream;
  const uint32_t world = (uint32_t)p->world;
  // (one 16-byte unit per thread: 64 lanes = one 1-KB row per wave.  The copy runs at the rate of its bytes — 163 MB in and
  // out per 16-batch call in ~70 us on the emulated 8-rank world = 4.7 TB/s; a wave-per-four-rows variant with four loads
  // in flight per lane was measured at 9.2 vs 8.5 us per rank-step and dropped: round 5)
  const bool a16 = (p->row_bytes & 15) == 0 && ((uintptr_t)table & 15) == 0 && (stride & 15) == 0;
  const uint32_t unit = a16 ? 16u : ((p->row_bytes & 3) == 0 ? 4u : 2u);
  const uint32_t upr = (uint32_t)(p->row_bytes / unit);
  hipLaunchKernelGGL(serve_rows_copy_kernel, dim3((unsigned)grid256(n_entries * upr)), dim3(256), 0, st, ids, n_entries,
                     world, table, p->feat->n, (uint32_t)p->row_bytes, unit, upr, out, self_lo, self_hi, self_out, stride);
  GIGL_HIP_CHECK(p->ctx, hipGetLastError());
  return GIGL_OK;
}

// hop k's answers -> tree slots, counts and the children's path sums (K of a root's path is the root id)
int32_t scatter_hop(gigl_dist_plan* p, int k, const uint32_t* roots) {
  const int64_t total = p->m[k] * p->fan[k];
  gigl_prof_scope ps(p->ctx, GIGL_K_DIST_PREP);
  hipLaunchKernelGGL(scatter_slots_kernel, dim3((unsigned)grid256(total)), dim3(256), 0, p->ctx->stream, p->resp_r[k],
                     p->hop_pos[k], k == 0 ? roots : p->child_ksum[k - 1], p->m[k], p->fan[k], p->tree.nbr[k],
                     p->tree.cnt[k], k + 1 < p->hops ? p->child_ksum[k] : (uint32_t*)nullptr);
  GIGL_HIP_CHECK(p->ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t phase_impl(gigl_dist_plan* p, int phase, const uint32_t* roots, int32_t seed, float* out) {
  gigl_ctx* ctx = p->ctx;
  hipStream_t st = ctx->stream;
  const int L = p->hops;
  const uint32_t world = (uint32_t)p->world;
  int32_t rc = GIGL_OK;
  // a LONE rank (world 1 on a transport that needs no self copy) owns every row: its hops are the single-GPU sampler's,
  // straight into the tree — no bucket, no request / answer blocks, no scatter (the same selection rule and table: the
  // owners' expansion IS that sampler, run on explicit frontiers)
  const bool peer_hops = p->peer_sample && world > 1;
  const bool lone_hops = (world == 1 && comm_self_in_place(p->comm) && !p->overlap) || peer_hops;
  if (lone_hops && phase < 2 * L) {
    if (phase != 0) return GIGL_OK;
    rc = clear_call(p);
    if (rc != GIGL_OK) return rc;
    if (peer_hops) {
      if (!p->peer_graphs_ready)
        return gigl_fail(ctx, GIGL_E_INVALID_ARG, "peer-sampled plan: the ranks' graph shards were not set (gigl_dist_plan_set_peer_graphs)");
      return gigl_sample_khop_peer(ctx, p->peer_rowptr_dev, p->peer_col_dev, p->world, p->n_global, p->mwe, roots, p->b, p->fan, L,
                                   seed, &p->tree);
    }
    return gigl_sample_khop(ctx, p->shard, roots, p->b, p->fan, L, seed, GIGL_MODE_SPARK_HASH, &p->tree);
  }
  if (phase < 2 * L && (phase & 1) == 0) {
    // ---- requester: (scatter the previous hop's answers,) bucket this hop's frontier by owner
    const int k = phase >> 1;
    if (k > 0) {
      rc = scatter_hop(p, k - 1, roots);
      if (rc != GIGL_OK) return rc;
    }
    const uint32_t* nodes = k == 0 ? roots : p->tree.nbr[k - 1];
    const uint32_t* ksums = k == 0 ? nullptr : p->child_ksum[k - 1];
    const bool in_place = comm_self_in_place(p->comm);
    if (k == 0) {
      rc = clear_call(p);
      if (rc != GIGL_OK) return rc;
    }
    {
      gigl_prof_scope ps(ctx, GIGL_K_DIST_PREP);
      hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)grid256(p->m[k])), dim3(256), 0, st, nodes, ksums, p->m[k],
                         (const int32_t*)nullptr, world, p->cap[k], p->rq_nodes_s[k], p->rq_ksum_s[k], (int32_t*)nullptr,
                         p->hop_pos[k], p->counts[k], (uint32_t)p->rank, in_place ? p->rq_nodes_r[k] : (uint32_t*)nullptr,
                         in_place ? p->rq_ksum_r[k] : (uint32_t*)nullptr);
    }
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    if (p->overlap) {
      // the own block sits in the receive buffer already (bucket_kernel wrote it there): expand it NOW, on the side
      // stream, while the exchange below moves the peers' blocks
      const int32_t hash_add = (int32_t)((uint32_t)seed * (uint32_t)(k + 1));
      const int64_t off = (int64_t)p->rank * p->cap[k];
      GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_bucket[k], st));
      GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(p->side->stream, p->ev_bucket[k], 0));
      rc = gigl_expand_frontier(p->side, p->shard, p->rq_nodes_r[k] + off, p->rq_ksum_r[k] + off, p->cap[k], p->fan[k],
                                hash_add, p->world, p->mwe, p->resp_r[k] + off * p->fan[k], p->side_cnt);
      if (rc != GIGL_OK) return gigl_fail(ctx, rc, "%s", gigl_last_error(p->side));
      GIGL_HIP_CHECK(ctx, hipEventRecord(p->ev_own[k], p->side->stream));
    }
    rc = comm_exchange(p->comm, p->rq_nodes_s[k], p->rq_nodes_r[k], p->cap[k] * 4, in_place);
    if (rc == GIGL_OK) rc = comm_exchange(p->comm, p->rq_ksum_s[k], p->rq_ksum_r[k], p->cap[k] * 4, in_place);
    return rc;
  }
  if (phase < 2 * L) {
    // ---- owner: expand the requests of every peer on this shard
    const int k = phase >> 1;
    const int32_t hash_add = (int32_t)((uint32_t)seed * (uint32_t)(k + 1));
    // (a one-rank world answers itself: straight into the receive buffer, no copy of the whole answer block)
    const bool solo = world == 1 && comm_self_in_place(p->comm);
    if (p->overlap) {  // the peers' blocks (before and after the own one); the own block's answers are the side stream's
      const int64_t cap = p->cap[k], f = p->fan[k];
      if (p->rank > 0) {
        rc = gigl_expand_frontier(ctx, p->shard, p->rq_nodes_r[k], p->rq_ksum_r[k], (int64_t)p->rank * cap, p->fan[k],
                                  hash_add, p->world, p->mwe, p->resp_s[k], p->own_cnt);
        if (rc != GIGL_OK) return rc;
      }
      if (p->rank + 1 < p->world) {
        const int64_t off = (int64_t)(p->rank + 1) * cap;
        rc = gigl_expand_frontier(ctx, p->shard, p->rq_nodes_r[k] + off, p->rq_ksum_r[k] + off,
                                  (int64_t)(p->world - p->rank - 1) * cap, p->fan[k], hash_add, p->world, p->mwe,
                                  p->resp_s[k] + off * f, p->own_cnt);
        if (rc != GIGL_OK) return rc;
      }
      GIGL_HIP_CHECK(ctx, hipStreamWaitEvent(st, p->ev_own[k], 0));
      return comm_exchange(p->comm, p->resp_s[k], p->resp_r[k], cap * f * 4, true);
    }
    rc = gigl_expand_frontier(ctx, p->shard, p->rq_nodes_r[k], p->rq_ksum_r[k], (int64_t)world * p->cap[k], p->fan[k],
                              hash_add, p->world, p->mwe, solo ? p->resp_r[k] : p->resp_s[k], p->own_cnt);
    if (rc != GIGL_OK) return rc;
    return comm_exchange(p->comm, p->resp_s[k], p->resp_r[k], p->cap[k] * p->fan[k] * 4, solo);
  }
  if (phase == 2 * L) {
    // ---- requester: last scatter, union graph, feature requests
    if (!lone_hops) {
      rc = scatter_hop(p, L - 1, roots);
      if (rc != GIGL_OK) return rc;
    }
    if (p->dense) {
      rc = gigl_union_build_impl(ctx, roots, &p->tree, p->group_roots, &p->un, 1 | (p->shard->multi ? 2 : 0));
      if (rc != GIGL_OK) return rc;
      if (++p->tag == DIST_HOT_STAMP) {  // (the tags are used up: forget every stamp but the hot marks)
        hipLaunchKernelGGL(hot_unmark_kernel, dim3((unsigned)grid256(p->n_global)), dim3(256), 0, st, p->stamp, p->n_global, 0);
        p->tag = 1;
      }
      if (p->peer) return GIGL_OK;  // (rows are read in place, from whichever rank holds them: nothing to request)
      const int32_t* n_inner = p->un.meta + GIGL_META_LEVEL0 + (L - 1);
      const int32_t own = p->own_in_place ? p->rank : -1;
      const bool self_ip = pull_self_in_place(p);
      const bool lone = world == 1 && self_ip;  // a lone rank owns every row: nothing to claim, request or serve
      gigl_prof_scope ps(ctx, GIGL_K_DIST_PREP);
      // the inner nodes first (their own rows: pos), then every sampled leaf
      if (!lone) {
        hipLaunchKernelGGL(claim_bucket_kernel, dim3((unsigned)grid256(p->act_rows)), dim3(256), 0, st, p->un.nodes,
                           p->act_rows, n_inner, world, p->pull_cap, p->ids_s, p->pull_counts, p->stamp, p->tag,
                           p->slot_map, p->n_global, own);
        hipLaunchKernelGGL(claim_bucket_kernel, dim3((unsigned)grid256(p->last_slots)), dim3(256), 0, st,
                           (const uint32_t*)p->tree.nbr[L - 1], p->last_slots, (const int32_t*)nullptr, world, p->pull_cap,
                           p->ids_s, p->pull_counts, p->stamp, p->tag, p->slot_map, p->n_global, own);
      }
      hipLaunchKernelGGL(pos_from_map_kernel, dim3((unsigned)grid256(p->act_rows)), dim3(256), 0, st, p->un.nodes, n_inner,
                         p->act_rows, p->slot_map, p->n_global, p->pos, world, own);
      if (p->preproj)  // the inner nodes' own W_r x rows: a second, small pull (a rank's own nodes: read in place)
        hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)grid256(p->act_rows)), dim3(256), 0, st, p->un.nodes,
                           (const uint32_t*)nullptr, p->act_rows, n_inner, world, p->pull_cap_b, p->idsb_s,
                           (uint32_t*)nullptr, (int32_t*)nullptr, p->posb, p->pullb_counts, 0u, (uint32_t*)nullptr,
                           (uint32_t*)nullptr, own);
      GIGL_HIP_CHECK(ctx, hipGetLastError());
      if (lone) return GIGL_OK;
      rc = comm_exchange(p->comm, p->ids_s, p->ids_r, p->pull_cap * 4, self_ip);
      if (rc == GIGL_OK) rc = comm_exchange(p->comm, p->pull_counts, p->req_counts, 4, self_ip);  // (the row blocks' sizes)
      if (rc == GIGL_OK && p->preproj) rc = comm_exchange(p->comm, p->idsb_s, p->idsb_r, p->pull_cap_b * 4, self_ip);
      return rc;
    }
    rc = gigl_union_build_groups(ctx, roots, &p->tree, p->group_roots, &p->un);
    if (rc != GIGL_OK) return rc;
    hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)grid256(p->un.cap_nodes)), dim3(256), 0, st, p->un.nodes,
                       (const uint32_t*)nullptr, p->un.cap_nodes, p->un.meta + GIGL_META_N_NODES, world, p->pull_cap,
                       p->ids_s, (uint32_t*)nullptr, (int32_t*)nullptr, p->pos, p->pull_counts, 0u, (uint32_t*)nullptr,
                       (uint32_t*)nullptr);
    if (p->project) {
      hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)grid256(p->act_rows)), dim3(256), 0, st, p->un.nodes,
                         (const uint32_t*)nullptr, p->act_rows, p->un.meta + GIGL_META_LEVEL0 + (L - 1), world,
                         p->pull_cap_b, p->idsb_s, (uint32_t*)nullptr, (int32_t*)nullptr, p->posb, p->pullb_counts,
                         0u, (uint32_t*)nullptr, (uint32_t*)nullptr);
    }
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    rc = comm_exchange(p->comm, p->ids_s, p->ids_r, p->pull_cap * 4);
    if (rc == GIGL_OK) rc = comm_exchange(p->comm, p->pull_counts, p->req_counts, 4);  // (the row blocks' sizes)
    if (rc == GIGL_OK && p->project) rc = comm_exchange(p->comm, p->idsb_s, p->idsb_r, p->pull_cap_b * 4);
    return rc;
  }
  if (phase == 2 * L + 1) {
    // ---- owner: the requested rows, gathered straight into the send buffer (or projected into it)
    const int64_t na = (int64_t)world * p->pull_cap, nb = (int64_t)world * p->pull_cap_b;
    if (p->peer) return GIGL_OK;
    if (!p->project) {
      const bool in_place = comm_self_in_place(p->comm);
      if (world == 1 && pull_self_in_place(p)) return GIGL_OK;  // (a lone rank requested nothing)
      const char* table = p->preproj ? (const char*)p->preproj : (const char*)p->feat->rows;
      const int64_t stride = p->preproj ? 2 * p->row_bytes : 0;  // ([W_l x | W_r x]: a served row is half a table row)
      rc = serve_rows(p, p->ids_r, na, table, stride, (char*)p->rows_s, (int64_t)p->rank * p->pull_cap,
                      (int64_t)(p->rank + 1) * p->pull_cap, in_place ? (char*)p->rows_r : (char*)nullptr);
      if (rc != GIGL_OK) return rc;
      // only the requested rows travel: the head of each block (comm_exchange_rows)
      rc = comm_exchange_rows(p->comm, p->rows_s, p->rows_r, p->pull_cap * p->row_bytes, p->req_counts, p->pull_counts,
                              p->row_bytes, in_place);
      if (rc != GIGL_OK || !p->preproj) return rc;
      rc = serve_rows(p, p->idsb_r, nb, table + p->row_bytes, stride, (char*)p->rowsb_s, (int64_t)0, (int64_t)0, (char*)nullptr);
      if (rc != GIGL_OK) return rc;
      return comm_exchange(p->comm, p->rowsb_s, p->rowsb_r, p->pull_cap_b * p->row_bytes, pull_self_in_place(p));
    }
    // (entries without a request keep whatever the operand held: their output rows are never read)
    hipLaunchKernelGGL(serve_rows_f32_kernel, dim3((unsigned)grid256(na * 64)), dim3(256), 0, st, p->ids_r, na, world,
                       p->feat->rows, p->feat->n, p->feat->d, p->feat->dtype, p->stage);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    rc = gigl_linear(ctx, p->stage, p->wl0, nullptr, p->n_entries_dev, na, p->dims[0], p->dims[1], 0,
                     (float*)p->rows_s);
    if (rc != GIGL_OK) return rc;
    hipLaunchKernelGGL(serve_rows_f32_kernel, dim3((unsigned)grid256(nb * 64)), dim3(256), 0, st, p->idsb_r, nb, world,
                       p->feat->rows, p->feat->n, p->feat->d, p->feat->dtype, p->stage);
    GIGL_HIP_CHECK(ctx, hipGetLastError());
    rc = gigl_linear(ctx, p->stage, p->wr0, nullptr, p->n_entries_dev + 1, nb, p->dims[0], p->dims[1], 0,
                     (float*)p->rowsb_s);
    if (rc != GIGL_OK) return rc;
    rc = comm_exchange_rows(p->comm, p->rows_s, p->rows_r, p->pull_cap * p->row_bytes, p->req_counts, p->pull_counts,
                            p->row_bytes, false);
    if (rc == GIGL_OK) rc = comm_exchange(p->comm, p->rowsb_s, p->rowsb_r, p->pull_cap_b * p->row_bytes);
    return rc;
  }
  // ---- requester: forward over the union graph, one row per root
  if (p->peer && !p->peers_ready)
    return gigl_fail(ctx, GIGL_E_INVALID_ARG, "peer-mapped plan: the ranks' tables were not set (gigl_dist_plan_set_peer_tables)");
  hipLaunchKernelGGL(fold_overflow_kernel, dim3(1), dim3(64), 0, st, p->un.meta, (const int32_t* const*)p->flag_ptrs,
                     p->n_flags, L, (int32_t)(p->act_rows < 0x7FFFFFFF ? p->act_rows : 0x7FFFFFFF));
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  for (int l = 0; l < L; ++l) {
    const int32_t* n_rows = p->un.meta + GIGL_META_LEVEL0 + (L - 1 - l);
    int64_t rows_cap = 0, width = p->b;
    for (int i = 0; i <= L - 1 - l; ++i) {
      rows_cap += width;
      width *= p->fan[i];
    }
    const int act = (l < L - 1 || p->act_last) ? 1 : 0;
    if (p->kind == 1) {
      if (l == 0) {  // the pulled rows sit in the receive buffer: local node i -> row pos[i]
        rc = gigl_gat_input_layer_fused(ctx, p->rows_r, p->feat->dtype, p->dims[0], (const uint32_t*)p->pos, nullptr,
                                        p->w[0], p->att_src[0], p->att_dst[0], p->heads[0], p->channels[0], p->slope,
                                        p->un.rowptr, p->un.rowend, p->un.col, n_rows, rows_cap, p->bias[0], act,
                                        p->gat_scratch, p->hbuf[0]);
        if (rc != GIGL_OK) return rc;
        continue;
      }
      const int32_t* n_src = p->un.meta + GIGL_META_LEVEL0 + (L - l);  // rows layer l - 1 computed
      const int64_t src_cap = rows_cap + width;
      rc = gigl_linear(ctx, p->hbuf[(l - 1) & 1], p->w[l], nullptr, n_src, src_cap, p->dims[l], p->dims[l + 1], 0, p->hw);
      if (rc != GIGL_OK) return rc;
      rc = gigl_gat_aggregate(ctx, p->hw, p->att_src[l], p->att_dst[l], p->heads[l], p->channels[l], p->slope, 1,
                              p->un.rowptr, p->un.rowend, p->un.col, n_src, src_cap, n_rows, rows_cap, p->bias[l], act,
                              p->alpha_scratch, p->hbuf[l & 1]);
      if (rc != GIGL_OK) return rc;
      continue;
    }
    if (l == 0 && p->project) {
      const int dout = p->dims[1];
      rc = gigl_gather_reduce(ctx, p->rows_r, GIGL_DTYPE_F32, dout, (const uint32_t*)p->pos, p->un.rowptr, p->un.rowend,
                              p->un.col, n_rows, rows_cap, p->aggr, p->abuf);
      if (rc != GIGL_OK) return rc;
      hipLaunchKernelGGL(projected_layer_kernel, dim3((unsigned)grid256(rows_cap * dout)), dim3(256), 0, st, p->abuf,
                         (const float*)p->rowsb_r, p->posb, p->bias[0], dout, act, n_rows, rows_cap, p->hbuf[0]);
      GIGL_HIP_CHECK(ctx, hipGetLastError());
      continue;
    }
    if (l == 0 && p->peer) {
      // every source by its global id (inner rows through un.nodes, leaf rows as they are), read from its owner's table
      const int32_t* hot_map = p->has_hot ? p->slot_map : nullptr;
      const int32_t* n_local = p->un.meta + GIGL_META_LEVEL0 + (L - 2);
      if (p->preproj) {
        const int dout = p->dims[1];
        rc = gigl_gather_project_mixed(ctx, nullptr, nullptr, dout, dout, p->un.nodes, p->un.rowptr, p->un.rowend, p->un.col,
                                       n_rows, rows_cap, p->aggr, n_local, p->bias[0], act, p->hbuf[0], hot_map,
                                       (const float*)p->hot_rows, nullptr, 2 * dout, nullptr, p->world, p->rank,
                                       (const float* const*)p->peers_dev);
        if (rc != GIGL_OK) return rc;
        continue;
      }
      rc = gigl_gather_reduce_mixed(ctx, nullptr, p->feat->dtype, p->dims[0], p->un.nodes, p->un.rowptr, p->un.rowend,
                                    p->un.col, n_rows, rows_cap, p->aggr, n_local, p->abuf, 0, hot_map, p->hot_rows, nullptr,
                                    0, p->world, p->rank, p->peers_dev);
      if (rc != GIGL_OK) return rc;
      rc = gigl_linear(ctx, p->abuf, p->w[0], p->bias[0], n_rows, rows_cap, 2 * p->dims[0], p->dims[1], act, p->hbuf[0]);
      if (rc != GIGL_OK) return rc;
      continue;
    }
    if (l == 0 && p->preproj) {  // one reduction over W_l x rows (receive buffer / own table / hot rows) + W_r x + bias
      const int dout = p->dims[1];
      rc = gigl_gather_project_mixed(ctx, (const float*)p->rows_r, (const float*)p->rowsb_r, dout, dout,
                                     (const uint32_t*)p->pos, p->un.rowptr, p->un.rowend, p->un.col, n_rows, rows_cap,
                                     p->aggr, p->un.meta + GIGL_META_LEVEL0 + (L - 2), p->bias[0], act, p->hbuf[0],
                                     p->slot_map, (const float*)p->hot_rows, p->preproj, 2 * dout, p->posb,
                                     p->own_in_place ? p->world : 0, p->rank);
      if (rc != GIGL_OK) return rc;
      continue;
    }
    if (l > 0 && p->tiled_layers) {
      // layers >= 1 as in the one-call plan (pipeline.hip): the gather writes the reduced half only, in the projection's
      // tiled operand layout; the projection reads the self half from the previous layer's rows (two-source operand)
      const int d = p->dims[l];
      rc = gigl_gather_reduce_mixed(ctx, p->hbuf[(l - 1) & 1], GIGL_DTYPE_F32, d, nullptr, p->un.rowptr, p->un.rowend,
                                    p->un.col, n_rows, rows_cap, p->aggr, nullptr, p->abuf, (d + 31) / 32, nullptr, nullptr,
                                    nullptr, 1);
      if (rc != GIGL_OK) return rc;
      rc = gigl_linear_tiled(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * d, p->dims[l + 1], act,
                             p->hbuf[l & 1], p->hbuf[(l - 1) & 1], nullptr, d, d, nullptr);
      if (rc != GIGL_OK) return rc;
      continue;
    }
    if (l == 0 && p->dense)  // rows of level L-1 hold global ids: located through slot_map
      rc = gigl_gather_reduce_mixed(ctx, p->rows_r, p->feat->dtype, p->dims[0], (const uint32_t*)p->pos, p->un.rowptr,
                                    p->un.rowend, p->un.col, n_rows, rows_cap, p->aggr,
                                    p->un.meta + GIGL_META_LEVEL0 + (L - 2), p->abuf, 0, p->slot_map, p->hot_rows,
                                    p->feat->rows, 0, p->own_in_place ? p->world : 0, p->rank);
    else if (l == 0)
      rc = gigl_gather_reduce(ctx, p->rows_r, p->feat->dtype, p->dims[0], (const uint32_t*)p->pos, p->un.rowptr,
                              p->un.rowend, p->un.col, n_rows, rows_cap, p->aggr, p->abuf);
    else
      rc = gigl_gather_reduce(ctx, p->hbuf[(l - 1) & 1], GIGL_DTYPE_F32, p->dims[l], nullptr, p->un.rowptr, p->un.rowend,
                              p->un.col, n_rows, rows_cap, p->aggr, p->abuf);
    if (rc != GIGL_OK) return rc;
    rc = gigl_linear(ctx, p->abuf, p->w[l], p->bias[l], n_rows, rows_cap, 2 * p->dims[l], p->dims[l + 1], act,
                     p->hbuf[l & 1]);
    if (rc != GIGL_OK) return rc;
  }
  const int dout = p->dims[L];
  // (a failed step must not hand out the previous step's activations as embeddings: its rows are NaN)
  gigl_take_rows(st, p->hbuf[(L - 1) & 1], p->un.root_local, p->b, dout, p->un.meta, out);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // namespace

extern "C" {

int32_t gigl_dist_plan_destroy(gigl_dist_plan* p) {
  if (!p) return GIGL_OK;
  if (p->ctx) {
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
  }
  if (p->side) {
    hipStreamSynchronize(p->side->stream);
    gigl_ctx_destroy(p->side);
  }
  for (int k = 0; k < GIGL_MAX_HOPS; ++k) {
    if (p->ev_bucket[k]) hipEventDestroy(p->ev_bucket[k]);
    if (p->ev_own[k]) hipEventDestroy(p->ev_own[k]);
  }
  for (void* q : p->owned) hipFree(q);
  delete p;
  return GIGL_OK;
}

static int32_t dist_plan_create_impl(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                                     const int32_t* fanouts, int32_t hops, const int32_t* dims, const float* const* w,
                                     const float* const* bias, int32_t act_last, const gigl_dist_plan_opts* opts,
                                     int32_t kind, gigl_dist_plan** out);

int32_t gigl_dist_plan_create(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                              const int32_t* fanouts, int32_t hops, const int32_t* dims, const float* const* w,
                              const float* const* bias, int32_t act_last, const gigl_dist_plan_opts* opts,
                              gigl_dist_plan** out) {
  return dist_plan_create_impl(comm, shard, shard_feat, b, fanouts, hops, dims, w, bias, act_last, opts, 0, out);
}

int32_t gigl_dist_gat_plan_create(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                                  const int32_t* fanouts, int32_t hops, const int32_t* heads, const int32_t* channels,
                                  const float* const* w, const float* const* att_src, const float* const* att_dst,
                                  const float* const* bias, float negative_slope, int32_t act_last,
                                  const gigl_dist_plan_opts* opts, gigl_dist_plan** out) {
  if (!comm || !out) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = comm->ctx;
  *out = nullptr;
  GIGL_REQUIRE(ctx, shard && shard_feat && fanouts && heads && channels && w && att_src && att_dst, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 1, "bad plan shape");
  GIGL_REQUIRE(ctx, !opts || !opts->project_on_owner, "the sharded GAT plan pulls raw rows");
  int32_t dims[GIGL_MAX_HOPS + 1];
  dims[0] = shard_feat->d;
  int32_t max_h = 1, max_hc = 1;
  for (int l = 0; l < hops; ++l) {
    GIGL_REQUIRE(ctx, heads[l] >= 1 && channels[l] >= 1 && att_src[l] && att_dst[l], "layer %d: bad heads / channels", l);
    dims[l + 1] = heads[l] * channels[l];
    if (heads[l] > max_h) max_h = heads[l];
    if (dims[l + 1] > max_hc) max_hc = dims[l + 1];
  }
  const int P = (shard_feat->d + 255) / 256;
  if ((shard_feat->d & 3) || (heads[0] != 1 && heads[0] != 2 && heads[0] != 4) || P > 4 ||
      (shard_feat->dtype != GIGL_DTYPE_F32 && shard_feat->dtype != GIGL_DTYPE_F16))
    return gigl_fail(ctx, GIGL_E_UNSUPPORTED, "sharded GAT plan: feature dim %d / %d heads outside the shapes the first "
                     "layer is built for (gigl_gat_input_layer_fused)", shard_feat->d, heads[0]);
  gigl_dist_plan* p = nullptr;
  int32_t rc = dist_plan_create_impl(comm, shard, shard_feat, b, fanouts, hops, dims, w, bias, act_last, opts, 1, &p);
  if (rc != GIGL_OK) return rc;
  p->slope = negative_slope;
  for (int l = 0; l < hops; ++l) {
    p->heads[l] = heads[l];
    p->channels[l] = channels[l];
    p->att_src[l] = att_src[l];
    p->att_dst[l] = att_dst[l];
  }
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    p->owned.push_back(q);
    return q;
  };
  p->gat_scratch = (float*)alloc((size_t)gigl_gat_input_layer_fused_scratch(shard_feat->d, heads[0], p->act_rows) * 4);
  p->alpha_scratch = (float*)alloc((size_t)2 * p->act_rows * max_h * 4);
  p->hw = (float*)alloc((size_t)p->act_rows * max_hc * 4);
  if (!p->gat_scratch || !p->alpha_scratch || !p->hw) {
    gigl_dist_plan_destroy(p);
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of the sharded GAT plan's workspace failed");
  }
  *out = p;
  return GIGL_OK;
}

int32_t gigl_dist_gat_plan_set_weights(gigl_dist_plan* p, const float* const* w, const float* const* att_src,
                                       const float* const* att_dst, const float* const* bias) {
  if (!p || !w || !att_src || !att_dst) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(p->ctx, p->kind == 1, "not a sharded GAT plan");
  for (int k = 0; k < p->hops; ++k) {
    if (!w[k] || !att_src[k] || !att_dst[k]) return gigl_fail(p->ctx, GIGL_E_INVALID_ARG, "layer %d: null weights", k);
    p->w[k] = w[k];
    p->att_src[k] = att_src[k];
    p->att_dst[k] = att_dst[k];
    p->bias[k] = bias ? bias[k] : nullptr;
  }
  return GIGL_OK;
}

static int32_t dist_plan_create_impl(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                                     const int32_t* fanouts, int32_t hops, const int32_t* dims, const float* const* w,
                                     const float* const* bias, int32_t act_last, const gigl_dist_plan_opts* opts,
                                     int32_t kind, gigl_dist_plan** out) {
  if (!comm || !out) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = comm->ctx;
  *out = nullptr;
  GIGL_REQUIRE(ctx, shard && shard_feat && fanouts && dims && w, "null argument");
  GIGL_REQUIRE(ctx, hops >= 1 && hops <= GIGL_MAX_HOPS && b >= 1, "bad plan shape");
  GIGL_REQUIRE(ctx, dims[0] == shard_feat->d, "dims[0]=%d != feature dim %d", dims[0], shard_feat->d);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_dist_plan* p = new (std::nothrow) gigl_dist_plan();
  if (!p) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  p->comm = comm;
  p->ctx = ctx;
  p->kind = kind;
  p->shard = shard;
  p->feat = shard_feat;
  p->world = comm->world;
  p->rank = comm->rank;
  p->b = b;
  p->hops = hops;
  p->act_last = act_last;
  p->group_roots = opts && opts->group_roots > 0 ? opts->group_roots : b;
  p->project = opts && opts->project_on_owner != 0;
  p->preproj = opts ? opts->projected : nullptr;
  p->mwe = opts ? opts->max_window_end : -1;
  const double slack = opts && opts->hop_slack > 0.f ? opts->hop_slack : 0.5;
  auto fail = [&](int32_t code, const char* msg) {
    gigl_dist_plan_destroy(p);
    return gigl_fail(ctx, code, "%s", msg);
  };
  if (p->group_roots < 1 || b % p->group_roots) return fail(GIGL_E_INVALID_ARG, "group_roots must divide b");
  for (int k = 0; k < hops; ++k) {
    if (fanouts[k] < 1 || fanouts[k] > GIGL_MAX_FANOUT) return fail(GIGL_E_UNSUPPORTED, "fanout outside [1,1024]");
    if (!w[k]) return fail(GIGL_E_INVALID_ARG, "null weight");
    p->fan[k] = fanouts[k];
    p->w[k] = w[k];
    p->bias[k] = bias ? bias[k] : nullptr;
  }
  int32_t max_in = 0, max_out = 0;
  for (int k = 0; k <= hops; ++k) {
    if (dims[k] < 1) return fail(GIGL_E_INVALID_ARG, "bad dims");
    p->dims[k] = dims[k];
    if (k < hops && dims[k] > max_in) max_in = dims[k];
    if (k > 0 && dims[k] > max_out) max_out = dims[k];
  }
  auto alloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
    p->owned.push_back(q);
    return q;
  };
  bool ok = true;
  const int64_t W = p->world;
  int64_t cap_nodes = 0, cap_edges = 0;
  gigl_union_capacity(b, fanouts, hops, &cap_nodes, &cap_edges);
  int64_t last_slots = b;
  for (int k = 0; k < hops; ++k) last_slots *= fanouts[k];
  p->last_slots = last_slots;
  p->n_global = shard->n * W;
  // (the GAT layers number every union node and read the pulled rows through pos[]: generic union)
  // (... and the leaf-global union numbers a level-1 row's leaves one per lane: second fan-outs past 64 take the generic one,
  // as in the one-call plan, pipeline.hip)
  p->dense = kind == 0 && hops == 2 && !p->project && p->n_global < ((int64_t)1 << 32) && (shard_feat->d & 3) == 0 &&
             fanouts[1] <= GIGL_FAST_FANOUT && getenv("GIGL_DIST_GENERIC_UNION") == nullptr && !(opts && opts->staged);
  p->own_in_place = p->dense && shard->n < ((int64_t)1 << 30) && getenv("GIGL_DIST_COPY_OWN_ROWS") == nullptr;
  p->peer = opts && opts->peer_direct != 0;
  if (p->peer && (!p->dense || p->n_global >= ((int64_t)1 << 31) || W > 64))
    return fail(GIGL_E_UNSUPPORTED, "the peer-mapped route needs the dense plan shape (SAGE layers, two hops, second fan-out "
                                    "<= 64, no owner-side projection, not staged), fewer than 2^31 nodes and a world <= 64");
  p->peer_sample = opts && opts->peer_sample != 0;
  if (p->peer_sample) {
    bool ok_f = p->peer && !shard->multi && p->mwe >= 0;
    for (int k = 0; k < hops; ++k) ok_f = ok_f && fanouts[k] <= GIGL_FAST_FANOUT;
    if (!ok_f)
      return fail(GIGL_E_UNSUPPORTED, "the peer-sampled route needs the peer-mapped feature route (peer_direct), fan-outs <= 64, no "
                                      "directed multi-edges and a window bound (max_window_end)");
  }
  if (p->preproj && (!p->dense || (dims[1] & 3) != 0 || dims[1] > 2048))
    return fail(GIGL_E_UNSUPPORTED, "pre-projected rows need the dense pull bookkeeping (two hops, second fan-out <= 64, no "
                                    "owner-side projection) and a first-layer width % 4 == 0, <= 2048");
  // (dense: the last hop's ids live right behind the union's col array so that rows can alias tree segments)
  p->un.col = (int32_t*)alloc((size_t)(cap_edges + (p->dense ? last_slots : 0)) * 4);
  ok = p->un.col != nullptr;
  // ---- tree + per-hop exchange buffers
  int64_t parents = b, max_served = 0;
  for (int k = 0; k < hops && ok; ++k) {
    p->m[k] = parents;
    // owner(v) is a uniform hash of distinct ids, but a frontier repeats hubs: buckets get `slack` headroom (ids and
    // path sums are 8 B per entry — cheap to pad, unlike feature rows); small worlds take the whole frontier
    int64_t cap = parents;
    if (W > 2) {
      cap = (int64_t)std::ceil((double)parents / (double)W * (1.0 + slack)) + 512;
      if (cap > parents) cap = parents;
    }
    p->cap[k] = cap;
    if (W * cap > max_served) max_served = W * cap;
    p->tree.cnt[k] = (int32_t*)alloc((size_t)parents * 4);
    parents *= fanouts[k];
    p->tree.nbr[k] = (p->dense && k == hops - 1) ? (uint32_t*)(p->un.col + cap_edges) : (uint32_t*)alloc((size_t)parents * 4);
    p->child_ksum[k] = (uint32_t*)alloc((size_t)parents * 4);
    p->rq_nodes_s[k] = (uint32_t*)alloc((size_t)W * cap * 4);
    p->rq_ksum_s[k] = (uint32_t*)alloc((size_t)W * cap * 4);
    p->rq_nodes_r[k] = (uint32_t*)alloc((size_t)W * cap * 4);
    p->rq_ksum_r[k] = (uint32_t*)alloc((size_t)W * cap * 4);
    p->resp_s[k] = (uint32_t*)alloc((size_t)W * cap * fanouts[k] * 4);
    p->resp_r[k] = (uint32_t*)alloc((size_t)W * cap * fanouts[k] * 4);
    p->hop_pos[k] = (int32_t*)alloc((size_t)p->m[k] * 4);
    p->counts[k] = (int32_t*)alloc((size_t)(W + 1) * 4);
    ok = p->tree.cnt[k] && p->tree.nbr[k] && p->child_ksum[k] && p->rq_nodes_s[k] && p->rq_ksum_s[k] &&
         p->rq_nodes_r[k] && p->rq_ksum_r[k] && p->resp_s[k] && p->resp_r[k] && p->hop_pos[k] && p->counts[k];
    if (parents >= ((int64_t)1 << 31)) return fail(GIGL_E_INVALID_ARG, "tree too large");
  }
  p->tree.hops = hops;
  p->tree.b = b;
  for (int k = 0; k < hops; ++k) p->tree.fanouts[k] = fanouts[k];
  p->own_cnt = (int32_t*)alloc((size_t)max_served * 4);
  // in-step overlap of the own block's expansion with the request exchange: OPT-IN (GIGL_DIST_OVERLAP=1).  Measured on
  // the emulated 8-rank world (bench.py --emulate-world 8): splitting a hop's expansion into own / before / after blocks
  // adds ~15 us of launches per step and rank (54.6 -> 70.0 us), while what it can hide is the REQUEST exchange alone
  // (8 B per frontier slot, ~1.8 MB per step: ~2 us on seven xGMI links) — the three plans in flight already cover it.
  const char* ov = getenv("GIGL_DIST_OVERLAP");
  if (W > 1 && comm_self_in_place(comm) && ov && ov[0] == '1') {
    int64_t cmax = 0;
    for (int k = 0; k < hops; ++k) cmax = p->cap[k] > cmax ? p->cap[k] : cmax;
    p->side_cnt = (int32_t*)alloc((size_t)cmax * 4);
    bool sok = p->side_cnt && gigl_ctx_create(ctx->device, &p->side) == GIGL_OK;
    for (int k = 0; k < hops && sok; ++k)
      sok = hipEventCreateWithFlags(&p->ev_bucket[k], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&p->ev_own[k], hipEventDisableTiming) == hipSuccess;
    if (!sok) return fail(GIGL_E_OOM, "dist plan: side stream for the in-step overlap");
    p->overlap = true;
  }
  // ---- union graph
  p->un.meta = (int32_t*)alloc(GIGL_META_LEN * 4);
  p->un.nodes = (uint32_t*)alloc((size_t)cap_nodes * 4);
  p->un.rowptr = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.rowend = (int32_t*)alloc((size_t)(cap_nodes + 2) * 4);
  p->un.root_local = (int32_t*)alloc((size_t)b * 4);
  if (p->dense && !p->peer) {  // (a peer-mapped plan claims nothing; its hot marks are allocated with the hot set)
    p->stamp = (uint32_t*)alloc((size_t)p->n_global * 4);
    p->slot_map = (int32_t*)alloc((size_t)p->n_global * 4);
    ok = ok && p->stamp && p->slot_map;
    if (ok && (hipMemsetAsync(p->stamp, 0, (size_t)p->n_global * 4, ctx->stream) != hipSuccess ||
               hipMemsetAsync(p->slot_map, 0, (size_t)p->n_global * 4, ctx->stream) != hipSuccess))
      ok = false;
  }
  p->un.cap_nodes = cap_nodes;
  p->un.cap_edges = cap_edges;
  int64_t act_rows = 0, width = b;
  for (int k = 0; k < hops; ++k) {
    act_rows += width;
    width *= fanouts[k];
  }
  p->act_rows = act_rows;
  // ---- feature pull.  Rows are the expensive bytes: the bucket capacity per peer is the caller's bound when given
  // (calibrated on warm-up steps, see GIGL_STATS_PULL_BUCKET_MAX), else the worst case / world + 10 %
  int64_t pc = opts && opts->pull_cap > 0 ? opts->pull_cap : (int64_t)std::ceil((double)cap_nodes / (double)W * 1.1) + 512;
  if (pc > cap_nodes) pc = cap_nodes;
  if (p->peer) pc = 0;  // (no row ever sits in a bucket)
  p->pull_cap = pc;
  p->row_bytes = (p->project || p->preproj) ? (int64_t)dims[1] * 4
                                            : (int64_t)dims[0] * (shard_feat->dtype == GIGL_DTYPE_F32 ? 4 : 2);
  p->ids_s = (uint32_t*)alloc((size_t)W * pc * 4);
  p->ids_r = (uint32_t*)alloc((size_t)W * pc * 4);
  p->pos = (int32_t*)alloc(p->peer ? 16 : (size_t)cap_nodes * 4);
  p->pull_counts = (int32_t*)alloc((size_t)(W + 1) * 4);
  p->req_counts = (int32_t*)alloc((size_t)(W + 1) * 4);
  p->rows_s = alloc((size_t)W * pc * p->row_bytes);
  p->rows_r = alloc((size_t)W * pc * p->row_bytes);
  ok = ok && p->own_cnt && p->un.meta && p->un.nodes && p->un.rowptr && p->un.rowend && p->un.col &&
       p->un.root_local && p->ids_s && p->ids_r && p->pos && p->pull_counts && p->req_counts && p->rows_s && p->rows_r;
  p->n_entries_dev = (int32_t*)alloc(16);
  if (p->peer && ok) {
    p->peers_dev = (const void**)alloc((size_t)W * sizeof(void*));
    ok = p->peers_dev != nullptr;
    if (ok && W == 1) {  // a lone rank's only table is its own
      const void* own = p->preproj ? (const void*)p->preproj : (const void*)shard_feat->rows;
      ok = hipMemcpy(p->peers_dev, &own, sizeof(void*), hipMemcpyHostToDevice) == hipSuccess;
      p->peers_ready = ok;
    }
  }
  if (p->peer_sample && ok) {
    p->peer_rowptr_dev = (const int64_t**)alloc((size_t)W * sizeof(void*));
    p->peer_col_dev = (const uint32_t**)alloc((size_t)W * sizeof(void*));
    ok = p->peer_rowptr_dev && p->peer_col_dev;
  }
  if (p->preproj && ok && !p->peer) {  // the second pull's buckets (W_r x of the inner nodes)
    int64_t pcb = opts && opts->pull_cap_b > 0 ? opts->pull_cap_b
                                               : (int64_t)std::ceil((double)act_rows / (double)W * (W > 1 ? 1.25 : 1.0)) + 512;
    if (pcb > act_rows) pcb = act_rows;
    p->pull_cap_b = pcb;
    p->idsb_s = (uint32_t*)alloc((size_t)W * pcb * 4);
    p->idsb_r = (uint32_t*)alloc((size_t)W * pcb * 4);
    p->posb = (int32_t*)alloc((size_t)act_rows * 4);
    p->pullb_counts = (int32_t*)alloc((size_t)(W + 1) * 4);
    p->rowsb_s = alloc((size_t)W * pcb * p->row_bytes);
    p->rowsb_r = alloc((size_t)W * pcb * p->row_bytes);
    ok = p->idsb_s && p->idsb_r && p->posb && p->pullb_counts && p->rowsb_s && p->rowsb_r;
  }
  if (p->project && ok) {
    int64_t pcb = (int64_t)std::ceil((double)act_rows / (double)W * (W > 1 ? 1.25 : 1.0)) + 512;
    if (pcb > act_rows) pcb = act_rows;
    p->pull_cap_b = pcb;
    p->idsb_s = (uint32_t*)alloc((size_t)W * pcb * 4);
    p->idsb_r = (uint32_t*)alloc((size_t)W * pcb * 4);
    p->posb = (int32_t*)alloc((size_t)act_rows * 4);
    p->pullb_counts = (int32_t*)alloc((size_t)(W + 1) * 4);
    p->rowsb_s = alloc((size_t)W * pcb * p->row_bytes);
    p->rowsb_r = alloc((size_t)W * pcb * p->row_bytes);
    p->stage = (float*)alloc((size_t)W * (pc > pcb ? pc : pcb) * dims[0] * 4);
    p->wl0 = (float*)alloc((size_t)dims[1] * dims[0] * 4);
    p->wr0 = (float*)alloc((size_t)dims[1] * dims[0] * 4);
    ok = p->idsb_s && p->idsb_r && p->posb && p->pullb_counts && p->rowsb_s && p->rowsb_r && p->stage && p->wl0 &&
         p->wr0;
    if (ok && W * pc >= ((int64_t)1 << 31)) ok = false;
  }
  // ---- activations
  // (a plan over pre-projected rows never forms the first layer's [mean | self] operand: its abuf serves layers >= 1 only —
  // 8 plans x 32 batches of a 768-wide first layer were 43 GB of workspace nobody wrote)
  if (p->preproj) {
    max_in = 0;
    for (int k = 1; k < hops; ++k) max_in = dims[k] > max_in ? dims[k] : max_in;
  }
  const int64_t a_cols = 2 * (int64_t)(max_in > max_out ? max_in : max_out);
  // (+ whole row tiles of 128 and whole K chunks of 32 for the tiled operand of layers >= 1)
  p->abuf = (float*)alloc((size_t)(act_rows + 128) * (a_cols + 64) * 4);
  p->tiled_layers = kind == 0 && getenv("GIGL_DIST_ROW_MAJOR") == nullptr;
  for (int k = 1; k < hops; ++k)
    if ((dims[k] & 3) != 0 || dims[k] > 2048) p->tiled_layers = false;
  p->hbuf[0] = (float*)alloc((size_t)act_rows * max_out * 4);
  p->hbuf[1] = hops > 1 ? (float*)alloc((size_t)act_rows * max_out * 4) : p->hbuf[0];
  // ---- overflow flags of the step
  std::vector<const int32_t*> flags;
  for (int k = 0; k < hops; ++k) flags.push_back(p->counts[k] + W);
  flags.push_back(p->pull_counts + W);
  if (p->pullb_counts) flags.push_back(p->pullb_counts + W);
  p->n_flags = (int)flags.size();
  p->flag_ptrs = (const int32_t**)alloc(flags.size() * sizeof(void*));
  ok = ok && p->abuf && p->hbuf[0] && p->hbuf[1] && p->n_entries_dev && p->flag_ptrs;
  if (!ok) return fail(GIGL_E_OOM, "hipMalloc of the sharded batch workspace failed");
  const int32_t ne[2] = {(int32_t)(W * pc), (int32_t)(W * p->pull_cap_b)};
  if (hipMemcpy(p->n_entries_dev, ne, sizeof(ne), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->flag_ptrs, flags.data(), flags.size() * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess)
    return fail(GIGL_E_HIP, "uploading plan constants failed");
  if (p->project) {
    int32_t rc = split_w0(p);
    if (rc != GIGL_OK) {
      gigl_dist_plan_destroy(p);
      return rc;
    }
    hipMemsetAsync(p->stage, 0, (size_t)W * (pc > p->pull_cap_b ? pc : p->pull_cap_b) * dims[0] * 4, ctx->stream);
  }
  hipMemsetAsync(p->rows_r, 0, (size_t)W * pc * p->row_bytes, ctx->stream);
  // (pull_self_in_place: the id buckets' self blocks are never written — a rank requests no row of its own — and must
  // read as empty on the owner side)
  hipMemsetAsync(p->ids_r, 0xFF, (size_t)W * pc * 4, ctx->stream);
  if (p->idsb_r) hipMemsetAsync(p->idsb_r, 0xFF, (size_t)W * p->pull_cap_b * 4, ctx->stream);
  hipStreamSynchronize(ctx->stream);
  *out = p;
  return GIGL_OK;
}

namespace {
// replicated ids carry DIST_HOT_STAMP in stamp[] and their permanent entry -1-h in slot_map[]
__global__ __launch_bounds__(256) void hot_scatter_kernel(const uint32_t* __restrict__ ids, int64_t n_hot, int64_t n_global,
                                                          uint32_t* __restrict__ stamp, int32_t* __restrict__ slot_map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_hot && (int64_t)ids[i] < n_global) {
    if (stamp) stamp[ids[i]] = DIST_HOT_STAMP;  // (a peer-mapped plan has no stamps: slot_map holds the marks alone)
    slot_map[ids[i]] = -1 - (int32_t)i;
  }
}
}  // namespace

int32_t gigl_dist_plan_set_hot_rows(gigl_dist_plan* p, const uint32_t* hot_ids, int64_t n_hot, const void* hot_rows) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, p->dense, "replicated hot rows need the dense pull bookkeeping (two hops, raw rows)");
  GIGL_REQUIRE(ctx, n_hot >= 0 && n_hot < ((int64_t)1 << 31) && (n_hot == 0 || (hot_ids && hot_rows)), "bad hot set");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (p->peer) {
    if (!p->slot_map && n_hot > 0) {
      void* q = nullptr;
      if (hipMalloc(&q, (size_t)p->n_global * 4) != hipSuccess) return gigl_fail(ctx, GIGL_E_OOM, "hot-row marks: hipMalloc failed");
      p->owned.push_back(q);
      p->slot_map = (int32_t*)q;
      p->has_hot = true;  // (cleared below)
    }
    if (p->has_hot) GIGL_HIP_CHECK(ctx, hipMemsetAsync(p->slot_map, 0, (size_t)p->n_global * 4, ctx->stream));
  } else if (p->has_hot)  // (the previous set's marks)
    hipLaunchKernelGGL(hot_unmark_kernel, dim3((unsigned)grid256(p->n_global)), dim3(256), 0, ctx->stream, p->stamp, p->n_global, 1);
  if (n_hot > 0)
    hipLaunchKernelGGL(hot_scatter_kernel, dim3((unsigned)grid256(n_hot)), dim3(256), 0, ctx->stream, hot_ids, n_hot,
                       p->n_global, p->stamp, p->slot_map);
  p->has_hot = n_hot > 0;
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // (hot_ids may be freed by the caller after this returns)
  p->hot_rows = n_hot > 0 ? hot_rows : nullptr;  // (n_hot == 0: back to "nothing replicated")
  return GIGL_OK;
}

int32_t gigl_dist_plan_set_peer_tables(gigl_dist_plan* p, const void* const* tables) {
  if (!p || !tables) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, p->peer, "not a peer-mapped plan (gigl_dist_plan_opts.peer_direct)");
  for (int r = 0; r < p->world; ++r) GIGL_REQUIRE(ctx, tables[r], "rank %d's table is null", r);
  const void* own = p->preproj ? (const void*)p->preproj : (const void*)p->feat->rows;
  GIGL_REQUIRE(ctx, tables[p->rank] == own, "tables[rank] must be this rank's own table (the plan's %s)",
               p->preproj ? "pre-projected rows" : "feature rows");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(p->peers_dev, tables, (size_t)p->world * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // (`tables` is the caller's host array)
  p->peers_ready = true;
  return GIGL_OK;
}

int32_t gigl_dist_plan_set_peer_graphs(gigl_dist_plan* p, const int64_t* const* rowptrs, const uint32_t* const* cols) {
  if (!p || !rowptrs || !cols) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, p->peer_sample, "not a peer-sampled plan (gigl_dist_plan_opts.peer_sample)");
  for (int r = 0; r < p->world; ++r) GIGL_REQUIRE(ctx, rowptrs[r] && cols[r], "rank %d's shard is null", r);
  GIGL_REQUIRE(ctx, rowptrs[p->rank] == p->shard->rowptr && cols[p->rank] == p->shard->col,
               "rowptrs[rank] / cols[rank] must be this rank's own shard (gigl_graph_device_ptrs)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(p->peer_rowptr_dev, rowptrs, (size_t)p->world * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(p->peer_col_dev, cols, (size_t)p->world * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  p->peer_graphs_ready = true;
  return GIGL_OK;
}

int32_t gigl_ipc_export(gigl_ctx* ctx, const void* dev_ptr, void* handle, int64_t* offset) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, dev_ptr && handle && offset, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == GIGL_IPC_HANDLE_BYTES, "ipc handle size");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // the handle names the ALLOCATION (a caching allocator hands out interior pointers): the offset travels with it
  void* base = nullptr;
  size_t size = 0;
  GIGL_HIP_CHECK(ctx, hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)dev_ptr));
  hipIpcMemHandle_t h;
  GIGL_HIP_CHECK(ctx, hipIpcGetMemHandle(&h, base));
  memcpy(handle, &h, sizeof(h));
  *offset = (int64_t)((const char*)dev_ptr - (const char*)base);
  return GIGL_OK;
}

int32_t gigl_ipc_open(gigl_ctx* ctx, const void* handle, int64_t offset, void** base, void** ptr) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, handle && base && ptr && offset >= 0, "bad arguments");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* b = nullptr;
  GIGL_HIP_CHECK(ctx, hipIpcOpenMemHandle(&b, h, hipIpcMemLazyEnablePeerAccess));
  *base = b;
  *ptr = (char*)b + offset;
  return GIGL_OK;
}

int32_t gigl_ipc_close(gigl_ctx* ctx, void* base) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  if (!base) return GIGL_OK;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipIpcCloseMemHandle(base));
  return GIGL_OK;
}

int32_t gigl_dist_plan_set_aggr(gigl_dist_plan* p, int32_t aggr) {
  if (!p) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(p->ctx, aggr == GIGL_AGGR_MEAN || aggr == GIGL_AGGR_SUM || aggr == GIGL_AGGR_MAX, "aggr %d", aggr);
  GIGL_REQUIRE(p->ctx, p->kind == 0, "the GAT plan has no segmented reduction to switch");
  // rows projected before the reduction (on the owner, or once per rank) lean on lin_l(reduce x) = reduce lin_l(x):
  // true for mean and sum, not for max
  if (aggr == GIGL_AGGR_MAX && (p->project || p->preproj))
    return gigl_fail(p->ctx, GIGL_E_UNSUPPORTED, "max aggregation pulls raw rows (no projection before the reduction)");
  p->aggr = aggr;
  return GIGL_OK;
}

int32_t gigl_dist_plan_set_weights(gigl_dist_plan* p, const float* const* w, const float* const* bias) {
  if (!p || !w) return GIGL_E_INVALID_ARG;
  for (int k = 0; k < p->hops; ++k) {
    if (!w[k]) return gigl_fail(p->ctx, GIGL_E_INVALID_ARG, "weight %d is null", k);
    p->w[k] = w[k];
    p->bias[k] = bias ? bias[k] : nullptr;
  }
  return p->project ? split_w0(p) : GIGL_OK;
}

int32_t gigl_dist_plan_phases(gigl_dist_plan* p, int32_t* n) {
  if (!p || !n) return GIGL_E_INVALID_ARG;
  *n = n_phases(p);
  return GIGL_OK;
}

int32_t gigl_dist_plan_phase(gigl_dist_plan* p, int32_t phase, const uint32_t* roots, int32_t sampling_seed,
                             float* out) {
  if (!p) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(p->ctx, roots && out && phase >= 0 && phase < n_phases(p), "bad phase arguments");
  GIGL_HIP_CHECK(p->ctx, hipSetDevice(p->ctx->device));
  return phase_impl(p, phase, roots, sampling_seed, out);
}

int32_t gigl_dist_plan_run(gigl_dist_plan* p, const uint32_t* roots, int32_t sampling_seed, float* out) {
  if (!p) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(p->ctx, roots && out, "null argument");
  GIGL_REQUIRE(p->ctx, p->comm->kind != GIGL_COMM_LOCAL,
               "ranks of an in-process group advance together: use gigl_dist_plan_run_local");
  GIGL_HIP_CHECK(p->ctx, hipSetDevice(p->ctx->device));
  for (int ph = 0; ph < n_phases(p); ++ph) {
    int32_t rc = phase_impl(p, ph, roots, sampling_seed, out);
    if (rc != GIGL_OK) return rc;
  }
  return GIGL_OK;
}

int32_t gigl_dist_plan_run_interleaved(gigl_dist_plan* const* plans, int32_t n, const uint32_t* const* roots,
                                       int32_t sampling_seed, float* const* out) {
  if (!plans || n < 1 || !plans[0]) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx0 = plans[0]->ctx;
  GIGL_REQUIRE(ctx0, roots && out, "null argument");
  const int np = n_phases(plans[0]);
  for (int i = 0; i < n; ++i)
    GIGL_REQUIRE(ctx0, plans[i] && roots[i] && out[i] && n_phases(plans[i]) == np && plans[i]->comm->kind != GIGL_COMM_LOCAL,
                 "plans[%d]: not a plan of this rank with the phases of plans[0]", i);
  GIGL_HIP_CHECK(ctx0, hipSetDevice(ctx0->device));
  for (int ph = 0; ph < np; ++ph)
    for (int i = 0; i < n; ++i) {
      int32_t rc = phase_impl(plans[i], ph, roots[i], sampling_seed, out[i]);
      if (rc != GIGL_OK) {
        if (plans[i]->ctx != ctx0) ctx0->err = plans[i]->ctx->err;
        return rc;
      }
    }
  return GIGL_OK;
}

int32_t gigl_dist_plan_run_local(gigl_dist_plan* const* plans, int32_t world, const uint32_t* const* roots,
                                 int32_t sampling_seed, float* const* out) {
  if (!plans || world < 1 || !plans[0]) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx0 = plans[0]->ctx;
  GIGL_REQUIRE(ctx0, roots && out && plans[0]->comm->kind == GIGL_COMM_LOCAL && plans[0]->world == world,
               "not the plans of one in-process group");
  const int np = n_phases(plans[0]);
  for (int ph = 0; ph < np; ++ph) {
    for (int r = 0; r < world; ++r) {
      GIGL_REQUIRE(ctx0, plans[r] && plans[r]->rank == r && plans[r]->comm->group == plans[0]->comm->group,
                   "plans[r] must be rank r of the same in-process group");
      int32_t rc = phase_impl(plans[r], ph, roots[r], sampling_seed, out[r]);
      if (rc != GIGL_OK) {
        if (plans[r]->ctx != ctx0) ctx0->err = plans[r]->ctx->err;
        return rc;
      }
    }
    if (ph + 1 < np) {
      int32_t rc = gigl_comm_flush_local(plans[0]->comm);
      if (rc != GIGL_OK) return rc;
    }
  }
  return GIGL_OK;
}

int32_t gigl_dist_plan_batch_features(gigl_dist_plan* p, float* x) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, x, "null argument");
  GIGL_REQUIRE(ctx, !p->dense && !p->project && !p->preproj,
               "the batch's feature matrix needs a staged plan (gigl_dist_plan_opts.staged: every union node numbered, raw rows)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(batch_features_kernel, dim3((unsigned)grid256(p->un.cap_nodes * 64)), dim3(256), 0, ctx->stream,
                     (const char*)p->rows_r, p->feat->dtype, p->feat->d, (const int32_t*)p->pos, p->un.cap_nodes,
                     (const int32_t*)(p->un.meta + GIGL_META_N_NODES), x);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_dist_plan_batch_graph(gigl_dist_plan* p, int32_t* rowptr, int32_t* rowend, int32_t* col, int32_t* root_local,
                                   int32_t* meta, uint32_t* nodes) {
  if (!p) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_REQUIRE(ctx, rowptr && rowend && col && root_local && meta, "null argument");
  GIGL_REQUIRE(ctx, !p->dense, "the batch graph with every node numbered needs a staged plan (gigl_dist_plan_opts.staged)");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t cn = (size_t)p->un.cap_nodes, ce = (size_t)p->un.cap_edges;
  // A staged step never runs the plan's last phase, which is where the hop / row bucket flags are folded into
  // meta[GIGL_META_OVERFLOW]: fold them here, before meta is handed out — a batch whose requests did not fit a bucket
  // (truncated neighbourhoods, NaN feature rows) must read as failed.  (No activation workspace is involved: the
  // level counts are not held against act_rows.)
  hipLaunchKernelGGL(fold_overflow_kernel, dim3(1), dim3(64), 0, st, p->un.meta, (const int32_t* const*)p->flag_ptrs,
                     p->n_flags, p->hops, (int32_t)0x7FFFFFFF);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(rowptr, p->un.rowptr, (cn + 1) * 4, hipMemcpyDeviceToDevice, st));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(rowend, p->un.rowend, (cn + 1) * 4, hipMemcpyDeviceToDevice, st));
  if (ce) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(col, p->un.col, ce * 4, hipMemcpyDeviceToDevice, st));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(root_local, p->un.root_local, (size_t)p->b * 4, hipMemcpyDeviceToDevice, st));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(meta, p->un.meta, (size_t)GIGL_META_LEN * 4, hipMemcpyDeviceToDevice, st));
  if (nodes) GIGL_HIP_CHECK(ctx, hipMemcpyAsync(nodes, p->un.nodes, cn * 4, hipMemcpyDeviceToDevice, st));
  return GIGL_OK;
}

int32_t gigl_dist_plan_buffers(gigl_dist_plan* p, gigl_tree* tree, gigl_union* un) {
  if (!p) return GIGL_E_INVALID_ARG;
  if (tree) *tree = p->tree;
  if (un) *un = p->un;
  return GIGL_OK;
}

namespace {
__global__ void bucket_fill_kernel(const int32_t* pull, const int32_t* pullb, int world, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int q = 0; q < 2; ++q) {
    const int32_t* c = q ? pullb : pull;
    if (!c) continue;
    long long most = 0, sum = 0;
    for (int r = 0; r < world; ++r) {
      sum += c[r];
      most = c[r] > most ? c[r] : most;
    }
    atomicMax(&out[2 * q], (unsigned long long)most);
    atomicAdd(&out[2 * q + 1], (unsigned long long)sum);
  }
}
}  // namespace

int32_t gigl_dist_plan_bucket_fill(gigl_dist_plan* p, int64_t* acc4) {
  if (!p || !acc4) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(p->ctx, hipSetDevice(p->ctx->device));
  hipLaunchKernelGGL(bucket_fill_kernel, dim3(1), dim3(64), 0, p->ctx->stream, p->pull_counts, p->pullb_counts, p->world,
                     (unsigned long long*)acc4);
  GIGL_HIP_CHECK(p->ctx, hipGetLastError());
  return GIGL_OK;
}

int32_t gigl_dist_plan_stats(gigl_dist_plan* p, int64_t* acc) {
  if (!p || !acc) return GIGL_E_INVALID_ARG;
  gigl_ctx* ctx = p->ctx;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  DistStatsArgs a{};
  a.hops = p->hops;
  int64_t most = p->b;
  for (int k = 0; k < p->hops; ++k) {
    a.cnt[k] = p->tree.cnt[k];
    a.parents[k] = p->m[k];
    if (p->m[k] > most) most = p->m[k];
  }
  a.meta = p->un.meta;
  a.rowptr = p->un.rowptr;
  a.rowend = p->un.rowend;
  a.pull_counts = p->pull_counts;
  a.world = p->world;
  a.peer = p->peer ? 1 : 0;
  a.rank = p->rank;
  a.self_rows = p->preproj ? 1 : 0;
  a.nodes = p->un.nodes;
  a.col = p->un.col;
  a.hot_map = p->has_hot ? p->slot_map : nullptr;
  int64_t blocks = grid256(most);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dist_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a, (unsigned long long*)acc);
  GIGL_HIP_CHECK(ctx, hipGetLastError());
  return GIGL_OK;
}

}  // extern "C"
