// ctx.hip — lifecycle, resident graph / feature tables (C ABI: include/gigl_hip.h).
#include "common.h"

#include <cstring>
#include <new>

int32_t gigl_fail(gigl_ctx* ctx, int32_t code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

int32_t gigl_arena_reset(gigl_ctx* ctx, int64_t need_bytes) {
  ctx->arena_off = 0;
  if (need_bytes <= ctx->arena_bytes) return GIGL_OK;
  int64_t want = gigl_align_up(need_bytes + (need_bytes >> 2), 1 << 20);
  if (ctx->arena) {
    GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    GIGL_HIP_CHECK(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_bytes = 0;
  }
  GIGL_HIP_CHECK(ctx, hipMalloc((void**)&ctx->arena, (size_t)want));
  ctx->arena_bytes = want;
  ++ctx->arena_gen;  // (captured launches that baked scratch addresses of the old arena are stale: pipeline.hip)
  return GIGL_OK;
}

void* gigl_arena_alloc(gigl_ctx* ctx, int64_t bytes) {
  int64_t off = gigl_align_up(ctx->arena_off, 256);
  if (off + bytes > ctx->arena_bytes) return nullptr;
  ctx->arena_off = off + bytes;
  return ctx->arena + off;
}

gigl_prof_scope::gigl_prof_scope(gigl_ctx* c, int id) : ctx(c), on(false), slot(0) {
  if (!(c->prof_mask & (1u << id)) || c->prof_used * 2 + 2 > c->prof_ev.size()) return;
  slot = c->prof_used++;
  c->prof_id[slot] = id;
  // inside a stream capture this becomes an event-record node of the graph (re-recorded by every replay)
  on = hipEventRecord(c->prof_ev[2 * slot], c->stream) == hipSuccess;
}

gigl_prof_scope::~gigl_prof_scope() {
  if (on) hipEventRecord(ctx->prof_ev[2 * slot + 1], ctx->stream);
}

extern "C" {

int32_t gigl_version(void) { return 100; }

int32_t gigl_profile_enable(gigl_ctx* ctx, uint32_t mask, int32_t capacity) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, capacity >= 0 && capacity <= (1 << 20), "bad profile capacity");
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_mask = mask;
  ctx->prof_used = 0;
  for (int i = 0; i < 16; ++i) {
    ctx->prof_acc_ms[i] = 0.0;
    ctx->prof_acc_n[i] = 0;
  }
  while (ctx->prof_ev.size() < (size_t)capacity * 2) {
    hipEvent_t e;
    GIGL_HIP_CHECK(ctx, hipEventCreate(&e));
    ctx->prof_ev.push_back(e);
  }
  ctx->prof_id.assign(ctx->prof_ev.size() / 2, -1);
  return GIGL_OK;
}

int32_t gigl_profile_reset(gigl_ctx* ctx) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof_used = 0;
  for (int i = 0; i < 16; ++i) {
    ctx->prof_acc_ms[i] = 0.0;
    ctx->prof_acc_n[i] = 0;
  }
  return GIGL_OK;
}

int32_t gigl_profile_read(gigl_ctx* ctx, int32_t kernel_id, double* total_ms, int64_t* launches) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, kernel_id >= 0 && kernel_id < GIGL_K_COUNT, "bad kernel id");
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = ctx->prof_acc_ms[kernel_id];
  int64_t n = ctx->prof_acc_n[kernel_id];
  for (size_t s = 0; s < ctx->prof_used; ++s) {
    if (ctx->prof_id[s] != kernel_id) continue;
    float ms = 0.f;
    GIGL_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[2 * s], ctx->prof_ev[2 * s + 1]));
    tot += ms;
    ++n;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return GIGL_OK;
}

int32_t gigl_ctx_create(int32_t device, gigl_ctx** out) {
  if (!out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GIGL_E_NO_DEVICE;
  if (device < 0 || device >= n) return GIGL_E_INVALID_ARG;
  gigl_ctx* c = new (std::nothrow) gigl_ctx();
  if (!c) return GIGL_E_OOM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return GIGL_E_HIP;
  }
  c->own_stream = true;
  *out = c;
  return GIGL_OK;
}

int32_t gigl_ctx_destroy(gigl_ctx* ctx) {
  if (!ctx) return GIGL_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  if (ctx->arena) hipFree(ctx->arena);
  gigl_sampler_table_free(ctx);
  if (ctx->crc_shift_tbl) hipFree(ctx->crc_shift_tbl);
  if (ctx->enc_tables) hipFree(ctx->enc_tables);
  for (hipEvent_t e : ctx->prof_ev) hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return GIGL_OK;
}

const char* gigl_last_error(gigl_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int32_t gigl_ctx_set_wide_workspaces(gigl_ctx* ctx, int32_t on) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  ctx->wide = on != 0;
  return GIGL_OK;
}

int32_t gigl_ctx_set_stream(gigl_ctx* ctx, void* hip_stream) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  if ((hipStream_t)hip_stream == ctx->stream && !ctx->own_stream) return GIGL_OK;  // already bound: nothing to drain
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  ctx->stream = (hipStream_t)hip_stream;  // NULL is a valid handle: the legacy default stream
  ctx->own_stream = false;
  return GIGL_OK;
}

int32_t gigl_ctx_synchronize(gigl_ctx* ctx) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GIGL_OK;
}

int32_t gigl_ctx_reserve(gigl_ctx* ctx, int64_t bytes) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return gigl_arena_reset(ctx, bytes);
}

int32_t gigl_memcpy(gigl_ctx* ctx, void* dst, int32_t dst_loc, const void* src, int32_t src_loc,
                    int64_t bytes) {
  if (!ctx) return GIGL_E_INVALID_ARG;
  GIGL_REQUIRE(ctx, bytes >= 0 && (bytes == 0 || (dst && src)), "bad memcpy arguments");
  if (bytes == 0) return GIGL_OK;
  hipMemcpyKind kind = dst_loc == GIGL_LOC_HOST
                           ? (src_loc == GIGL_LOC_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost)
                           : (src_loc == GIGL_LOC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, kind, ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GIGL_OK;
}

static int32_t copy_in(gigl_ctx* ctx, void* dst, const void* src, size_t bytes, int32_t loc) {
  if (bytes == 0) return GIGL_OK;
  GIGL_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes,
                                     loc == GIGL_LOC_HOST ? hipMemcpyHostToDevice
                                                          : hipMemcpyDeviceToDevice,
                                     ctx->stream));
  GIGL_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GIGL_OK;
}

int32_t gigl_graph_load_csc(gigl_ctx* ctx, int64_t n, int64_t e, const int64_t* rowptr,
                            const uint32_t* col, int32_t loc, gigl_graph** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, n >= 0 && e >= 0 && rowptr && (col || e == 0), "bad CSC arguments");
  GIGL_REQUIRE(ctx, n < (int64_t)GIGL_INVALID, "node ids must fit uint32 (n=%lld)", (long long)n);
  GIGL_REQUIRE(ctx, loc == GIGL_LOC_HOST || loc == GIGL_LOC_DEVICE, "bad loc %d", loc);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  gigl_graph* g = new (std::nothrow) gigl_graph();
  if (!g) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  g->ctx = ctx;
  g->n = n;
  g->e = e;
  hipError_t e1 = hipMalloc((void**)&g->rowptr, (size_t)(n + 1) * sizeof(int64_t));
  hipError_t e2 = e1 == hipSuccess ? hipMalloc((void**)&g->col, (size_t)(e > 0 ? e : 1) * sizeof(uint32_t))
                                   : e1;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    if (g->rowptr) hipFree(g->rowptr);
    delete g;
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of CSC (%lld nodes, %lld edges) failed",
                     (long long)n, (long long)e);
  }
  int32_t rc = copy_in(ctx, g->rowptr, rowptr, (size_t)(n + 1) * sizeof(int64_t), loc);
  if (rc == GIGL_OK) rc = copy_in(ctx, g->col, col, (size_t)e * sizeof(uint32_t), loc);
  if (rc == GIGL_OK) rc = gigl_graph_compute_maxdeg(ctx, g);
  if (rc != GIGL_OK) {
    gigl_graph_destroy(g);
    return rc;
  }
  *out = g;
  return GIGL_OK;
}

int32_t gigl_graph_info(gigl_graph* g, int64_t* n, int64_t* e) {
  if (!g) return GIGL_E_INVALID_ARG;
  if (n) *n = g->n;
  if (e) *e = g->e;
  return GIGL_OK;
}

int32_t gigl_graph_device_ptrs(gigl_graph* g, const int64_t** rowptr, const uint32_t** col) {
  if (!g) return GIGL_E_INVALID_ARG;
  if (rowptr) *rowptr = g->rowptr;
  if (col) *col = g->col;
  return GIGL_OK;
}

int32_t gigl_graph_destroy(gigl_graph* g) {
  if (!g) return GIGL_OK;
  if (g->ctx) {
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->ctx->stream);
  }
  if (g->rowptr) hipFree(g->rowptr);
  if (g->col) hipFree(g->col);
  delete g;
  return GIGL_OK;
}

int32_t gigl_features_load(gigl_ctx* ctx, int64_t n, int32_t d, int32_t dtype, const void* rows,
                           int32_t loc, gigl_feat** out) {
  if (!ctx || !out) return GIGL_E_INVALID_ARG;
  *out = nullptr;
  GIGL_REQUIRE(ctx, n >= 0 && d > 0 && rows, "bad feature table arguments");
  GIGL_REQUIRE(ctx, dtype == GIGL_DTYPE_F32 || dtype == GIGL_DTYPE_F16, "bad dtype %d", dtype);
  GIGL_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t es = dtype == GIGL_DTYPE_F32 ? 4 : 2;
  gigl_feat* f = new (std::nothrow) gigl_feat();
  if (!f) return gigl_fail(ctx, GIGL_E_OOM, "host OOM");
  f->ctx = ctx;
  f->n = n;
  f->d = d;
  f->dtype = dtype;
  size_t bytes = (size_t)n * (size_t)d * es;
  if (hipMalloc(&f->rows, bytes ? bytes : 16) != hipSuccess) {
    delete f;
    return gigl_fail(ctx, GIGL_E_OOM, "hipMalloc of feature table (%zu bytes) failed", bytes);
  }
  int32_t rc = copy_in(ctx, f->rows, rows, bytes, loc);
  if (rc != GIGL_OK) {
    gigl_features_destroy(f);
    return rc;
  }
  *out = f;
  return GIGL_OK;
}

int32_t gigl_features_device_ptr(gigl_feat* f, const void** rows, int64_t* n, int32_t* d,
                                 int32_t* dtype) {
  if (!f) return GIGL_E_INVALID_ARG;
  if (rows) *rows = f->rows;
  if (n) *n = f->n;
  if (d) *d = f->d;
  if (dtype) *dtype = f->dtype;
  return GIGL_OK;
}

int32_t gigl_features_destroy(gigl_feat* f) {
  if (!f) return GIGL_OK;
  if (f->ctx) {
    hipSetDevice(f->ctx->device);
    hipStreamSynchronize(f->ctx->stream);
  }
  if (f->rows) hipFree(f->rows);
  if (f->row_crc) hipFree(f->row_crc);
  delete f;
  return GIGL_OK;
}

}  // extern "C"
