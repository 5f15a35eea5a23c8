// common.h — internal definitions shared by the HIP translation units of libgigl_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gigl_hip.h"

#define GIGL_WAVE 64

struct gigl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // scratch arena: one device allocation, bump-allocated per call, grown (hipMalloc) on demand
  char* arena = nullptr;
  int64_t arena_bytes = 0;
  int64_t arena_off = 0;
  uint64_t arena_gen = 0;  // bumped whenever the arena is reallocated (a larger request): scratch addresses change
  // optional per-kernel HIP-event timing (gigl_profile_*): events are recorded on ctx->stream
  // around the launches whose id bit is set in prof_mask
  uint32_t prof_mask = 0;
  std::vector<hipEvent_t> prof_ev;   // 2 per recorded launch
  std::vector<int32_t> prof_id;      // kernel id per recorded launch
  size_t prof_used = 0;              // launches recorded so far
  bool capturing = false;            // stream capture in progress: events become EXTERNAL event-record nodes
  double prof_acc_ms[16] = {0};      // + durations harvested from hipGraph replays (pipeline.hip)
  int64_t prof_acc_n[16] = {0};
  // sampler: threshold-list table over the xxhash sequence (sample.hip), built lazily, shared per device
  void* sampler_table = nullptr;
  // record encoder: x^(8*b*256^j) mod P tables of the CRC-32C combine step (serialize.hip), built lazily
  uint32_t* crc_shift_tbl = nullptr;
  // ... the write pass's LDS tables (slicing-by-4 CRC | x^(8*4d) byte-sliced | the shift tables again), for row width
  // enc_tables_d
  uint32_t* enc_tables = nullptr;
  int32_t enc_tables_d = -1;
  // gigl_ctx_set_wide_workspaces: plans created on this ctx provision for the WORST batch — every node of the tree an inner
  // node (roots that are each other's sampled neighbours) — instead of b*(1 + f0 + ...) inner rows: the workspace of the
  // redo of a batch that overflowed a regular plan
  bool wide = false;
};

// rows a plan provisions for the nodes of level <= level_max of a batch of b roots: b*(1 + f0 + ... ) over level_max
// fan-outs when no root is another root's sampled neighbour; wide: every child of a root-valued slot moves up a level,
// up to the whole tree (level 0 = the distinct roots: never more than b)
inline int64_t gigl_level_rows(bool wide, int64_t b, const int32_t* fanouts, int hops, int level_max) {
  int64_t rows = 0, width = b;
  const int top = wide && level_max >= 1 ? hops : level_max;
  for (int i = 0; i <= top; ++i) {
    rows += width;
    if (i < hops) width *= fanouts[i];
  }
  return rows;
}

void gigl_sampler_table_free(gigl_ctx* ctx);

// RAII bracket: records start/stop events around a launch when profiling of `id` is on
struct gigl_prof_scope {
  gigl_ctx* ctx;
  bool on;
  size_t slot;
  gigl_prof_scope(gigl_ctx* c, int id);
  ~gigl_prof_scope();
};

struct gigl_graph {
  gigl_ctx* ctx = nullptr;
  int64_t n = 0, e = 0;
  int64_t maxdeg = 0;         // largest in-degree (bounds the sampler's hash windows)
  bool multi = false;         // some row repeats an id (directed multi-edges kept at ingest): the sampler draws over
                              // the multiset and writes every sampled id once
  int64_t* rowptr = nullptr;  // device [n+1]
  uint32_t* col = nullptr;    // device [e]
};

struct gigl_feat {
  gigl_ctx* ctx = nullptr;
  int64_t n = 0;
  int32_t d = 0;
  int32_t dtype = 0;
  void* rows = nullptr;  // device [n][d]
  // [n] raw CRC-32C state of every row's packed feature_values bytes (serialize.hip, built on first use)
  uint32_t* row_crc = nullptr;
  std::mutex row_crc_mu;
  // largest |value| of the table (gigl_feat_absmax, found on first use; < 0: not looked at yet): decides whether the
  // half-split projection applies to operands made of its rows
  float absmax = -1.f;
  float absmean_nz = 0.f;  // mean |value| over the non-zero values (same pass)
};

int32_t gigl_fail(gigl_ctx* ctx, int32_t code, const char* fmt, ...);

#define GIGL_HIP_CHECK(ctx, expr)                                                          \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return gigl_fail((ctx), _e == hipErrorOutOfMemory ? GIGL_E_OOM : GIGL_E_HIP,         \
                       "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                       __LINE__);                                                          \
  } while (0)

#define GIGL_REQUIRE(ctx, cond, ...)                                  \
  do {                                                                \
    if (!(cond)) return gigl_fail((ctx), GIGL_E_INVALID_ARG, __VA_ARGS__); \
  } while (0)

// arena: reset at the start of every public per-batch call, then bump-allocate (256 B aligned)
int32_t gigl_arena_reset(gigl_ctx* ctx, int64_t need_bytes);
void* gigl_arena_alloc(gigl_ctx* ctx, int64_t bytes);

// fills g->maxdeg from the resident rowptr (synchronises the ctx stream)
int32_t gigl_graph_compute_maxdeg(gigl_ctx* ctx, gigl_graph* g);
// the weight gradient's per-chunk partial sums without the reduction launch (agg.hip): see gigl_linear_weight_grad_parts
int64_t gigl_linear_weight_grad_chunks(int64_t m_cap, int32_t n, int32_t k, int32_t* rows_per_chunk);
int32_t gigl_linear_weight_grad_sum(gigl_ctx* ctx, const float* part, const float* partb, const int32_t* m_dev, int32_t n, int32_t k,
                                    int32_t rows_per_chunk, float* dw, float* db);
int32_t gigl_linear_weight_grad_parts(gigl_ctx* ctx, const float* dy, const float* a, const float* relu_y, const int32_t* m_dev,
                                      int64_t m_cap, int32_t n, int32_t k, float* part, float* partb);
// union build with the plan-internal leaf-global option (union.hip)
int32_t gigl_union_build_impl(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree, int32_t group_roots,
                              gigl_union* out, int32_t leaf_global);
// exclusive scan of n int64 sizes on ctx->stream (one workgroup); out[n] = total, *status = total > cap (serialize.hip)
void gigl_scan_i64(gigl_ctx* ctx, const int64_t* sizes, int64_t n, int64_t cap, int64_t* out, int32_t* status);
// gather + reduce where the rows i >= *n_local_rows_dev hold GLOBAL source ids (no gather_ids translation): the
// plan's layer-0 gather over a leaf-global union (agg.hip); n_local_rows_dev == NULL: every row is translated
int32_t gigl_gather_reduce_mixed(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                                 const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                 const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr,
                                 const int32_t* n_local_rows_dev, float* out, int32_t tiled_nkc = 0,
                                 const int32_t* global_map = nullptr, const void* src2 = nullptr,
                                 const void* src3 = nullptr, int32_t no_self = 0, int32_t own_world = 0,
                                 int32_t own_rank = 0, const void* const* peers = nullptr);
// peers (DEVICE array of own_world table pointers; the sharded plan's peer-mapped route): every non-negative source index
// is a GLOBAL id v, read as row v / own_world of peers[v % own_world] (rows as far apart as src3's); global_map then holds
// only the replicated rows' marks (-1-h) and may be NULL; `src` is not read
// own_world > 0 (with global_map): a global id with id % own_world == own_rank is row id / own_world of src3 by arithmetic
// (the rank's own nodes are never claimed into global_map)
// no_self (tiled layout): only the reduced half is written, tiled_nkc = ceil(d / 32) chunks per row tile; the
// projection then takes the self half from the source rows (gigl_linear_tiled with self_src)
// global_map: the rows i >= *n_local_rows_dev hold GLOBAL ids whose row in `src` is global_map[id] (the sharded plan's
// receive buffer, dist.hip); a NEGATIVE row index -1-h (from gather_ids or global_map) is row h of `src2` (replicated
// hot rows) when h < 2^30, else row h - 2^30 of `src3` (the rank's own feature table)
// the plan's projected-input first layer: src_l / src_r hold W_l x / W_r x for every node (fp32 rows of width d, `ld`
// floats apart, by global id); out[i][0:d] = act(reduce_{e in row i} src_l[idx(col[e])] + src_r[gather_ids[i]] + bias)
int32_t gigl_gather_project_mixed(gigl_ctx* ctx, const float* src_l, const float* src_r, int32_t ld, int32_t d,
                                  const uint32_t* gather_ids, const int32_t* rowptr, const int32_t* rowend,
                                  const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr,
                                  const int32_t* n_local_rows_dev, const float* bias, int32_t act, float* out,
                                  const int32_t* global_map = nullptr, const float* src2 = nullptr,
                                  const float* src3 = nullptr, int32_t ld3 = 0, const int32_t* self_ids = nullptr,
                                  int32_t own_world = 0, int32_t own_rank = 0, const float* const* peers = nullptr);
// (peers: as in gigl_gather_reduce_mixed — the tables are the ranks' [W_l x | W_r x] rows, ld3 floats apart; destination
// i's W_r x is the right half of the row of ITS global id gather_ids[i]; src_l / src_r / self_ids are not read)
// (self_ids[i] < 0, = -1 - (2^30 + r): destination i is a node of this rank — its W_r x is the right half of row r of src3)
// (sharded plan: global_map / src2 / src3 as in gigl_gather_reduce_mixed — src3's rows ld3 floats apart — and
// self_ids[i] = the row of src_r that holds destination i's W_r x)
// tiled_nkc > 0: `out` is written in the projection's tiled operand layout ([row tile of 128][K chunk of 32][128 rows]
// [32 floats], tiled_nkc = ceil(2d / 32); capacity: whole row tiles) and read by gigl_linear_tiled (agg.hip)
int32_t gigl_linear_tiled(gigl_ctx* ctx, const float* a_tiled, const float* w, const float* bias, const int32_t* m_dev,
                          int64_t m_cap, int32_t k, int32_t n, int32_t act, float* y, const float* self_src = nullptr,
                          const uint32_t* self_ids = nullptr, int32_t d_mean = 0, int32_t self_ld = 0,
                          const float* hs_scale = nullptr, bool self_half = false);
// self_half (with hs_scale): self_src points at fp16 rows (self_ld halves apart)
// hs_scale != NULL (half split): operands as two fp16 planes, three MFMAs per accumulator (linear_split_kernel<.., HS>);
// hs_scale = device {s_a, s_w, 1 / (s_a s_w)}: powers of two that bring each operand's largest magnitude into
// [2^14, 2^15) on its way into the planes, undone in the epilogue (all exact) — written by gigl_hs_scale_update
// gigl_feat_half_split_scale: *s_a = the power of two for operands made of the table's rows (times `fan` under a sum
// reduction), or 0 when such operands stay on the bf16 planes (a NaN / inf / all-zero table, a largest magnitude outside
// [2^-46, 2^74], a table whose typical non-zero magnitude lies 2^10 below its largest, GIGL_GEMM_SPLIT=bf16);
// looks at the table once (cached in gigl_feat; that first look synchronises the ctx's stream)
int32_t gigl_feat_half_split_scale(gigl_ctx* ctx, gigl_feat* feat, float fan, float* s_a);
// hs_dev[0..2] = {s_a, s_w, 1 / (s_a s_w)} with s_w from the largest |w[i]| as the n weights are when the launch runs —
// one small launch on the ctx's stream, no synchronisation, may be captured
int32_t gigl_hs_scale_update(gigl_ctx* ctx, const float* w, int64_t n, float s_a, float* hs_dev);
int32_t gigl_gat_input_layer_fused_hs(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                      const uint32_t* gather_ids, const int32_t* n_local_dev, const float* w,
                                      const float* att_src, const float* att_dst, int32_t heads, int32_t channels,
                                      float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                      const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                      int32_t act, float* scratch, float* out, const float* hs_scale);
// The fused two-layer projection (agg.hip, linear_fused2_kernel): layer 0's [mean | self] projection (half split, two-source
// tiled operand) with the LAST layer's [W_l | W_r] applied to the hidden rows before they leave the workgroup — y2 =
// [2 K-split partial planes][rows][gigl_fused2_row_floats()] of p = [W_l h | W_r h]; gigl_sage_fused_out is the last
// layer over those rows (one reduction + self half + bias per root, written straight into the caller's `out`);
// gigl_fused2_prepare (per run, after gigl_hs_scale_update on the same stream) finds the second product's scales and
// lays out W2's fp16 planes.  gigl_fused2_shape_ok: hidden width 256, 2 * out <= row floats, d0 % 4 == 0.
// half-split scales of a layer >= 1 from the previous layer's (its outputs are bounded by K max|W| max|a| + max|b|): hs_out
// = {s_h, s_w, 1 / (s_h s_w)} for gigl_linear_tiled(.., hs_scale); fan > 1 under a sum reduction.  Per run, on the stream.
int32_t gigl_hs_chain_update(gigl_ctx* ctx, const float* hs_prev, const float* b_prev, int32_t n_b, int32_t k_prev, float fan,
                             const float* w, int64_t n_w, float* hs_out);
bool gigl_fused2_shape_ok(int32_t d0, int32_t hid, int32_t n_out);
int64_t gigl_fused2_w2h_bytes(int32_t k1);
int32_t gigl_fused2_row_floats();
int32_t gigl_fused2_planes();  // partial p planes per node: 1 (linear_fused2x_kernel, whole rows) or 2 (K-split over the hidden tiles)
int32_t gigl_fused2_prepare(gigl_ctx* ctx, const float* hs_dev, const float* b1, const float* w1, const float* w2, int32_t n_out,
                            int32_t k1, float* f2, void* w2h);
int32_t gigl_linear_fused2(gigl_ctx* ctx, const float* a_tiled, const float* w, const float* bias, const int32_t* m_dev,
                           int64_t m_cap, int32_t k, float* y2, int64_t plane_stride, const float* self_src,
                           const uint32_t* self_ids, int32_t d_mean, int32_t self_ld, const float* hs_scale,
                           const float* f2, const void* w2h, const int32_t* n_root_rows = nullptr);
// (n_root_rows: DEVICE count of the leading rows that are roots — only their W_r half is ever read; NULL: every row's)
int32_t gigl_sage_fused_out(gigl_ctx* ctx, const float* p, int64_t plane_stride, const int32_t* rowptr,
                            const int32_t* rowend, const int32_t* col, const int32_t* root_local, int32_t b,
                            int32_t n_out, const float* bias, int32_t act, int32_t aggr, const int32_t* meta, float* out);
bool gigl_half_split_enabled();  // (GIGL_GEMM_SPLIT=bf16 keeps every projection on the bf16 planes)
int32_t gigl_dev_absmax_f32(gigl_ctx* ctx, const float* p, int64_t n, float* out);  // synchronises the ctx's stream
int32_t gigl_feat_absmax(gigl_ctx* ctx, gigl_feat* feat, float* out);
// self_src != NULL: two-source operand — columns k >= d_mean are element k - d_mean of row self_ids[row] (or row) of
// self_src (fp32 rows, self_ld floats apart); a_tiled then holds ceil(d_mean / 32) chunks per row tile

static inline int64_t gigl_align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// n_words 32-bit words at p = value, as a KERNEL on the stream.  The launches of a plan are replayed from captured hipGraphs;
// a memset NODE was seen to run out of order with the kernels around it on this runtime (round 4: counters cleared by
// hipMemsetAsync were read stale by the next kernel of the replayed graph), a kernel node is ordered like any other.
static __global__ __launch_bounds__(256) void gigl_fill_u32_kernel(uint32_t* __restrict__ p, uint32_t value, int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}
static inline void gigl_fill_u32(hipStream_t st, void* p, uint32_t value, int64_t n_words) {
  if (n_words <= 0) return;
  int64_t blocks = (n_words + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gigl_fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, value, n_words);
}

// out[i] = h[root_local[i]] for the b roots of a batch set (rows of d floats) — the last stage of the one-call plans.
// (a failed batch set — meta[GIGL_META_OVERFLOW] != 0: levels zeroed, nothing computed — hands out NaN rows, never the
// previous call's activations)
template <int V>  // V = 4: d % 4 == 0, a thread moves 16 bytes (rows of h and out are 16-byte aligned); V = 1: any d
__global__ __launch_bounds__(256) void gigl_take_rows_kernel(const float* __restrict__ h, const int32_t* __restrict__ root_local,
                                                        int b, int d, const int32_t* __restrict__ meta,
                                                        float* __restrict__ out) {
  const uint32_t dv = (uint32_t)d / V;  // items per row (32-bit index math: b * d < 2^31 is checked at plan creation)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)b * dv) return;
  const uint32_t r = i / dv, c = (i - r * dv) * V;
  const bool failed = meta[GIGL_META_OVERFLOW] != 0;
  const int32_t l = failed ? -1 : root_local[r];
  float* dst = out + (int64_t)r * d + c;
  if constexpr (V == 4) {
    float4 v = failed ? make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""))
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    if (l >= 0) v = *reinterpret_cast<const float4*>(h + (int64_t)l * d + c);
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    *dst = failed ? __builtin_nanf("") : (l >= 0 ? h[(int64_t)l * d + c] : 0.f);
  }
}
static inline void gigl_take_rows(hipStream_t st, const float* h, const int32_t* root_local, int b, int d,
                                  const int32_t* meta, float* out) {
  const int64_t total = (int64_t)b * d;  // (< 2^31: the plans' activation buffers are smaller than that)
  // 16-byte items when the rows of both buffers are 16-byte aligned (hipMalloc'ed activations; torch allocations)
  if ((d & 3) == 0 && (((uintptr_t)out | (uintptr_t)h) & 15) == 0)
    hipLaunchKernelGGL(gigl_take_rows_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, h, root_local, b,
                       d, meta, out);
  else
    hipLaunchKernelGGL(gigl_take_rows_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, h, root_local, b, d,
                       meta, out);
}

#if defined(__HIPCC__)
// inclusive prefix sum over the 64 lanes of a wave on the VALU (DPP row shifts inside the 16-lane rows, then the row
// totals broadcast into the rows after them): no LDS-pipeline instruction (__shfl_up lowers to ds_bpermute)
__device__ __forceinline__ int gigl_wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source read 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2 and 3
  return v;
}

// Sum over aligned groups of `group` adjacent lanes (a power of two <= 64), returned to every lane of the group, on the
// VALU / SALU only: DPP quad permutes for 2 and 4, the half-row / row mirrors for 8 and 16, v_readlane of the four row
// totals for 32 and 64.  (__shfl_xor lowers the steps of width >= 4 to ds_bpermute, i.e. LDS-pipeline instructions.)
__device__ __forceinline__ float gigl_group_sum(float v, int group) {
#define GIGL_DPP_ADD(CTRL)                                                                                            \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true))
  if (group >= 2) GIGL_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  if (group >= 4) GIGL_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  if (group >= 8) GIGL_DPP_ADD(0x141);   // row_half_mirror
  if (group >= 16) GIGL_DPP_ADD(0x140);  // row_mirror
#undef GIGL_DPP_ADD
  if (group >= 32) {
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    if (group >= 64) return (r0 + r1) + (r2 + r3);
    return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
  }
  return v;
}
#endif
