/* abi_parity.c — the drop-in boundary exercised from plain C (no Python, no torch): links libgigl_hip.so through
 * include/gigl_hip.h exactly as a cgo / JNI shim would, and checks the device results bit-for-bit against the C
 * oracle (oracle/gigl_oracle.c, test infrastructure).  Built and run by tests/test_gpu_c_abi.py on the GPU box:
 *   gcc -O2 -Iinclude tests/c/abi_parity.c -o abi_parity -Lgigl_amd -lgigl_hip -Loracle -lgigl_oracle -Wl,-rpath,...
 * Exit code 0 = parity, nonzero = the failing step (message on stderr). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gigl_hip.h"

int gigl_oracle_sample_khop(int64_t n_nodes, const int64_t* rowptr, const uint32_t* col, const uint32_t* roots,
                            int32_t b, const int32_t* fanouts, int32_t hops, int32_t sampling_seed,
                            int32_t first_counter, int32_t canonical_order, uint32_t** nbr, int32_t** cnt);
int gigl_oracle_build_csc(int64_t n, int64_t e, const uint32_t* src, const uint32_t* dst, int32_t is_directed,
                          int64_t* rowptr, uint32_t* col, int64_t* e_out);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}

#define CHECK(call, step)                                                               \
  do {                                                                                  \
    int32_t rc_ = (call);                                                               \
    if (rc_ != GIGL_OK) {                                                               \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? gigl_last_error(ctx) : ""); \
      return step;                                                                      \
    }                                                                                   \
  } while (0)

int main(void) {
  const int64_t n = 50000, e = 400000;
  const int32_t b = 512, hops = 2, fanouts[2] = {25, 10};
  uint32_t* src = malloc(e * 4);
  uint32_t* dst = malloc(e * 4);
  for (int64_t i = 0; i < e; ++i) {
    /* skewed: a few hubs so that long rows (table path) and short rows both occur */
    uint32_t a = rnd() % n, c = rnd() % n;
    if ((rnd() & 7) == 0) c = rnd() % 16;
    src[i] = a;
    dst[i] = c;
  }
  int64_t* rowptr = malloc((n + 1) * 8);
  uint32_t* col = malloc(2 * e * 4);
  int64_t e_csc = 0;
  if (gigl_oracle_build_csc(n, e, src, dst, 0, rowptr, col, &e_csc) != 0) return 1;
  uint32_t roots[512];
  for (int i = 0; i < b; ++i) roots[i] = i < 16 ? (uint32_t)i : rnd() % n;

  gigl_ctx* ctx = NULL;
  gigl_graph* g = NULL;
  CHECK(gigl_ctx_create(0, &ctx), 2);
  /* ingest on the device from the same COO: must give the oracle's CSC */
  CHECK(gigl_graph_build_from_coo(ctx, n, e, src, dst, GIGL_LOC_HOST, 0, &g), 3);
  int64_t gn = 0, ge = 0;
  CHECK(gigl_graph_info(g, &gn, &ge), 4);
  if (gn != n || ge != e_csc) {
    fprintf(stderr, "CSC size: device %lld edges, oracle %lld\n", (long long)ge, (long long)e_csc);
    return 4;
  }
  const int64_t* d_rowptr;
  const uint32_t* d_col;
  CHECK(gigl_graph_device_ptrs(g, &d_rowptr, &d_col), 5);
  uint32_t* col_back = malloc(e_csc * 4);
  CHECK(gigl_memcpy(ctx, col_back, GIGL_LOC_HOST, d_col, GIGL_LOC_DEVICE, e_csc * 4), 5);
  if (memcmp(col_back, col, e_csc * 4) != 0) {
    fprintf(stderr, "CSC columns differ from the oracle\n");
    return 5;
  }

  /* k-hop sampling through the C ABI: device buffers allocated by copying zeros (the ABI has no allocator of its
   * own: a caller owns its buffers; here they are carved from one features_load allocation for brevity) */
  const int64_t s0 = (int64_t)b * fanouts[0], s1 = s0 * fanouts[1];
  const int64_t words = b + s0 + s1 + b + s0;
  float* zeros = calloc(words, 4);
  gigl_feat* buf = NULL;
  CHECK(gigl_features_load(ctx, words, 1, GIGL_DTYPE_F32, zeros, GIGL_LOC_HOST, &buf), 6);
  const void* base = NULL;
  CHECK(gigl_features_device_ptr(buf, &base, NULL, NULL, NULL), 6);
  uint32_t* d = (uint32_t*)base;
  uint32_t* d_roots = d;
  gigl_tree tree;
  memset(&tree, 0, sizeof tree);
  tree.nbr[0] = d + b;
  tree.nbr[1] = d + b + s0;
  tree.cnt[0] = (int32_t*)(d + b + s0 + s1);
  tree.cnt[1] = (int32_t*)(d + b + s0 + s1 + b);
  CHECK(gigl_memcpy(ctx, d_roots, GIGL_LOC_DEVICE, roots, GIGL_LOC_HOST, b * 4), 7);
  CHECK(gigl_sample_khop(ctx, g, d_roots, b, fanouts, hops, 42, GIGL_MODE_SPARK_HASH, &tree), 8);
  uint32_t* h_nbr0 = malloc(s0 * 4);
  uint32_t* h_nbr1 = malloc(s1 * 4);
  CHECK(gigl_memcpy(ctx, h_nbr0, GIGL_LOC_HOST, tree.nbr[0], GIGL_LOC_DEVICE, s0 * 4), 9);
  CHECK(gigl_memcpy(ctx, h_nbr1, GIGL_LOC_HOST, tree.nbr[1], GIGL_LOC_DEVICE, s1 * 4), 9);

  uint32_t* o_nbr[2] = {malloc(s0 * 4), malloc(s1 * 4)};
  int32_t* o_cnt[2] = {malloc(b * 4), malloc(s0 * 4)};
  if (gigl_oracle_sample_khop(n, rowptr, col, roots, b, fanouts, hops, 42, 1, 1, o_nbr, o_cnt) != 0) return 10;
  if (memcmp(h_nbr0, o_nbr[0], s0 * 4) != 0 || memcmp(h_nbr1, o_nbr[1], s1 * 4) != 0) {
    fprintf(stderr, "sampled trees differ from the oracle\n");
    return 11;
  }
  int64_t sampled = 0;
  for (int64_t i = 0; i < b; ++i) sampled += o_cnt[0][i];
  for (int64_t i = 0; i < s0; ++i) sampled += o_cnt[1][i];

  /* edge hydration keys (gigl_edge_ids): every sampled hop-1 edge resolves to its position in the resident CSC */
  const int64_t extra_words = s0 + 2 * s0;
  float* zeros2 = calloc(extra_words, 4);
  gigl_feat* buf2 = NULL;
  CHECK(gigl_features_load(ctx, extra_words, 1, GIGL_DTYPE_F32, zeros2, GIGL_LOC_HOST, &buf2), 15);
  const void* base2 = NULL;
  CHECK(gigl_features_device_ptr(buf2, &base2, NULL, NULL, NULL), 15);
  uint32_t* d_dst = (uint32_t*)base2;
  int64_t* d_eid = (int64_t*)((uint32_t*)base2 + s0);
  uint32_t* h_dst = malloc(s0 * 4);
  for (int64_t i = 0; i < s0; ++i) h_dst[i] = roots[i / fanouts[0]];
  CHECK(gigl_memcpy(ctx, d_dst, GIGL_LOC_DEVICE, h_dst, GIGL_LOC_HOST, s0 * 4), 16);
  CHECK(gigl_edge_ids(ctx, g, tree.nbr[0], d_dst, s0, d_eid), 17);
  int64_t* h_eid = malloc(s0 * 8);
  CHECK(gigl_memcpy(ctx, h_eid, GIGL_LOC_HOST, d_eid, GIGL_LOC_DEVICE, s0 * 8), 18);
  for (int64_t i = 0; i < s0; ++i) {
    if (h_nbr0[i] == GIGL_INVALID) {
      if (h_eid[i] != -1) return 19;
      continue;
    }
    const int64_t p = h_eid[i];
    if (p < rowptr[h_dst[i]] || p >= rowptr[h_dst[i] + 1] || col[p] != h_nbr0[i]) return 20;
  }
  /* SamplingOp-DAG frontier union (gigl_rows_dedup): hop-2 slots of a root as one row */
  const int32_t width = fanouts[0] * fanouts[1];
  CHECK(gigl_rows_dedup(ctx, tree.nbr[1], b, width), 21);
  uint32_t* h_dd = malloc(s1 * 4);
  CHECK(gigl_memcpy(ctx, h_dd, GIGL_LOC_HOST, tree.nbr[1], GIGL_LOC_DEVICE, s1 * 4), 22);
  for (int64_t r = 0; r < b; ++r)
    for (int32_t q = 0; q < width; ++q) {
      const uint32_t before = h_nbr1[r * width + q], after = h_dd[r * width + q];
      int first = 1;
      for (int32_t t = 0; t < q; ++t)
        if (h_nbr1[r * width + t] == before) first = 0;
      if (after != (before != GIGL_INVALID && first ? before : GIGL_INVALID)) return 23;
    }
  /* Avro layout is host arithmetic: 128 floats + "user" -> 16 kB blocks of 30 records */
  int32_t per_block = 0;
  int64_t n_blocks = 0, avro_bytes = 0;
  CHECK(gigl_avro_embeddings_layout(1000, 128, 4, &per_block, &n_blocks, &avro_bytes), 24);
  if (per_block < 1 || n_blocks != (1000 + per_block - 1) / per_block || avro_bytes < 1000 * (128 * 4 + 8)) return 25;
  if (gigl_avro_embeddings_layout(10, 4, 100000, &per_block, &n_blocks, &avro_bytes) != GIGL_E_INVALID_ARG) return 26;
  gigl_features_destroy(buf2);

  /* error behaviour of the boundary */
  int32_t bad_fan[2] = {GIGL_MAX_FANOUT + 1, 1};
  if (gigl_sample_khop(ctx, g, d_roots, b, bad_fan, 2, 42, 0, &tree) != GIGL_E_UNSUPPORTED) return 12;
  if (gigl_sample_khop(ctx, NULL, d_roots, b, fanouts, 2, 42, 0, &tree) != GIGL_E_INVALID_ARG) return 13;
  if (strlen(gigl_last_error(ctx)) == 0) return 14;

  gigl_features_destroy(buf);
  gigl_graph_destroy(g);
  gigl_ctx_destroy(ctx);
  printf("C ABI parity OK: %lld CSC edges, %lld sampled edges bit-identical to the oracle\n", (long long)e_csc,
         (long long)sampled);
  return 0;
}
