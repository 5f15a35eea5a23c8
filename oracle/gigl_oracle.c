/*
 * gigl_oracle.c — CPU restatement of the reference algorithm for the sampler + collate hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * path (gigl_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / timed CPU baseline.
 *
 * Parity status: the reference sampler is Scala on Spark SQL 3.1.3 (no JVM in
 * this image -> unbuildable here) and no reference test or fixture pins the deterministic
 * xxhash64 permutation, so the *exact sample* is "parity unpinned": this file restates
 * SamplingStrategy.scala:16-82 and is pinned (a) on XXH64 against the canonical xxhash library
 * vectors (tests/golden/xxh64_int32.json, generated with python-xxhash 3.x = xxHash 0.8.x) and
 * (b) on the reference's own validity properties over its real sampler-output fixtures
 * (tests/golden/sgs_*).  The collate half follows abstract_graph_builder.py and is pinned by the
 * reference's known-answer unit tests (restated in tests/test_collate_oracle.py).
 *
 * All paths below are relative to the reference root.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GIGL_INVALID 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------
 * XXH64 of one little-endian int32 — Spark SQL `xxhash64(int)` == XXH64.hashInt(input, seed)
 * (Spark 3.1.3 catalyst XXH64.java, third-party, not in the reference tree; pinned version from
 * scala/build.sbt:37-38).  Identical to canonical XXH64 over the 4 LE bytes of the int.
 * Spark's SQL function uses seed 42 (XxHash64 expression default).
 * ---------------------------------------------------------------------------------------- */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

uint64_t gigl_oracle_xxh64_int32(int32_t input, uint64_t seed) {
  uint64_t h = seed + P5 + 4ULL;
  h ^= ((uint64_t)(uint32_t)input) * P1;
  h = rotl64(h, 23) * P2 + P3;
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

/* ------------------------------------------------------------------------------------------
 * S3  SamplingStrategy.hashBasedUniformPermutation
 *     (scala/subgraph_sampler/src/main/scala/libs/task/SamplingStrategy.scala:16-82)
 *   currentSeed   = seed * _counter                          (:36)   [Scala Int, wraps]
 *   _internal_seed= sum of the row's integer key columns     (:38-45) [Spark int add, wraps]
 *   _indices      = sequence(1, size)                        (:50)
 *   _hash         = xxhash64(x + _internal_seed + currentSeed)(:55)   [int add wraps; hash is int64]
 *   _permuted     = array_sort(arrays_zip(_hash, _indices))  (:63)   [signed hash asc, then index asc]
 *   out           = element_at(arr, i) for i in _permuted._indices (:72)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t key;
  int32_t idx;
} keyed_t;

static int keyed_cmp(const void* a, const void* b) {
  const keyed_t* x = (const keyed_t*)a;
  const keyed_t* y = (const keyed_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return 0;
}

static inline int32_t wrap_add(int32_t a, int32_t b) {
  return (int32_t)((uint32_t)a + (uint32_t)b);
}
static inline int32_t wrap_mul(int32_t a, int32_t b) {
  return (int32_t)((uint32_t)a * (uint32_t)b);
}

/* writes the FULL permutation of sorted_arr[0..n) to out[0..n) */
int gigl_oracle_hash_permutation(const uint32_t* sorted_arr, int64_t n, int32_t internal_seed,
                                 int32_t sampling_seed, int32_t counter, uint32_t* out) {
  if (n <= 0) return 0;
  keyed_t* k = (keyed_t*)malloc((size_t)n * sizeof(keyed_t));
  if (!k) return -3;
  int32_t current_seed = wrap_mul(sampling_seed, counter);
  for (int64_t i = 0; i < n; ++i) {
    int32_t x = (int32_t)(i + 1);
    int32_t arg = wrap_add(wrap_add(x, internal_seed), current_seed);
    k[i].key = (int64_t)gigl_oracle_xxh64_int32(arg, 42ULL);
    k[i].idx = x;
  }
  qsort(k, (size_t)n, sizeof(keyed_t), keyed_cmp);
  for (int64_t i = 0; i < n; ++i) out[i] = sorted_arr[k[i].idx - 1];
  free(k);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * S4 / S5 (+ S12 per-hop fanouts)  k-hop rooted sampling in the tree layout of include/gigl_hip.h.
 *   hop 1: sampleOnehopSrcNodesUniformly   SGSPureSparkV1Task.scala:313-388
 *          GROUP BY _dst_node -> array_sort(collect_list(_src_node)) -> permute(K={_dst_node},
 *          counter 1) -> slice(1, f)
 *   hop 2: sampleTwohopSrcNodesUniformly   :390-494
 *          explode hop 1 -> JOIN edges ON _dst_node=_1_hop -> GROUP BY (_0_hop,_1_hop) -> sort ->
 *          permute(K={_0_hop,_1_hop}, counter 2) -> slice(1, f)
 *   hop k>2: same rule continued along the path (the reference hard-wires 2 hops; per-hop fanouts
 *          come from SamplingOp DAGs, GraphDBSampler.scala:40-148).
 * rowptr/col: CSC by destination, rows ascending.  canonical_order == 0: each parent's samples in
 * PERMUTATION order (the reference's slice order); == 1: the same SET written ascending (the HIP
 * path's canonical form; the reference output is a set).  The tree positions of deeper hops follow
 * the order chosen here; the sampled sets do not depend on it (K is a sum over the path).
 * first_counter = value of the process-global _counter at the first hop (1 for a fresh JVM).
 * ---------------------------------------------------------------------------------------- */
int gigl_oracle_sample_khop(int64_t n_nodes, const int64_t* rowptr, const uint32_t* col,
                            const uint32_t* roots, int32_t b, const int32_t* fanouts, int32_t hops,
                            int32_t sampling_seed, int32_t first_counter, int32_t canonical_order,
                            uint32_t** nbr, int32_t** cnt) {
  if (hops < 1 || hops > 4) return -1;
  int64_t parents = b;
  const uint32_t* parent_ids = roots;
  int32_t* parent_ksum = (int32_t*)malloc((size_t)(b > 0 ? b : 1) * sizeof(int32_t));
  if (!parent_ksum) return -3;
  for (int32_t i = 0; i < b; ++i) parent_ksum[i] = (int32_t)roots[i];
  int64_t maxdeg = 0;
  for (int64_t v = 0; v < n_nodes; ++v)
    if (rowptr[v + 1] - rowptr[v] > maxdeg) maxdeg = rowptr[v + 1] - rowptr[v];
  uint32_t* perm = (uint32_t*)malloc((size_t)(maxdeg > 0 ? maxdeg : 1) * sizeof(uint32_t));
  if (!perm) {
    free(parent_ksum);
    return -3;
  }
  for (int32_t k = 0; k < hops; ++k) {
    int32_t f = fanouts[k];
    int64_t slots = parents * f;
    int32_t* ksum_next = (int32_t*)malloc((size_t)(slots > 0 ? slots : 1) * sizeof(int32_t));
    if (!ksum_next) return -3;
    for (int64_t p = 0; p < parents; ++p) {
      uint32_t v = parent_ids[p];
      uint32_t* o = nbr[k] + p * f;
      int32_t c = 0;
      if (v != GIGL_INVALID && (int64_t)v < n_nodes) {
        int64_t s = rowptr[v], deg = rowptr[v + 1] - rowptr[v];
        if (deg > 0) {
          gigl_oracle_hash_permutation(col + s, deg, parent_ksum[p], sampling_seed,
                                       first_counter + k, perm);
          c = (int32_t)(deg < f ? deg : f);
          for (int32_t j = 0; j < c; ++j) o[j] = perm[j];
          if (canonical_order) { /* the sampled SET in ascending id order (insertion sort, c <= f) */
            for (int32_t j = 1; j < c; ++j) {
              uint32_t x = o[j];
              int32_t q = j - 1;
              while (q >= 0 && o[q] > x) {
                o[q + 1] = o[q];
                --q;
              }
              o[q + 1] = x;
            }
            /* a multiset row (directed multi-edges) can put one id at several sampled positions: the canonical
             * form is the SET of sampled ids */
            int32_t w = 0;
            for (int32_t j = 0; j < c; ++j)
              if (j == 0 || o[j] != o[w - 1]) o[w++] = o[j];
            c = w;
          }
        }
      }
      for (int32_t j = c; j < f; ++j) o[j] = GIGL_INVALID;
      for (int32_t j = 0; j < f; ++j)
        ksum_next[p * f + j] = j < c ? wrap_add(parent_ksum[p], (int32_t)o[j]) : 0;
      cnt[k][p] = c;
    }
    free(parent_ksum);
    parent_ksum = ksum_next;
    parent_ids = nbr[k];
    parents = slots;
  }
  free(parent_ksum);
  free(perm);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * open-addressing u64 -> i64 map used by the collate restatements
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t* keys;
  int64_t* vals;
  uint64_t mask;
} map_t;
#define MAP_EMPTY 0xFFFFFFFFFFFFFFFFULL

static int map_init(map_t* m, int64_t cap_items) {
  uint64_t cap = 16;
  while (cap < (uint64_t)cap_items * 2 + 2) cap <<= 1;
  m->keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
  m->vals = (int64_t*)malloc(cap * sizeof(int64_t));
  if (!m->keys || !m->vals) return -3;
  memset(m->keys, 0xFF, cap * sizeof(uint64_t));
  m->mask = cap - 1;
  return 0;
}
static void map_free(map_t* m) {
  free(m->keys);
  free(m->vals);
}
static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
/* returns slot; *found tells whether key was present */
static inline uint64_t map_find(map_t* m, uint64_t key, int* found) {
  uint64_t s = mix64(key) & m->mask;
  while (m->keys[s] != MAP_EMPTY && m->keys[s] != key) s = (s + 1) & m->mask;
  *found = m->keys[s] == key;
  return s;
}

/* ------------------------------------------------------------------------------------------
 * T2 / T3  reference collate: GraphBuilder first-seen remap + edge dedup, then coalesce.
 *   __remap_node            abstract_graph_builder.py:16-24   local id = first-seen order
 *   add_graph_data          :49-100   per sample: nodes (sample order) then edges
 *   add_edge skip_if_exists :144-145  drop (local src, local dst) already present
 *   PygGraphBuilder.build   pyg_graph_builder.py:20-69  edge_index = [[src...],[dst...]] in insertion order
 *   coalesce()              rooted_node_neighborhood_data_loader.py:144 (PyG: sort by (src,dst), unique)
 * Homogeneous graphs only (one node type, one edge type).
 * Inputs: per-sample node id lists and edge lists, concatenated with offsets.
 * Outputs: out_nodes[*n_nodes] global id per local id; edges as local (src,dst); returns 0.
 * An edge whose endpoint was never registered as a node is the reference's TypeError (:26-30): -1.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t s, d;
} pair_t;
static int pair_cmp(const void* a, const void* b) {
  const pair_t* x = (const pair_t*)a;
  const pair_t* y = (const pair_t*)b;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  return 0;
}

int gigl_oracle_collate_reference(int32_t n_samples, const int64_t* node_off,
                                  const uint32_t* node_ids, const int64_t* edge_off,
                                  const uint32_t* edge_src, const uint32_t* edge_dst,
                                  int32_t do_coalesce, uint32_t* out_nodes, int64_t* n_nodes,
                                  int64_t* out_src, int64_t* out_dst, int64_t* n_edges) {
  map_t nm, em;
  int64_t tot_nodes = node_off[n_samples], tot_edges = edge_off[n_samples];
  if (map_init(&nm, tot_nodes) || map_init(&em, tot_edges)) return -3;
  int64_t nn = 0, ne = 0;
  int found;
  for (int32_t s = 0; s < n_samples; ++s) {
    for (int64_t i = node_off[s]; i < node_off[s + 1]; ++i) {
      uint64_t slot = map_find(&nm, node_ids[i], &found);
      if (!found) {
        nm.keys[slot] = node_ids[i];
        nm.vals[slot] = nn;
        out_nodes[nn++] = node_ids[i];
      }
    }
    for (int64_t i = edge_off[s]; i < edge_off[s + 1]; ++i) {
      uint64_t a = map_find(&nm, edge_src[i], &found);
      if (!found) goto bad;
      uint64_t c = map_find(&nm, edge_dst[i], &found);
      if (!found) goto bad;
      int64_t ls = nm.vals[a], ld = nm.vals[c];
      uint64_t ek = ((uint64_t)ls << 32) | (uint64_t)ld;
      uint64_t es = map_find(&em, ek, &found);
      if (found) continue;
      em.keys[es] = ek;
      em.vals[es] = ne;
      out_src[ne] = ls;
      out_dst[ne] = ld;
      ++ne;
    }
  }
  if (do_coalesce && ne > 1) {
    pair_t* p = (pair_t*)malloc((size_t)ne * sizeof(pair_t));
    if (!p) return -3;
    for (int64_t i = 0; i < ne; ++i) {
      p[i].s = out_src[i];
      p[i].d = out_dst[i];
    }
    qsort(p, (size_t)ne, sizeof(pair_t), pair_cmp);
    for (int64_t i = 0; i < ne; ++i) {
      out_src[i] = p[i].s;
      out_dst[i] = p[i].d;
    }
    free(p);
  }
  *n_nodes = nn;
  *n_edges = ne;
  map_free(&nm);
  map_free(&em);
  return 0;
bad:
  map_free(&nm);
  map_free(&em);
  return -1;
}

/* ------------------------------------------------------------------------------------------
 * Union graph in this library's LEVEL-ORDERED numbering (include/gigl_hip.h, gigl_union_build):
 * same node set / edge set as the reference collate above (checked in tests), different local ids.
 *   stream position: roots 0..b-1, then hop-0 slots, hop-1 slots, ... (valid slots only matter)
 *   level(root)=0; level(src) = min over union edges (src->dst) of level(dst)+1
 *   local id = rank by (level, first stream position)
 *   edges: unique (dst_local, src_local), CSR by dst, ascending src_local within a row.
 * meta[0]=n_nodes, meta[1]=n_edges, meta[2+l]=cumulative count through level l (l=0..hops).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t level;
  int64_t pos;
  uint32_t id;
} lnode_t;
static int lnode_cmp(const void* a, const void* b) {
  const lnode_t* x = (const lnode_t*)a;
  const lnode_t* y = (const lnode_t*)b;
  if (x->level != y->level) return x->level < y->level ? -1 : 1;
  if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
  return 0;
}

int gigl_oracle_union_build(const uint32_t* roots, int32_t b, const int32_t* fanouts, int32_t hops,
                            uint32_t* const* nbr, int32_t* meta, uint32_t* out_nodes,
                            int32_t* out_rowptr, int32_t* out_col, int32_t* root_local) {
  int64_t slots[4], total = b, parents = b;
  for (int32_t k = 0; k < hops; ++k) {
    slots[k] = parents * fanouts[k];
    total += slots[k];
    parents = slots[k];
  }
  map_t nm;
  if (map_init(&nm, total)) return -3;
  lnode_t* ln = (lnode_t*)malloc((size_t)(total > 0 ? total : 1) * sizeof(lnode_t));
  int64_t nn = 0, pos = 0;
  int found;
  /* pass 1: first positions */
  for (int32_t i = 0; i < b; ++i, ++pos) {
    uint64_t s = map_find(&nm, roots[i], &found);
    if (!found) {
      nm.keys[s] = roots[i];
      nm.vals[s] = nn;
      ln[nn].level = 0;
      ln[nn].pos = pos;
      ln[nn].id = roots[i];
      ++nn;
    }
  }
  for (int32_t k = 0; k < hops; ++k)
    for (int64_t j = 0; j < slots[k]; ++j, ++pos) {
      uint32_t v = nbr[k][j];
      if (v == GIGL_INVALID) continue;
      uint64_t s = map_find(&nm, v, &found);
      if (!found) {
        nm.keys[s] = v;
        nm.vals[s] = nn;
        ln[nn].level = 1 << 20;
        ln[nn].pos = pos;
        ln[nn].id = v;
        ++nn;
      }
    }
  /* pass 2: levels by relaxation over all (src -> dst) occurrences, `hops` rounds */
  for (int32_t round = 0; round < hops; ++round)
    for (int32_t k = 0; k < hops; ++k)
      for (int64_t j = 0; j < slots[k]; ++j) {
        uint32_t v = nbr[k][j];
        if (v == GIGL_INVALID) continue;
        int64_t p = j / fanouts[k];
        uint32_t d = k == 0 ? roots[p] : nbr[k - 1][p];
        int64_t is = nm.vals[map_find(&nm, v, &found)];
        int64_t id = nm.vals[map_find(&nm, d, &found)];
        if (ln[id].level + 1 < ln[is].level) ln[is].level = ln[id].level + 1;
      }
  /* rank */
  lnode_t* sorted = (lnode_t*)malloc((size_t)(nn > 0 ? nn : 1) * sizeof(lnode_t));
  memcpy(sorted, ln, (size_t)nn * sizeof(lnode_t));
  qsort(sorted, (size_t)nn, sizeof(lnode_t), lnode_cmp);
  for (int32_t l = 0; l < 14; ++l) meta[2 + l] = 0;
  for (int64_t i = 0; i < nn; ++i) {
    out_nodes[i] = sorted[i].id;
    nm.vals[map_find(&nm, sorted[i].id, &found)] = i;
    for (int32_t l = sorted[i].level; l <= hops; ++l) meta[2 + l] += 1;
  }
  for (int32_t i = 0; i < b; ++i) root_local[i] = (int32_t)nm.vals[map_find(&nm, roots[i], &found)];
  /* edges */
  int64_t cap_e = total - b, ne = 0;
  pair_t* ep = (pair_t*)malloc((size_t)(cap_e > 0 ? cap_e : 1) * sizeof(pair_t));
  for (int32_t k = 0; k < hops; ++k)
    for (int64_t j = 0; j < slots[k]; ++j) {
      uint32_t v = nbr[k][j];
      if (v == GIGL_INVALID) continue;
      int64_t p = j / fanouts[k];
      uint32_t d = k == 0 ? roots[p] : nbr[k - 1][p];
      ep[ne].s = nm.vals[map_find(&nm, d, &found)]; /* primary key: dst */
      ep[ne].d = nm.vals[map_find(&nm, v, &found)]; /* secondary: src */
      ++ne;
    }
  qsort(ep, (size_t)ne, sizeof(pair_t), pair_cmp);
  int64_t ue = 0;
  for (int64_t i = 0; i <= nn; ++i) out_rowptr[i] = 0;
  for (int64_t i = 0; i < ne; ++i) {
    if (i > 0 && ep[i].s == ep[i - 1].s && ep[i].d == ep[i - 1].d) continue;
    out_col[ue++] = (int32_t)ep[i].d;
    out_rowptr[ep[i].s + 1] += 1;
  }
  for (int64_t i = 0; i < nn; ++i) out_rowptr[i + 1] += out_rowptr[i];
  meta[0] = (int32_t)nn;
  meta[1] = (int32_t)ue;
  free(ep);
  free(sorted);
  free(ln);
  map_free(&nm);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * S1  enforceBidirectionalization + CSC build (SGSPureSparkV1Task.scala:218-258, :338-345)
 *   undirected: (LEAST, GREATEST) -> DISTINCT/dropDuplicates -> UNION reversed copy
 *   then per destination: ascending source list.  Duplicate (src,dst) pairs are dropped in both
 *   modes (this library's CSC is a simple graph; the reference's directed path keeps multi-edges
 *   as repeated list entries — see DESIGN.md "multi-edges").
 * Two-call protocol: pass col == NULL to get *e_out only.
 * ---------------------------------------------------------------------------------------- */
static int u64_cmp(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
int gigl_oracle_build_csc(int64_t n, int64_t e, const uint32_t* src, const uint32_t* dst,
                          int32_t is_directed, int64_t* rowptr, uint32_t* col, int64_t* e_out) {
  int64_t m = is_directed ? e : 2 * e;
  uint64_t* keys = (uint64_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(uint64_t));
  if (!keys) return -3;
  int64_t c = 0;
  for (int64_t i = 0; i < e; ++i) {
    if ((int64_t)src[i] >= n || (int64_t)dst[i] >= n) {
      free(keys);
      return -1;
    }
    keys[c++] = ((uint64_t)dst[i] << 32) | src[i];
    if (!is_directed) keys[c++] = ((uint64_t)src[i] << 32) | dst[i];
  }
  qsort(keys, (size_t)c, sizeof(uint64_t), u64_cmp);
  int64_t u = 0;
  /* is_directed == 2: the directed path's collect_list keeps repeated (src, dst) rows (SGSPureSparkV1Task.scala:337,
   * 442): rows are ascending multisets */
  for (int64_t i = 0; i < c; ++i)
    if (is_directed == 2 || i == 0 || keys[i] != keys[i - 1]) keys[u++] = keys[i];
  *e_out = u;
  if (col) {
    for (int64_t i = 0; i <= n; ++i) rowptr[i] = 0;
    for (int64_t i = 0; i < u; ++i) {
      col[i] = (uint32_t)(keys[i] & 0xFFFFFFFFu);
      rowptr[(keys[i] >> 32) + 1] += 1;
    }
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  }
  free(keys);
  return 0;
}

/* ---- split generator hash slots (scala/split_generator/src/main/scala/lib/assigners/AbstractAssigners.scala:30-111).
 * scala.util.hashing.MurmurHash3.bytesHash(data) == MurmurHash3_x86_32(data, seed = 0x3c074a61 "arraySeed") — the
 * published algorithm of Austin Appleby's MurmurHash3 (scala-library is a third-party dependency absent from
 * /root/reference: restated from the public reference implementation, pinned on its public known answers in
 * tests/test_split_generator.py).  slot = Math.floorMod(hash, 10000). */
int32_t gigl_oracle_murmur3_x86_32(const uint8_t* data, int64_t len, uint32_t seed) {
  uint32_t h = seed;
  const int64_t nblocks = len / 4;
  for (int64_t i = 0; i < nblocks; ++i) {
    uint32_t k;
    memcpy(&k, data + 4 * i, 4); /* little-endian host */
    k *= 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    h ^= k;
    h = (h << 13) | (h >> 19);
    h = h * 5u + 0xe6546b64u;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fall through */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fall through */
    case 1:
      k1 ^= tail[0];
      k1 *= 0xcc9e2d51u;
      k1 = (k1 << 15) | (k1 >> 17);
      k1 *= 0x1b873593u;
      h ^= k1;
  }
  h ^= (uint32_t)len;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return (int32_t)h;
}

/* key = "<a>-<type>" (has_b == 0) or "<a>-<type>-<b>" with (a, b) ordered (min, max) first when symmetric */
int gigl_oracle_split_slots(const uint32_t* a, const uint32_t* b, int64_t n, int32_t type, int32_t has_b,
                            int32_t symmetric, int32_t* slots) {
  char key[64];
  for (int64_t i = 0; i < n; ++i) {
    uint32_t x = a[i], y = has_b ? b[i] : 0;
    if (has_b && symmetric && x > y) {
      uint32_t t = x;
      x = y;
      y = t;
    }
    int len = has_b ? snprintf(key, sizeof(key), "%u-%d-%u", x, type, y) : snprintf(key, sizeof(key), "%u-%d", x, type);
    int32_t h = gigl_oracle_murmur3_x86_32((const uint8_t*)key, len, 0x3c074a61u);
    int32_t m = h % 10000;
    slots[i] = m < 0 ? m + 10000 : m;
  }
  return 0;
}
