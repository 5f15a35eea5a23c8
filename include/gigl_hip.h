/*
 * gigl_hip.h — C ABI of libgigl_hip.so: the MI355X (gfx950) k-hop subgraph sampler +
 * GNN aggregation hot path of GiGL.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own for
 * this path (it is Scala-on-Spark + PyTorch-Geometric); the closest operator interface is the
 * Scala trait `KHopSamplerService` and the Python collate / conv calls.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - every function returns int32 status: 0 = OK, <0 = error (GIGL_E_*); the message for the
 *    last error on a ctx is `gigl_last_error(ctx)` (owned by the ctx, valid until next call).
 *  - a ctx is bound to ONE device and ONE HIP stream and is single-threaded — this mirrors
 *    "one sampler service per Spark partition" (setup()/teardown() per partition:
 *    scala_spark35/subgraph_sampler/src/main/scala/libs/task/graphdb/
 *    GraphDBNodeAnchorBasedLinkPredictionTask.scala:62-78).  Different ctxs are independent.
 *  - pointers tagged DEVICE are HBM addresses on the ctx's device; HOST are host addresses.
 *    Per-batch calls take DEVICE pointers only and never synchronise with the host, so a whole
 *    step can be captured into a hipGraph.
 *  - node ids are uint32 (proto/snapchat/research/gbml/graph_schema.proto:7,20-21), edge
 *    offsets int64, batch-local ids int32.
 *  - GIGL_INVALID (0xFFFFFFFF) marks an empty slot in the tree layout.
 */
#ifndef GIGL_HIP_H
#define GIGL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GIGL_OK 0
#define GIGL_E_INVALID_ARG (-1)
#define GIGL_E_HIP (-2)
#define GIGL_E_OOM (-3)
#define GIGL_E_UNSUPPORTED (-4)
#define GIGL_E_NO_DEVICE (-5)

#define GIGL_INVALID 0xFFFFFFFFu
#define GIGL_MAX_HOPS 4
/* fanouts up to GIGL_FAST_FANOUT take the wave-resident selection (one candidate per lane: the tuned path of every
 * benchmark configuration); larger ones, up to GIGL_MAX_FANOUT, a workgroup-per-row selection with the same results
 * contract (parity mode; gigl_sample_khop, gigl_expand_frontier, the one-call / sharded / typed plans built on them and
 * the record encoders: the reference's numNeighborsToSample is any int, SGSPureSparkV1Task.scala:313-388) */
#define GIGL_FAST_FANOUT 64
#define GIGL_MAX_FANOUT 1024

#define GIGL_LOC_HOST 0
#define GIGL_LOC_DEVICE 1

#define GIGL_DTYPE_F32 0
#define GIGL_DTYPE_F16 1

/* serialized training-sample kinds (training_samples_schema.proto) */
#define GIGL_REC_ROOTED_NODE_NEIGHBORHOOD 0 /* also SupervisedNodeClassificationSample: same fields + labels */
#define GIGL_REC_NODE_ANCHOR_LINK_PRED 1

/* sampling modes */
#define GIGL_MODE_SPARK_HASH 0 /* parity: xxhash64-keyed permutation, SamplingStrategy.scala:16-82 */
#define GIGL_MODE_FAST 1       /* NOT parity: counter-based RNG positions; labelled as such everywhere */
#define GIGL_MODE_REPLACE 2    /* sampling WITH replacement (experimental_flags.sample_with_replacement: the reference's
                                  sampleWithReplacementUDF, SGSPureSparkV1Task.scala:42-50,355-364, draws from an
                                  unseeded java.util.Random — no parity definition): f uniform draws per parent from a
                                  counter-based generator keyed by the path sum, repeats possible, cnt = f whenever
                                  the parent has in-edges.  gigl_sample_khop only: trees with repeated ids are not
                                  valid input of the one-call plan */

typedef struct gigl_ctx gigl_ctx;
typedef struct gigl_graph gigl_graph;
typedef struct gigl_feat gigl_feat;

/* ---- lifecycle: KHopSamplerService.setup()/teardown()
 *      (scala_spark35/common/src/main/scala/graphdb/KHopSamplerService.scala:10-33) ---- */
int32_t gigl_version(void);
int32_t gigl_ctx_create(int32_t device, gigl_ctx** out);
int32_t gigl_ctx_destroy(gigl_ctx* ctx);
const char* gigl_last_error(gigl_ctx* ctx);
/* bind the ctx to a caller-owned hipStream_t; NULL is the legacy default stream.  Until this is
 * called the ctx runs on a private non-blocking stream created by gigl_ctx_create. */
int32_t gigl_ctx_set_stream(gigl_ctx* ctx, void* hip_stream);
/* Plans created on this ctx AFTER the call (gigl_sage_plan_create, gigl_gat_plan_create, the three training plans)
 * provision their batch workspace for the WORST batch instead of the typical one.  A regular plan holds
 * b*(1 + f0 + ...) inner rows — every batch in which no root is another root's sampled neighbour; a batch beyond that
 * (children of root-valued slots move up a level: up to the whole tree, common on graphs of a few thousand nodes) is
 * reported through meta[GIGL_META_OVERFLOW] (NaN rows / NaN loss, nothing trained).  on = 1: rows for every node of the
 * tree and the generic union build (every node numbered, tables sized by the tree): no batch overflows.  The entry
 * points create such a plan the first time a batch overflows the regular one and redo the batch through it — the
 * reference's collate has no such bound (rooted_node_neighborhood_data_loader.py:78-158). */
int32_t gigl_ctx_set_wide_workspaces(gigl_ctx* ctx, int32_t on);
int32_t gigl_ctx_synchronize(gigl_ctx* ctx);
/* pre-size the ctx scratch arena (bytes); grows on demand otherwise (grow = hipMalloc, so
 * warm up once before capturing a hipGraph) */
int32_t gigl_ctx_reserve(gigl_ctx* ctx, int64_t bytes);
/* synchronous copy between host/device buffers on the ctx stream (`*_loc` = GIGL_LOC_*): the
 * "block_to_host" of the boundary — results leave HBM only through this or the caller's own copies */
int32_t gigl_memcpy(gigl_ctx* ctx, void* dst, int32_t dst_loc, const void* src, int32_t src_loc,
                    int64_t bytes);

/* ---- per-kernel timing with HIP events recorded on the ctx stream (the reference's analogue is the
 *      @profileit wall timer, python/gigl/common/metrics/decorators.py:68-111).  `mask` selects
 *      kernel ids (bit i = GIGL_K_*); capacity = max launches kept.  gigl_profile_read synchronises
 *      the stream and returns the summed duration and launch count of one kernel id. */
#define GIGL_K_EXPAND 0        /* frontier-expand, wave-per-parent (all hops) */
#define GIGL_K_EXPAND_HEAVY 1  /* frontier-expand, workgroup-per-parent rows */
#define GIGL_K_FIND_HEAVY 2
#define GIGL_K_UNION_INSERT 3
#define GIGL_K_UNION_RELAX 4
#define GIGL_K_UNION_NODES 5   /* flag + scan + assign */
#define GIGL_K_UNION_EDGE_SORT 6
#define GIGL_K_UNION_CSR 7     /* unique + scan + col/rowptr */
#define GIGL_K_GATHER_MEAN 8
#define GIGL_K_LINEAR 9
#define GIGL_K_GATHER_BWD 10   /* gigl_gather_reduce_backward (scatter of the layer's input gradient: fp32 atomics) */
#define GIGL_K_DIST_PREP 11    /* sharded plan: clear / bucket / claim / scatter / locate (what the exchanges need around them) */
#define GIGL_K_DIST_SERVE 12   /* sharded plan, owner side: the requested feature rows gathered into the send buffer */
#define GIGL_K_COUNT 13
int32_t gigl_profile_enable(gigl_ctx* ctx, uint32_t mask, int32_t capacity);
int32_t gigl_profile_read(gigl_ctx* ctx, int32_t kernel_id, double* total_ms, int64_t* launches);
int32_t gigl_profile_reset(gigl_ctx* ctx);

/* ---- graph ingest: replaces loadEdgeDataframeIntoSparkSql + enforceBidirectionalization
 *      (scala/subgraph_sampler/src/main/scala/libs/task/pureSpark/SGSPureSparkV1Task.scala:120-286)
 * CSC by destination: rowptr[n+1] (int64), col[e] (uint32 in-neighbour ids), each row sorted
 * ascending and duplicate-free (== array_sort(collect_list(_src_node)) per _dst_node on a simple
 * graph, SGSPureSparkV1Task.scala:338-345). `loc` says where rowptr/col live; data is copied. */
int32_t gigl_graph_load_csc(gigl_ctx* ctx, int64_t n, int64_t e, const int64_t* rowptr,
                            const uint32_t* col, int32_t loc, gigl_graph** out);
/* COO edge list (src -> dst).  is_directed == 0 applies the reference's bidirectionalisation:
 * canonicalise (min,max), drop duplicates, union with the reversed copy (:218-258).  Always
 * sorts rows ascending and drops duplicate (src,dst) pairs.  Runs on the device. */
int32_t gigl_graph_build_from_coo(gigl_ctx* ctx, int64_t n, int64_t e, const uint32_t* src,
                                  const uint32_t* dst, int32_t loc, int32_t is_directed,
                                  gigl_graph** out);
/* The same ingest for ONE rank of a graph hash-partitioned over `world` ranks — replaces the partitioner of the
 * reference's distributed path (python/gigl/distributed/dist_link_prediction_data_partitioner.py:666-714: owner(v) =
 * v % world; edges follow the DESTINATION node, the INCOMING-sampling choice of :692-695) applied to the edge table as
 * loadEdgeDataframeIntoSparkSql reads it: every record in one (directed) or both (undirected) orientations, the
 * orientations whose destination `rank` owns are kept as row dst / world of the shard; source ids stay global.  The
 * result has ceil((n - rank) / world) rows and equals the owned rows of gigl_graph_build_from_coo's graph (what
 * gigl_expand_frontier / gigl_dist_plan_create take as `shard`).  `src` / `dst` hold the WHOLE edge list (every rank
 * reads it, each keeps 1 / world of it). */
int32_t gigl_graph_build_shard_from_coo(gigl_ctx* ctx, int64_t n, int32_t rank, int32_t world, int64_t e,
                                        const uint32_t* src, const uint32_t* dst, int32_t loc, int32_t is_directed,
                                        gigl_graph** out);
/* edge hydration keys: replaces the JOIN on (_from, _to) of hydrateEdges (SGSPureSparkV1Task.scala:549-593).
 * eid[i] = position of the edge src[i] -> dst[i] in the resident `col` array (= rowptr[dst] + index of src in the
 * ascending row), -1 when the graph has no such edge.  An edge-feature table stored in `col` order (row p = the
 * `_edge_features` of edge p; both directions of a bidirectionalised edge carry the same row, :218-258) is then
 * indexed by eid.  All pointers DEVICE.  gigl_union_edge_ids does the lookup for every edge of a union graph:
 * eid[p] for the positions p of u->col that hold an edge (global src = nodes[col[p]], dst = nodes[row]),
 * -1 elsewhere; eid: [u->cap_edges]. */
int32_t gigl_edge_ids(gigl_ctx* ctx, gigl_graph* g, const uint32_t* src, const uint32_t* dst, int64_t m,
                      int64_t* eid);
struct gigl_union;
int32_t gigl_union_edge_ids(gigl_ctx* ctx, gigl_graph* g, const struct gigl_union* u, int64_t* eid);
int32_t gigl_graph_info(gigl_graph* g, int64_t* n, int64_t* e);
/* DEVICE pointers to the resident CSC (borrowed; valid until gigl_graph_destroy) */
int32_t gigl_graph_device_ptrs(gigl_graph* g, const int64_t** rowptr, const uint32_t** col);
int32_t gigl_graph_destroy(gigl_graph* g);

/* ---- preprocessed-table reader (HOST code, no ctx): TFRecord framing + tf.Example decoding into dense columns.
 *      Replaces the spark-tfrecord reads and the column selection / casts of loadNodeDataframeIntoSparkSql and
 *      loadEdgeDataframeIntoSparkSql (SGSPureSparkV1Task.scala:52-118, :120-216; TFRecordIO.scala:20-51).
 * gigl_tfrecord_index: walks the frames of a TFRecord file image (HOST buffer), optionally verifying both masked
 *   CRC-32C words; writes payload offset / length of the first `cap` records and the record count (call with
 *   cap = 0 to count).  GIGL_E_INVALID_ARG on a truncated file or a CRC mismatch.
 * gigl_tfexample_decode: column c of record i <- the values of feature `name` (Int64List -> int64, or cast to
 *   float for an F32 column, as the reference concatenates integer feature columns into array<float>; FloatList ->
 *   float), first `width` values, zero-filled when the key is absent or shorter; counts[i] (optional) = values
 *   present.  Multi-threaded over records.  A malformed record or a kind mismatch returns GIGL_E_INVALID_ARG and
 *   *bad_record = its index. */
#define GIGL_COL_I64 0
#define GIGL_COL_F32 1
typedef struct gigl_column {
  const char* name;
  int32_t kind;    /* GIGL_COL_* */
  int32_t width;   /* values per record */
  void* out;       /* HOST [n][width] int64 or float */
  int32_t* counts; /* HOST [n] or NULL */
} gigl_column;
int32_t gigl_tfrecord_index(const uint8_t* buf, int64_t len, int32_t verify_crc, int64_t cap, int64_t* payload_off,
                            int64_t* payload_len, int64_t* n_records);
int32_t gigl_tfexample_decode(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t n,
                              const gigl_column* cols, int32_t n_cols, int32_t n_threads, int64_t* bad_record);

/* ---- serialized training samples -> one batch graph (HOST code, no ctx): the collate step for samples that
 *      arrive as TFRecords.  Replaces GbmlProtosTranslator.graph_data_from_GraphPb
 *      (python/gigl/src/common/translators/gbml_protos_translator.py:101-121), GraphBuilder.add_node/add_edge
 *      (python/gigl/src/common/graph_builder/abstract_graph_builder.py:16-24,49-150), PygGraphBuilder.build
 *      (.../pyg_graph_builder.py:20-69) and the loops of the three collate functions
 *      (python/gigl/src/training/v1/lib/data_loaders/{rooted_node_neighborhood,supervised_node_classification,
 *      node_anchor_based_link_prediction}_data_loader.py).
 * `kind` GIGL_REC_ROOTED_NODE_NEIGHBORHOOD reads root_node=1, neighborhood=2 and, when present, root_node_labels=3
 * (a SupervisedNodeClassificationSample); GIGL_REC_NODE_ANCHOR_LINK_PRED reads root_node=1, neighborhood=3,
 * pos_edges=4, hard_neg_edges=2.  Result: local ids in first-seen order (a sample's nodes, then its edges), a node
 * seen again must carry allclose features, duplicate (src,dst) edges dropped, edge_index sorted by (src,dst),
 * root_local[b], labels[b]/has_label[b] (root_node_labels[0]), pos/hard-neg targets as offsets[b+1] + local ids.
 * Errors (GIGL_E_INVALID_ARG + message in `err`): malformed record, node re-added with different features, edge /
 * root / supervision target naming a node the batch has no information on — the reference's AssertionError /
 * TypeError / KeyError cases. */
typedef struct gigl_collated gigl_collated;
int32_t gigl_collate_records(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t b,
                             int32_t kind, int32_t n_threads, gigl_collated** out, char* err, int32_t err_cap);
int32_t gigl_collated_info(const gigl_collated* c, int64_t* n_nodes, int64_t* n_edges, int32_t* feat_dim,
                           int64_t* n_pos, int64_t* n_hard_neg);
/* any destination may be NULL; edge_index is [2][n_edges] int64 */
int32_t gigl_collated_copy(const gigl_collated* c, uint32_t* node_ids, float* x, int64_t* edge_index,
                           int64_t* root_local, int64_t* labels, uint8_t* has_label, int64_t* pos_off,
                           int64_t* pos_dst, int64_t* neg_off, int64_t* neg_dst);
/* edge features of the collated batch (GraphBuilder.add_edge(feature_values), abstract_graph_builder.py:102-150;
 * PygGraphBuilder.build stacks them into edge_attr, pyg_graph_builder.py:41-56): *edge_dim = floats per edge (0: the
 * samples carry none), edge_attr [n_edges][edge_dim] in the order of edge_index (an edge keeps the features of the
 * FIRST sample edge that registered it).  edge_attr may be NULL to query the dimension.  A batch whose edges do not
 * all carry the same number of feature values fails in gigl_collate_records (the reference raises TypeError). */
int32_t gigl_collated_edge_attr(const gigl_collated* c, int32_t* edge_dim, float* edge_attr);
int32_t gigl_collated_destroy(gigl_collated* c);

/* Typed (heterogeneous) samples: RootedNodeNeighborhood / NodeAnchorBasedLinkPredictionSample payloads whose nodes
 * and edges carry condensed_node_type / condensed_edge_type (graph_schema.proto:5-25) -> one batch graph per type, as
 * the reference's collate builds it through GraphBuilder (abstract_graph_builder.py:16-24: one first-seen counter per
 * node type; :100-150: one ordered, de-duplicated edge list per edge type) and PygGraphBuilder.build
 * (pyg_graph_builder.py:20-69: x_dict / edge_index_dict of a HeteroData), then coalesce() per edge type.
 * et_src_nt[t] / et_dst_nt[t]: condensed node types of the endpoints of condensed edge type t
 * (GraphMetadataPbWrapper.condensed_edge_type_to_edge_type_map).  Host pointers, host C++ (like
 * gigl_collate_records).  Errors as there: malformed record, a node outside the metadata's types, a node re-added
 * with other features, an edge whose endpoint is unknown, mixed edge-feature registration, a root / supervision
 * target outside the batch graph. */
typedef struct gigl_collated_typed gigl_collated_typed;
int32_t gigl_collate_typed_records(const uint8_t* buf, const int64_t* payload_off, const int64_t* payload_len, int64_t b,
                                   int32_t kind, int32_t n_node_types, int32_t n_edge_types, const int32_t* et_src_nt,
                                   const int32_t* et_dst_nt, int32_t n_threads, gigl_collated_typed** out, char* err,
                                   int32_t err_cap);
/* per node type: nodes, feature dim; per edge type: edges, edge-feature dim (arrays sized by the caller) */
int32_t gigl_collated_typed_info(const gigl_collated_typed* c, int64_t* nodes_per_type, int32_t* feat_dim_per_type,
                                 int64_t* edges_per_type, int32_t* edge_dim_per_type, int64_t* n_pos, int64_t* n_hard_neg);
int32_t gigl_collated_typed_nodes(const gigl_collated_typed* c, int32_t node_type, uint32_t* node_ids, float* x);
/* edge_index = [2, E] (row 0 sources, row 1 destinations), local to the edge type's endpoint node types */
int32_t gigl_collated_typed_edges(const gigl_collated_typed* c, int32_t edge_type, int64_t* edge_index, float* edge_attr);
/* per sample: root (type, local id), root label; supervision targets local to their edge type's destination type */
int32_t gigl_collated_typed_samples(const gigl_collated_typed* c, int32_t* root_type, int64_t* root_local, int64_t* labels,
                                    uint8_t* has_label, int64_t* pos_off, int64_t* pos_dst, int32_t* pos_type,
                                    int64_t* neg_off, int64_t* neg_dst, int32_t* neg_type);
int32_t gigl_collated_typed_destroy(gigl_collated_typed* c);

/* ---- node features: replaces loadNodeDataframeIntoSparkSql (SGSPureSparkV1Task.scala:52-118):
 * dense row-major [n][d], row index == node id. */
int32_t gigl_features_load(gigl_ctx* ctx, int64_t n, int32_t d, int32_t dtype, const void* rows,
                           int32_t loc, gigl_feat** out);
int32_t gigl_features_device_ptr(gigl_feat* f, const void** rows, int64_t* n, int32_t* d,
                                 int32_t* dtype);
int32_t gigl_features_destroy(gigl_feat* f);
/* table[id] = raw CRC-32C state (initial state 0, no final inversion) of the 4*d bytes node id's packed
 * Node.feature_values carry in a record (fp16 tables: of the fp32 values the proto carries).  Built on the first call
 * (one pass over the table on ctx's stream, then synchronised; not inside a stream capture) and owned by `f`.
 * gigl_records_encode uses it to checksum TFRecord payloads (TFRecordIO.scala:53-69) without reading them back: a
 * node's payload bytes are the same in every record that holds the node, and CRC-32C is linear. */
int32_t gigl_features_row_crc(gigl_ctx* ctx, gigl_feat* f, const uint32_t** table);

/* ---- k-hop rooted sampling: replaces sampleOnehopSrcNodesUniformly / sampleTwohopSrcNodesUniformly
 *      (SGSPureSparkV1Task.scala:313-388, :390-494), SamplingStrategy.hashBasedUniformPermutation
 *      (scala/subgraph_sampler/src/main/scala/libs/task/SamplingStrategy.scala:16-82) and, for
 *      per-hop fanouts, KHopSamplerService.getKHopSubgraphForRootNodes / GraphDBSampler
 *      (scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/GraphDBSampler.scala:40-148).
 *
 * Tree layout.  slots[0] = b*fanouts[0]; slots[k] = slots[k-1]*fanouts[k].
 *   nbr[k][p*fanouts[k] + j]  = j-th sampled in-neighbour of the node in parent slot p
 *                               (parent of hop 0 = root p), GIGL_INVALID past cnt.
 *   cnt[k][p]                 = number of valid entries under parent slot p (0 if the parent slot
 *                               is itself invalid or has no in-edges).
 * Within a parent the sampled ids are written in ascending id order: the reference's output is a
 * set (collect_list order after its joins is undefined), ascending is this library's canonical form.
 * Parity mode selects, for a parent with sorted in-neighbours A[1..n] and n > f, the f indices i
 * with the smallest (xxhash64_int32(i + K + seed*(k+1), 42) as signed int64, i), where K is the
 * int32 wrapping sum of the ids on the path root..parent — K={_dst_node} at hop 1, {_0_hop,_1_hop}
 * at hop 2; `seed*(k+1)` is samplingSeed*_counter with _counter = 1,2 (SamplingStrategy.scala:14,36).
 * All pointers DEVICE, caller-allocated. */
typedef struct gigl_tree {
  int32_t hops;
  int32_t b;
  int32_t fanouts[GIGL_MAX_HOPS];
  uint32_t* nbr[GIGL_MAX_HOPS]; /* [slots[k]] */
  int32_t* cnt[GIGL_MAX_HOPS];  /* [k==0 ? b : slots[k-1]] */
} gigl_tree;

int32_t gigl_sample_khop(gigl_ctx* ctx, gigl_graph* g, const uint32_t* roots, int32_t b,
                         const int32_t* fanouts, int32_t hops, int32_t sampling_seed,
                         int32_t mode, gigl_tree* out);

/* k-hop sampling (parity mode) over PEER-MAPPED graph shards of a hash-partitioned graph: node v's in-edge row is row
 * v / world of rank (v % world)'s CSC shard — peer_rowptr[r] / peer_col[r]: DEVICE arrays of `world` pointers, each valid in
 * this process (a rank's own: gigl_graph_device_ptrs; the others': mapped with gigl_ipc_export / gigl_ipc_open).  The
 * requester expands its own frontier and reads the owners' adjacency where it lives: what the sharded plan's peer-sampled
 * route runs instead of the per-hop request / answer exchange (distributed_neighborloader.py:162-192).  Same selection rule
 * (SamplingStrategy.scala:16-82), threshold table and tree layout as gigl_sample_khop: the trees are bit-identical to a
 * single process sampling the whole graph.  Fan-outs <= 64, rows without repeated ids, every hash window of the job below
 * max_window_end (inside the threshold table): GIGL_E_UNSUPPORTED otherwise. */
int32_t gigl_sample_khop_peer(gigl_ctx* ctx, const int64_t* const* peer_rowptr, const uint32_t* const* peer_col,
                              int32_t world, int64_t n_global, int64_t max_window_end, const uint32_t* roots, int32_t b,
                              const int32_t* fanouts, int32_t hops, int32_t sampling_seed, gigl_tree* out);
/* one hop over an EXPLICIT frontier on a hash-partitioned graph (owner(v) = v % world,
 * python/gigl/distributed/dist_link_prediction_data_partitioner.py:692-695): `shard` holds only the CSC rows of
 * the nodes this rank owns (row v / world, ids inside rows stay global); nodes[i] (all owned by this rank, or
 * GIGL_INVALID) and ksums[i] (K of the path root..nodes[i], computed by the requesting rank) arrive through the
 * frontier all-to-all.  Same selection rule as gigl_sample_khop with hash_add = seed * counter; output
 * out_nbr[i*f + j], out_cnt[i].  max_window_end >= 0 bounds ksum + hash_add + degree over all requests (lets
 * every window use the hash threshold table); -1 = unknown. */
int32_t gigl_expand_frontier(gigl_ctx* ctx, gigl_graph* shard, const uint32_t* nodes, const uint32_t* ksums,
                             int64_t m, int32_t f, int32_t hash_add, int32_t world, int64_t max_window_end,
                             uint32_t* out_nbr, int32_t* out_cnt);

/* the requester side of a hop on a hash-partitioned graph (see gigl_expand_frontier for the owner side): bucket the
 * frontier by owner for ONE equal-split all_to_all, and scatter the owners' answers back into the tree layout.
 * Replaces the per-batch RPC fan-out of the reference's distributed loader
 * (python/gigl/distributed/distributed_neighborloader.py:26-192; owner(v) = v % world,
 * dist_link_prediction_data_partitioner.py:692-695).  All pointers DEVICE; nothing synchronises with the host.
 *   bucket:  slot i (nodes[i], ksums[i] or nodes[i] itself when ksums == NULL) goes to bucket r = nodes[i] % world at
 *            the next free position p (GIGL_INVALID slots are not sent):
 *              req[(2r) * cap + p] = node, req[(2r+1) * cap + p] = K  (req is [world][2][cap], unused entries
 *              GIGL_INVALID), slot_idx[r * cap + p] = i, counts[r] = requests for peer r,
 *              counts[world] != 0 iff some bucket overflowed `cap` (the batch must be redone with a larger cap)
 *   scatter: resp is [world][cap][f] (the owners' out_nbr for the requests in bucket order): answer p of bucket r
 *            becomes out_nbr[i*f .. i*f+f), out_cnt[i] = its valid ids, child_ksums[i*f + j] = parent_ksums[i] + id
 *            (uint32 wrap; optional) for i = slot_idx[r*cap+p]; slots that were not sent get GIGL_INVALID / 0. */
int32_t gigl_frontier_bucket(gigl_ctx* ctx, const uint32_t* nodes, const uint32_t* ksums, int64_t m, int32_t world,
                             int64_t cap, uint32_t* req, int32_t* slot_idx, int32_t* counts);
int32_t gigl_frontier_scatter(gigl_ctx* ctx, const uint32_t* resp, const int32_t* slot_idx, const int32_t* counts,
                              const uint32_t* parent_ksums, int64_t m, int32_t world, int64_t cap, int32_t f,
                              uint32_t* out_nbr, int32_t* out_cnt, uint32_t* child_ksums);

/* SamplingOp DAG frontiers (SamplingOpDAG.from, scala_spark35/common/src/main/scala/types/SamplingOpDAG.scala:19-53;
 * GraphDBSampler.getKHopSubgraphForRootNode, scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/
 * GraphDBSampler.scala:40-148): an op's input frontier for a root is the SET UNION of the node sets its parent ops
 * returned for that root.  ids: DEVICE [rows][width] (row r = the parents' results for root r back to back,
 * GIGL_INVALID = empty): later occurrences of an id inside a row are overwritten with GIGL_INVALID in place.
 * width <= 8192.  The op itself is then one gigl_expand_frontier call over rows*width slots on the op's edge-type
 * graph (world = 1). */
int32_t gigl_rows_dedup(gigl_ctx* ctx, uint32_t* ids, int64_t rows, int32_t width);

/* positives for node-anchor link prediction: `f` OUT-neighbours of each root, counter = 3
 * (sampleDstNodesUniformly, NodeAnchorBasedLinkPredictionBaseTask.scala:19-104).  `g_out` is the
 * CSR-by-source graph loaded through gigl_graph_load_csc with the roles of src/dst swapped. */
int32_t gigl_sample_positives(gigl_ctx* ctx, gigl_graph* g_out, const uint32_t* roots, int32_t b,
                              int32_t f, int32_t sampling_seed, int32_t mode, uint32_t* pos,
                              int32_t* cnt);
/* the same with an explicit permutation call counter (hashBasedUniformPermutation's process-global `_counter`,
 * SamplingStrategy.scala:14,79): the user-defined-labels task samples positives with counter 3 and hard negatives
 * with counter 4 from the user-defined edge lists (UserDefinedLabelsNodeAnchorBasedLinkPredictionTask.scala:171-199). */
int32_t gigl_sample_out_neighbors(gigl_ctx* ctx, gigl_graph* g_out, const uint32_t* roots, int32_t b,
                                  int32_t f, int32_t sampling_seed, int32_t counter, int32_t mode, uint32_t* pos,
                                  int32_t* cnt);

/* ---- sampled trees -> serialized training samples in TFRecord framing, encoded on the device.
 *      Replaces the per-root assembly, hydration, proto cast and record writer of the Spark sampler:
 *      hydrateNodes / hydrateEdges / createSubgraph (SGSPureSparkV1Task.scala:496-593, :671-820),
 *      createNodeAnchorBasedLinkPredictionSubgraph (pureSpark/NodeAnchorBasedLinkPredictionTask.scala:146-312),
 *      castToRootedNodeNeighborhoodProtoSchema (SGSPureSparkV1Task.scala:1019-1040) and
 *      TFRecordIO.writeDatasetToTfrecord (scala/common/src/main/scala/utils/TFRecordIO.scala:53-69).
 *
 * Record r is built from trees_per_record consecutive trees of `tree` (tree r*T is the root's; for
 * GIGL_REC_NODE_ANCHOR_LINK_PRED trees r*T+1 .. are the rooted samples of the root's positives, root id
 * GIGL_INVALID = no such positive, lookupDstNodeNeighborhood NodeAnchorBasedLinkPredictionBaseTask.scala:106-198):
 *   nodes  = distinct ids in stream order: per tree the hop-1 slots, hop-2 slots, ..., then the tree's root
 *            (array_distinct(hop nodes ++ [root]), SGSPureSparkV1Task.scala:737-780; across trees
 *            array_distinct(root nbhd ++ positives' nbhds)), each written once with its feature row
 *   edges  = per tree the hop-1 edges (src = slot, dst = root), then hop-2 edges (dst = parent slot), ...;
 *            distinct (src,dst) pairs across the trees of a record
 *   GIGL_REC_ROOTED_NODE_NEIGHBORHOOD: RootedNodeNeighborhood{root_node=1, neighborhood=2}
 *            (training_samples_schema.proto:16-19) followed by the opaque `suffix` bytes of the record —
 *            already-encoded `root_node_labels` (field 3) turn it into a SupervisedNodeClassificationSample (:23-27)
 *   GIGL_REC_NODE_ANCHOR_LINK_PRED: NodeAnchorBasedLinkPredictionSample{root_node=1, neighborhood=3,
 *            pos_edges=4: root -> root of tree j} (:31-43)
 * Encoding is byte-identical to ScalaPB / protobuf (field-number order, proto3 zero elision for node/edge ids,
 * explicit presence for the `optional` condensed types, packed floats).  tfrecord_frame != 0 wraps every
 * record as u64 length | masked crc32c(length) | payload | masked crc32c(payload).
 * All pointers DEVICE.  rec_off[r] = byte offset of record r in `out`, rec_off[n_records] = total bytes;
 * *status = 1 (and nothing is written) when the total exceeds out_cap.  Never synchronises with the host.
 * One pass: a persistent grid takes records from a ticket counter; a record's plan (distinct nodes / edges and their
 * byte offsets) lives in LDS while trees_per_record * (1 + sum of slots per tree) stays below ~5,000 positions and
 * in per-workgroup scratch beyond (up to 2^20 positions; a record must stay below 16 MiB); record offsets come from
 * a decoupled look-back over the sizes of the records before it.  status = 1: `out` was too small (its contents are
 * then undefined; nothing is written beyond out_cap). */
typedef struct gigl_record_opts {
  int32_t kind;                /* GIGL_REC_* */
  int32_t trees_per_record;    /* 1, or 1 + num_positive_samples for GIGL_REC_NODE_ANCHOR_LINK_PRED */
  int32_t condensed_node_type; /* >= 0: written in every Node; < 0: field absent */
  int32_t condensed_edge_type; /* >= 0: written in every Edge; < 0: field absent */
  int32_t tfrecord_frame;
  const uint8_t* emit;         /* [n_records] or NULL; 0 = the record is skipped (takes 0 bytes) */
  const uint8_t* suffix;       /* bytes appended to the payload of record r: suffix[suffix_off[r] .. suffix_off[r+1]) */
  const int64_t* suffix_off;   /* [n_records+1] or NULL (with suffix) */
  /* edge features (hydrateEdges, SGSPureSparkV1Task.scala:549-593; Edge.feature_values, graph_schema.proto:29-30):
   * `edge_feat` = fp32 table with one row per edge of `graph`, in the order of its `col` array (gigl_features_load
   * with n = #edges).  Every emitted Edge — the neighbourhood's and the pos_edges — then carries the row of its
   * (src, dst) pair, found by binary search in the destination's row.  NULL / d == 0: no feature_values. */
  gigl_graph* graph;
  gigl_feat* edge_feat;
  /* user-defined label edges (UserDefinedLabelsNodeAnchorBasedLinkPredictionTask.scala:171-232, :385-486): the LAST
   * n_neg_trees trees of a record are the sampled hard negatives — their neighbourhoods are merged like the
   * positives' and `hard_neg_edges` (= 2: root -> root of the tree) is written between root_node and neighborhood.
   * pos_edges_graph / neg_edges_graph: the user-defined edge list as CSR by SOURCE (gigl_graph_build_from_coo with
   * src/dst swapped, is_directed = 1); pos_edge_feat / neg_edge_feat: its feature rows in that graph's `col` order
   * (NULL: the label edges carry no feature_values).  pos_edges_graph == NULL: positives are main edges and take
   * their features from graph / edge_feat. */
  int32_t n_neg_trees;
  gigl_graph* pos_edges_graph;
  gigl_feat* pos_edge_feat;
  gigl_graph* neg_edges_graph;
  gigl_feat* neg_edge_feat;
} gigl_record_opts;

/* upper bound of the bytes n_records records can take (d = feature dim, suffix_total = all suffix bytes) */
int32_t gigl_records_capacity(const int32_t* fanouts, int32_t hops, int32_t d, const gigl_record_opts* opts,
                              int64_t n_records, int64_t suffix_total, int64_t* bytes);
int32_t gigl_records_encode(gigl_ctx* ctx, const uint32_t* tree_roots, const gigl_tree* tree, gigl_feat* feat,
                            const gigl_record_opts* opts, int64_t n_records, uint8_t* out, int64_t out_cap,
                            int64_t* rec_off, int32_t* status);

/* ---- typed (heterogeneous) RootedNodeNeighborhood records of a SamplingOp DAG, encoded on the device.
 *      Replaces GraphDBSampler's per-root assembly (scala_spark35/common/src/main/scala/graphdb/GraphDBSampler.scala:
 *      45-113: the union, as sets, of every op's edges and nodes + the root) and the proto cast / TFRecord write of the
 *      task around it (GraphDBNodeAnchorBasedLinkPredictionTask.scala:62-78).  One gigl_typed_op per sampling op, as
 *      gigl_rows_dedup + gigl_expand_frontier produced it: `frontier` [b, w] node ids (GIGL_INVALID = none), `nbr`
 *      [b, w, f] sampled neighbours, the op's condensed edge type, the condensed node type of the ids in `nbr`, and the
 *      direction (outgoing: edges are frontier -> nbr; else nbr -> frontier).  feats[t]: fp32 feature table of
 *      condensed node type t (x NULL: nodes of that type carry no feature_values).  Records: root_node = 1 (id, type,
 *      features), neighborhood = 2 with the distinct nodes ascending by (id, type) and the distinct edges ascending by
 *      (src, dst, type) — byte-identical to the host assembly in gigl_amd/graphdb_sampler.py.  efeats[t] (optional):
 *      Edge.feature_values of condensed edge type t — the type's edge list as a CSR-by-source graph
 *      (gigl_graph_build_from_coo with the roles swapped) and one fp32 row per edge in that graph's `col` order; the
 *      row of edge (s -> d) is found by binary search in row s (hydrateEdges' join).  Up to 16 ops, 16 node types, 16
 *      edge types, 2^20 - 1 sampled slots (sum of w*f) per root (GIGL_E_UNSUPPORTED beyond; up to 4095 the per-root sort
 *      runs in LDS, above it in global scratch, <= 1.5 GB of the context's arena).  out / rec_off / status as
 *      gigl_records_encode. */
typedef struct gigl_typed_op {
  const uint32_t* frontier;
  const uint32_t* nbr;
  int32_t w, f;
  int32_t condensed_edge_type;
  int32_t result_node_type;
  int32_t outgoing; /* bit 0: outgoing; GIGL_TYPED_OP_POSITIVE: the op sampled the sample's positive edges */
} gigl_typed_op;
#define GIGL_TYPED_OP_POSITIVE 2
typedef struct gigl_typed_feat {
  const float* x; /* device, [n, d] */
  int32_t d;
  int64_t n;
} gigl_typed_feat;
typedef struct gigl_typed_edge_feat {
  gigl_graph* by_source; /* the edge type's edges, CSR by source */
  const float* feat;     /* device, [edges, d] in by_source's col order; NULL: no features for this type */
  int32_t d;
} gigl_typed_edge_feat;
int32_t gigl_typed_records_capacity(const gigl_typed_op* ops, int32_t n_ops, const gigl_typed_feat* feats,
                                    int32_t n_node_types, const gigl_typed_edge_feat* efeats, int32_t n_edge_types,
                                    int64_t n_records, int32_t tfrecord_frame, int64_t* bytes);
int32_t gigl_typed_records_encode(gigl_ctx* ctx, const uint32_t* roots, int32_t root_node_type, const gigl_typed_op* ops,
                                  int32_t n_ops, const gigl_typed_feat* feats, int32_t n_node_types,
                                  const gigl_typed_edge_feat* efeats, int32_t n_edge_types, int64_t n_records,
                                  int32_t tfrecord_frame, uint8_t* out, int64_t out_cap, int64_t* rec_off,
                                  int32_t* status);
/* The same encoder for the typed TRAINING samples of GraphDBNodeAnchorBasedLinkPredictionTask.scala:118-496
 * (kind GIGL_REC_NODE_ANCHOR_LINK_PRED): NodeAnchorBasedLinkPredictionSample{root_node = 1, neighborhood = 3,
 * pos_edges = 4}.  The ops of the root's own DAG, one op flagged GIGL_TYPED_OP_POSITIVE (the root's sampled positive
 * edges, samplePositiveEdgeNeighborhoods, GraphDBSampler.scala:175-218: its edges become pos_edges — distinct, ascending
 * by (src, dst, type), features joined like any edge — and its result nodes join the neighbourhood's nodes), and the
 * ops of the positives' DAG with `frontier` / `nbr` holding the results of the root's P positives side by side
 * ([b][P*w] / [b][P*w][f]): the neighbourhood is the union over all of them = mergeGraphs of the root's and the
 * positives' neighbourhoods.  kind GIGL_REC_ROOTED_NODE_NEIGHBORHOOD = gigl_typed_records_encode. */
int32_t gigl_typed_samples_encode(gigl_ctx* ctx, int32_t kind, const uint32_t* roots, int32_t root_node_type,
                                  const gigl_typed_op* ops, int32_t n_ops, const gigl_typed_feat* feats,
                                  int32_t n_node_types, const gigl_typed_edge_feat* efeats, int32_t n_edge_types,
                                  int64_t n_records, int32_t tfrecord_frame, uint8_t* out, int64_t out_cap,
                                  int64_t* rec_off, int32_t* status);

/* ---- the typed (heterogeneous) batch graph of a SamplingOp DAG in ONE host call (csrc/typed_plan.hip).
 *      Replaces, per batch of roots of one node type: GraphDBSampler.getKHopSubgraphForRootNode for every root
 *      (scala_spark35/subgraph_sampler/src/main/scala/libs/sampler/GraphDBSampler.scala:40-148: an op's frontier is the
 *      set union of its parents' results; it runs for a root only when all its parents ran and the frontier is not
 *      empty), SamplingOpDAG (scala_spark35/common/src/main/scala/types/SamplingOpDAG.scala:19-53) and the trainer-side
 *      collate of the typed samples into one batch graph (python/gigl/src/common/graph_builder/
 *      abstract_graph_builder.py:49-150, pyg_graph_builder.py:20-69): per node type the distinct nodes, per edge type
 *      the distinct edges as local ids.  Ops are listed parents first; op.graph = the op's (edge type, direction) edge
 *      list with one row per FRONTIER node (what gigl_expand_frontier takes, world = 1); hash_add = seed * (1 + the
 *      op's position in the config's op order); edge_slot = which output edge list the op's edges join (ops of one
 *      edge type share a slot); outgoing: edges are frontier -> result, else result -> frontier.
 *      Results (DEVICE, owned by the plan, valid until the next run): nodes[t][0 .. n_nodes[t]) = the distinct ids of
 *      node type t, ASCENDING (position = local id); edges[s][0 .. n_edges[s]) = the distinct edges of slot s as
 *      (src_local << 32 | dst_local), ascending; root_index[i] = local id of root i; and every op's frontier
 *      [b][w] / neighbours [b][w][f] / counts [b][w] (the inputs of gigl_typed_records_encode).  Nothing in a run
 *      synchronises with the host; the caller reads n_nodes / n_edges when it needs the sizes. */
#define GIGL_DAG_MAX_OPS 16
#define GIGL_DAG_MAX_PARENTS 8
typedef struct gigl_dag_op {
  gigl_graph* graph;
  int32_t fanout;
  int32_t n_parents;
  int32_t parents[GIGL_DAG_MAX_PARENTS]; /* indices of earlier ops; n_parents == 0: the op starts at the roots */
  int32_t hash_add;
  int32_t frontier_node_type, result_node_type; /* condensed node types */
  int32_t edge_slot;
  int32_t outgoing;
} gigl_dag_op;
typedef struct gigl_typed_plan gigl_typed_plan;
typedef struct gigl_typed_plan_out {
  const int32_t* n_nodes;  /* [n_node_types] */
  const int32_t* n_edges;  /* [n_edge_slots] */
  const int32_t* root_index; /* [b] */
  const uint32_t* nodes[16];
  int64_t nodes_cap[16];
  const unsigned long long* edges[32];
  int64_t edges_cap[32];
  const uint32_t* op_frontier[GIGL_DAG_MAX_OPS];
  const uint32_t* op_nbr[GIGL_DAG_MAX_OPS];
  const int32_t* op_cnt[GIGL_DAG_MAX_OPS];
  int32_t op_width[GIGL_DAG_MAX_OPS];
} gigl_typed_plan_out;
int32_t gigl_typed_plan_create(gigl_ctx* ctx, const gigl_dag_op* ops, int32_t n_ops, int32_t n_node_types,
                               int32_t root_node_type, int32_t n_edge_slots, int32_t b_max, gigl_typed_plan** out);
int32_t gigl_typed_plan_run(gigl_typed_plan* plan, const uint32_t* roots, int32_t b);
/* gigl_typed_plan_run in two halves — the ops, the node numbering and the roots' positions; then the edge lists — so that a
 * caller can start work that needs only the nodes (feature rows, input projections) on another stream while the edges
 * are still being numbered (csrc/hgt_plan.hip) */
int32_t gigl_typed_plan_run_nodes(gigl_typed_plan* plan, const uint32_t* roots, int32_t b);
int32_t gigl_typed_plan_run_edges(gigl_typed_plan* plan, int32_t b);
/* a second workspace of the same DAG on another ctx (its own stream): what lets batch i + 1 be built while batch i is read */
int32_t gigl_typed_plan_clone(gigl_typed_plan* plan, gigl_ctx* ctx, gigl_typed_plan** out);
int32_t gigl_typed_plan_buffers(gigl_typed_plan* plan, gigl_typed_plan_out* out);
/* The batch graph's edges of ALL listed slots as ONE CSR by destination — the operand of the typed attention layers
 * (torch_geometric HGTConv's per-destination softmax over every incoming edge type; python/gigl/src/common/models/pyg/
 * heterogeneous.py:18-120 builds it inside HGTConv._construct_src_node_feat / propagate).  After gigl_typed_plan_run,
 * same stream, no host read: destinations are numbered type after type in `type_order` (type t's node i = offset of t +
 * i), sources slot after slot in `slot_order` (source index = sum of the earlier slots' SOURCE-type node counts +
 * src_local: the layer's per-edge-type K / V blocks laid side by side), a destination's edges in slot order, then
 * ascending (src, dst) — the order of a stable sort by destination.  etype[e] = slot_etype[position of e's slot]
 * (NULL: the position).  root_* = the rows of the batch's roots alone, in root order (an inference pass's last layer).
 * type_order / slot_order / slot_etype: HOST.  The buffers belong to the plan and are valid until its next run. */
typedef struct gigl_typed_csr_out {
  const int32_t* rowptr;      /* [n_dst + 1] */
  const int32_t* col;         /* [E] */
  const int32_t* etype;       /* [E] */
  const int32_t* counts;      /* [2]: n_dst, E (= the sums of the counts the caller already reads) */
  const int32_t* root_rowptr; /* [b + 1] */
  const int32_t* root_col;
  const int32_t* root_etype;
  int64_t edges_cap, rows_cap;
} gigl_typed_csr_out;
int32_t gigl_typed_plan_merged_csr(gigl_typed_plan* plan, int32_t b, const int32_t* type_order, int32_t n_types_used,
                                   const int32_t* slot_order, const int32_t* slot_etype, int32_t n_slots_used,
                                   gigl_typed_csr_out* out);
/* The same with capacity_layout != 0: destination blocks and source blocks start at CAPACITY prefixes (type t's block =
 * nodes_cap[t] rows, a slot's source block = its source type's nodes_cap) instead of at the batch's counts, rows beyond a
 * type's count are empty, counts[0] = the sum of the listed types' capacities.  Every address of a layer over this CSR
 * is then known when the plan is made: what gigl_hgt_infer_* runs over, captured once and replayed. */
int32_t gigl_typed_plan_merged_csr_ex(gigl_typed_plan* plan, int32_t b, const int32_t* type_order, int32_t n_types_used,
                                      const int32_t* slot_order, const int32_t* slot_etype, int32_t n_slots_used,
                                      int32_t capacity_layout, gigl_typed_csr_out* out);
int32_t gigl_typed_plan_destroy(gigl_typed_plan* plan);

/* ---- the typed inference step in ONE call (csrc/hgt_plan.hip): gigl_typed_plan_run -> merged CSR at capacity prefixes
 *      -> HGT encoder -> the roots' rows; no host read inside; the step is two captured parts (graph part on the plan's own
 *      stream, layers on the caller's), replayed from their second run on.  Replaces the HGT forward of python/gigl/src/common/models/pyg/heterogeneous.py:18-120
 *      (torch_geometric HGTConv layers: hgt_conv.py) driven per batch by the inferencer
 *      (python/gigl/src/inference/v1/gnn_inferencer.py:234-340).
 *      Weights are DEVICE pointers the caller keeps alive, in the COMPOSED inference form (linear stages multiplied
 *      together once per parameter state — gigl_amd/models_hetero.py::HGTConv._composed):
 *        per used node type j   feat[j] [n_j][feat_dim[j]] fp32 rows by global id (NULL: a constant 1-wide input),
 *                               w_in / b_in: lin_dict[t] ([hid][feat_dim], [hid]; ReLU follows)
 *        per layer, per type    wq / bq ([hid][hid], [hid]);  wout / bout = a * out_lin, keep = device float (1 - a) with
 *                               a = sigmoid(skip[t]) (NULL: no skip)
 *        per layer, per slot    wk / bk = blockdiag(k_rel[etype]) @ K-part of kqv_lin[src type], wv / bv likewise
 *        per layer              p_rel [model edge types][heads]
 *        w_final / b_final      the encoder's output Linear ([out_dim][hid]); l2_normalize: F.normalize(p=2, dim=1)
 *      type_order[j] = the plan's node type of used type j (every type an op touches, the roots' type among them:
 *      root_type); slot_order[s] = the plan's edge slot of listed slot s, slot_etype[s] = its row of p_rel.
 *      The LAST layer computes the roots' rows only (identical rows: hgt_conv is row-wise per destination). */
#define GIGL_HGT_MAX_LAYERS 4
typedef struct gigl_hgt_layer_weights {
  const float* wq[16];
  const float* bq[16];
  const float* wout[16];
  const float* bout[16];
  const float* keep[16];
  const float* wk[32];
  const float* bk[32];
  const float* wv[32];
  const float* bv[32];
  const float* p_rel;
} gigl_hgt_layer_weights;
typedef struct gigl_hgt_model {
  int32_t n_types, n_slots, n_layers, heads, hid, out_dim, l2_normalize, root_type;
  int32_t type_order[16];
  int32_t slot_order[32];
  int32_t slot_etype[32];
  int32_t feat_dim[16];
  const float* feat[16];
  const float* w_in[16];
  const float* b_in[16];
  gigl_hgt_layer_weights layer[GIGL_HGT_MAX_LAYERS];
  const float* w_final;
  const float* b_final;
} gigl_hgt_model;
typedef struct gigl_hgt_infer gigl_hgt_infer;
/* slot_src_type / slot_dst_type [n_slots]: the plan's node types each listed slot joins (HOST).  b_max <= the plan's. */
int32_t gigl_hgt_infer_create(gigl_ctx* ctx, gigl_typed_plan* plan, int32_t b_max, const gigl_hgt_model* model,
                              const int32_t* slot_src_type, const int32_t* slot_dst_type, gigl_hgt_infer** out);
/* roots: DEVICE uint32 [b]; out: DEVICE fp32 [b][out_dim], on the ctx's stream.  roots_next (NULL: none) = the roots the NEXT
 * call will pass (same pointer, b_next of them): their GRAPH part — ops, numbering, merged CSR — is enqueued on the plan's
 * own stream before this batch's layers, so the two overlap; a call whose roots were announced finds its graph built.
 * `plan` of _create is a template: the step keeps two workspaces of its DAG of its own. */
int32_t gigl_hgt_infer_run(gigl_hgt_infer* infer, const uint32_t* roots, int32_t b, const uint32_t* roots_next,
                           int32_t b_next, float* out);
/* new weight pointers of the same shapes (after a parameter update); a changed pointer re-captures the step */
int32_t gigl_hgt_infer_set_model(gigl_hgt_infer* infer, const gigl_hgt_model* model);
int32_t gigl_hgt_infer_use_graph(gigl_hgt_infer* infer, int32_t enable);
int32_t gigl_hgt_infer_destroy(gigl_hgt_infer* infer);

/* The plan's numbering step as a call of its own (csrc/sortscan.h: the library's LSD radix sort + ordered distinct —
 * kernels only, so a plan that uses them can be captured into a hipGraph; rocPRIM's sort issues memsets).  Replaces
 * torch.unique(sorted=True) in gigl_amd/graphdb_sampler.py::batch_graph, i.e. the first-seen / sorted numbering of
 * python/gigl/src/common/graph_builder/abstract_graph_builder.py:100-150.  out[0 .. *count) = the distinct keys of
 * keys[0 .. n) other than `pad`, ascending; keys, out, count: DEVICE.  Only the bit fields that can differ are sorted:
 * u64 keys by bits [0, low_bits) and [32, 32 + high_bits), u32 keys by bits [0, bits); every real key must be below the
 * all-ones value of its fields and `pad` all ones in them (it is dropped wherever it lands last). */
int32_t gigl_sort_distinct_u64(gigl_ctx* ctx, const unsigned long long* keys, int64_t n, int32_t low_bits, int32_t high_bits,
                               unsigned long long pad, unsigned long long* out, int32_t* count);
int32_t gigl_sort_distinct_u32(gigl_ctx* ctx, const uint32_t* keys, int64_t n, int32_t bits, uint32_t pad, uint32_t* out,
                               int32_t* count);

/* ---- inference output: (node id, embedding row) batches -> Avro object-container DATA BLOCKS, encoded on the device.
 *      Replaces the record loop of EmbeddingExporter.add_embedding (python/gigl/common/data/export.py:103-135:
 *      {"node_id": int(id), "node_type": type, "emb": row.tolist()} through fastavro.writer) for AVRO_SCHEMA
 *      (export.py:34-43: record Embedding {node_id: long, node_type: string, emb: array<float>}).
 * Encoding per the Apache Avro 1.x specification (fastavro, the reference's writer, is a third-party package):
 * long = zig-zag varint, string = long length + UTF-8, float = 4 bytes LE, array = long count + items + long 0
 * (only the 0 when empty); a data block = long record count | long byte size | records | 16-byte sync marker.
 * The file header (magic, metadata map with the schema, sync marker) is host work (gigl_amd/export.py).
 * gigl_avro_embeddings_layout: records per data block (~16 kB blocks, the writer's sync interval), the number
 * of blocks and an upper bound of the bytes `n` records of dimension `dim` take.
 * gigl_avro_embeddings_encode: ids [n] int64, emb [n, dim] fp32 with `emb_stride` floats between rows, out: all
 * DEVICE; type_utf8 / sync_marker (16 bytes): HOST.  rec_off[i] = byte offset of record i in `out` (device, [n]);
 * *total_bytes (device) = bytes written; *status (device) = 1 and nothing written when they exceed out_cap.
 * Never synchronises with the host (the type / marker bytes travel as a kernel argument). */
int32_t gigl_avro_embeddings_layout(int64_t n, int32_t dim, int32_t type_len, int32_t* records_per_block,
                                    int64_t* n_blocks, int64_t* bytes);
int32_t gigl_avro_embeddings_encode(gigl_ctx* ctx, const int64_t* ids, const float* emb, int64_t emb_stride,
                                    int64_t n, int32_t dim, const uint8_t* type_utf8, int32_t type_len,
                                    const uint8_t* sync_marker, uint8_t* out, int64_t out_cap, int64_t* rec_off,
                                    int64_t* total_bytes, int32_t* status);

/* line-per-root JSON rows of an inference batch — the local form of the rows the reference's inferencer emits per root
 * ({"node_id", "emb"} / {"node_id", "pred"}: python/gigl/src/inference/v1/lib/base_inference_blueprint.py:76-103,
 * loaded into BigQuery there).  HOST pointers: ids [n], emb [n, dim] fp32 with `emb_stride` floats between rows (or
 * NULL), pred [n] (or NULL; with emb both keys go into one object); out holds >= gigl_json_rows_capacity(n, dim) bytes;
 * *bytes = bytes written.  Floats print as the shortest decimal that parses back to the same fp32.  Formats row
 * ranges on worker threads. */
int64_t gigl_json_rows_capacity(int64_t n, int32_t dim);
int32_t gigl_json_rows_format(const int64_t* ids, const float* emb, int64_t emb_stride, const int32_t* pred, int64_t n,
                              int32_t dim, char* out, int64_t out_cap, int64_t* bytes);

/* ---- batch union graph ("collate"): replaces GraphBuilder.add_graph_data/add_edge dedup
 *      (python/gigl/src/common/graph_builder/abstract_graph_builder.py:49-150), the collate
 *      functions (python/gigl/src/training/v1/lib/data_loaders/
 *      rooted_node_neighborhood_data_loader.py:78-158) and coalesce().
 * Merges the b rooted trees into ONE deduplicated graph: unique nodes, unique (src,dst) edges.
 * Local numbering is level-ordered: level 0 = roots, level l = nodes first reached as a source of
 * an in-edge of a level l-1 node in the UNION graph; within a level: first occurrence in the
 * canonical stream (roots, then hop-1 slots, then hop-2 slots ...).  So the rows a layer must
 * compute are always a prefix (layer-wise trimmed schedule, exact for root outputs).
 * Outputs (DEVICE, caller-allocated, capacities from gigl_union_capacity):
 *   meta[0]=n_nodes, meta[1]=n_edges (unique), meta[2+l]=cumulative node count through level l
 *           (meta[2]=#distinct roots), l=0..hops; meta[GIGL_META_OVERFLOW] != 0 reports a workspace
 *           that did not hold the batch (a full node table of the leaf-global builds): treat the
 *           batch as failed.  Rows have no length bound: more than 16,384 distinct in-edges of one
 *           node in one batch (past the LDS sort) are sorted and made distinct in global memory
 *   nodes[n_nodes]    global id of local node i
 *   rowptr[i], rowend[i]  row i = col[rowptr[i] .. rowend[i]) — CSR by destination over local ids
 *                     whose rows keep their pre-dedup capacity (rowptr is monotone, rowend[i] <=
 *                     rowptr[i+1]); only nodes of level < hops have in-edges
 *   col[...]          local source ids, ascending and duplicate-free within a row
 *   root_local[b]     local id of roots[i] (duplicates in `roots` map to the same local id) */
#define GIGL_META_N_NODES 0
#define GIGL_META_N_EDGES 1
#define GIGL_META_LEVEL0 2 /* meta[2+l], l = 0..hops */
#define GIGL_META_OVERFLOW 8
#define GIGL_META_LEN 16

typedef struct gigl_union {
  int32_t* meta;       /* [GIGL_META_LEN] */
  uint32_t* nodes;     /* [cap_nodes] */
  int32_t* rowptr;     /* [cap_nodes+1] */
  int32_t* rowend;     /* [cap_nodes+1] */
  int32_t* col;        /* [cap_edges] */
  int32_t* root_local; /* [b] */
  int64_t cap_nodes;
  int64_t cap_edges;
} gigl_union;

int32_t gigl_union_capacity(int32_t b, const int32_t* fanouts, int32_t hops, int64_t* cap_nodes,
                            int64_t* cap_edges);
int32_t gigl_union_build(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree,
                         gigl_union* out);
/* Several INDEPENDENT batches in one set of launches: the b = n_groups*group_roots trees are treated as
 * n_groups consecutive batches of group_roots roots; nodes and edges are deduplicated WITHIN a batch only
 * (the same global id in two batches is two local nodes), i.e. the result is the disjoint union of the
 * n_groups per-batch union graphs — what calling the reference collate once per batch produces — in ONE
 * level-ordered local numbering (level, first stream position), so layer prefixes still hold and the
 * relative order of a batch's nodes equals its stand-alone numbering.  Every batch has its own hash
 * sub-tables.  group_roots must divide b; group_roots == b is gigl_union_build. */
int32_t gigl_union_build_groups(gigl_ctx* ctx, const uint32_t* roots, const gigl_tree* tree,
                                int32_t group_roots, gigl_union* out);

/* ---- message passing over the union graph: replaces PyG SAGEConv / GATConv / GCNConv as used by
 *      python/gigl/src/common/models/pyg/homogeneous.py:107-153,171-202,300-343,488-546.
 *
 * gigl_gather_mean: the segmented gather + mean reduce, fused with feature hydration.
 *   rows i in [0, n_rows), row i = col[rowptr[i] .. rowend[i]) (rowend == rowptr + 1 for a packed CSR):
 *                           out[i][0:d]  = mean_{e in row i} src[ idx(col[e]) ][0:d]   (0 if empty)
 *                           out[i][d:2d] = src[ idx(i) ][0:d]
 *   idx(j) = gather_ids ? gather_ids[j] : j   (gather_ids = union.nodes reads the global feature
 *   table directly: hydrateNodes, SGSPureSparkV1Task.scala:496-547, without materialising x).
 *   n_rows is read on the device from *n_rows_dev (a union.meta entry); rows_cap bounds the grid.
 *   out is f32 [rows_cap][2d] — the A operand of the SAGE projection [mean | self]·[W_l ; W_r]^T. */
int32_t gigl_gather_mean(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                         const uint32_t* gather_ids, const int32_t* rowptr, const int32_t* rowend,
                         const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, float* out);

/* the same with the reduction chosen: PyG SAGEConv(aggr = "mean" | "sum" | "max") via conv_kwargs
 * (python/gigl/src/common/models/pyg/homogeneous.py:171-202); an empty row reduces to 0 */
#define GIGL_AGGR_MEAN 0
#define GIGL_AGGR_SUM 1
#define GIGL_AGGR_MAX 2
int32_t gigl_gather_reduce(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                           const int32_t* n_rows_dev, int64_t rows_cap, int32_t aggr, float* out);
/* its backward w.r.t. a dense local fp32 source (dsrc zero-filled by the caller): mean / sum scatter dout[i][0:d]
 * (over deg_i for mean); max shares dout[i][k] evenly among the sources attaining the maximum (torch amax) and
 * needs the forward's `src` */
int32_t gigl_gather_reduce_backward(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                    const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                    int64_t rows_cap, int32_t aggr, const float* src, float* dsrc);

/* feature hydration alone (hydrateNodes, SGSPureSparkV1Task.scala:496-547): out[i][0:d] = src[ids[i]][0:d] as fp32
 * for i < *n_dev.  Needed when a layer projects before it aggregates (GAT). */
int32_t gigl_gather_rows(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* ids,
                         const int32_t* n_dev, int64_t cap, float* out);

/* GCNConv aggregation (PyG 2.5.3 GCNConv as used by TwoLayerGCN, python/gigl/src/common/models/pyg/homogeneous.py:
 * 488-546): symmetric normalisation with one self loop per node,
 *   out[i] = act( sum_{j in N(i) u {i}} h[idx(j)] / sqrt(deg_i deg_j) + bias ),  deg = 1 + #non-self in-edges,
 * for rows i < *n_rows_dev; degrees are taken over the first *n_nodes_dev nodes of the union graph.
 * h: [*, d] fp32/fp16 rows (gather_ids = union.nodes reads the resident feature table; NULL = local matrix).
 * dinv_scratch: DEVICE fp32 [nodes_cap].  (A X) W == A (X W): apply gigl_linear before or after. */
int32_t gigl_gcn_aggregate(gigl_ctx* ctx, const void* h, int32_t h_dtype, int32_t d, const uint32_t* gather_ids,
                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                           const int32_t* n_nodes_dev, int64_t nodes_cap, const int32_t* n_rows_dev,
                           int64_t rows_cap, const float* bias, int32_t act, float* dinv_scratch, float* out);

/* GATConv attention + aggregation (PyG 2.5.3 GATConv as configured by GAT.init_conv_layers, homogeneous.py:300-343):
 *   h: fp32 [nodes][heads*channels] = x W (gigl_linear), att_src/att_dst: [heads*channels];
 *   e_ij = leaky_relu(<h_j, att_src> + <h_i, att_dst>, negative_slope) over the in-edges of i (self loops removed)
 *   plus one self loop; alpha = softmax_j e_ij; out[i] = sum_j alpha_ij h_j, heads concatenated (concat=1,
 *   [rows][heads*channels]) or averaged (concat=0, [rows][channels]); + bias; optional relu.
 * alpha_scratch: DEVICE fp32 [2*nodes_cap*heads]. */
int32_t gigl_gat_aggregate(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst, int32_t heads,
                           int32_t channels, float negative_slope, int32_t concat, const int32_t* rowptr,
                           const int32_t* rowend, const int32_t* col, const int32_t* n_nodes_dev, int64_t nodes_cap,
                           const int32_t* n_rows_dev, int64_t rows_cap, const float* bias, int32_t act,
                           float* alpha_scratch, float* out);

/* The FIRST GATConv layer of an inference pass, computed from the input side (same layer as gigl_gather_rows +
 * gigl_linear + gigl_gat_aggregate with concat=1 and no edge features; GAT.forward's first conv, homogeneous.py:300-343):
 *   <W_h x_j, att_h> = <x_j, W_h^T att_h>  — the logits come from two folded d-vectors per head, and
 *   sum_j alpha_ij W_h x_j = W_h (sum_j alpha_ij x_j)  — the projection runs on the *n_rows_dev aggregated rows, not on
 *   all *n_src_dev source rows; x rows are read from the table `src` (fp32 / fp16, [*, d]) through ids (union.nodes).
 * w: [heads*channels][d]; out: [rows_cap][heads*channels] (rows < *n_rows_dev written); fp32-class results that differ
 * from the projection-first order by rounding only.  Built shapes: d % 4 == 0, heads in {1, 2, 4}
 * (GIGL_E_UNSUPPORTED otherwise: take the projection-first entry points).
 * cap_edges: length of `col`.  scratch: DEVICE fp32 [gigl_gat_input_layer_scratch(d, heads, cap_nodes, rows_cap,
 * cap_edges)]. */
int64_t gigl_gat_input_layer_scratch(int32_t d, int32_t heads, int64_t cap_nodes, int64_t rows_cap, int64_t cap_edges);
int32_t gigl_gat_input_layer(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* ids,
                             const float* w, const float* att_src, const float* att_dst, int32_t heads,
                             int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                             const int32_t* col, int64_t cap_edges, const int32_t* n_src_dev, int64_t cap_nodes,
                             const int32_t* n_rows_dev, int64_t rows_cap, const float* bias, int32_t act,
                             float* scratch, float* out);

/* TRAINING over in-HBM batches: the aggregation half of that layer and its backward.  u [2*heads][d] (DEVICE fp32) holds
 * the folded vectors: u[h] = W_h^T att_src_h, u[heads + h] = W_h^T att_dst_h (the caller folds them — a small
 * differentiable product — and projects the aggregated rows itself: out_h = z_h W_h^T, gigl_linear /
 * gigl_linear_weight_grad).
 *   gigl_gat_input_aggregate           z[h][i][0:d] = sum_e alpha_e^h x_e over row i's in-edges and its self loop
 *                                      (z: DEVICE fp32 [heads][rows_cap][d], plain rows), alpha the layer's attention
 *   gigl_gat_input_aggregate_backward  given dz (same shape): du[0:heads] += d u_src, du[heads:2 heads] += d u_dst
 *                                      (du [2*heads][d] is ADDED to: zero it first; fp32 atomics).  The feature rows are
 *                                      inputs: no gradient.  edge_scratch: DEVICE fp32 [2 * heads * cap_edges].
 * Every `col` entry is a local id translated through gather_ids (all rows numbered: a staged union graph).  Rows whose
 * dz is all zero are skipped.  PyG's autograd of GATConv reaches the same gradients through x W first; this order reads
 * each d-wide stored row per edge instead of projecting every source row (python/gigl/src/common/models/pyg/
 * homogeneous.py:300-343 under node_anchor_based_link_prediction_modeling_task_spec.py:334-451). */
int32_t gigl_gat_input_aggregate(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d, const uint32_t* gather_ids,
                                 const float* u, int32_t heads, float negative_slope, const int32_t* rowptr,
                                 const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap,
                                 float* z);
int32_t gigl_gat_input_aggregate_backward(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                          const uint32_t* gather_ids, const float* u, int32_t heads, float negative_slope,
                                          const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                          const int32_t* n_rows_dev, int64_t rows_cap, const float* dz,
                                          float* edge_scratch, float* du);
/* gigl_gat_input_layer in ONE pass over the destination rows (the one-call plan's first GAT layer): every edge's logit is
 * formed from the feature row as it is read for the aggregation (online softmax), so the sources need no dense
 * numbering: rows >= *n_local_dev hold GLOBAL source ids in `col` (the leaf-global union of gigl_sage_plan), rows below
 * it — all rows when n_local_dev is NULL — local ids translated through gather_ids (gather_ids[i] is also row i's own
 * feature row).  Built shapes: d % 4 == 0, d <= 1024, heads in {1, 2, 4}.
 * scratch: DEVICE fp32 [gigl_gat_input_layer_fused_scratch(d, heads, rows_cap)]. */
int64_t gigl_gat_input_layer_fused_scratch(int32_t d, int32_t heads, int64_t rows_cap);
int32_t gigl_gat_input_layer_fused(gigl_ctx* ctx, const void* src, int32_t src_dtype, int32_t d,
                                   const uint32_t* gather_ids, const int32_t* n_local_dev, const float* w,
                                   const float* att_src, const float* att_dst, int32_t heads, int32_t channels,
                                   float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                   const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                   int32_t act, float* scratch, float* out);

/* GATConv with edge features (edge_dim = De; GAT.init_conv_layers passes edge_dim, homogeneous.py:300-343) and
 * EdgeAttrGATConv (python/gigl/src/common/models/pyg/nn/conv/edge_attr_gat_conv.py:11-144):
 *   e_ij = leaky_relu(<h_j, att_src> + <h_i, att_dst> + <W_e e_ij, att_edge>); the added self loop carries the mean
 *   attribute of the row's in-edges (PyG add_self_loops fill_value="mean"; 0 for a row without in-edges);
 *   w_edge_msg != NULL (EdgeAttrGATConv): messages are h_j + W_msg e_ij  ([heads*channels][De]; lin_edge's weight
 *   when share_edge_att_message_weight, lin_edge_message's otherwise).
 * edge_attr: [cap_edges][De] fp32 rows aligned with the positions of `col` (gigl_union_edge_ids -> feature table);
 * att_edge_folded: [heads][De] = W_e^T att_edge per head (the caller folds: <W_e e, att> == <e, W_e^T att>).
 * alpha_scratch: DEVICE fp32 [2*nodes_cap*heads + cap_edges*heads]. */
int32_t gigl_gat_aggregate_edge(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst,
                                int32_t heads, int32_t channels, float negative_slope, int32_t concat,
                                const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                const int32_t* n_nodes_dev, int64_t nodes_cap, const int32_t* n_rows_dev,
                                int64_t rows_cap, const float* bias, int32_t act, const float* edge_attr,
                                int32_t edge_dim, int64_t cap_edges, const float* att_edge_folded,
                                const float* w_edge_msg, float* alpha_scratch, float* out);

/* GATv2Conv attention + aggregation (PyG 2.5.3 GATv2Conv as configured by GATv2.init_conv_layers, homogeneous.py:
 * 346-386; no edge features): xl = lin_l(x), xr = lin_r(x) as fp32 [nodes][heads*channels] rows (gigl_linear; the same
 * pointer twice for share_weights), att: [heads*channels];
 *   z_ij = <att_h, leaky_relu(xl_j + xr_i, negative_slope)> per head over the in-edges of i (self loops removed) plus one
 *   self loop; alpha = softmax_j z_ij; out[i] = sum_j alpha_ij xl_j, heads concatenated, + bias; optional relu.
 * Built shapes: channels % 4 == 0, channels/4 a power of two <= 64, heads*channels <= 1024 (GIGL_E_UNSUPPORTED
 * otherwise).  The backward takes out_pre (the forward's rows before bias / activation) and dout (the gradient with the
 * activation already peeled off): dxl [nodes][H*C] and datt [H*C] are ACCUMULATED into (zero them first), dxr
 * [rows_cap][H*C] is written for rows < *n_rows_dev. */
int32_t gigl_gatv2_aggregate(gigl_ctx* ctx, const float* xl, const float* xr, const float* att, int32_t heads,
                             int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                             const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                             int32_t act, float* out);
int32_t gigl_gatv2_aggregate_backward(gigl_ctx* ctx, const float* xl, const float* xr, const float* att,
                                      int32_t heads, int32_t channels, float negative_slope, const int32_t* rowptr,
                                      const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                      int64_t rows_cap, const float* out_pre, const float* dout, float* dxl,
                                      float* dxr, float* datt);
/* GATv2Conv with edge features (edge_dim): edge_rows = lin_edge(edge_attr) as fp32 [edges][heads*channels] rows in the
 * CSR's edge order (lin_edge has no bias), added inside the leaky_relu: z_ij = <att, leaky_relu(xl_j + xr_i + edge_rows_e)>;
 * the added self loop carries the mean of the row's edge rows (fill_value="mean").  dedge_rows [edges][H*C] is written
 * for the rows' edges. */
int32_t gigl_gatv2_aggregate_edge(gigl_ctx* ctx, const float* xl, const float* xr, const float* att, int32_t heads,
                                  int32_t channels, float negative_slope, const int32_t* rowptr, const int32_t* rowend,
                                  const int32_t* col, const int32_t* n_rows_dev, int64_t rows_cap, const float* bias,
                                  int32_t act, const float* edge_rows, float* out);
int32_t gigl_gatv2_aggregate_edge_backward(gigl_ctx* ctx, const float* xl, const float* xr, const float* att,
                                           int32_t heads, int32_t channels, float negative_slope,
                                           const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                           const int32_t* n_rows_dev, int64_t rows_cap, const float* out_pre,
                                           const float* dout, const float* edge_rows, float* dxl, float* dxr,
                                           float* datt, float* dedge_rows);

/* TransformerConv attention with edge features (PyG 2.5.3 TransformerConv(edge_dim) as configured by
 * Transformer.init_conv_layers, homogeneous.py:440-487): q / k / v = lin_query / lin_key / lin_value rows fp32
 * [nodes][heads*channels], edge_rows = lin_edge(edge_attr) [edges][H*C] in the CSR's edge order;
 *   alpha_e = softmax over the in-edges of i of <q_i, k_j + edge_rows_e> / sqrt(channels),
 *   out[i] = sum_e alpha_e (v_j + edge_rows_e)   (no self loops are added; 0 for a row without in-edges).
 * Shapes as gigl_gatv2_aggregate.  Backward: dq [rows_cap][H*C] and dedge_rows [edges][H*C] are written (rows <
 * *n_rows_dev / their edges), dk and dv [nodes][H*C] ACCUMULATED into (zero them first).  Without edge features the
 * layer is gigl_hgt_aggregate with one edge type. */
int32_t gigl_transformer_aggregate_edge(gigl_ctx* ctx, const float* q, const float* k, const float* v,
                                        const float* edge_rows, int32_t heads, int32_t channels, const int32_t* rowptr,
                                        const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                        int64_t rows_cap, float* out);
int32_t gigl_transformer_aggregate_edge_backward(gigl_ctx* ctx, const float* q, const float* k, const float* v,
                                                 const float* edge_rows, int32_t heads, int32_t channels,
                                                 const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                                 const int32_t* n_rows_dev, int64_t rows_cap, const float* out,
                                                 const float* dout, float* dq, float* dk, float* dv,
                                                 float* dedge_rows);

/* GINEConv aggregation (PyG 2.5.3 GINEConv as configured by GINE.init_conv_layers, homogeneous.py:252-297):
 *   out[i] = (1 + *eps) x[i] + sum over the in-edges e = (j -> i) of relu(x[j] + edge_rows[e]),
 * x: fp32 [nodes][d], edge_rows: fp32 [edges][d] = lin(edge_attr) in the CSR's edge order (`col` positions), eps: DEVICE
 * scalar; rows < *n_rows_dev are written (the conv's MLP is gigl_linear).  Backward: dx [nodes][d] and *deps are
 * ACCUMULATED into (zero them first), dedge_rows [edges][d] is written for the rows' edges. */
int32_t gigl_gine_aggregate(gigl_ctx* ctx, const float* x, const float* edge_rows, const float* eps, int32_t d,
                            const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                            const int32_t* n_rows_dev, int64_t rows_cap, float* out);
int32_t gigl_gine_aggregate_backward(gigl_ctx* ctx, const float* x, const float* edge_rows, const float* eps, int32_t d,
                                     const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                     const int32_t* n_rows_dev, int64_t rows_cap, const float* dout, float* dx,
                                     float* dedge_rows, float* deps);

/* backward of gigl_gat_aggregate / gigl_gat_aggregate_edge (concatenated heads or one head; no W_msg messages) for
 * training through the reference's plugins (GnnTrainingProcess, training_process.py:153-370): given dout = dL/d(out
 * before bias and activation) and out_pre = that output, both [rows][heads*channels]:
 *   dh[nodes][heads*channels]  += the gradient w.r.t. the projected rows through the messages (zero-filled by the
 *                                 caller, accumulated with fp32 atomics),
 *   d_alpha_src / d_alpha_dst [nodes][heads] += the gradients w.r.t. <h, att_src> and <h, att_dst> (zero-filled),
 *   d_alpha_edge [cap_edges][heads] = the gradient w.r.t. the per-edge attention term (with edge_attr only).
 * The caller finishes with dense algebra: dh += d_alpha_src (x) att_src + d_alpha_dst (x) att_dst, d att_src =
 * sum_i d_alpha_src[i] h[i], d att_edge_folded = d_alpha_edge^T edge_attr, and the projection's own backward.
 * Shapes: channels % 4 == 0 with channels/4 a power of two <= 64, or channels % 256 == 0; heads*channels <= 1024
 * (GIGL_E_UNSUPPORTED otherwise).  alpha_scratch as for gigl_gat_aggregate_edge.
 * EdgeAttrGATConv's messages (out_i += W_msg sum_e alpha_e e_e): pass u_msg [rows][heads][De] = W_msg^T dout per
 * head (dense, the caller's) and receive z_out [rows][heads][De] = sum_e alpha_e e_e (self loop: the row's mean
 * attribute), from which d W_msg = sum_i dout_i (x) z_i; both NULL for plain GATConv.  Needs heads*channels % 256 == 0
 * and De <= 4 * (lanes per head).  Rows whose dout row is all zero add nothing anywhere and are skipped: their z_out
 * rows are left as the caller filled them (zero-fill z_out). */
int32_t gigl_gat_aggregate_backward(gigl_ctx* ctx, const float* h, const float* att_src, const float* att_dst,
                                    int32_t heads, int32_t channels, float negative_slope, const int32_t* rowptr,
                                    const int32_t* rowend, const int32_t* col, const int32_t* n_nodes_dev,
                                    int64_t nodes_cap, const int32_t* n_rows_dev, int64_t rows_cap,
                                    const float* out_pre, const float* dout, const float* edge_attr, int32_t edge_dim,
                                    int64_t cap_edges, const float* att_edge_folded, float* alpha_scratch, float* dh,
                                    float* d_alpha_src, float* d_alpha_dst, float* d_alpha_edge,
                                    const float* u_msg, float* z_out);
/* the dense tail of that backward, one pass over the projected rows xw [nodes][heads*channels] (heads*channels <= 1024):
 *   dh[r][c] += d_alpha_src[r][h] att_src[c] + d_alpha_dst[r][h] att_dst[c]   (dh becomes the gradient w.r.t. xw)
 *   d_att_src[c] += sum_r d_alpha_src[r][h] xw[r][c];  d_att_dst[c] += sum_r d_alpha_dst[r][h] xw[r][c]   (h = c / channels)
 * over the first *n_nodes_dev rows; d_att_src / d_att_dst [heads*channels] are ADDED to (zero them first; fp32 atomics).
 * What PyG's autograd of GATConv spreads over a dozen elementwise / reduction kernels. */
int32_t gigl_gat_backward_epilogue(gigl_ctx* ctx, float* dh, const float* d_alpha_src, const float* d_alpha_dst,
                                   const float* xw, const float* att_src, const float* att_dst,
                                   const int32_t* n_nodes_dev, int64_t nodes_cap, int32_t heads, int32_t channels,
                                   float* d_att_src, float* d_att_dst);

/* backward of gigl_gather_mean w.r.t. a dense local fp32 source (gather_ids == NULL; layers >= 2):
 *   dsrc[i][0:d] += dout[i][d:2d];  dsrc[col[e]][0:d] += dout[i][0:d] / deg_i  for e in row i, i < n_rows.
 * dsrc must be zero-filled by the caller.  (PyG's autograd of MessagePassing.propagate + scatter-mean.) */
int32_t gigl_gather_mean_backward(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                  const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                  int64_t rows_cap, float* dsrc);

/* The same backward as a gather over the transposed rows: every row of dsrc below *n_src_dev is WRITTEN once (no zero-fill,
 * no float atomics): dsrc[j] = dout[j][d:2d] (j < *n_rows_dev) + sum over the rows i < *n_rows_dev that list j of
 * dout[i][0:d] / deg_i (aggr = GIGL_AGGR_MEAN; GIGL_AGGR_SUM: without the division).  The transposed lists are built per
 * call from the rows' CSR (scratch from the ctx arena).  d % 4 == 0; every col[] entry of those rows must be < *n_src_dev
 * <= src_cap; edges_cap >= the number of edges of those rows.  The sum over a source's readers runs in arrival order
 * (as the atomics of gigl_gather_mean_backward did). */
int32_t gigl_gather_mean_backward_transposed(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                             const int32_t* rowend, const int32_t* col, const int32_t* n_rows_dev,
                                             int64_t rows_cap, const int32_t* n_src_dev, int64_t src_cap,
                                             int64_t edges_cap, int32_t aggr, float* dsrc);
/* ... in two halves: the transposed lists depend on the batch graph alone, so a training plan builds them beside the
 * previous step's layers (with the sampling and the union build) and the backward only gathers.  `lists`: DEVICE int32
 * [gigl_transposed_rows_words(src_cap, edges_cap)], written by _build, read by _lists. */
int64_t gigl_transposed_rows_words(int64_t src_cap, int64_t edges_cap);
int32_t gigl_transposed_rows_build(gigl_ctx* ctx, const int32_t* rowptr, const int32_t* rowend, const int32_t* col,
                                   const int32_t* n_rows_dev, int64_t rows_cap, const int32_t* n_src_dev, int64_t src_cap,
                                   int64_t edges_cap, int32_t* lists);
int32_t gigl_gather_mean_backward_lists(gigl_ctx* ctx, const float* dout, int32_t d, const int32_t* rowptr,
                                        const int32_t* rowend, const int32_t* n_rows_dev, const int32_t* n_src_dev,
                                        int64_t src_cap, const int32_t* lists, int32_t aggr, float* dsrc);

/* weight gradient of gigl_linear (training: the backward of PyG's Linear inside SAGEConv, which torch autograd computes
 * as dy^T @ a): dw[n][k] += sum_{i < *m_dev} dy[i][n] a[i][k] and, when db != NULL, db[n] += sum_i dy[i][n] — the rows
 * are the inner dimension and only the first *m_dev of the m_cap allocated rows are read (the rest may hold anything).
 * relu_y != NULL ([m_cap][n], the layer's activated output): dy is masked by relu_y > 0 on the way in.  dw / db are
 * ADDED to (zero them first); partial sums over chunks of 256 rows are combined in chunk order (reproducible). */
int32_t gigl_linear_weight_grad(gigl_ctx* ctx, const float* dy, const float* a, const float* relu_y, const int32_t* m_dev,
                                int64_t m_cap, int32_t n, int32_t k, float* dw, float* db);

/* dense projection  y[i][0:n] = act( a[i][0:k] · w[0:n][0:k]^T + bias )  — fp32 MFMA
 * (v_mfma_f32_32x32x2_f32, exact f32).  w is row-major [n][k] (torch Linear layout; for SAGE
 * w = cat(lin_l.weight, lin_r.weight, dim=1)).  act: 0 none, 1 relu.  m read from *m_dev. */
int32_t gigl_linear(gigl_ctx* ctx, const float* a, const float* w, const float* bias,
                    const int32_t* m_dev, int64_t m_cap, int32_t k, int32_t n, int32_t act,
                    float* y);

/* `batch` independent products of equal shape in one launch (row-major operands, k % 4 == 0): product b reads
 * a + b * a_bstride ([m][k]) and w + b * w_bstride ([n][k]) and writes columns [b * n, (b + 1) * n) of y's rows, which
 * are ldy >= batch * n floats apart.  The inner-product decoder of a group of batches (decoder.py:64-66: one torch.mm
 * per batch in the reference) is one call. */
int32_t gigl_linear_batched(gigl_ctx* ctx, const float* a, const float* w, const float* bias, const int32_t* m_dev,
                            int64_t m_cap, int32_t k, int32_t n, int32_t act, int32_t batch, int64_t a_bstride,
                            int64_t w_bstride, int32_t ldy, float* y);
/* `n_groups` independent products of equal K / N in one launch, every one with its OWN operands, row count and output
 * rows (row-major, n floats apart): the per-node-type / per-edge-type projections of a typed layer (HGTConv's k_lin /
 * q_lin / v_lin / a_lin ModuleDicts, python/gigl/src/common/models/pyg/heterogeneous.py:18-120 — one torch Linear call per
 * type in the reference, one launch per type here before).  groups_dev: DEVICE array; m_cap_max >= every group's row
 * capacity.  Same kernel and summation order as gigl_linear on each product (k % 4 == 0). */
typedef struct gigl_linear_group {
  const float* a;       /* [m][k] */
  const float* w;       /* [n][k] */
  const float* bias;    /* [n] or NULL */
  const int32_t* m_dev; /* DEVICE row count */
  float* y;             /* [m][n] */
} gigl_linear_group;
int32_t gigl_linear_grouped(gigl_ctx* ctx, const gigl_linear_group* groups_dev, int32_t n_groups, int64_t m_cap_max,
                            int32_t k, int32_t n, int32_t act);

/* ---- retrieval loss of the link-prediction head, fused (temperature -> sampling-probability correction -> duplicate
 *      / accidental-hit masking -> log-softmax -> cross-entropy against the diagonal, one pass over the scores).
 *      Replaces RetrievalLoss.calculate_batch_retrieval_loss with _mask_by_query_ids / _mask_by_candidate_ids
 *      (python/gigl/src/common/models/layers/loss.py:209-277, :279-305, :307-331) and the CrossEntropyLoss(
 *      reduction="sum") against eye(Q, C) it feeds.
 * scores: DEVICE fp32 [q][c] with `ld` floats between rows, q <= c (row i's positive is column i);
 *   s_ij = scores_ij / temperature (temperature <= 0: none) - log(max(cand_prob[j], 1e-10)) (cand_prob NULL: none);
 *   column j != i is masked (the reference adds finfo.min: softmax term 0) when query_ids != NULL and j < q and
 *   query_ids[j] == query_ids[i], or cand_ids != NULL (remove_accidental_hits) and cand_ids[j] == cand_ids[i];
 *   row_lse[i] = logsumexp_j s_ij, row_loss[i] = row_lse[i] - s_ii, *loss = sum_i row_loss[i] (fixed order).
 * masked_scores (optional, [q][c]): the masked logits themselves (for a caller-supplied loss module).
 * backward: dscores[q][c] = *grad_loss * d loss / d scores (grad_loss NULL: 1).  All pointers DEVICE. */
int32_t gigl_retrieval_loss(gigl_ctx* ctx, const float* scores, int64_t ld, int32_t q, int32_t c, float temperature,
                            const float* cand_prob, const int64_t* query_ids, const int64_t* cand_ids,
                            float* masked_scores, float* row_lse, float* row_loss, float* loss);
/* `batches` independent retrieval losses of equal shape in one pass (the forward of gigl_retrieval_loss per batch;
 * RetrievalLoss.calculate_batch_retrieval_loss, loss.py:209-277, called once per batch by the reference): batch g's
 * scores start batch_stride floats after batch g-1's (rows ld apart), its query_ids / cand_ids / cand_prob / row_lse /
 * row_loss are the g-th block of q or c entries, loss[g] its summed cross-entropy. */
int32_t gigl_retrieval_loss_batched(gigl_ctx* ctx, const float* scores, int64_t ld, int64_t batch_stride, int32_t q,
                                    int32_t c, int32_t batches, float temperature, const float* cand_prob,
                                    const int64_t* query_ids, const int64_t* cand_ids, float* row_lse, float* row_loss,
                                    float* loss);
int32_t gigl_retrieval_loss_backward(gigl_ctx* ctx, const float* scores, int64_t ld, int32_t q, int32_t c,
                                     float temperature, const float* cand_prob, const int64_t* query_ids,
                                     const int64_t* cand_ids, const float* row_lse, const float* grad_loss,
                                     float* dscores);

/* ---- one-call batch pipeline: sample -> union -> GraphSAGE forward -> one output row per root.
 *      Replaces `infer_batch` of the reference's task specs for a RootedNodeNeighborhood batch
 *      (python/gigl/src/common/modeling_task_specs/node_anchor_based_link_prediction_modeling_task_spec.py:626-655,
 *      node_classification_modeling_task_spec.py:176-187: `model(data)[root_node_indices]`) together with
 *      the sampler + collate that feed it.  The plan owns the per-batch workspace (tree, union graph,
 *      activations) sized for batch `b`; graph / feature table / weights are borrowed and may belong
 *      to another ctx on the same device.
 *        dims[0..hops]: dims[0] = feature dim, dims[l+1] = out dim of layer l
 *        w[l]:   DEVICE fp32 [dims[l+1]][2*dims[l]] = cat(lin_l.weight, lin_r.weight, dim=1)
 *        bias[l]: DEVICE fp32 [dims[l+1]] or NULL;  relu between layers (and after the last if act_last)
 *      gigl_sage_plan_run enqueues the whole batch on the ctx stream with no host synchronisation and
 *      writes out[b][dims[hops]] (row i = embedding of roots[i]). */
typedef struct gigl_sage_plan gigl_sage_plan;
int32_t gigl_sage_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b,
                              const int32_t* fanouts, int32_t hops, const int32_t* dims,
                              const float* const* w, const float* const* bias, int32_t act_last,
                              gigl_sage_plan** out);
int32_t gigl_sage_plan_set_weights(gigl_sage_plan* plan, const float* const* w, const float* const* bias);
/* Treat the plan's b roots as b/group_roots INDEPENDENT batches of group_roots roots (the reference's
 * dataloader batch size): every batch gets its own union graph (gigl_union_build_groups), and all of them go
 * through ONE set of launches, which amortises the per-launch cost over several batches.  Row i of the output is
 * bit-identical to what a plan of group_roots roots returns for the batch that holds roots[i].
 * group_roots == b (the default) is a single batch. */
int32_t gigl_sage_plan_set_groups(gigl_sage_plan* plan, int32_t group_roots);
/* borrow the plan's device buffers (valid until the next run overwrites them / destroy) */
int32_t gigl_sage_plan_buffers(gigl_sage_plan* plan, gigl_tree* tree, gigl_union* un);
int32_t gigl_sage_plan_run(gigl_sage_plan* plan, const uint32_t* roots, int32_t sampling_seed,
                           int32_t mode, float* out);
/* the same call in two parts (eager launches), for callers that pipeline batch sets across two plans on two streams:
 * GIGL_PLAN_PART_GRAPH = sample + union graph of `roots` (latency-bound: dependent loads), GIGL_PLAN_PART_LAYERS = the
 * layers over that union graph + the roots' rows (bandwidth- / MFMA-bound).  PART_LAYERS of a batch set must follow
 * PART_GRAPH of the same set on the plan's stream; while it runs, another plan can build the next set's graph. */
#define GIGL_PLAN_PART_GRAPH 1
#define GIGL_PLAN_PART_LAYERS 2
int32_t gigl_sage_plan_run_part(gigl_sage_plan* plan, const uint32_t* roots, int32_t sampling_seed, int32_t mode,
                                float* out, int32_t part);
/* exact work counts of the batch set the plan ran LAST (its tree and union graph are still in the workspace), added
 * to acc[GIGL_STATS_LEN] (DEVICE int64, caller-zeroed, accumulates over calls; enqueued on the ctx stream, no host
 * synchronisation) — the units of BASELINE.json's metric and the algorithmic bytes of SURVEY.md 8(d):
 *   SAMPLED       sum of cnt[k][*]: (src -> dst) pairs emitted by the hop expansions, before batch dedup
 *   AGGREGATED    sum over layers l of the in-edges of the rows layer l computes (levels <= hops-1-l): edges actually
 *                 consumed by a segmented reduce (the reference executes hops * UNION_EDGES)
 *   EXPAND_BYTES  sum over frontier nodes of what a position-keyed sampler must move: 16 (row bounds) + 4*deg for a row of
 *                 <= fanout neighbours, else 8*min(deg, lambda) + 4*fanout (the lambda = f + 4 sqrt(f) + 4 threshold-list
 *                 pairs that can hold the f smallest keys, and the f chosen ids) + 8*min(deg, fanout) written.  (SURVEY
 *                 8(d)'s 4*deg for every row charges an adjacency read the reference's rule never needs: the key of
 *                 entry i depends on the position i only, SamplingStrategy.scala:55.)
 *   AGG_LAYER0+l / ROWS_LAYER0+l   aggregated edges / computed rows of layer l
 * `roots` = the roots passed to that run. */
#define GIGL_STATS_SAMPLED 0
#define GIGL_STATS_AGGREGATED 1
#define GIGL_STATS_UNION_EDGES 2
#define GIGL_STATS_UNION_NODES 3
#define GIGL_STATS_EXPAND_BYTES 4
#define GIGL_STATS_AGG_LAYER0 5  /* + l, l < GIGL_MAX_HOPS */
#define GIGL_STATS_ROWS_LAYER0 9 /* + l */
#define GIGL_STATS_OVERFLOW 13
#define GIGL_STATS_LEN 16
int32_t gigl_sage_plan_stats(gigl_sage_plan* plan, const uint32_t* roots, int64_t* acc);
/* hipGraph replay: on != 0 captures the plan's launches once (per seed/mode/profile mask) and replays them
 * with one graph launch per batch plus the D2D copies of the root ids in and the rows out.  Weights and the
 * graph/feature tables are baked into the captured kernels: call again (or set_weights + use_graph) after
 * changing them.  gigl_sage_plan_flush_profile folds the timing events of in-flight replays into
 * gigl_profile_read (synchronises). */
/* The same one-call pipeline with GAT layers (GAT.init_conv_layers, homogeneous.py:300-343; no edge features):
 * sample -> union -> layer 0 from the input side in one row pass (gigl_gat_input_layer_fused: the plan's leaf-global
 * union needs no dense numbering of the leaves) -> layers >= 1 as projection + gigl_gat_aggregate -> one row per root.
 * w[l]: lin weight [heads[l]*channels[l]][in_l] (in_0 = the feature dim, in_l = heads[l-1]*channels[l-1]: heads are
 * concatenated), att_src / att_dst [heads[l]*channels[l]], bias[l] (may be NULL): DEVICE, borrowed.  The plan handle is
 * a gigl_sage_plan: run / set_groups / use_graph / stats / buffers / destroy are the gigl_sage_plan_* calls.
 * GIGL_E_UNSUPPORTED when the first layer's shape is outside gigl_gat_input_layer_fused's. */
int32_t gigl_gat_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                             int32_t hops, const int32_t* heads, const int32_t* channels, const float* const* w,
                             const float* const* att_src, const float* const* att_dst, const float* const* bias,
                             float negative_slope, int32_t act_last, gigl_sage_plan** out);
int32_t gigl_gat_plan_set_weights(gigl_sage_plan* plan, const float* const* w, const float* const* att_src,
                                  const float* const* att_dst, const float* const* bias);
/* the SAGE layers' reduction: GIGL_AGGR_MEAN (default) | GIGL_AGGR_SUM | GIGL_AGGR_MAX (PyG SAGEConv aggr) */
int32_t gigl_sage_plan_set_aggr(gigl_sage_plan* plan, int32_t aggr);
/* Projected input (inference passes over a feature table whose rows are wider than the first layer's output — MAG240M:
 * 768 fp16 in, 256 out): lin_l(mean_j x_j) = mean_j lin_l(x_j), so [X W_l^T | X W_r^T] is computed ONCE per model over
 * the resident table (gigl_sage_project_features: one gigl_linear pass per row chunk against the stacked weight; w_fused
 * = the first layer's [W_l | W_r], out = DEVICE fp32 [feature rows][2 * n_out]) instead of once per OCCURRENCE of a node
 * in a batch, and the plan's first layer becomes one reduction over the projected rows + the destination's own W_r row +
 * bias (activation fused) — no [mean | self] operand, no per-batch projection; results equal the unprojected order up
 * to fp32 rounding (the reference computes lin_l AFTER the mean: python/gigl/src/common/models/pyg/homogeneous.py:
 * 171-202 via PyG SAGEConv).  The table is borrowed; set_projected_input(plan, NULL) returns to the unprojected layer;
 * it must be recomputed whenever the first layer's weights change (gigl_sage_plan_set_weights does NOT touch it).
 * Mean / sum reductions only. */
int32_t gigl_sage_project_features(gigl_ctx* ctx, gigl_feat* feat, const float* w_fused, int32_t n_out, float* out);
int32_t gigl_sage_plan_set_projected_input(gigl_sage_plan* plan, const float* proj);
/* 1 when the plan's first projection runs over TWO fp16 planes per operand (x = h1 + h2, three MFMAs per accumulator
 * instead of the six of the three-bf16-plane split: same 1e-5 parity class, 2^-22 relative per product).  fp16's range
 * is narrow on both sides, so each operand is multiplied by a power of two that brings its largest magnitude into
 * [2^14, 2^15) on its way into the planes and the accumulators are multiplied by the inverse in the epilogue (all
 * exact): the element error is max(2^-22 |x|, 2^-39 max|x|) whatever the operand's scale.  The table's factor is fixed
 * at create / set_weights / set_aggr from the feature table's largest magnitude (times the largest fan-out, for a sum
 * reduction; looked at once per table); the WEIGHTS' factor is found on the device at every run from the weights as
 * they are, so weights rewritten in place (a training loop) need no call and gigl_sage_plan_set_weights neither reads
 * the weights nor synchronises.  0 = bf16 planes: a table with a NaN / inf, an all-zero table, a largest magnitude
 * outside [2^-46, 2^74], a table whose typical non-zero magnitude lies more than 2^10 below its largest (outlier-
 * dominated: most elements would sit next to the subnormal halves), shapes outside the tiled projection, or
 * GIGL_GEMM_SPLIT=bf16 in the environment.  GAT plans (gigl_gat_plan_create / _set_weights) take the same decision for
 * their first layer's projection. */
int32_t gigl_sage_plan_half_split(gigl_sage_plan* plan);
/* Non-zero when the plan runs its two SAGE layers' projections in ONE kernel (the value = the partial planes of p rows that
 * kernel writes per node: 1 = whole rows, linear_fused2x_kernel, the default; 2 = K-split over the hidden width's two column
 * tiles, the round-5 kernel, GIGL_F2_VARIANT=0 / 4): layer 0's hidden rows are multiplied by the last
 * layer's [W_l | W_r] before they leave the workgroup (W_l mean_j h_j = mean_j W_l h_j), 2 x out floats per node leave
 * instead of the hidden row, and the last layer (homogeneous.py:122-126 -> SAGEConv: W_l mean + b + W_r self) is one
 * reduction over those rows.  Two layers, hidden width 256, 2 x out <= 96, half-split first layer over an fp32 table, a
 * linear reduction (mean / sum); GIGL_PLAN_NO_FUSE2=1 at plan creation keeps the layers apart (A/B).  Same rows up to
 * rounding (1e-5 of the fp32 CPU forward: tests/test_gpu_plan.py). */
int32_t gigl_sage_plan_fused_layers(gigl_sage_plan* plan);
/* *acc (DEVICE int32, caller-zeroed) += 1 when the batch set the plan ran LAST failed (meta[GIGL_META_OVERFLOW] != 0:
 * its rows are NaN) — enqueued on the ctx stream, no synchronisation: callers that stream many calls add every call's
 * flag into one counter and read it once (gigl_amd/hbm.py) */
int32_t gigl_sage_plan_overflow_add(gigl_sage_plan* plan, int32_t* acc);
int32_t gigl_sage_plan_use_graph(gigl_sage_plan* plan, int32_t on);
/* The GRAPH part of every later call (sample + union: latency-bound integer launches that move a few percent of the
 * step's bytes) is issued on `hip_stream` — a stream of the caller's, typically created with a HIGHER priority than the
 * ctx stream — and the layers wait for it on an event; the call still returns without synchronising and is complete when
 * the ctx stream is.  With several plans in flight (one per ctx / stream, as bench.py and the inferencer run them) the
 * graph part of one call is then dispatched ahead of the other calls' bandwidth-bound layers instead of queueing for
 * compute units behind them.  Rows are unchanged (same launches, same order within the call).  hipGraph replay
 * (gigl_sage_plan_use_graph) captures the two parts separately; eager calls ignore the setting.  on = 0: back to one
 * stream.  Synchronises both streams. */
int32_t gigl_sage_plan_set_graph_stream(gigl_sage_plan* plan, void* hip_stream, int32_t on);
int32_t gigl_sage_plan_flush_profile(gigl_sage_plan* plan);
int32_t gigl_sage_plan_destroy(gigl_sage_plan* plan);

/* ---- one TRAINING step per call, all of it in the library: the loop body of NodeClassificationModelingTaskSpec._train
 * (python/gigl/src/common/modeling_task_specs/node_classification_modeling_task_spec.py:134-173 — zero_grad, forward,
 * cross-entropy on `out[root_node_indices]`, backward, Adam step) for batches sampled in HBM:
 *   k-hop sample -> batch union graph (the one-call plan's build) -> GraphSAGE (mean) forward keeping every layer's
 *   operand and output -> mean cross-entropy over the batch's real roots -> backward (gigl_linear_weight_grad per layer;
 *   the input gradient of layers >= 1 by one projection over the transposed weights + gigl_gather_mean_backward) ->
 *   Adam with L2 weight decay (torch.optim.Adam: the decay joins the gradient),
 * ~25 launches, no host read, replayed from hipGraphs after the first two calls.
 * w[l]: fused [dims[l+1]][2*dims[l]] (= [W_l | W_r]), bias[l] (array or entries may be NULL): DEVICE fp32, borrowed and
 * UPDATED IN PLACE by every step (the Adam moments live in the plan, zero at creation).  act_last as gigl_sage_plan_create.
 * gigl_sage_train_plan_step: roots [b] (uint32; a batch of fewer than b real roots is padded by the caller, e.g. with its
 * first root), labels [n_valid] int64 class ids of the first n_valid roots, both DEVICE; loss_out (DEVICE float, may be
 * NULL) receives the step's loss; gigl_sage_train_plan_loss returns the plan's own device word.  A batch whose union graph
 * does not fit the workspace (meta[GIGL_META_OVERFLOW]) trains nothing and reports a NaN loss.  Parity mode and
 * GIGL_MODE_FAST; no with-replacement trees. */
typedef struct gigl_sage_train_plan gigl_sage_train_plan;
int32_t gigl_sage_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b, const int32_t* fanouts,
                                    int32_t hops, const int32_t* dims, float* const* w, float* const* bias,
                                    int32_t act_last, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    gigl_sage_train_plan** out);
/* The step has two parts: the GRAPH part (sample + union: independent of the weights) and the LAYERS part (forward,
 * loss, backward, Adam).  roots_next (DEVICE [b], may be NULL) = the roots of the batch the NEXT call will be given: its
 * graph part then runs on a stream of the plan's own, into the second of the plan's two workspaces, while this batch's
 * layers part runs on the ctx's stream (the next call's `roots` are then not read again).  Results do not depend on it. */
int32_t gigl_sage_train_plan_step(gigl_sage_train_plan* plan, const uint32_t* roots, const int64_t* labels, int32_t n_valid,
                                  const uint32_t* roots_next, int32_t sampling_seed, int32_t mode, float* loss_out);
/* the same with the batch AFTER the next announced too (roots_next2: what the call after the next will pass; NULL: none):
 * two graph parts — small latency-bound launches — are then in flight on streams of their own next to this batch's layers.
 * CONTRACT of an announced batch (roots_next / roots_next2, and main_roots_next / rn_roots_next of
 * gigl_nablp_train_plan_step2): the plan samples from the announced DEVICE buffer on its own stream right away and
 * recognises the batch at the next call by the buffer's ADDRESS — the buffer must keep its contents, and must not be
 * handed over again for a different batch, until the step that consumes it has been issued.  A caller that refills one
 * static buffer per step must not announce it (pass NULL: the step then samples when it is called). */
int32_t gigl_sage_train_plan_step2(gigl_sage_train_plan* plan, const uint32_t* roots, const int64_t* labels, int32_t n_valid,
                                   const uint32_t* roots_next, const uint32_t* roots_next2, int32_t sampling_seed,
                                   int32_t mode, float* loss_out);
const float* gigl_sage_train_plan_loss(gigl_sage_train_plan* plan);
int32_t gigl_sage_train_plan_destroy(gigl_sage_train_plan* plan);
/* Steps are issued without a host read in between, so a batch that fails (NaN loss: its union graph did not fit the
 * workspace, or a label outside the output width) HALTS the plan on the device: every later step of the queue trains
 * nothing and reports NaN too — the batches behind a failed one are never applied ahead of it.  The host finds the first
 * NaN in the losses it collected, deals with that batch (a wider plan: gigl_ctx_set_wide_workspaces +
 * gigl_sage_train_plan_adopt, or skipping it), and resumes: the halt flag is cleared on the plan's stream. */
int32_t gigl_sage_train_plan_resume(gigl_sage_train_plan* plan);
/* dst takes over src's optimiser state (Adam's moments and step counter; the weights are the caller's buffers, shared by
 * construction): a plan re-created with wider workspaces (gigl_ctx_set_wide_workspaces) continues the training run of
 * the one it replaces.  Same shape and widths; synchronises both plans' streams. */
int32_t gigl_sage_train_plan_adopt(gigl_sage_train_plan* dst, gigl_sage_train_plan* src);
/* Adam's first / second moments of `layer`'s fused weight [dims[l+1]][2 dims[l]] and bias [dims[l+1]] copied into the
 * caller's DEVICE buffers on the ctx stream (any may be NULL): what torch.optim.Adam keeps as exp_avg / exp_avg_sq —
 * checkpointing, and the parity tests (moments are linear / quadratic in the gradients: comparable where the raw
 * parameters of elements with rounding-noise gradients are not) */
int32_t gigl_sage_train_plan_moments(gigl_sage_train_plan* plan, int32_t layer, float* m_w, float* v_w, float* m_b, float* v_b);

/* ---- one LINK-PREDICTION training step per call, all of it in the library (round 5).  Replaces the loop body of
 * NodeAnchorBasedLinkPredictionModelingTaskSpec.train (python/gigl/src/common/modeling_task_specs/
 * node_anchor_based_link_prediction_modeling_task_spec.py:334-451) for the reference's default encoder (GraphSAGE, mean
 * aggregation, optional L2-normalised output) and its default task (Retrieval: inner-product scores of every (anchor,
 * positive) query row against cat(positives, random negatives), RetrievalLoss with temperature, same-query and
 * accidental-hit masks, summed cross-entropy / query rows: utils/infer.py, decoder.py:64-70, loss.py:209-331), Adam with L2
 * weight decay: sample + union of the main batch and of the random-negative batch, two encoder forwards over the SHARED
 * weights, the head, the backward of both encodes (their weight gradients are added), the update — no torch kernel, no
 * host read, replayed as two hipGraphs per step (graph part, layers part).
 * create: b_anchors anchors x (1 + num_positives) rooted trees per main batch, n_random_negatives roots per negative
 *   batch; w[l] = fused [dims[l+1]][2 dims[l]] (= [W_l | W_r]), bias[l] (may be NULL): DEVICE, borrowed and UPDATED IN
 *   PLACE by every step; dims[hops] <= 512.
 * step: main_roots DEVICE uint32 [b_anchors * (1 + P)], anchor-major (anchor, its P positive slots; a slot without a
 *   positive repeats the anchor's id — or any id — and lies at or beyond pos_cnt[anchor]; an absent anchor of a short batch
 *   is 0xFFFFFFFF with pos_cnt 0), pos_cnt DEVICE int32 [b_anchors], rn_roots DEVICE uint32 [n_random_negatives]
 *   (0xFFFFFFFF = none).  loss_out (DEVICE float[2], may be NULL) = {loss, query rows that took part};
 *   gigl_nablp_train_plan_loss returns the plan's own words.  A batch that does not fit its workspace trains nothing and
 *   reports a NaN loss.  All work is enqueued on the ctx's stream — or, since round 6, on private streams forked from it and
 *   joined back into it before the step's last launch (the random negatives' encode; the main batch's weight gradients of
 *   the GraphSAGE encoder): nothing a caller synchronises differently.  Environment, read at plan creation (A/B knobs):
 *   GIGL_LP_FORK=0 (no private streams), GIGL_LP_WGRAD_STREAM=0 (weight gradients stay in the chain),
 *   GIGL_TRAIN_PLAN_UNFUSED=1 (a reduce launch per weight gradient instead of partial sums inside the Adam kernel). */
typedef struct gigl_nablp_train_plan gigl_nablp_train_plan;
int32_t gigl_nablp_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b_anchors,
                                     int32_t num_positives, int32_t n_random_negatives, const int32_t* fanouts, int32_t hops,
                                     const int32_t* dims, float* const* w, float* const* bias, int32_t act_last,
                                     int32_t l2_normalize, float temperature, int32_t remove_accidental_hits, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, gigl_nablp_train_plan** out);
int32_t gigl_nablp_train_plan_step(gigl_nablp_train_plan* plan, const uint32_t* main_roots, const int32_t* pos_cnt,
                                   const uint32_t* rn_roots, int32_t sampling_seed, int32_t mode, float* loss_out);
/* step2 = step + the NEXT step's root buffers (may be NULL: then it is step): their graph part — sample + union of both root
 * sets, latency-bound launches — is enqueued on a side stream into the plan's second workspace and runs beside this
 * step's layers; the next call finds it done if it is handed the same two buffers (otherwise it samples again).  The
 * buffers must stay untouched until that call. */
int32_t gigl_nablp_train_plan_step2(gigl_nablp_train_plan* plan, const uint32_t* main_roots, const int32_t* pos_cnt,
                                    const uint32_t* rn_roots, const uint32_t* next_main_roots,
                                    const uint32_t* next_rn_roots, int32_t sampling_seed, int32_t mode, float* loss_out);
const float* gigl_nablp_train_plan_loss(gigl_nablp_train_plan* plan);
/* the LAST step's parameter gradients of layer `layer` (the two encodes' added): gw DEVICE [dims[l+1]][2 dims[l]] (= d loss
 * / d [W_l | W_r]), gb DEVICE [dims[l+1]] (may be NULL) — what the step's Adam update consumed; for gradient parity tests */
int32_t gigl_nablp_train_plan_grads(gigl_nablp_train_plan* plan, int32_t layer, float* gw, float* gb);
/* as gigl_sage_train_plan_adopt, for the link-prediction plans (GraphSAGE and GAT encoders alike) */
int32_t gigl_nablp_train_plan_adopt(gigl_nablp_train_plan* dst, gigl_nablp_train_plan* src);
/* moments of one parameter tensor of a link-prediction plan: GraphSAGE encoder: index = 2 * layer (fused weight) | 2 * layer
 * + 1 (bias); GAT encoder: index 0..7 = w0, att_src0, att_dst0, bias0, w1, att_src1, att_dst1, bias1 */
int32_t gigl_nablp_train_plan_moments(gigl_nablp_train_plan* plan, int32_t index, float* m, float* v);
int32_t gigl_nablp_train_plan_destroy(gigl_nablp_train_plan* plan);
/* The same plan with the GAT encoder configs[4] names (GAT.init_conv_layers, python/gigl/src/common/models/pyg/
 * homogeneous.py:300-343): hops == 2, heads[0] in {1, 2, 4} concatenated heads of channels[0] in the first layer, one head of
 * channels[1] (= the embedding width, <= 512) in the second, no edge features, no activation after the last layer; feature
 * rows of d % 4 == 0, d <= 1024 floats (fp32 or fp16 table).  The first layer runs from the INPUT side (attention-weighted
 * sums of the stored rows under the folded attention vectors, then one projection per head) for the nodes of level <= 1, the
 * second for the roots; the batch graphs number every node (generic union).  Parameters per layer: w [heads*channels][in],
 * att_src / att_dst [heads*channels], bias [heads*channels] (may be NULL) — DEVICE fp32, borrowed and UPDATED IN PLACE.
 * step / step2 / loss / destroy are gigl_nablp_train_plan's; _grads hands out the LAST step's gradients of one layer
 * (g_att_src / g_att_dst [heads*channels]; gb may be NULL). */
int32_t gigl_gat_nablp_train_plan_create(gigl_ctx* ctx, gigl_graph* graph, gigl_feat* feat, int32_t b_anchors,
                                         int32_t num_positives, int32_t n_random_negatives, const int32_t* fanouts,
                                         int32_t hops, const int32_t* heads, const int32_t* channels, float* const* w,
                                         float* const* att_src, float* const* att_dst, float* const* bias,
                                         float negative_slope, int32_t l2_normalize, float temperature,
                                         int32_t remove_accidental_hits, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, gigl_nablp_train_plan** out);
int32_t gigl_gat_nablp_train_plan_grads(gigl_nablp_train_plan* plan, int32_t layer, float* gw, float* g_att_src,
                                        float* g_att_dst, float* gb);

/* Count-min sketch of candidate ids for the Retrieval task's candidate-sampling correction
 * (python/gigl/src/common/models/layers/count_min_sketch.py:11-95, used by task.py:140-205): table = DEVICE int32
 * [depth][width], zeroed by the caller; cell of (id, row) = hash((id, row)) % width with CPython's tuple hash of two
 * ints and Python's non-negative remainder, so the table equals the reference's for the same ids.
 * gigl_cms_add: table[row][cell] += 1 for every id and row (duplicates count).  gigl_cms_estimate: counts[k] = min over
 * rows of the id's cells.  ids / counts: DEVICE int64 [n]. */
int32_t gigl_cms_add(gigl_ctx* ctx, int32_t* table, int32_t width, int32_t depth, const int64_t* ids, int64_t n);
int32_t gigl_cms_estimate(gigl_ctx* ctx, const int32_t* table, int32_t width, int32_t depth, const int64_t* ids,
                          int64_t n, int64_t* counts);

/* ---- heterogeneous encoders: the attention-weighted segmented reductions of HGTConv and SimpleHGNConv
 *      (python/gigl/src/common/models/pyg/nn/conv/hgt_conv.py:161-244, simplehgn_conv.py:113-180; the models are
 *      python/gigl/src/common/models/pyg/heterogeneous.py:18-273).  Rows are [heads*dim] fp32; dim % 4 == 0 with dim/4
 *      a power of two <= 64, heads*dim <= 1024 (GIGL_E_UNSUPPORTED otherwise).  The merged ("bipartite") graph of all
 *      edge types is a CSR by destination: rowptr[n_dst+1], col[e] = row of the source in k / v.  All pointers DEVICE.
 * gigl_hgt_aggregate: alpha_e = <q_i, k_col[e]> * p_rel[etype[e]][h] / sqrt(dim) per head, softmax over ALL in-edges
 *      of i, out_i = sum_e alpha_e v_col[e] (0 for a row without in-edges).  k / v are the relation-transformed source
 *      rows (one block of rows per edge type, :115-159); p_rel: [n_edge_types][heads] (NULL: 1), etype NULL: type 0.
 * gigl_simplehgn_alpha: alpha[e][h] = softmax over the edges that share the SOURCE node (softmax(alpha, row), :154-155)
 *      of leaky_relu(hl[src[e]][h] + hr[dst[e]][h] + het[etype[e]][h] (+ hef[e][h])); group_scratch: [2*n_nodes*heads].
 * gigl_weighted_aggregate: out_i = sum_{e in row i} alpha[e][h] * v[col[e]] (alpha in the CSR's edge order). */
int32_t gigl_hgt_aggregate(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads, int32_t dim,
                           const int32_t* rowptr, const int32_t* col, const int32_t* etype, const float* p_rel,
                           int64_t n_dst, float* out);
/* gigl_hgt_aggregate with the layer's GELU (exact erf form, hgt_conv.py: F.gelu before the output projection) applied in
 * the reduce's epilogue when gelu != 0 */
int32_t gigl_hgt_aggregate_act(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads, int32_t dim,
                               const int32_t* rowptr, const int32_t* col, const int32_t* etype, const float* p_rel,
                               int64_t n_dst, int32_t gelu, float* out);
int32_t gigl_simplehgn_alpha(gigl_ctx* ctx, const float* hl, const float* hr, const float* het, const float* hef,
                             const int32_t* src, const int32_t* dst, const int32_t* etype, int64_t n_edges,
                             int64_t n_nodes, int32_t heads, float negative_slope, float* group_scratch, float* alpha);
int32_t gigl_weighted_aggregate(gigl_ctx* ctx, const float* alpha, const float* v, int32_t heads, int32_t dim,
                                const int32_t* rowptr, const int32_t* col, int64_t n_dst, float* out);
/* Backward of gigl_hgt_aggregate (training HGT: heterogeneous.py:18-119 under autograd): `out` = what the forward
 * wrote, `dout` its gradient.  dq [n_dst, heads*dim] is written; dk, dv [n_src, heads*dim] and dp_rel [n_types, heads]
 * (may be NULL) are ACCUMULATED into (zero them first): a source row has many destinations.  The softmax statistics
 * are recomputed from q / k, nothing else needs saving. */
int32_t gigl_hgt_aggregate_backward(gigl_ctx* ctx, const float* q, const float* k, const float* v, int32_t heads,
                                    int32_t dim, const int32_t* rowptr, const int32_t* col, const int32_t* etype,
                                    const float* p_rel, int64_t n_dst, const float* out, const float* dout, float* dq,
                                    float* dk, float* dv, float* dp_rel);
/* Backward of gigl_weighted_aggregate: dalpha [E, heads] is written, dv [n_src, heads*dim] accumulated into. */
int32_t gigl_weighted_aggregate_backward(gigl_ctx* ctx, const float* alpha, const float* v, int32_t heads, int32_t dim,
                                         const int32_t* rowptr, const int32_t* col, int64_t n_dst, const float* dout,
                                         float* dalpha, float* dv);

/* ---- split generator: hash slots of the assigners, in bulk.  Replaces HashingAssigner.assign's per-object hashing
 *      (scala/split_generator/src/main/scala/lib/assigners/AbstractAssigners.scala:30-111):
 *      slots[i] = floorMod(MurmurHash3.bytesHash(key_i), 10000), key = "<a>-<type>" for nodes (b == NULL;
 *      NodeToDatasetSplitHashingAssigner.scala) or "<src>-<type>-<dst>" for edges (a = src, b = dst; endpoints ordered
 *      (min, max) first when symmetric != 0: TransductiveEdgeToLinkSplitHashingAssigner.scala:66-78).  The bucket a
 *      slot belongs to (cumulative float32 weights) stays with the caller.  All pointers DEVICE. */
int32_t gigl_split_hash_slots(gigl_ctx* ctx, const uint32_t* a, const uint32_t* b, int64_t n, int32_t condensed_type,
                              int32_t symmetric, int32_t* slots);

/* ---- the hash-partitioned (multi-GPU) step inside the library.
 *      Replaces the reference's distributed loader path: DistLinkPredictionDataPartitioner (owner(v) = v % world,
 *      python/gigl/distributed/dist_link_prediction_data_partitioner.py:692-695) + DistNeighborLoader's per-batch RPC
 *      fan-out to the partitions' sampling workers (python/gigl/distributed/distributed_neighborloader.py:26-192) and
 *      the feature lookup that follows — here one process per GPU, the exchanges are equal-split all-to-alls issued
 *      from C++ on the ctx stream, and nothing in a step reads device memory from the host.
 *
 * Communicators.  A gigl_comm belongs to ONE ctx (its stream carries the exchanges).
 *   gigl_comm_unique_id   rank 0 creates the 128-byte RCCL id; the caller hands it to the other ranks (its own
 *                         channel: torch.distributed store, MPI, a file)
 *   gigl_dist_init        RCCL communicator (librccl is opened at run time; GIGL_E_UNSUPPORTED when absent)
 *   gigl_dist_init_local  every rank of a world is a ctx of THIS process (same device, same stream): exchanges are
 *                         device copies, performed when all ranks have registered theirs (gigl_comm_flush_local);
 *                         ranks advance phase by phase (gigl_dist_plan_run_local) — tests, single-process drivers
 *   gigl_dist_init_callback  the caller moves the bytes (MPI, gloo, ...): fn(user, send, recv, bytes_per_peer) gets
 *                         DEVICE pointers ([world][bytes_per_peer] each) after the stream was synchronised, returns 0
 *   gigl_comm_all_to_all  block p of `send` goes to rank p, block r of `recv` comes from rank r */
#define GIGL_COMM_RCCL 0
#define GIGL_COMM_LOCAL 1
#define GIGL_COMM_CALLBACK 2
#define GIGL_COMM_ID_BYTES 128
typedef struct gigl_comm gigl_comm;
typedef int32_t (*gigl_exchange_fn)(void* user, const void* send, void* recv, int64_t bytes_per_peer);
int32_t gigl_comm_unique_id(void* id /* HOST [GIGL_COMM_ID_BYTES] */);
int32_t gigl_dist_init(gigl_ctx* ctx, int32_t rank, int32_t world, const void* rccl_unique_id, gigl_comm** out);
int32_t gigl_dist_init_local(gigl_ctx* const* ctxs, int32_t world, gigl_comm** out /* [world] */);
int32_t gigl_dist_init_callback(gigl_ctx* ctx, int32_t rank, int32_t world, gigl_exchange_fn fn, void* user,
                                gigl_comm** out);
int32_t gigl_comm_info(gigl_comm* comm, int32_t* rank, int32_t* world, int32_t* kind);
int32_t gigl_comm_all_to_all(gigl_comm* comm, const void* send, void* recv, int64_t bytes_per_peer);
int32_t gigl_comm_flush_local(gigl_comm* any_member);
/* The feature-row exchange moves only the REQUESTED rows of each block, which costs one host read of 2 * world counts per
 * call (RCCL needs the sizes on the host).  on = 1: whole blocks travel instead — more bytes on the links, no host read
 * anywhere in a step, so that a step can be captured into a hipGraph (the emulated world of bench/sharded.py replays its
 * ranks' steps that way).  Set it on every rank of the communicator alike. */
int32_t gigl_comm_set_fixed_blocks(gigl_comm* comm, int32_t on);
/* bytes this rank has sent to OTHER ranks since the communicator was created: as moved, and as they would have been
 * with every block sent at its full capacity.  The sharded plans' feature-row exchange moves only the requested rows
 * of each block (the counts travel with the id request; RCCL and in-process groups — the host-callback transport has a
 * fixed-size contract and moves full blocks; GIGL_DIST_FIXED_BLOCKS=1 forces full blocks everywhere). */
int32_t gigl_comm_traffic(gigl_comm* comm, int64_t* moved_bytes, int64_t* full_block_bytes);
int32_t gigl_comm_destroy(gigl_comm* comm);

/* The sharded batch plan: gigl_sage_plan's step on a hash-partitioned graph.  `shard` holds the CSC rows of the nodes
 * this rank owns (row v / world, ids inside rows global), `shard_feat` their feature rows (row v / world).  A step =
 * 2*hops + 3 phases; every phase but the last ends in an exchange:
 *   2k    requester: scatter hop k-1's answers into the tree, bucket hop k's frontier by owner  -> (ids, path sums)
 *   2k+1  owner: gigl_expand_frontier over the requests of every peer                           -> f ids per request
 *   2L    requester: last scatter, gigl_union_build_groups, bucket the union graph's node ids   -> ids
 *   2L+1  owner: the requested feature rows gathered straight into the send buffer              -> rows
 *   2L+2  requester: GraphSAGE forward over the union graph (first layer reads the rows where they arrived), out[b]
 * Buckets have fixed capacities, so no count ever travels and the host never reads the device: hop buckets hold
 * ceil(m/world * (1 + hop_slack)) + 512 requests per peer (8 B each), the row buckets pull_cap rows per peer (0: the
 * worst case / world + 10 %; tune it with GIGL_STATS_PULL_BUCKET_MAX from warm-up steps — rows are the bytes that
 * matter on xGMI).  A bucket that overflows fails the batch: meta[GIGL_META_OVERFLOW] != 0, levels zeroed.
 * project_on_owner: the owners apply the first layer's weights before sending — lin_l's for every requested node, lin_r's
 * for the nodes of level < hops — so dims[1] fp32 travel per row instead of the raw row (exact up to fp32 rounding:
 * lin_l(mean_j x_j) == mean_j lin_l(x_j)); pays when dims[1]*4 < the raw row bytes and the links, not the MFMAs, bound
 * the step.  max_window_end: see gigl_expand_frontier.  Same weights layout as gigl_sage_plan_create. */
typedef struct gigl_dist_plan gigl_dist_plan;
typedef struct gigl_dist_plan_opts {
  int32_t group_roots;       /* roots per independent batch (0: b) */
  int32_t project_on_owner;
  int64_t pull_cap;          /* feature rows per peer and step (0: default bound) */
  float hop_slack;           /* 0: 0.5 */
  int64_t max_window_end;    /* -1: unknown */
  const float* projected;    /* NULL, or this rank's PRE-PROJECTED rows: gigl_sage_project_features over the shard's
                                feature table with the plan's first-layer weight ([shard rows][2*dims[1]] fp32 =
                                [W_l x | W_r x], DEVICE, borrowed; recompute after a weight update).  The pull then moves
                                W_l x rows (dims[1] fp32 instead of the raw row: MAG240M 1.5 KB -> 1 KB), the W_r x rows
                                of the nodes of level < hops come in a second small pull, and the first layer is one
                                reduction — no projection per step, on owner or requester.  Two-hop plans, not with
                                project_on_owner; replicated hot rows (gigl_dist_plan_set_hot_rows) are then W_l x rows
                                ([n_hot][dims[1]] fp32). */
  int64_t pull_cap_b;        /* rows per peer and step of the SECOND pull of a pre-projected plan (0: the worst case,
                                every node of level < hops distinct; size it from gigl_dist_plan_bucket_fill) */
  int32_t staged;            /* 1: the plan serves TRAINING batches — every union node is numbered and its pulled row is
                                located through pos[] (the generic union build; raw rows: not with project_on_owner /
                                projected).  Run the phases before the last one (gigl_dist_plan_phase 0 .. n-2: sample,
                                union, feature pull), then read the batch union graph (gigl_dist_plan_buffers) and its
                                dense feature matrix (gigl_dist_plan_batch_features): what the trainer's collate hands
                                the encoder (pyg_graph_builder.py:20-69), for a graph sharded over the ranks */
  int32_t peer_direct;       /* 1: the PEER-MAPPED route of the feature pull — the first layer reads every source row where it
                                lives: row v / world of rank (v % world)'s table, mapped into this process (hipIpc handles
                                between processes: gigl_ipc_export / gigl_ipc_open; plain device pointers inside one) and
                                handed over with gigl_dist_plan_set_peer_tables before the first step.  No claim, no id
                                exchange, no owner-side gather, no receive buffer: the rows cross xGMI inside the
                                aggregation kernel's loads, once per OCCURRENCE (per-call dedup goes with the buckets;
                                replicated hot rows still stay local).  Same rows summed in the same order as the bucketed
                                route: bit-identical results.  Needs the dense plan shape (SAGE layers, two hops, second
                                fan-out <= 64, raw or pre-projected rows), < 2^31 nodes, world <= 64, and ranks whose memory
                                is peer-accessible (one node).  Replaces the same chunked scatter / RPC feature lookup as
                                the bucketed pull: dist_link_prediction_data_partitioner.py:560-664,
                                distributed_neighborloader.py:162-192 */
  int32_t peer_sample;       /* 1 (with peer_direct): the PEER-SAMPLED route — the ranks' CSC shards are mapped too
                                (gigl_dist_plan_set_peer_graphs) and every rank expands its OWN frontier, reading the owners'
                                adjacency rows where they live: no request / answer buckets, no hop exchange, no scatter — the
                                step has no collective left.  The same selection rule over the same rows: trees and rows
                                bit-identical to the exchange route.  Needs fan-outs <= 64, no directed multi-edges, a window
                                bound (max_window_end).  Replaces the per-hop sampling RPC of
                                distributed_neighborloader.py:162-192 like the exchange route does */
} gigl_dist_plan_opts;
int32_t gigl_dist_plan_create(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                              const int32_t* fanouts, int32_t hops, const int32_t* dims, const float* const* w,
                              const float* const* bias, int32_t act_last, const gigl_dist_plan_opts* opts,
                              gigl_dist_plan** out);
/* the SAGE layers' reduction (PyG SAGEConv `aggr`: homogeneous.py:107-153 passes it through): GIGL_AGGR_MEAN (default) |
 * _SUM | _MAX.  MAX needs raw pulled rows (not with project_on_owner / projected: lin_l does not commute with max). */
int32_t gigl_dist_plan_set_aggr(gigl_dist_plan* plan, int32_t aggr);
/* The sharded plan with GAT layers (BASELINE configs[4]: link-prediction GAT over the hash-partitioned MAG240M-shaped
 * graph; python/gigl/src/common/models/pyg/homogeneous.py:300-343 via PyG GATConv): the sampling / union / feature-pull
 * phases of gigl_dist_plan_create (raw rows, every union node numbered), then the layer stages of gigl_gat_plan_create
 * over the pulled rows — first layer from the input side in one row pass (gigl_gat_input_layer_fused reading the
 * receive buffer through pos[]), layers >= 1 projection + attention.  Same weight layout as gigl_gat_plan_create; run /
 * phase / run_local / stats / buffers / destroy are the gigl_dist_plan_* calls. */
int32_t gigl_dist_gat_plan_create(gigl_comm* comm, gigl_graph* shard, gigl_feat* shard_feat, int32_t b,
                                  const int32_t* fanouts, int32_t hops, const int32_t* heads, const int32_t* channels,
                                  const float* const* w, const float* const* att_src, const float* const* att_dst,
                                  const float* const* bias, float negative_slope, int32_t act_last,
                                  const gigl_dist_plan_opts* opts, gigl_dist_plan** out);
int32_t gigl_dist_gat_plan_set_weights(gigl_dist_plan* plan, const float* const* w, const float* const* att_src,
                                       const float* const* att_dst, const float* const* bias);
int32_t gigl_dist_plan_set_weights(gigl_dist_plan* plan, const float* const* w, const float* const* bias);
int32_t gigl_dist_plan_phases(gigl_dist_plan* plan, int32_t* n);
int32_t gigl_dist_plan_phase(gigl_dist_plan* plan, int32_t phase, const uint32_t* roots, int32_t sampling_seed,
                             float* out);
/* all phases of one step on an RCCL / callback communicator */
int32_t gigl_dist_plan_run(gigl_dist_plan* plan, const uint32_t* roots, int32_t sampling_seed, float* out);
/* one step of EACH of `n` plans of this rank (their own ctx / stream / communicator each: the plans a rank keeps in
 * flight), the phases issued interleaved — phase 0 of every plan, then phase 1 of every plan, ... — by one host call, so
 * that one plan's exchange overlaps another's expansion / forward and every rank issues its collectives in the same order
 * (what the reference's loader gets from worker processes: python/gigl/distributed/distributed_neighborloader.py:162-192).
 * RCCL / callback communicators; roots[i] / out[i] belong to plans[i]. */
int32_t gigl_dist_plan_run_interleaved(gigl_dist_plan* const* plans, int32_t n, const uint32_t* const* roots,
                                       int32_t sampling_seed, float* const* out);
/* one step of every rank of an in-process group, phase by phase; plans[r] = rank r's plan */
int32_t gigl_dist_plan_run_local(gigl_dist_plan* const* plans, int32_t world, const uint32_t* const* roots,
                                 int32_t sampling_seed, float* const* out);
/* Replicated hot rows (hub-row replication): feature rows that every rank keeps a copy of are read locally and never
 * requested over the links.  hot_ids: n_hot distinct GLOBAL node ids (device), hot_rows: their feature rows [n_hot, d]
 * in the shard's feature dtype (device; BORROWED: keep it alive while the plan runs) — the same set on every rank (the
 * job replicates it at setup, e.g. the nodes that occur most often as in-neighbours).  Results are unchanged; needs
 * the plan's dense pull bookkeeping (two hops, raw rows; GIGL_E_INVALID_ARG otherwise).  n_hot = 0 clears the set. */
int32_t gigl_dist_plan_set_hot_rows(gigl_dist_plan* plan, const uint32_t* hot_ids, int64_t n_hot, const void* hot_rows);
/* peer-mapped plans (opts->peer_direct): tables[r] = rank r's table as a DEVICE pointer valid in THIS process — its feature
 * rows, or its pre-projected [W_l x | W_r x] rows when the plan was created over `projected` — for r = 0 .. world-1 (HOST
 * array; tables[rank] = the plan's own table).  Once, before the first step (a one-rank world needs no call). */
int32_t gigl_dist_plan_set_peer_tables(gigl_dist_plan* plan, const void* const* tables);
/* peer-sampled plans (opts->peer_sample): rowptrs[r] / cols[r] = rank r's CSC shard (gigl_graph_device_ptrs on rank r, mapped
 * here through gigl_ipc_export / gigl_ipc_open — two allocations per rank) as DEVICE pointers valid in THIS process (HOST
 * arrays; entries [rank] = the plan's own shard).  Once, before the first step. */
int32_t gigl_dist_plan_set_peer_graphs(gigl_dist_plan* plan, const int64_t* const* rowptrs, const uint32_t* const* cols);
/* Sharing a device allocation with the other ranks' processes of the node (what gigl_dist_plan_set_peer_tables is fed
 * with): export = (handle of the allocation dev_ptr lies in, dev_ptr's offset inside it) — ship both to the peers over any
 * host channel (torch.distributed all_gather); open = map a peer's allocation into this process, *ptr = the peer's dev_ptr
 * here, *base = what gigl_ipc_close takes when the plans that read it are gone.  HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf). */
#define GIGL_IPC_HANDLE_BYTES 64
int32_t gigl_ipc_export(gigl_ctx* ctx, const void* dev_ptr, void* handle /* HOST [GIGL_IPC_HANDLE_BYTES] */, int64_t* offset);
int32_t gigl_ipc_open(gigl_ctx* ctx, const void* handle, int64_t offset, void** base, void** ptr);
int32_t gigl_ipc_close(gigl_ctx* ctx, void* base);
int32_t gigl_dist_plan_buffers(gigl_dist_plan* plan, gigl_tree* tree, gigl_union* un);
/* staged plans, after the feature-pull phases of a step: x[i][0:d] (DEVICE fp32 [cap_nodes][d], d = the shard's feature
 * width) = the feature row of union node i for i < meta[GIGL_META_N_NODES] — own rows and rows pulled from their owners
 * alike; NaN for a node whose request overflowed its bucket (the step is flagged: meta[GIGL_META_OVERFLOW]).
 * Replaces the feature hydration of the trainer's batches (training_process.py:86-119 over
 * distributed_neighborloader.py:26-192) when the graph is hash-partitioned. */
int32_t gigl_dist_plan_batch_features(gigl_dist_plan* plan, float* x);
/* ... and the batch union graph itself copied into the caller's DEVICE arrays on the plan's stream (no synchronisation;
 * the plan's own buffers are reused by the next step): rowptr / rowend [cap_nodes + 1], col [cap_edges], root_local [b],
 * meta [GIGL_META_LEN], nodes [cap_nodes] (global ids; may be NULL) — capacities from gigl_dist_plan_buffers. */
int32_t gigl_dist_plan_batch_graph(gigl_dist_plan* plan, int32_t* rowptr, int32_t* rowend, int32_t* col,
                                   int32_t* root_local, int32_t* meta, uint32_t* nodes);
/* like gigl_sage_plan_stats for the step run last, plus the feature rows requested (PULLED_ROWS, summed) and the
 * fullest row bucket seen (PULL_BUCKET_MAX, a running maximum) */
#define GIGL_STATS_PULLED_ROWS 14
#define GIGL_STATS_PULL_BUCKET_MAX 15
/* fill of the feature-pull buckets of the step run last, folded into acc4 (DEVICE int64[4], caller-zeroed): [0] max
 * and [1] sum over peers of the rows requested in the first pull, [2] / [3] the same for a pre-projected plan's second
 * pull — what pull_cap / pull_cap_b are calibrated from on warm-up steps (fixed-capacity buckets travel whole). */
int32_t gigl_dist_plan_bucket_fill(gigl_dist_plan* plan, int64_t* acc4);
int32_t gigl_dist_plan_stats(gigl_dist_plan* plan, int64_t* acc);
int32_t gigl_dist_plan_destroy(gigl_dist_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* GIGL_HIP_H */
