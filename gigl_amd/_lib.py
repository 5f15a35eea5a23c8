"""ctypes binding of libgigl_hip.so (C ABI: include/gigl_hip.h).

There is deliberately NO fallback: if the library is missing, cannot be loaded, or there is no GPU,
every compute entry point raises — a silent CPU path would void the parity claims.
"""
from __future__ import annotations

import ctypes as C
import os

GIGL_INVALID = 0xFFFFFFFF
GIGL_MAX_HOPS = 4
GIGL_FAST_FANOUT = 64
GIGL_MAX_FANOUT = 1024
GIGL_META_LEN = 16
GIGL_META_N_NODES, GIGL_META_N_EDGES, GIGL_META_LEVEL0, GIGL_META_OVERFLOW = 0, 1, 2, 8
LOC_HOST, LOC_DEVICE = 0, 1
DTYPE_F32, DTYPE_F16 = 0, 1
MODE_SPARK_HASH, MODE_FAST, MODE_REPLACE = 0, 1, 2
AGGR = {"mean": 0, "sum": 1, "add": 1, "max": 2}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgigl_hip.so")

# every symbol include/gigl_hip.h declares (tests/test_abi.py checks the .so exports exactly these)
SYMBOLS = [
    "gigl_version", "gigl_ctx_create", "gigl_ctx_destroy", "gigl_last_error", "gigl_ctx_set_stream",
    "gigl_ctx_synchronize", "gigl_ctx_reserve", "gigl_memcpy", "gigl_graph_load_csc", "gigl_graph_build_from_coo",
    "gigl_graph_info", "gigl_graph_device_ptrs", "gigl_graph_destroy", "gigl_features_load",
    "gigl_features_device_ptr", "gigl_features_destroy", "gigl_sample_khop", "gigl_sample_positives",
    "gigl_union_capacity", "gigl_union_build", "gigl_gather_mean", "gigl_linear",
    "gigl_linear_batched", "gigl_retrieval_loss_batched",
    "gigl_profile_enable", "gigl_profile_read", "gigl_profile_reset",
    "gigl_sage_plan_create", "gigl_sage_plan_set_weights", "gigl_sage_plan_buffers", "gigl_sage_plan_run",
    "gigl_sage_plan_destroy", "gigl_gather_mean_backward", "gigl_gather_mean_backward_transposed", "gigl_transposed_rows_words", "gigl_transposed_rows_build",
    "gigl_gather_mean_backward_lists", "gigl_linear_grouped", "gigl_expand_frontier", "gigl_gcn_aggregate",
    "gigl_gat_aggregate", "gigl_gather_rows", "gigl_sage_plan_use_graph", "gigl_sage_plan_flush_profile",
    "gigl_union_build_groups", "gigl_sage_plan_set_groups", "gigl_records_capacity", "gigl_records_encode",
    "gigl_tfrecord_index", "gigl_tfexample_decode", "gigl_collate_records", "gigl_collated_info", "gigl_collated_copy",
    "gigl_collated_destroy", "gigl_gather_reduce", "gigl_gather_reduce_backward",
    "gigl_frontier_bucket", "gigl_frontier_scatter", "gigl_avro_embeddings_layout", "gigl_avro_embeddings_encode",
    "gigl_edge_ids", "gigl_union_edge_ids", "gigl_gat_aggregate_edge", "gigl_collated_edge_attr", "gigl_sample_out_neighbors", "gigl_rows_dedup", "gigl_gat_aggregate_backward", "gigl_gat_backward_epilogue", "gigl_gat_input_aggregate", "gigl_gat_input_aggregate_backward",
    "gigl_cms_add", "gigl_cms_estimate", "gigl_gine_aggregate", "gigl_gine_aggregate_backward", "gigl_gat_input_layer", "gigl_gat_input_layer_scratch", "gigl_gat_input_layer_fused",
    "gigl_gat_input_layer_fused_scratch", "gigl_gat_plan_create", "gigl_gat_plan_set_weights", "gigl_sage_plan_set_aggr", "gigl_gatv2_aggregate", "gigl_gatv2_aggregate_backward", "gigl_gatv2_aggregate_edge",
    "gigl_gatv2_aggregate_edge_backward", "gigl_transformer_aggregate_edge", "gigl_transformer_aggregate_edge_backward",
    "gigl_sage_plan_stats", "gigl_retrieval_loss", "gigl_retrieval_loss_backward",
    "gigl_comm_unique_id", "gigl_dist_init", "gigl_dist_init_local", "gigl_dist_init_callback", "gigl_comm_info",
    "gigl_comm_all_to_all", "gigl_comm_flush_local", "gigl_comm_traffic", "gigl_comm_set_fixed_blocks", "gigl_ctx_set_wide_workspaces", "gigl_sage_train_plan_adopt", "gigl_sage_train_plan_moments", "gigl_nablp_train_plan_moments", "gigl_sage_train_plan_resume", "gigl_nablp_train_plan_adopt", "gigl_comm_destroy", "gigl_dist_plan_batch_features", "gigl_dist_plan_batch_graph", "gigl_dist_plan_create",
    "gigl_dist_plan_set_weights", "gigl_dist_plan_phases", "gigl_dist_plan_phase", "gigl_dist_plan_run",
    "gigl_dist_plan_run_local", "gigl_dist_plan_run_interleaved", "gigl_dist_plan_buffers", "gigl_dist_plan_stats", "gigl_dist_plan_destroy",
    "gigl_dist_plan_set_hot_rows", "gigl_dist_plan_set_peer_tables", "gigl_dist_plan_set_peer_graphs", "gigl_sample_khop_peer", "gigl_ipc_export", "gigl_ipc_open", "gigl_ipc_close",
    "gigl_split_hash_slots", "gigl_hgt_aggregate", "gigl_simplehgn_alpha", "gigl_weighted_aggregate",
    "gigl_collate_typed_records", "gigl_collated_typed_info", "gigl_collated_typed_nodes", "gigl_collated_typed_edges",
    "gigl_collated_typed_samples", "gigl_collated_typed_destroy", "gigl_typed_records_capacity",
    "gigl_typed_records_encode", "gigl_typed_samples_encode", "gigl_hgt_aggregate_backward", "gigl_weighted_aggregate_backward",
    "gigl_graph_build_shard_from_coo", "gigl_json_rows_capacity", "gigl_json_rows_format",
    "gigl_sage_project_features", "gigl_sage_plan_set_projected_input",
    "gigl_dist_gat_plan_create", "gigl_dist_gat_plan_set_weights", "gigl_dist_plan_bucket_fill",
    "gigl_linear_weight_grad", "gigl_features_row_crc",
    "gigl_typed_plan_create", "gigl_typed_plan_run", "gigl_typed_plan_buffers", "gigl_typed_plan_destroy",
    "gigl_typed_plan_merged_csr", "gigl_sage_plan_half_split", "gigl_sage_plan_fused_layers", "gigl_sort_distinct_u64", "gigl_sort_distinct_u32",
    "gigl_typed_plan_merged_csr_ex", "gigl_hgt_aggregate_act", "gigl_dist_plan_set_aggr", "gigl_typed_plan_run_nodes", "gigl_typed_plan_run_edges", "gigl_typed_plan_clone", "gigl_hgt_infer_create", "gigl_hgt_infer_run",
    "gigl_hgt_infer_set_model", "gigl_hgt_infer_use_graph", "gigl_hgt_infer_destroy",
    "gigl_sage_plan_run_part", "gigl_sage_plan_set_graph_stream", "gigl_sage_plan_overflow_add",
    "gigl_sage_train_plan_create", "gigl_sage_train_plan_step", "gigl_sage_train_plan_step2", "gigl_sage_train_plan_loss", "gigl_sage_train_plan_destroy",
    "gigl_nablp_train_plan_create", "gigl_nablp_train_plan_step", "gigl_nablp_train_plan_step2", "gigl_gat_nablp_train_plan_create", "gigl_gat_nablp_train_plan_grads", "gigl_nablp_train_plan_loss", "gigl_nablp_train_plan_destroy", "gigl_nablp_train_plan_grads",
]

KERNEL_IDS = {
    "expand": 0, "expand_heavy": 1, "find_heavy": 2, "union_insert": 3, "union_relax": 4, "union_nodes": 5,
    "union_edge_sort": 6, "union_csr": 7, "gather_mean": 8, "linear": 9, "gather_bwd": 10,
    "dist_prep": 11, "dist_serve": 12,
}


class GiglTree(C.Structure):
    _fields_ = [
        ("hops", C.c_int32),
        ("b", C.c_int32),
        ("fanouts", C.c_int32 * GIGL_MAX_HOPS),
        ("nbr", C.c_void_p * GIGL_MAX_HOPS),
        ("cnt", C.c_void_p * GIGL_MAX_HOPS),
    ]


class GiglUnion(C.Structure):
    _fields_ = [
        ("meta", C.c_void_p),
        ("nodes", C.c_void_p),
        ("rowptr", C.c_void_p),
        ("rowend", C.c_void_p),
        ("col", C.c_void_p),
        ("root_local", C.c_void_p),
        ("cap_nodes", C.c_int64),
        ("cap_edges", C.c_int64),
    ]


class GiglRecordOpts(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("trees_per_record", C.c_int32),
        ("condensed_node_type", C.c_int32),
        ("condensed_edge_type", C.c_int32),
        ("tfrecord_frame", C.c_int32),
        ("emit", C.c_void_p),
        ("suffix", C.c_void_p),
        ("suffix_off", C.c_void_p),
        ("graph", C.c_void_p),
        ("edge_feat", C.c_void_p),
        ("n_neg_trees", C.c_int32),
        ("pos_edges_graph", C.c_void_p),
        ("pos_edge_feat", C.c_void_p),
        ("neg_edges_graph", C.c_void_p),
        ("neg_edge_feat", C.c_void_p),
    ]


GIGL_DAG_MAX_OPS, GIGL_DAG_MAX_PARENTS = 16, 8


class GiglDagOp(C.Structure):
    _fields_ = [("graph", C.c_void_p), ("fanout", C.c_int32), ("n_parents", C.c_int32),
                ("parents", C.c_int32 * GIGL_DAG_MAX_PARENTS), ("hash_add", C.c_int32),
                ("frontier_node_type", C.c_int32), ("result_node_type", C.c_int32), ("edge_slot", C.c_int32),
                ("outgoing", C.c_int32)]


class GiglTypedPlanOut(C.Structure):
    _fields_ = [("n_nodes", C.c_void_p), ("n_edges", C.c_void_p), ("root_index", C.c_void_p),
                ("nodes", C.c_void_p * 16), ("nodes_cap", C.c_int64 * 16),
                ("edges", C.c_void_p * 32), ("edges_cap", C.c_int64 * 32),
                ("op_frontier", C.c_void_p * GIGL_DAG_MAX_OPS), ("op_nbr", C.c_void_p * GIGL_DAG_MAX_OPS),
                ("op_cnt", C.c_void_p * GIGL_DAG_MAX_OPS), ("op_width", C.c_int32 * GIGL_DAG_MAX_OPS)]


class GiglTypedCsrOut(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("etype", C.c_void_p), ("counts", C.c_void_p),
                ("root_rowptr", C.c_void_p), ("root_col", C.c_void_p), ("root_etype", C.c_void_p),
                ("edges_cap", C.c_int64), ("rows_cap", C.c_int64)]


GIGL_HGT_MAX_LAYERS = 4


class GiglHgtLayerWeights(C.Structure):
    _fields_ = [("wq", C.c_void_p * 16), ("bq", C.c_void_p * 16), ("wout", C.c_void_p * 16), ("bout", C.c_void_p * 16),
                ("keep", C.c_void_p * 16), ("wk", C.c_void_p * 32), ("bk", C.c_void_p * 32), ("wv", C.c_void_p * 32),
                ("bv", C.c_void_p * 32), ("p_rel", C.c_void_p)]


class GiglHgtModel(C.Structure):
    _fields_ = [("n_types", C.c_int32), ("n_slots", C.c_int32), ("n_layers", C.c_int32), ("heads", C.c_int32),
                ("hid", C.c_int32), ("out_dim", C.c_int32), ("l2_normalize", C.c_int32), ("root_type", C.c_int32),
                ("type_order", C.c_int32 * 16), ("slot_order", C.c_int32 * 32), ("slot_etype", C.c_int32 * 32),
                ("feat_dim", C.c_int32 * 16), ("feat", C.c_void_p * 16), ("w_in", C.c_void_p * 16),
                ("b_in", C.c_void_p * 16), ("layer", GiglHgtLayerWeights * GIGL_HGT_MAX_LAYERS),
                ("w_final", C.c_void_p), ("b_final", C.c_void_p)]


class GiglTypedOp(C.Structure):
    _fields_ = [("frontier", C.c_void_p), ("nbr", C.c_void_p), ("w", C.c_int32), ("f", C.c_int32),
                ("condensed_edge_type", C.c_int32), ("result_node_type", C.c_int32), ("outgoing", C.c_int32)]


class GiglTypedFeat(C.Structure):
    _fields_ = [("x", C.c_void_p), ("d", C.c_int32), ("n", C.c_int64)]


class GiglTypedEdgeFeat(C.Structure):
    _fields_ = [("by_source", C.c_void_p), ("feat", C.c_void_p), ("d", C.c_int32)]


REC_ROOTED_NODE_NEIGHBORHOOD, REC_NODE_ANCHOR_LINK_PRED = 0, 1
STATS = {"sampled": 0, "aggregated": 1, "union_edges": 2, "union_nodes": 3, "expand_bytes": 4, "agg_layer0": 5,
         "rows_layer0": 9, "overflow": 13, "pulled_rows": 14, "pull_bucket_max": 15}
COMM_RCCL, COMM_LOCAL, COMM_CALLBACK = 0, 1, 2
COMM_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class GiglDistPlanOpts(C.Structure):
    _fields_ = [
        ("group_roots", C.c_int32),
        ("project_on_owner", C.c_int32),
        ("pull_cap", C.c_int64),
        ("hop_slack", C.c_float),
        ("max_window_end", C.c_int64),
        ("projected", C.c_void_p),
        ("pull_cap_b", C.c_int64),
        ("staged", C.c_int32),
        ("peer_direct", C.c_int32),
        ("peer_sample", C.c_int32),
    ]
STATS_LEN = 16
STATS_SAMPLED, STATS_AGGREGATED = 0, 1  # GIGL_STATS_* slots of gigl_sage_plan_stats
COL_I64, COL_F32 = 0, 1


class GiglColumn(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("kind", C.c_int32),
        ("width", C.c_int32),
        ("out", C.c_void_p),
        ("counts", C.c_void_p),
    ]


class GiglError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libgigl_hip error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gigl_amd/csrc`. gigl_amd has no CPU fallback.")
    # torch first: the library must share the HIP runtime (libamdhip64) torch already mapped, or the
    # two would each own a runtime and device pointers could not be exchanged
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    P = C.POINTER
    lib.gigl_version.restype = i32
    lib.gigl_last_error.restype = C.c_char_p
    lib.gigl_last_error.argtypes = [vp]
    sig = {
        "gigl_ctx_create": [i32, P(vp)],
        "gigl_ctx_destroy": [vp],
        "gigl_ctx_set_stream": [vp, vp],
        "gigl_ctx_synchronize": [vp],
        "gigl_ctx_reserve": [vp, i64],
        "gigl_memcpy": [vp, vp, i32, vp, i32, i64],
        "gigl_graph_load_csc": [vp, i64, i64, vp, vp, i32, P(vp)],
        "gigl_graph_build_from_coo": [vp, i64, i64, vp, vp, i32, i32, P(vp)],
        "gigl_graph_build_shard_from_coo": [vp, i64, i32, i32, i64, vp, vp, i32, i32, P(vp)],
        "gigl_json_rows_format": [vp, vp, i64, vp, i64, i32, vp, i64, P(i64)],
        "gigl_graph_info": [vp, P(i64), P(i64)],
        "gigl_graph_device_ptrs": [vp, P(vp), P(vp)],
        "gigl_graph_destroy": [vp],
        "gigl_features_load": [vp, i64, i32, i32, vp, i32, P(vp)],
        "gigl_features_device_ptr": [vp, P(vp), P(i64), P(i32), P(i32)],
        "gigl_features_row_crc": [vp, vp, P(vp)],
        "gigl_typed_plan_create": [vp, P(GiglDagOp), i32, i32, i32, i32, i32, P(vp)],
        "gigl_typed_plan_run": [vp, vp, i32],
        "gigl_typed_plan_buffers": [vp, P(GiglTypedPlanOut)],
        "gigl_typed_plan_destroy": [vp],
        "gigl_typed_plan_merged_csr_ex": [vp, i32, P(i32), i32, P(i32), P(i32), i32, i32, P(GiglTypedCsrOut)],
        "gigl_hgt_aggregate_act": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, i32, vp],
        "gigl_dist_plan_set_aggr": [vp, i32],
        "gigl_typed_plan_run_nodes": [vp, vp, i32],
        "gigl_typed_plan_run_edges": [vp, i32],
        "gigl_typed_plan_clone": [vp, vp, P(vp)],
        "gigl_hgt_infer_create": [vp, vp, i32, P(GiglHgtModel), P(i32), P(i32), P(vp)],
        "gigl_hgt_infer_run": [vp, vp, i32, vp, i32, vp],
        "gigl_hgt_infer_set_model": [vp, P(GiglHgtModel)],
        "gigl_hgt_infer_use_graph": [vp, i32],
        "gigl_hgt_infer_destroy": [vp],
        "gigl_sort_distinct_u64": [vp, vp, i64, i32, i32, C.c_uint64, vp, vp],
        "gigl_sort_distinct_u32": [vp, vp, i64, i32, C.c_uint32, vp, vp],
        "gigl_typed_plan_merged_csr": [vp, i32, P(i32), i32, P(i32), P(i32), i32, P(GiglTypedCsrOut)],
        "gigl_features_destroy": [vp],
        "gigl_sample_khop": [vp, vp, vp, i32, P(i32), i32, i32, i32, P(GiglTree)],
        "gigl_sample_positives": [vp, vp, vp, i32, i32, i32, i32, vp, vp],
        "gigl_sample_out_neighbors": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
        "gigl_union_capacity": [i32, P(i32), i32, P(i64), P(i64)],
        "gigl_union_build": [vp, vp, P(GiglTree), P(GiglUnion)],
        "gigl_union_build_groups": [vp, vp, P(GiglTree), i32, P(GiglUnion)],
        "gigl_sage_plan_set_groups": [vp, i32],
        "gigl_gather_mean": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i64, vp],
        "gigl_linear": [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp],
        "gigl_linear_batched": [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i64, i64, i32, vp],
        "gigl_retrieval_loss_batched": [vp, vp, i64, i64, i32, i32, i32, C.c_float, vp, vp, vp, vp, vp, vp],
        "gigl_profile_enable": [vp, C.c_uint32, i32],
        "gigl_profile_read": [vp, i32, P(C.c_double), P(i64)],
        "gigl_profile_reset": [vp],
        "gigl_sage_plan_create": [vp, vp, vp, i32, P(i32), i32, P(i32), P(vp), P(vp), i32, P(vp)],
        "gigl_sage_plan_set_weights": [vp, P(vp), P(vp)],
        "gigl_sage_plan_buffers": [vp, P(GiglTree), P(GiglUnion)],
        "gigl_sage_plan_run": [vp, vp, i32, i32, vp],
        "gigl_sage_plan_run_part": [vp, vp, i32, i32, vp, i32],
        "gigl_sage_plan_destroy": [vp],
        "gigl_sage_plan_use_graph": [vp, i32],
        "gigl_sage_plan_flush_profile": [vp],
        "gigl_sage_plan_stats": [vp, vp, vp],
        "gigl_split_hash_slots": [vp, vp, vp, i64, i32, i32, vp],
        "gigl_hgt_aggregate": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp],
        "gigl_simplehgn_alpha": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, C.c_float, vp, vp],
        "gigl_weighted_aggregate": [vp, vp, vp, i32, i32, vp, vp, i64, vp],
        "gigl_hgt_aggregate_backward": [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp],
        "gigl_weighted_aggregate_backward": [vp, vp, vp, i32, i32, vp, vp, i64, vp, vp, vp],
        "gigl_comm_unique_id": [vp],
        "gigl_dist_init": [vp, i32, i32, vp, P(vp)],
        "gigl_dist_init_local": [P(vp), i32, P(vp)],
        "gigl_dist_init_callback": [vp, i32, i32, EXCHANGE_FN, vp, P(vp)],
        "gigl_comm_info": [vp, P(i32), P(i32), P(i32)],
        "gigl_comm_all_to_all": [vp, vp, vp, i64],
        "gigl_comm_flush_local": [vp],
        "gigl_comm_traffic": [vp, vp, vp],
        "gigl_comm_set_fixed_blocks": [vp, i32],
        "gigl_ctx_set_wide_workspaces": [vp, i32],
        "gigl_sage_train_plan_adopt": [vp, vp],
        "gigl_sage_train_plan_moments": [vp, i32, vp, vp, vp, vp],
        "gigl_nablp_train_plan_moments": [vp, i32, vp, vp],
        "gigl_sage_train_plan_resume": [vp],
        "gigl_nablp_train_plan_adopt": [vp, vp],
        "gigl_dist_plan_batch_features": [vp, vp],
        "gigl_dist_plan_batch_graph": [vp, vp, vp, vp, vp, vp, vp],
        "gigl_comm_destroy": [vp],
        "gigl_dist_plan_create": [vp, vp, vp, i32, P(i32), i32, P(i32), P(vp), P(vp), i32, P(GiglDistPlanOpts), P(vp)],
        "gigl_dist_plan_set_weights": [vp, P(vp), P(vp)],
        "gigl_dist_gat_plan_create": [vp, vp, vp, i32, P(i32), i32, P(i32), P(i32), vp, vp, vp, vp, C.c_float, i32,
                                      P(GiglDistPlanOpts), P(vp)],
        "gigl_dist_gat_plan_set_weights": [vp, vp, vp, vp, vp],
        "gigl_dist_plan_bucket_fill": [vp, vp],
        "gigl_linear_weight_grad": [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp],
        "gigl_dist_plan_phases": [vp, P(i32)],
        "gigl_dist_plan_phase": [vp, i32, vp, i32, vp],
        "gigl_dist_plan_run": [vp, vp, i32, vp],
        "gigl_dist_plan_run_local": [P(vp), i32, P(vp), i32, P(vp)],
        "gigl_dist_plan_run_interleaved": [P(vp), i32, P(vp), i32, P(vp)],
        "gigl_dist_plan_buffers": [vp, P(GiglTree), P(GiglUnion)],
        "gigl_dist_plan_stats": [vp, vp],
        "gigl_dist_plan_set_hot_rows": [vp, vp, i64, vp],
        "gigl_dist_plan_set_peer_tables": [vp, P(vp)],
        "gigl_dist_plan_set_peer_graphs": [vp, P(vp), P(vp)],
        "gigl_sample_khop_peer": [vp, vp, vp, i32, i64, i64, vp, i32, P(i32), i32, i32, P(GiglTree)],
        "gigl_ipc_export": [vp, vp, vp, P(i64)],
        "gigl_ipc_open": [vp, vp, i64, P(vp), P(vp)],
        "gigl_ipc_close": [vp, vp],
        "gigl_dist_plan_destroy": [vp],
        "gigl_retrieval_loss": [vp, vp, i64, i32, i32, C.c_float, vp, vp, vp, vp, vp, vp, vp],
        "gigl_retrieval_loss_backward": [vp, vp, i64, i32, i32, C.c_float, vp, vp, vp, vp, vp, vp],
        "gigl_gather_mean_backward": [vp, vp, i32, vp, vp, vp, vp, i64, vp],
        "gigl_gather_mean_backward_transposed": [vp, vp, i32, vp, vp, vp, vp, i64, vp, i64, i64, i32, vp],
        "gigl_transposed_rows_build": [vp, vp, vp, vp, vp, i64, vp, i64, i64, vp],
        "gigl_gather_mean_backward_lists": [vp, vp, i32, vp, vp, vp, vp, i64, vp, i32, vp],
        "gigl_linear_grouped": [vp, vp, i32, i64, i32, i32, i32],
        "gigl_expand_frontier": [vp, vp, vp, vp, i64, i32, i32, i32, i64, vp, vp],
        "gigl_gather_rows": [vp, vp, i32, i32, vp, vp, i64, vp],
        "gigl_frontier_bucket": [vp, vp, vp, i64, i32, i64, vp, vp, vp],
        "gigl_frontier_scatter": [vp, vp, vp, vp, vp, i64, i32, i64, i32, vp, vp, vp],
        "gigl_gather_reduce": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i64, i32, vp],
        "gigl_gather_reduce_backward": [vp, vp, i32, vp, vp, vp, vp, i64, i32, vp, vp],
        "gigl_tfrecord_index": [vp, i64, i32, i64, vp, vp, P(i64)],
        "gigl_collate_records": [vp, vp, vp, i64, i32, i32, P(vp), C.c_char_p, i32],
        "gigl_collated_info": [vp, P(i64), P(i64), P(i32), P(i64), P(i64)],
        "gigl_collated_copy": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "gigl_collated_destroy": [vp],
        "gigl_collated_edge_attr": [vp, P(i32), vp],
        "gigl_collate_typed_records": [vp, vp, vp, i64, i32, i32, i32, vp, vp, i32, P(vp), C.c_char_p, i32],
        "gigl_collated_typed_info": [vp, vp, vp, vp, vp, P(i64), P(i64)],
        "gigl_collated_typed_nodes": [vp, i32, vp, vp],
        "gigl_collated_typed_edges": [vp, i32, vp, vp],
        "gigl_collated_typed_samples": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "gigl_collated_typed_destroy": [vp],
        "gigl_typed_records_capacity": [P(GiglTypedOp), i32, P(GiglTypedFeat), i32, P(GiglTypedEdgeFeat), i32, i64, i32,
                                        P(i64)],
        "gigl_typed_records_encode": [vp, vp, i32, P(GiglTypedOp), i32, P(GiglTypedFeat), i32, P(GiglTypedEdgeFeat), i32,
                                      i64, i32, vp, i64, vp, vp],
        "gigl_typed_samples_encode": [vp, i32, vp, i32, P(GiglTypedOp), i32, P(GiglTypedFeat), i32, P(GiglTypedEdgeFeat),
                                      i32, i64, i32, vp, i64, vp, vp],
        "gigl_tfexample_decode": [vp, vp, vp, i64, P(GiglColumn), i32, i32, P(i64)],
        "gigl_gat_aggregate_edge": [vp, vp, vp, vp, i32, i32, C.c_float, i32, vp, vp, vp, vp, i64, vp, i64, vp, i32, vp, i32,
                                    i64, vp, vp, vp, vp],
        "gigl_gat_input_aggregate": [vp, vp, i32, i32, vp, vp, i32, C.c_float, vp, vp, vp, vp, i64, vp],
        "gigl_gat_input_aggregate_backward": [vp, vp, i32, i32, vp, vp, i32, C.c_float, vp, vp, vp, vp, i64, vp, vp, vp],
        "gigl_gat_backward_epilogue": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp],
        "gigl_gat_aggregate_backward": [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i32,
                                        i64, vp, vp, vp, vp, vp, vp, vp, vp],
        "gigl_rows_dedup": [vp, vp, i64, i32],
        "gigl_edge_ids": [vp, vp, vp, vp, i64, vp],
        "gigl_union_edge_ids": [vp, vp, P(GiglUnion), vp],
        "gigl_avro_embeddings_layout": [i64, i32, i32, P(i32), P(i64), P(i64)],
        "gigl_avro_embeddings_encode": [vp, vp, vp, i64, i64, i32, C.c_char_p, i32, C.c_char_p, vp, i64, vp, vp, vp],
        "gigl_records_capacity": [P(i32), i32, i32, P(GiglRecordOpts), i64, i64, P(i64)],
        "gigl_records_encode": [vp, vp, P(GiglTree), vp, P(GiglRecordOpts), i64, vp, i64, vp, vp],
        "gigl_gcn_aggregate": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i64, vp, i64, vp, i32, vp, vp],
        "gigl_gat_aggregate": [vp, vp, vp, vp, i32, i32, C.c_float, i32, vp, vp, vp, vp, i64, vp, i64, vp, i32, vp, vp],
        "gigl_gine_aggregate": [vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, vp],
        "gigl_gine_aggregate_backward": [vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp],
        "gigl_cms_add": [vp, vp, i32, i32, vp, i64],
        "gigl_cms_estimate": [vp, vp, i32, i32, vp, i64, vp],
        "gigl_gatv2_aggregate": [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, i32, vp],
        "gigl_gatv2_aggregate_backward": [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp],
        "gigl_transformer_aggregate_edge": [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp],
        "gigl_transformer_aggregate_edge_backward": [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                                     vp],
        "gigl_gatv2_aggregate_edge": [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, i32, vp, vp],
        "gigl_gatv2_aggregate_edge_backward": [vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, vp, vp, vp,
                                               vp, vp, vp],
        "gigl_gat_plan_create": [vp, vp, vp, i32, P(i32), i32, P(i32), P(i32), vp, vp, vp, vp, C.c_float, i32, vp],
        "gigl_gat_plan_set_weights": [vp, vp, vp, vp, vp],
        "gigl_sage_plan_set_aggr": [vp, i32],
        "gigl_sage_project_features": [vp, vp, vp, i32, vp],
        "gigl_sage_plan_set_projected_input": [vp, vp],
        "gigl_sage_plan_half_split": [vp],
        "gigl_sage_plan_fused_layers": [vp],
        "gigl_sage_plan_overflow_add": [vp, vp],
        "gigl_sage_plan_set_graph_stream": [vp, vp, i32],
        "gigl_sage_train_plan_create": [vp, vp, vp, i32, P(i32), i32, P(i32), vp, vp, i32, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, vp],
        "gigl_sage_train_plan_step": [vp, vp, vp, i32, vp, i32, i32, vp],
        "gigl_sage_train_plan_step2": [vp, vp, vp, i32, vp, vp, i32, i32, vp],
        "gigl_sage_train_plan_destroy": [vp],
        "gigl_nablp_train_plan_create": [vp, vp, vp, i32, i32, i32, P(i32), i32, P(i32), vp, vp, i32, i32, C.c_float, i32,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp],
        "gigl_gat_nablp_train_plan_create": [vp, vp, vp, i32, i32, i32, P(i32), i32, P(i32), P(i32), vp, vp, vp, vp, C.c_float,
                                             i32, C.c_float, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp],
        "gigl_gat_nablp_train_plan_grads": [vp, i32, vp, vp, vp, vp],
        "gigl_nablp_train_plan_step": [vp, vp, vp, vp, i32, i32, vp],
        "gigl_nablp_train_plan_step2": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
        "gigl_nablp_train_plan_destroy": [vp],
        "gigl_nablp_train_plan_grads": [vp, i32, vp, vp],
        "gigl_gat_input_layer_fused": [vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, i64, vp, i32,
                                       vp, vp],
        "gigl_gat_input_layer": [vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp, i64, vp, i64, vp, i64,
                                 vp, i32, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = args
    lib.gigl_gat_input_layer_scratch.argtypes = [i32, i32, i64, i64, i64]
    lib.gigl_gat_input_layer_scratch.restype = i64
    lib.gigl_gat_input_layer_fused_scratch.argtypes = [i32, i32, i64]
    lib.gigl_gat_input_layer_fused_scratch.restype = i64
    lib.gigl_transposed_rows_words.argtypes = [i64, i64]
    lib.gigl_transposed_rows_words.restype = i64
    lib.gigl_json_rows_capacity.argtypes = [i64, i32]
    lib.gigl_json_rows_capacity.restype = i64
    lib.gigl_sage_train_plan_loss.argtypes = [vp]
    lib.gigl_sage_train_plan_loss.restype = vp
    lib.gigl_nablp_train_plan_loss.argtypes = [vp]
    lib.gigl_nablp_train_plan_loss.restype = vp
    _lib = lib
    return lib


def check(rc: int, ctx=None) -> None:
    if rc != 0:
        msg = ""
        if ctx:
            raw = load().gigl_last_error(ctx)
            msg = raw.decode("utf-8", "replace") if raw else ""
        raise GiglError(rc, msg or {-5: "no HIP device visible", -1: "invalid argument"}.get(rc, "failed"))
