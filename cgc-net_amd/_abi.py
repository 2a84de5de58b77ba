"""ctypes prototypes of include/cgc_hip.h (one line per exported symbol; kept in the header's order)."""
import ctypes as C

P, I, F, D, L = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64

ABI_VERSION = 5      # = CGC_ABI_VERSION of include/cgc_hip.h these prototypes were written against (tests compare the two)

PROTOTYPES = {
    'cgc_abi_version': [],
    'cgc_csr_bad_edges_offset': [L, I, I],
    'cgc_csr_build': [P, L, I, I, P, P, P, P, P, P, P, P],
    'cgc_collate': [P, I, I, P, P, P, I, P, P, L, P, P],
    'cgc_farthest_point_sample': [P, P, I, I, P, P, P, P],
    'cgc_farthest_point_sample_table16': [P, P, I, I, P, P, P, P],
    'cgc_radius_knn_ws_ints': [I, I],
    'cgc_radius_knn': [P, P, I, I, F, I, I, P, P, P, P, P],
    'cgc_knn_emit_edges': [P, P, I, I, L, P, P],
    'cgc_edge_renorm': [P, P, I, F, P, P],
    'cgc_csr_transpose_vals': [P, P, P, I, P, P],
    'cgc_csr_invdeg': [P, P, I, P, P],
    'cgc_graph_build': [P, L, I, F, P, P, P, P, P, P, P, P, P, P, P],
    'cgc_graph_build_local': [P, L, I, P, P, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P],
    'cgc_graph_local_max_nodes': [],
    'cgc_spmm': [P, P, P, P, P, P, P, P, I, I, P],
    'cgc_spmm_graphs': [P, P, P, P, P, P, P, P, I, I, I, P, I, I, I, P],
    'cgc_spmm_graphs_ordered': [P, P, P, P, P, P, P, P, I, I, I, P, I, I, I, P, P],
    'cgc_gemm_f32': [I, I, I, I, I, F, P, I, P, I, F, P, I, P, I, L, L, L, P, I, I, P],
    'cgc_gemm_f32_cat': [I, I, I, I, I, F, P, I, P, I, F, P, I, P, I, L, L, L, P, I, I, I, P, P, P, P, P, P, P, P],
    'cgc_gemm_ws_floats': [],
    'cgc_gemm_split_count': [],
    'cgc_gemm_half_count': [],
    'cgc_gemm_half_ws_floats': [],
    'cgc_gemm_half_min_work': [L],
    'cgc_gemm_f32_ws': [I, I, I, I, I, F, P, I, P, I, F, P, I, P, I, L, L, L, P, I, I, P, L, I, P],
    'cgc_gemm_f32_cat_ws': [I, I, I, I, I, F, P, I, P, I, F, P, I, P, I, L, L, L, P, I, I, I, P, P, P, P, P, P, P, P, L, I, P],
    'cgc_gemm_tuning': [I],
    'cgc_reduce_batch_sum': [P, P, I, L, F, P],
    'cgc_reduce_batched': [P, P, I, I, I, F, P],
    'cgc_stats_blocks': [I, I],
    'cgc_stats_ws_floats': [I, I],
    'cgc_l2norm_act_stats': [P, I, I, I, I, P, P, P, P, P],
    'cgc_bn_finalize': [P, I, D, F, F, P, P, P, P, P],
    'cgc_l2norm_act_bn': [P, I, I, I, I, P, P, P, D, F, F, P, P, P, P, P, P],
    'cgc_sage_wide_fwd': [P, I, P, P, I, I, I, I, I, P, I, P, I, P, D, F, F, P, P, P, P, P, P],
    'cgc_bn_running_stats': [P, P, I, F, P, P, P],
    'cgc_bn_act_apply': [P, I, I, I, P, P, P, P, P, I, P],
    'cgc_bn_act_apply2': [P, I, I, I, P, P, P, P, P, I, P, I, P],
    'cgc_bn_bwd_reduce': [P, I, P, I, I, I, P, P, P, P, P],
    'cgc_bn_act_l2_bwd': [P, I, P, P, I, I, I, I, I, P, P, P, P, D, P, P, P, P],
    'cgc_sage_narrow_fwd': [P, I, P, P, I, I, I, I, I, P, P, I, P, D, F, F, P, P, P, P, P, P],
    'cgc_sage_narrow_ws_floats': [I, I, I],
    'cgc_sage_narrow_bwd': [P, I, P, P, I, I, I, I, I, P, P, P, P, D, P, I, I, P, P, P, P, P],
    'cgc_sage_narrow_bwd_ld': [P, I, P, P, I, I, I, I, I, P, P, P, P, D, P, I, I, P, P, I, P, P, P],
    'cgc_colsum': [P, I, I, I, P, P, P],
    'cgc_softmax_fwd': [P, I, I, I, P, P],
    'cgc_softmax_bwd': [P, P, I, I, I, P, P, P, P],
    'cgc_segment_max_fwd': [P, P, I, I, I, P, P, P],
    'cgc_segment_max_bwd': [P, P, I, I, P, P],
    'cgc_segment_max_bwd_full': [P, P, P, I, I, I, P, P],
    'cgc_jk_supported': [I],
    'cgc_jk_matrix_core': [I],
    'cgc_jk_lstm_fwd': [P, I, I, I, P, P, P, P, P, P, P],
    'cgc_jk_lstm_bwd': [P, P, I, I, I, P, P, P, P, P, P, P, P, P, P],
    'cgc_jk_bwd_ws_floats': [I],
    'cgc_jk_lstm_bwd_params': [P, P, I, I, I, P, P, P, P, P, P, P, P, P],
    'cgc_jk_lstm_bwd_flat': [P, P, I, I, I, P, P, P, P, P, P, P, P, P],
    'cgc_jk_param_grad_floats': [I],
    'cgc_jk_unpack_param_grads': [P, I, P, P],
    'cgc_dense_rownorm_fwd': [P, I, I, P, P, P, P],
    'cgc_dense_rownorm_bwd': [P, P, P, P, I, I, P, P],
    'cgc_dense_renorm_fwd': [P, I, I, F, P, P],
    'cgc_dense_renorm_bwd': [P, P, I, I, F, P, P],
    'cgc_adj_prep_fwd': [P, I, I, F, P, P, P, P, P],
    'cgc_adj_prep_bwd': [P, P, P, P, P, P, I, I, F, P, P],
    'cgc_head_fwd': [P, I, I, I, I, I, I, P, P, P, P, P, F, C.c_uint64, P, P, P, P],
    'cgc_head_bwd': [P, I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P],
    'cgc_adam_step': [P, P, I, P, D, D, D, D, D, F, F, P],
    'cgc_timing_create': [I],
    'cgc_timing_attach': [P],
    'cgc_timing_count': [P],
    'cgc_timing_read': [P, I, P, P],
    'cgc_timing_destroy': [P],
    'cgc_cat_cols': [P, I, I, I, P, P, P, I, P],
    'cgc_transpose': [P, I, I, I, P, I, P],
    # step sequencer (csrc/exec.hip; the struct arguments are ctypes Structures of native.py passed by reference)
    'cgc_level_supported': [P],
    'cgc_level_saved_floats': [P],
    'cgc_level_scratch_floats': [P],
    'cgc_level_grad_layout_of': [P, P],
    'cgc_level_fwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    'cgc_level_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
}


def declare(lib):
    """Attach argtypes/restype to every symbol; raises AttributeError if the library lacks one and RuntimeError if the library
    was built from another revision of the header than these prototypes (a stale libcgc_hip.so next to newer Python)."""
    lib.cgc_abi_version.argtypes, lib.cgc_abi_version.restype = [], C.c_int
    got = lib.cgc_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError('libcgc_hip.so reports ABI version %d, the bindings were written for %d: rebuild it (make -C cgc-net_amd/csrc)'
                           % (got, ABI_VERSION))
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        if name == 'cgc_timing_create':
            fn.restype = P          # a handle, not a status
            continue
        fn.restype = C.c_int64 if name.endswith(('_ws_ints', '_ws_floats', '_offset', '_grad_floats', '_saved_floats', '_scratch_floats', '_split_count', '_half_count', '_min_work')) else C.c_int
