"""ctypes binding of the C ABI declared in ``include/spk_hip.h`` (libspk_hip.so).

The library is built in-tree by ``schnetpack_amd/csrc/build.py`` (hipcc, gfx950).  There is no
CPU fallback: if the shared library is missing, or a tensor is not a contiguous fp32 tensor on a
ROCm device, the call raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libspk_hip.so")

SPK_ACT_NONE, SPK_ACT_SSP, SPK_ACT_SILU = 0, 1, 2
SPK_RBF_GAUSSIAN, SPK_RBF_BESSEL = 0, 1
VARIANT_AUTO, VARIANT_SIMPLE, VARIANT_MFMA, VARIANT_MFMA_DIRECTED, VARIANT_MFMA_PAIR, VARIANT_MFMA_MOL = 0, 1, 2, 3, 4, 5

c_f = ctypes.c_void_p  # device pointers travel as void*
c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32


class SpkHipError(RuntimeError):
    pass


class RadialT(ctypes.Structure):
    _fields_ = [("kind", c_i32), ("n_rbf", c_i32), ("p0", c_f), ("p1", c_f),
                ("cutoff", ctypes.c_float)]


class GraphT(ctypes.Structure):
    _fields_ = [("n_atoms", c_i64), ("n_edges", c_i64), ("idx_i", c_f), ("idx_j", c_f),
                ("rowptr", c_f), ("sorted", c_i32), ("symmetric", c_i32), ("rev", c_f), ("half", c_f),
                ("n_half", c_i64), ("grp_atom0", c_f), ("grp_pair0", c_f), ("grp_tile0", c_f),
                ("n_groups", c_i32), ("max_group_atoms", c_i32), ("n_tiles_grouped", c_i64),
                ("filter_pairs", c_i32), ("reserved0", c_i32), ("n_half_dev", c_f), ("edge_pair", c_f),
                ("max_group_pairs", c_i32), ("reserved1", c_i32), ("blocks", c_f), ("transposed", c_f)]


class TransposedT(ctypes.Structure):
    """``spk_transposed_t``: the list sorted by neighbour (asymmetric lists: transposed sums as row passes instead of atomics)."""
    _fields_ = [("idx_i", c_f), ("idx_j", c_f), ("rowptr", c_f), ("perm", c_f), ("r_perm", c_f)]


class BlocksT(ctypes.Structure):
    """``spk_blocks_t``: block plan of a large sorted list for the PaiNN message kernels of the box regime (spk_painn_blk.hip)."""
    _fields_ = [("n_groups", c_i32), ("max_unique", c_i32), ("n_tiles", c_i32), ("cap", c_i32), ("ks", c_i32), ("ok", c_i32),
                ("n_blocks", c_i32), ("reserved", c_i32), ("blk_desc", c_f), ("sub_n", c_f), ("sub_u", c_f), ("uniq", c_f), ("jl", c_f), ("atom_tile0", c_f), ("tile_info", c_f),
                ("apack", c_f), ("adpack", c_f), ("rec", c_f), ("part", c_f)]


class SchnetLayerT(ctypes.Structure):
    _fields_ = [(n, c_f) for n in ("in2f_w", "fn_w1", "fn_b1", "fn_w2", "fn_b2", "f2out_w1",
                                   "f2out_b1", "f2out_w2", "f2out_b2", "in2f_wT", "f2out_w1T", "f2out_w2T")]


class SchnetT(ctypes.Structure):
    _fields_ = [("n_atom_basis", c_i32), ("n_filters", c_i32), ("n_interactions", c_i32),
                ("reserved", c_i32), ("layers", ctypes.POINTER(SchnetLayerT)), ("wpack", c_f)]


class HeadT(ctypes.Structure):
    _fields_ = [("w1", c_f), ("w1t", c_f), ("b1", c_f), ("w2", c_f), ("b2", c_f), ("n_hidden", c_i32), ("act", c_i32)]


class ChainLayerT(ctypes.Structure):
    _fields_ = [("w", c_f), ("b", c_f), ("res", c_f), ("out", c_f), ("pre_out", c_f), ("post_pre", c_f),
                ("k", c_i32), ("n_out", c_i32), ("act", c_i32), ("trans", c_i32), ("post_act", c_i32)]


class ChainT(ctypes.Structure):
    _fields_ = [("n_layers", c_i32), ("in_act", c_i32), ("m", c_i64), ("inp", c_f), ("in_pre", c_f),
                ("zero_ptr", c_f), ("zero_count", c_i64), ("tmp", c_f * 2), ("layers", ChainLayerT * 3)]


class DenseDualT(ctypes.Structure):
    """spk_dense_dual_t: a Dense layer on a (value, tangent) pair of activations (include/spk_hip.h)."""
    _fields_ = [(n, c_f) for n in ("x_v", "x_t", "w", "b", "res_v", "res_t", "pre_v_in", "pre_t_in", "fc", "fc1", "y_v", "y_t", "pre_v", "pre_t")] + \
               [("m", c_i64), ("k_in", c_i32), ("n_out", c_i32), ("act", c_i32), ("mode", c_i32), ("trans", c_i32)]


class PainnLayerT(ctypes.Structure):
    _fields_ = [(n, c_f) for n in ("ctx_w1", "ctx_b1", "ctx_w2", "ctx_b2", "filt_w", "filt_b",
                                   "mix_w", "ictx_w1", "ictx_b1", "ictx_w2", "ictx_b2",
                                   "ctx_w1T", "ctx_w2T", "mix_wT", "ictx_w1T", "ictx_w2T")]


class FmBatchT(ctypes.Structure):
    _fields_ = [("n_atoms", c_i64), ("n_edges", c_i64), ("n_mol", c_i64), ("Z", c_f), ("idx_i", c_f), ("idx_j", c_f), ("idx_m", c_f), ("R", c_f),
                ("offsets", c_f), ("embedding", c_f), ("n_types", c_i32), ("reserved", c_i32)]


class PainnT(ctypes.Structure):
    _fields_ = [("n_atom_basis", c_i32), ("n_interactions", c_i32), ("epsilon", ctypes.c_float),
                ("reserved", c_i32), ("layers", ctypes.POINTER(PainnLayerT)), ("wpack", c_f)]


P = ctypes.POINTER
# name -> (restype, argtypes); mirrors include/spk_hip.h one to one
_PROTOS = {
    "spk_version": (ctypes.c_int, []),
    "spk_last_error": (ctypes.c_char_p, []),
    "spk_device_info": (ctypes.c_int, [P(c_i32)]),
    "spk_set_variant": (None, [ctypes.c_int]),
    "spk_get_variant": (ctypes.c_int, []),
    "spk_set_split": (None, [ctypes.c_int]),
    "spk_get_split": (ctypes.c_int, []),
    "spk_profile_enable": (None, [ctypes.c_int]),
    "spk_profile_report": (ctypes.c_char_p, []),
    "spk_edge_plan": (ctypes.c_int, [c_f, c_f, c_f, c_i64, c_i64, c_f, c_f, c_f, P(c_i32), c_f]),
    "spk_scatter_add_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i64, c_i64, c_i64, c_i64, c_f, c_f]),
    "spk_gather_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i64, c_i64, c_i64, c_f, c_f]),
    "spk_pairwise_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i64, c_f, c_f]),
    "spk_pairwise_n_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, c_f]),
    "spk_pairwise_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i64, c_i64, c_f, c_f]),
    "spk_nbl_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "spk_nbl_count_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, ctypes.c_float, c_f, c_f, ctypes.POINTER(ctypes.c_int64), c_f]),
    "spk_nbl_fill_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i64, ctypes.c_float, c_f, c_f, c_i64, c_f, c_f, c_f, c_f, c_f]),
    "spk_md_half_step_f32": (ctypes.c_int, [c_f, c_f, ctypes.c_float, c_i64, c_f]),
    "spk_md_kick_drift_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, ctypes.c_float, c_i64, c_f, ctypes.c_float, c_f, c_f]),
    "spk_md_ring_polymer_step_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i32, c_i64, c_i32, c_i32, c_f, c_f, c_f, ctypes.c_float, c_f, c_f]),
    "spk_md_pile_f32": (ctypes.c_int, [c_f, c_f, c_f, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, c_f, c_i32, c_i32, c_i64, c_i32, c_i32, c_f, c_f]),
    "spk_pack_weight_f32": (ctypes.c_int, [c_f, c_i32, c_i32, c_i32, c_f, c_f]),
    "spk_schnet_packed_floats": (c_i64, [P(SchnetT)]),
    "spk_schnet_pack_weights_f32": (ctypes.c_int, [P(SchnetT), c_f, c_f]),
    "spk_painn_packed_floats": (c_i64, [P(PainnT)]),
    "spk_painn_pack_weights_f32": (ctypes.c_int, [P(PainnT), c_f, c_f]),
    "spk_chain_set_rows": (None, [c_i32]),
    "spk_chain_set_debug_buffer": (None, [c_f]),
    "spk_atomwise_supported": (ctypes.c_int, [c_i32, c_i32, c_i32]),
    "spk_atomwise_fwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i64, c_f, c_f, c_f, c_f]),
    "spk_atomwise_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_i64, c_f, c_f]),
    "spk_filter_table_set": (ctypes.c_int, [c_f, c_f, c_i32, ctypes.c_float]),
    "spk_filter_table_clear": (None, []),
    "spk_cfconv_tab_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i32, ctypes.c_float, ctypes.c_float, c_i32, c_f, c_f]),
    "spk_segment_rowptr_i32": (ctypes.c_int, [c_f, c_i64, c_i64, c_f, c_f, c_f]),
    "spk_index_range_check": (ctypes.c_int, [c_f, c_i64, c_i64, c_f, c_f]),
    "spk_pairwise_bwd_graph_f32": (ctypes.c_int, [c_f, P(GraphT), c_f, c_f]),
    "spk_radial_cutoff_f32": (ctypes.c_int, [c_f, c_i64, P(RadialT), c_f, c_f, c_f]),
    "spk_radial_cutoff_bwd_f32": (ctypes.c_int, [c_f, c_i64, P(RadialT), c_f, c_f, c_f, c_f]),
    "spk_edge_norm_f32": (ctypes.c_int, [c_f, c_i64, c_f, c_f, c_f]),
    "spk_act_mul_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i64, c_i32, c_i32, c_f, c_f]),
    "spk_gemm_tn_plan": (ctypes.c_int, [c_i64, c_i32, c_i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]),
    "spk_gemm_tn_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f]),
    "spk_gemm_tn_nb_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, c_i32, c_f, c_f, c_i64, c_f, c_f, c_f]),
    "spk_gemm_pair_f32": (ctypes.c_int, [c_f, c_f, c_i32, c_i64, c_i32, c_i32, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_f, c_f, c_f, c_f, c_f]),
    "spk_cfconv_edge_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_i64, c_i32, c_f, c_f]),
    "spk_edge_mul_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_i64, c_i32, c_f, c_f]),
    "spk_radial_d_f32": (ctypes.c_int, [c_f, c_f, c_i64, P(RadialT), c_i32, c_f, c_f]),
    "spk_radial_c_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i64, P(RadialT), c_i32, c_f, c_f]),
    "spk_rowscale_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, c_f, c_f]),
    "spk_rowdot_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, c_f, c_f]),
    "spk_fm_loss_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_f, c_f, c_i64, ctypes.c_float, ctypes.c_float, c_f, c_f, c_f]),
    "spk_fm_loss_bwd_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_f, c_i64, c_f, c_f]),
    "spk_schnet_fm_workspace_bytes": (c_i64, [P(SchnetT), P(HeadT), P(RadialT), c_i64, c_i64, c_i64, c_i32]),
    "spk_schnet_fm_grad_floats": (c_i64, [P(SchnetT), P(HeadT), P(RadialT), c_i32]),
    "spk_schnet_fm_forward_f32": (ctypes.c_int, [P(SchnetT), P(HeadT), P(RadialT), P(FmBatchT), c_f, c_f, c_f, c_f, c_f]),
    "spk_schnet_fm_backward_f32": (ctypes.c_int, [P(SchnetT), P(HeadT), P(RadialT), P(FmBatchT), c_f, c_f, c_f, c_f, c_f]),
    "spk_painn_fm_workspace_bytes": (c_i64, [P(PainnT), P(HeadT), P(RadialT), c_i64, c_i64, c_i64, c_i32]),
    "spk_painn_fm_grad_floats": (c_i64, [P(PainnT), P(HeadT), P(RadialT), c_i32]),
    "spk_painn_fm_forward_f32": (ctypes.c_int, [P(PainnT), P(HeadT), P(RadialT), P(FmBatchT), c_f, c_f, c_f, c_f, c_f]),
    "spk_painn_fm_backward_f32": (ctypes.c_int, [P(PainnT), P(HeadT), P(RadialT), P(FmBatchT), c_f, c_f, c_f, c_f, c_f]),
    "spk_transpose_plan_bytes": (c_i64, [c_i64, c_i64]),
    "spk_transpose_plan": (ctypes.c_int, [c_f, c_i64, c_i64, c_f, c_f, c_f, c_f]),
    "spk_vec3_f32": (ctypes.c_int, [c_i32, c_f, c_i64, c_f, c_i64, c_i64, c_i32, c_f, c_f]),
    "spk_dense_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_f]),
    "spk_dense_bwd_input_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_f]),
    "spk_adamw_f32": (ctypes.c_int, [c_f, c_i64, c_f, c_f, c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f]),
    "spk_adamw_devlr_f32": (ctypes.c_int, [c_f, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f]),
    "spk_filter_table_set_stamp": (ctypes.c_int, [c_f, ctypes.c_uint64]),
    "spk_filter_table_drop_if_stale": (ctypes.c_int, [c_f, ctypes.c_uint64]),
    "spk_dense_dual_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "spk_dense_dual_fwd_supported": (ctypes.c_int, [c_i64, c_i32, c_i32]),
    "spk_dense_dual_f32": (ctypes.c_int, [P(DenseDualT), c_f]),
    "spk_dense_chain_f32": (ctypes.c_int, [P(ChainT), c_f]),
    "spk_schnet_cfconv_fwd_f32": (ctypes.c_int, [P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f]),
    "spk_schnet_cfconv_bwd_f32": (ctypes.c_int, [P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_f]),
    "spk_cfconv_set_debug_buffer": (None, [c_f]),
    "spk_schnet_mol_set_debug_buffer": (None, [c_f]),
    "spk_painn_mol_set_debug_buffer": (None, [c_f]),
    "spk_schnet_saved_floats": (c_i64, [P(SchnetT), c_i64]),
    "spk_schnet_saved_floats_graph": (c_i64, [P(SchnetT), P(GraphT), P(RadialT)]),
    "spk_schnet_scratch_floats": (c_i64, [P(SchnetT), c_i64]),
    "spk_schnet_forward_f32": (ctypes.c_int, [P(SchnetT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_schnet_backward_f32": (ctypes.c_int, [P(SchnetT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_schnet_potential_supported": (ctypes.c_int, [P(SchnetT), P(HeadT), P(GraphT), P(RadialT)]),
    "spk_schnet_potential_forward_f32": (ctypes.c_int, [P(SchnetT), P(HeadT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_i64, c_f, c_f, c_f, c_f, c_f]),
    "spk_schnet_potential_forces_f32": (ctypes.c_int, [P(SchnetT), P(HeadT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_i32, c_f, c_f, c_f, c_i64, c_i32, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_schnet_potential_backward_f32": (ctypes.c_int, [P(SchnetT), P(HeadT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_painn_potential_supported": (ctypes.c_int, [P(PainnT), P(HeadT), P(GraphT), P(RadialT)]),
    "spk_painn_potential_forces_f32": (ctypes.c_int, [P(PainnT), P(HeadT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_i32, c_f, c_f, c_f, c_i64, c_i32, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_painn_message_fwd_f32": (ctypes.c_int, [P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_f]),
    "spk_painn_message_bwd_f32": (ctypes.c_int, [P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_f, c_f]),
    "spk_painn_set_tile": (None, [c_i32]),
    "spk_painn_set_row_table": (None, [c_i32]),
    "spk_painn_set_rowtile": (None, [c_i32]),
    "spk_painn_set_block": (None, [c_i32]),
    "spk_transpose_plan_bytes": (c_i64, [c_i64, c_i64]),
    "spk_fm_set_chain": (None, [ctypes.c_int32]),
    "spk_index_jobs": (ctypes.c_int, [c_f, ctypes.c_int32, c_f, c_f]),
    "spk_transposed_build": (ctypes.c_int, [c_f, c_f, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_blocks_group_atoms": (ctypes.c_int, []),
    "spk_painn_blk_set_debug_buffer": (None, [c_f, c_i32]),
    "spk_blocks_sizes": (ctypes.c_int, [c_i64, c_i64, c_i32, c_i32, ctypes.POINTER(c_i64)]),
    "spk_blocks_build": (ctypes.c_int, [P(GraphT), c_i32, c_i32, P(BlocksT), c_f, ctypes.POINTER(c_i32), c_f]),
    "spk_blocks_prepare_f32": (ctypes.c_int, [P(GraphT), P(RadialT), c_f, c_f]),
    "spk_painn_mix_ctx_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, ctypes.c_float, c_f, c_f]),
    "spk_painn_mix_update_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i32, c_f, c_f, c_f]),
    "spk_painn_mix_update_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i32, c_f, c_f, c_f]),
    "spk_painn_mix_ctx_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i64, c_i32, ctypes.c_float, c_f, c_f, c_f]),
    "spk_painn_saved_floats": (c_i64, [P(PainnT), c_i64]),
    "spk_painn_scratch_floats": (c_i64, [P(PainnT), c_i64]),
    "spk_painn_forward_f32": (ctypes.c_int, [P(PainnT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_painn_backward_f32": (ctypes.c_int, [P(PainnT), P(GraphT), P(RadialT), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "spk_embedding_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_i32, c_f, c_f]),
    "spk_add_f32": (ctypes.c_int, [c_f, c_f, c_i64, c_f, c_f]),
    # deployment runtime (host pointers throughout)
    "spk_potential_load": (ctypes.c_int, [ctypes.c_char_p, P(ctypes.c_void_p)]),
    "spk_potential_from_memory": (ctypes.c_int, [ctypes.c_char_p, c_i64, P(ctypes.c_void_p)]),
    "spk_potential_free": (None, [ctypes.c_void_p]),
    "spk_potential_info": (ctypes.c_int, [ctypes.c_void_p, P(c_i32), P(ctypes.c_float)]),
    "spk_potential_compute": (ctypes.c_int, [ctypes.c_void_p, c_i64, P(ctypes.c_int64), P(ctypes.c_float), c_i64, P(ctypes.c_int64),
                                             P(ctypes.c_int64), P(ctypes.c_float), c_i64, P(ctypes.c_int64), P(ctypes.c_float),
                                             P(ctypes.c_float)]),
    "spk_potential_compute_cell": (ctypes.c_int, [ctypes.c_void_p, c_i64, P(ctypes.c_int64), P(ctypes.c_float), c_i64, P(ctypes.c_int64),
                                                  P(ctypes.c_float), P(ctypes.c_uint8), ctypes.c_float, P(ctypes.c_float),
                                                  P(ctypes.c_float), P(ctypes.c_int64)]),
}

_lib = None
_lock = threading.Lock()


def exported_symbols():
    """Names declared by include/spk_hip.h (used by the CPU symbol test)."""
    return sorted(_PROTOS)


def lib():
    """Load libspk_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SpkHipError(
                    "libspk_hip.so not found at %s -- build it with "
                    "`python -m schnetpack_amd.csrc.build` (hipcc, gfx950); there is no CPU "
                    "fallback" % LIB_PATH)
            # torch must be imported first so that the HIP runtime already loaded by torch
            # (soname libamdhip64.so.7) is the one this library binds to.
            handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, (res, args) in _PROTOS.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().spk_last_error()
        raise SpkHipError("spk_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise SpkHipError(
                "schnetpack_amd: the HIP path needs tensors on a ROCm device (got %s); there is "
                "no CPU fallback" % t.device)


def fptr(t):
    """Device pointer of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise SpkHipError("expected a contiguous float32 ROCm tensor, got %s %s contiguous=%s"
                          % (t.device, t.dtype, t.is_contiguous()))
    return ctypes.c_void_p(t.data_ptr())


def iptr(t, dtype=torch.int64):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise SpkHipError("expected a contiguous %s ROCm tensor, got %s %s" % (dtype, t.device, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_variant(v):
    lib().spk_set_variant(int(v))


def get_variant():
    return int(lib().spk_get_variant())


def set_split(on):
    """Split-precision matrix path (csrc/spk_split.h) on / off; off = v_mfma_f32_32x32x2_f32 everywhere."""
    lib().spk_set_split(1 if on else 0)


def get_split():
    return int(lib().spk_get_split())


def device_info():
    arr = (c_i32 * 4)()
    check(lib().spk_device_info(arr))
    return {"compute_units": arr[0], "wavefront": arr[1], "lds_bytes": arr[2], "gfx": arr[3]}


def profile_enable(on=True):
    lib().spk_profile_enable(1 if on else 0)


def profile_report():
    """{tag: (count, total_ms)} of the kernels launched since the last report."""
    txt = lib().spk_profile_report().decode()
    out = {}
    for line in txt.splitlines():
        tag, cnt, ms = line.split()
        out[tag] = (int(cnt), float(ms))
    return out
