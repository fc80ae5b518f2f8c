"""ctypes binding of libxeve_hip.so (C-ABI: include/xeve_hip.h).  Fails loudly when the library is absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XEVE_HIP_LIB_PATH") or os.path.join(_HERE, "lib", "libxeve_hip.so")  # (the override: experiment builds, tools/gpu/)

c_int, c_void_p, c_i64 = C.c_int, C.c_void_p, C.c_int64


class XeveHipError(RuntimeError):
    pass


class Job(C.Structure):  # xeve_hip_job
    _fields_ = [("off1", C.c_int32), ("off2", C.c_int32)]


class McJob(C.Structure):  # xeve_hip_mc_job
    _fields_ = [("gmv_x", C.c_int32), ("gmv_y", C.c_int32), ("pred_off", C.c_int32), ("frac", C.c_int32)]


class RdoqEst(C.Structure):  # xeve_hip_rdoq_est
    _fields_ = [("cbf", C.c_int32 * 2), ("run", (C.c_int32 * 2) * 24), ("level", (C.c_int32 * 2) * 24), ("last", (C.c_int32 * 2) * 2)]


class MeParams(C.Structure):  # xeve_hip_me_params
    _fields_ = [("lambda_mv", C.c_uint32), ("refi_bits", C.c_int32), ("extra_bits", C.c_int32), ("bi", C.c_int32),
                ("faststep", C.c_int32), ("max_search_range", C.c_int32), ("range_recentre", C.c_int32),
                ("min_clip", C.c_int32 * 2), ("max_clip", C.c_int32 * 2), ("reserved", C.c_int32)]


class SpelParams(C.Structure):  # xeve_hip_spel_params
    _fields_ = [("lambda_mv", C.c_uint32), ("refi_bits", C.c_int32), ("extra_bits", C.c_int32), ("bi", C.c_int32),
                ("hpel_cnt", C.c_int32), ("qpel_cnt", C.c_int32)]


class EpzsParams(C.Structure):  # xeve_hip_epzs_params
    _fields_ = [("me", MeParams), ("hpel_cnt", C.c_int32), ("qpel_cnt", C.c_int32)]


class RdoParams(C.Structure):  # xeve_hip_rdo_params
    _fields_ = [("log2_cuw", C.c_int32), ("log2_cuh", C.c_int32), ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("slice_type", C.c_int32),
                ("num_refp", C.c_int32 * 2), ("chroma_format_idc", C.c_int32), ("bit_depth", C.c_int32), ("tool_iqt", C.c_int32),
                ("qp", C.c_int32 * 3), ("pad_", C.c_int32), ("lambda_", C.c_double * 3), ("dist_chroma_weight", C.c_double * 2)]


class EncConfig(C.Structure):  # xeve_hip_enc_config
    _fields_ = [(n, C.c_int32) for n in "w h fps_num fps_den qp keyint bframes closed_gop preset threads inter_slice_type ref".split()] + [("reserved", C.c_int32 * 4)]


class InterParams(C.Structure):  # xeve_hip_inter_params
    _fields_ = [("rdo", RdoParams), ("me", EpzsParams), ("refi_bits", (C.c_int32 * 8) * 2), ("range_recentre", (C.c_int32 * 8) * 2), ("max_cand", C.c_int32),
                ("poc", C.c_int32), ("col_list_poc0", C.c_int32), ("pad_", C.c_int32), ("skip_th", C.c_double)]


class IntraParams(C.Structure):  # xeve_hip_intra_params
    _fields_ = [("log2_cuw", C.c_int32), ("log2_cuh", C.c_int32), ("w_scu", C.c_int32), ("h_scu", C.c_int32), ("slice_type", C.c_int32),
                ("chroma_format_idc", C.c_int32), ("bit_depth", C.c_int32), ("tool_iqt", C.c_int32), ("constrained_intra_pred", C.c_int32), ("qp", C.c_int32 * 3),
                ("lambda_", C.c_double * 3), ("sqrt_lambda0", C.c_double), ("dist_chroma_weight", C.c_double * 2)]


class TreeParams(C.Structure):  # xeve_hip_tree_params
    _fields_ = [("ip", IntraParams), ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("log2_ctu", C.c_int32), ("max_cu", C.c_int32), ("min_cu", C.c_int32),
                ("min_cuwh", C.c_int32), ("slice_qp", C.c_int32), ("slice_num", C.c_int32), ("pad_", C.c_int32)]


CU_DEPTHS = 10  # XEVE_HIP_CU_DEPTHS
CTU_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("sbac", "<i4"), ("pic", "<i4")]  # xeve_hip_ctu_job (16 B)
CTU_DATA_DTYPE = [("split_mode", "i1", (CU_DEPTHS, 256)), ("pred_mode", "u1", (256,)), ("ipm", "i1", (2, 256)), ("depth", "i1", (256,)), ("nnz", "<i4", (3, 256)),
                  ("map_scu", "<u4", (256,)), ("map_cu_mode", "<u4", (256,)), ("coef", "<i2", (3, 4096)), ("reco", "<i2", (3, 4096)), ("mv", "<i2", (256, 2, 2)),
                  ("mvd", "<i2", (256, 2, 2)), ("refi", "i1", (256, 2)), ("mvp_idx", "u1", (256, 2))]  # xeve_hip_ctu_data (62976 B)
CTU_DATA_BYTES = 62976


class EcoParams(C.Structure):  # xeve_hip_eco_params
    _fields_ = [("chroma_format_idc", C.c_int32), ("slice_type", C.c_int32), ("log2_ctu", C.c_int32), ("pic_w", C.c_int32), ("pic_h", C.c_int32), ("w_scu", C.c_int32),
                ("h_scu", C.c_int32), ("num_refp", C.c_int32 * 2), ("pad_", C.c_int32)]


class TreeInter(C.Structure):  # xeve_hip_tree_inter
    _fields_ = [("refp", C.c_void_p), ("s_ref_l", C.c_int32), ("s_ref_c", C.c_int32), ("ipar", InterParams), ("map_mv", C.c_void_p), ("map_refi", C.c_void_p),
                ("col_mv0", C.c_void_p), ("col_mv1", C.c_void_p), ("coef_l", C.c_void_p), ("coef_c", C.c_void_p), ("ecu_depth", C.c_int32), ("pad_", C.c_int32)]

INTRA_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("inter_satd", "<u4"), ("sbac", "<i4"), ("pic", "<i4"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1"),
                   ("pad_", "u1", (2,))]  # xeve_hip_intra_job (24 B)
INTRA_RESULT_DTYPE = [("cost", "<f8"), ("dist_cu", "<i4"), ("nnz", "<i4", (3,)), ("pred_cnt", "<i4"), ("ipm", "i1", (2,)), ("pad_", "i1", (2,))]  # xeve_hip_intra_result (32 B)
INTER_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (2, 4, 2)), ("mv_col", "<i2", (2,)), ("sbac", "<i4"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1"),
                   ("pad_", "u1", (2,))]  # xeve_hip_inter_job (52 B)
INTER_RESULT_DTYPE = [("cost", "<f8"), ("cost_inter", "<f8", (5,)), ("cu_mode", "<i4"), ("best_idx", "<i4"), ("mv", "<i2", (2, 2)), ("mvd", "<i2", (2, 2)),
                      ("refi", "i1", (2,)), ("mvp_idx", "u1", (2,)), ("nnz", "<i4", (3,)), ("pad_", "<i4", (2,))]  # xeve_hip_inter_result (96 B)
RDO_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 2)), ("mvd", "<i2", (2, 2)), ("refi", "i1", (2,)), ("mvp_idx", "u1", (2,)),
                 ("dir_flag", "u1"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1"), ("pad_", "u1"), ("sbac", "<i4")]  # xeve_hip_rdo_job (36 B)
RDO_RESULT_DTYPE = [("cost", "<f8"), ("nnz", "<i4", (3,)), ("pad_", "<i4"), ("dist", "<i8", (2, 3))]  # xeve_hip_rdo_result (72 B)
SKIP_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("mvp", "<i2", (2, 4, 2)), ("refi_pred", "i1", (2, 4)), ("ncand", "<i4"), ("sbac", "<i4"),
                  ("ctx_skip", "u1"), ("pad_", "u1", (3,))]  # xeve_hip_skip_job (60 B)
SKIP_RESULT_DTYPE = [("cost", "<f8"), ("best_ssd", "<i8"), ("idx0", "<i4"), ("idx1", "<i4"), ("mv", "<i2", (2, 2)), ("refi", "i1", (2,)),
                     ("pad_", "i1", (6,))]  # xeve_hip_skip_result (40 B)


class DeblockParams(C.Structure):  # xeve_hip_deblock_params
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("w_scu", C.c_int32), ("h_scu", C.c_int32), ("log2_max_cuwh", C.c_int32),
                ("bit_depth_luma", C.c_int32), ("bit_depth_chroma", C.c_int32), ("chroma_format_idc", C.c_int32),
                ("qp_u_offset", C.c_int32), ("qp_v_offset", C.c_int32), ("qp_chroma", (C.c_int32 * 100) * 2)]


class CuBitsParams(C.Structure):  # xeve_hip_cu_bits_params
    _fields_ = [("log2_cuw", C.c_int32), ("log2_cuh", C.c_int32), ("slice_type", C.c_int32), ("num_refp", C.c_int32 * 2),
                ("cm_init", C.c_int32), ("chroma_format_idc", C.c_int32)]


EST_FULL_INTS = 108  # xeve_hip_rdoq_est_full: cbf_all, cbf_luma, cbf_cb, cbf_cr [2] each, run[24][2], level[24][2], last[2][2]
REFPIC_DTYPE = [("y", "<u8"), ("u", "<u8"), ("v", "<u8"), ("poc", "<i4"), ("pad_", "<i4")]  # xeve_hip_refpic (device addresses), host array
CU_MC_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("mv", "<i2", (2, 2)), ("refi", "i1", (2,)), ("pad_", "i1", (2,))]  # xeve_hip_cu_mc_job (20 B)
SBAC_NCTX = 72
SBAC_DTYPE = [("range", "<u4"), ("code", "<u4"), ("code_bits", "<u4"), ("stacked_ff", "<u4"), ("stacked_zero", "<u4"), ("pending_byte", "<u4"),
              ("is_pending_byte", "<u4"), ("bitcounter", "<u4"), ("bin_counter", "<u4"), ("ctx", "<u2", (SBAC_NCTX,))]  # xeve_hip_sbac (180 B)
CU_BITS_JOB_DTYPE = [("coef_off", "<i4", (3,)), ("nnz", "<i4", (3,)), ("sbac", "<i4"), ("mvd", "<i2", (2, 2)), ("refi", "i1", (2,)),
                     ("mvp_idx", "u1", (2,)), ("mode", "u1"), ("dir_flag", "u1"), ("ctx_skip", "u1"), ("ctx_pred_mode", "u1")]  # xeve_hip_cu_bits_job (44 B)
EPZS_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("org_off", "<i4"), ("mvp", "<i2", 2), ("mv_start", "<i2", 2)]  # xeve_hip_epzs_job
SPEL_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("org_off", "<i4"), ("gmvp", "<i2", 2), ("mvi", "<i2", 2)]  # xeve_hip_spel_job
ME_JOB_DTYPE = [("x", "<i4"), ("y", "<i4"), ("org_off", "<i4"), ("range", "<i2", 4), ("gmvp", "<i2", 2), ("mvi", "<i2", 2), ("beststep_in", "<i4")]  # xeve_hip_me_job
ME_RESULT_DTYPE = [("mv", "<i2", 2), ("cost", "<u4"), ("beststep", "<i4"), ("best_mv_bits", "<i4")]  # xeve_hip_me_result


# reference: src_base/xeve_sad.h:41-45, xeve_mc.h:85-87, xeve_type.h:169-170
FN_SAD = C.CFUNCTYPE(c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int)
FN_SATD = FN_SAD
FN_SSD = C.CFUNCTYPE(c_i64, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int)
FN_DIFF = C.CFUNCTYPE(None, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int)
FN_MC = C.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p)
FN_AVG = C.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int)
FN_TXB = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int, c_int)
FN_MCM = C.CFUNCTYPE(None, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int)  # XEVEM_MC (src_main/xevem_mc.h:45)
FN_TX = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int)  # XEVE_TX / XEVE_ITX
FN_ANG = C.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, C.c_uint16, c_void_p, c_int, c_int, c_int, c_int)  # XEVE_INTRA_PRED_ANG (src_main/xevem_ipred.h:104)
FN_ITR = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int, c_int, c_int)  # XEVE_INV_TRANS (src_main/xevem_type.h:47)
FN_RECON = C.CFUNCTYPE(None, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int)

# every symbol include/xeve_hip.h declares: name -> (restype, argtypes) for functions, ctypes array type for tables
_JOB_ARGS = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int]
FUNCTIONS = {
    "xeve_hip_init": (c_int, [c_int]),
    "xeve_hip_shutdown": (None, []),
    "xeve_hip_last_error": (C.c_char_p, []),
    "xeve_hip_table_calls": (C.c_uint64, []),
    "xeve_hip_table_calls_main": (C.c_uint64, []),
    "xeve_hip_install_tables": (c_int, [c_void_p]),
    "xeve_hip_install_tables_main": (c_int, [c_void_p]),
    "xevem_scaled_horizontal_sobel_filter_hip": (None, [c_void_p, c_int, c_void_p, c_int, c_int, c_int]),
    "xevem_scaled_vertical_sobel_filter_hip": (None, [c_void_p, c_int, c_void_p, c_int, c_int, c_int]),
    "xevem_equal_coeff_computer_hip": (None, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int]),
    # Main profile: the adaptive loop filter's sample kernels (include/xeve_hip.h, src_main/xevem_alf.c)
    "xeve_hip_affine_mc_host": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int]),
    "xeve_hip_affine_me_jobs": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.c_uint32, c_int, c_int, c_void_p]),
    "xeve_hip_affine_me_host": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, C.c_uint32,
                                        c_int, c_int, c_void_p]),
    "xeve_hip_affine_mc_jobs": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_alf_copy_and_extend": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "xeve_hip_alf_classify": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "xeve_hip_alf_filter_jobs": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "xeve_hip_alf_blk_stats_jobs": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_alf_get_blk_stats_host": (None, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "xeve_hip_alf_derive_classification_blk_host": (None, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int]),
    "xeve_hip_alf_filter_blk_7_host": (None, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, C.c_uint8, c_void_p, c_void_p]),
    "xeve_hip_alf_filter_blk_5_host": (None, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, C.c_uint8, c_void_p, c_void_p]),
    "xeve_average_16b_no_clip_hip": (None, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "xeve_recon_blk_hip": (None, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "xeve_hip_sad_jobs": (c_int, _JOB_ARGS + [c_int, c_void_p, c_void_p]),
    "xeve_hip_sad_jobs_dual": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p, c_void_p]),
    "xeve_hip_plane_shift1": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "xeve_hip_ssd_jobs": (c_int, _JOB_ARGS + [c_void_p, c_void_p]),
    "xeve_hip_satd_jobs": (c_int, _JOB_ARGS + [c_void_p, c_void_p]),
    "xeve_hip_diff_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_mc_l_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_mc_c_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_mc_l_sad_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_mc_ssd_jobs": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_avg": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "xeve_hip_trans": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "xeve_hip_itrans": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "xeve_hip_quant": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_rdoq_zero_test": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_rdoq": (c_int, [c_void_p, c_int, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_rdoq_zt": (c_int, [c_void_p, c_int, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "xeve_hip_residual_rdoq": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, C.c_double,
                                       c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_dquant": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "xeve_hip_residual_rdo": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_me_ipel_diamond_jobs": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p]),
    "xeve_hip_me_spel_workspace": (C.c_size_t, [c_int]),
    "xeve_hip_me_spel_pattern_jobs": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_me_epzs_workspace": (C.c_size_t, [c_int]),
    "xeve_hip_me_epzs_jobs": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_rdoq_bit_est": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "xeve_hip_rdoq_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_void_p, c_void_p]),
    "xeve_hip_deblock": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_deblock_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_picbuf_expand_host": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9),
    "xeve_hip_picbuf_expand": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "xeve_hip_mc_cu_workspace": (C.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "xeve_hip_mc_cu_jobs": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_residue_rdo_workspace": (C.c_size_t, [c_int, c_int, c_void_p, c_int, c_int]),
    "xeve_hip_residue_rdo_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_analyze_skip_workspace": (C.c_size_t, [c_int, c_void_p, c_int]),
    "xeve_hip_analyze_skip_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_pinter_analyze_cu_workspace": (C.c_size_t, [c_int, c_int, c_void_p, c_int, c_int]),
    "xeve_hip_pinter_analyze_cu_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_pintra_analyze_cu_workspace": (C.c_size_t, [c_int, c_int, c_void_p]),
    "xeve_hip_pintra_analyze_cu_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 5 +
                                        [C.c_size_t, c_void_p]),
    "xeve_hip_pintra_analyze_cu_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 14),
    "xeve_hip_mode_analyze_ctu_intra_workspace": (C.c_size_t, [c_int, c_void_p]),
    "xeve_hip_mode_analyze_ctu_intra_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int] + [c_void_p] * 4 +
                                             [C.c_size_t, c_void_p]),
    "xeve_hip_mode_analyze_ctu_workspace": (C.c_size_t, [c_int, c_void_p, c_void_p, c_int, c_int]),
    "xeve_hip_mode_analyze_ctu_jobs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 4 +
                                       [C.c_size_t, c_void_p]),
    "xeve_hip_mode_analyze_ctu_intra_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 3),
    "xeve_hip_walk_prof_enable": (c_int, [c_int]),
    "xeve_hip_walk_prof": (c_int, [c_void_p, c_int]),
    "xeve_hip_walk_fused": (c_int, [c_int]),
    "xeve_hip_walk_select": (c_int, [c_int]),
    "xeve_hip_walk_team": (c_int, [c_int]),
    "xeve_hip_walk_side": (c_int, [c_int]),
    "xeve_hip_mode_analyze_ctu_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int] * 4 + [c_void_p] * 3),
    "xeve_hip_eco_ctu_jobs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, C.c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "xeve_hip_eco_tile_end_jobs": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "xeve_hip_sizeof": (c_int, [c_int]),
    "xeve_hip_picture_begin": (c_int, []),
    "xeve_hip_picture_end": (c_int, []),
    "xeve_hip_resident_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_prof_enable": (c_int, [c_int]),
    "xeve_hip_prof_read": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "xeve_hip_prof_cu_bits_hist": (c_int, [c_void_p, c_int]),
    "xeve_hip_inter_candidates": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_int, c_void_p]),
    "xeve_hip_pinter_analyze_cu_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 14),
    "xeve_hip_me_epzs_jobs_x": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, C.c_size_t, c_void_p]),
    "xeve_hip_cu_bits_workspace": (C.c_size_t, [c_int, C.c_size_t]),
    "xeve_hip_cu_bits_jobs": (c_int, [c_void_p, C.c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_size_t, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_cu_bits_jobs_chain": (c_int, [c_void_p, C.c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p, C.c_size_t, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_me_epzs_host": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_tq_nnz_host": (c_int, [c_void_p, c_int, c_int, c_int, C.c_double, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "xeve_hip_itdq_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "xeve_hip_eco_coef_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int]),
    "xeve_hip_mc_cu_host": (c_int, [c_void_p] + [c_int] * 8 + [c_void_p] + [c_int] * 5 + [c_void_p] * 5),
    "xeve_hip_recon": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    # the closed-GOP batch encoder (xeve_amd/encode.py)
    "xeve_hip_enc_create": (c_void_p, [c_void_p, c_int, c_int]),
    "xeve_hip_enc_delete": (None, [c_void_p]),
    "xeve_hip_enc_push": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int]),
    "xeve_hip_enc_encode": (c_int, [c_void_p]),
    "xeve_hip_enc_begin": (c_int, [c_void_p]),
    "xeve_hip_enc_advance": (c_int, [c_void_p, c_i64, c_void_p]),
    "xeve_hip_enc_sync": (c_int, [c_void_p]),
    "xeve_hip_enc_flush": (c_int, [c_void_p]),
    "xeve_hip_enc_bitstream": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "xeve_hip_enc_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "xeve_hip_enc_footprint": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
}
TABLES = {
    "xeve_tbl_sad_16b_hip": FN_SAD * 64,
    "xeve_tbl_ssd_16b_hip": FN_SSD * 64,
    "xeve_tbl_diff_16b_hip": FN_DIFF * 64,
    "xeve_tbl_satd_16b_hip": FN_SATD * 1,
    "xeve_tbl_mc_l_hip": FN_MC * 4,
    "xeve_tbl_mc_c_hip": FN_MC * 4,
    "xeve_tbl_txb_hip": FN_TXB * 6,
    "xeve_tbl_itxb_hip": FN_TXB * 6,
    # Main profile, first slice (src_main/xevem_mc.c:465-485, xevem_tq.c:702, xevem_itdq.c:549)
    "xevem_tbl_dmvr_mc_l_hip": FN_MCM * 4,
    "xevem_tbl_dmvr_mc_c_hip": FN_MCM * 4,
    "xevem_tbl_bl_mc_l_hip": FN_MCM * 4,
    "xeve_tbl_tx_hip": FN_TX * 6,
    "xeve_tbl_itx_hip": FN_TX * 6,
    "xeve_itrans_map_tbl_hip": FN_ITR * 80,  # [16][5]
    "xeve_trans_map_tbl_hip": FN_ITR * 80,  # [16][5]
    "xeve_tbl_intra_pred_ang_hip": FN_ANG * 6,  # [3][2]
}

_lib = None


def load():
    """dlopen libxeve_hip.so and bind every symbol of the C-ABI.  No GPU is touched here."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XeveHipError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C xeve_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in FUNCTIONS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.tables = {name: ty.in_dll(L, name) for name, ty in TABLES.items()}
        _lib = L
    return _lib


def last_error():
    return (load().xeve_hip_last_error() or b"").decode()


def check(rc):
    if rc != 0:
        raise XeveHipError("libxeve_hip rc=%d: %s" % (rc, last_error()))


def init(device=0):
    """Bind this process to GPU `device` (must be gfx950).  Raises if that is impossible."""
    check(load().xeve_hip_init(int(device)))


def table_calls():
    return int(load().xeve_hip_table_calls())


PROF_CLASSES = ("search", "spel", "cu_bits", "mc", "resid", "rdoq", "cu_bits_slow", "walk")  # include/xeve_hip.h: xeve_hip_prof_*


def prof_enable(classes=PROF_CLASSES):
    """switch the kernel-class timers on for the named classes (an empty list / None: all off)"""
    mask = 0
    for c in classes or ():
        mask |= 1 << PROF_CLASSES.index(c)
    check(load().xeve_hip_prof_enable(mask))


def prof_cu_bits_hist(reset=False):
    """{bucket label: jobs} -- the bins-per-job histogram of the CABAC bit-count launches that ran with their class timer on (xeve_hip_prof_cu_bits_hist)"""
    h = (C.c_uint64 * 32)()
    check(load().xeve_hip_prof_cu_bits_hist(h, 1 if reset else 0))
    lab = lambda b: "0" if b == 0 else "1" if b == 1 else "%d-%d" % (1 << (b - 1), (1 << b) - 1)
    top = max([b for b in range(32) if h[b]] or [0])
    return {lab(b): int(h[b]) for b in range(top + 1)}


def prof_read():
    """{class: (ms, launches, units)} since the last read (waits for the device)"""
    ms, n, u = (C.c_double * 8)(), (C.c_uint64 * 8)(), (C.c_uint64 * 8)()
    check(load().xeve_hip_prof_read(ms, n, u, 8))
    return {name: (ms[i], int(n[i]), int(u[i])) for i, name in enumerate(PROF_CLASSES)}
