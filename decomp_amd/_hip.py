"""ctypes binding of libdecomp_hip.so (the C ABI declared in include/decomp_hip.h).

The HIP library IS the compute path: there is no CPU fallback.  If the shared object
is missing or no MI355X is visible, the first call raises (loudly) instead of
computing somewhere else.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libdecomp_hip.so')

OK = 0
ERR_NAMES = {-1: 'DCP_ERR_INVALID', -2: 'DCP_ERR_HIP', -3: 'DCP_ERR_NOMEM',
             -4: 'DCP_ERR_INTERNAL', -5: 'DCP_ERR_UNSUPPORTED', -6: 'DCP_ERR_REF_TYPEERROR',
             -7: 'DCP_ERR_COMM'}
LIK_L2, LIK_KL = 0, 1
PROF_NLABELS = 10
PROF_XUPDATE, PROF_STATS = 2, 4
LASSO_ISTA, LASSO_ACC_ISTA, LASSO_FISTA, LASSO_CD = 0, 1, 2, 3
LASSO_PARALLEL_CD, LASSO_ADMM = 4, 5
LASSO_POSITIVE = 0x100
ERR_REF_TYPEERROR = -6
ERR_COMM = -7
COMM_ID_BYTES = 128
# int fn(void* buf, int64 count, int dtype, void* hip_stream, void* user)  (dcp_comm_set_external)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)

_c_int, _c_i64, _c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
_c_f32, _c_f64 = ctypes.c_float, ctypes.c_double
_P = ctypes.POINTER

# name -> (restype, argtypes).  Kept in step with include/decomp_hip.h; the CPU test
# tests/test_abi.py checks that every symbol the header declares is listed and exported.
SIGNATURES = {
    'dcp_create': (_c_int, [_P(_c_vp), _c_int]),
    'dcp_destroy': (_c_int, [_c_vp]),
    'dcp_set_stream': (_c_int, [_c_vp, _c_vp]),
    'dcp_last_error_string': (ctypes.c_char_p, [_c_vp]),
    'dcp_build_info': (ctypes.c_char_p, []),
    'dcp_comm_unique_id': (_c_int, [_c_vp, _c_i64]),
    'dcp_comm_init': (_c_int, [_c_vp, _c_vp, _c_int, _c_int]),
    'dcp_comm_destroy': (_c_int, [_c_vp]),
    'dcp_comm_set_external': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int]),
    'dcp_memcpy': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64]),
    'dcp_comm_info': (_c_int, [_c_vp, _P(_c_int), _P(_c_int)]),
    'dcp_comm_allreduce_sum_f32': (_c_int, [_c_vp, _c_vp, _c_i64]),
    'dcp_comm_allreduce_sum_f64': (_c_int, [_c_vp, _c_vp, _c_i64]),
    'dcp_nmf_mu_sharded_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                        _c_int, _c_f32, _c_int, _P(_c_int), _P(_c_f32)]),
    'dcp_nmf_mu_sharded_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                        _c_int, _c_f64, _c_int, _P(_c_int), _P(_c_f64)]),
    'dcp_profile_enable': (_c_int, [_c_vp, _c_int]),
    'dcp_profile_reset': (_c_int, [_c_vp]),
    'dcp_profile_select': (_c_int, [_c_vp, ctypes.c_uint]),
    'dcp_profile_read': (_c_int, [_c_vp, _c_int, _P(_c_f64), _P(_c_i64)]),
    'dcp_profile_label_name': (ctypes.c_char_p, [_c_int]),
    'dcp_l2_normalize_f32': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int]),
    'dcp_l2_normalize_f64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int]),
    'dcp_l2_normalize_c64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int]),
    'dcp_l2_normalize_c128': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int]),
    'dcp_count_negative_f32': (_c_int, [_c_vp, _c_vp, _c_i64, _P(_c_i64)]),
    'dcp_count_negative_f64': (_c_int, [_c_vp, _c_vp, _c_i64, _P(_c_i64)]),
    'dcp_gemm_f32': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                              _c_int, _c_int]),
    'dcp_gemm_f64': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                              _c_int, _c_int]),
    'dcp_gather_rows_bytes': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_scatter_rows_bytes': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_prefetch_rows_bytes': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_set_pcd_order': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64]),
    'dcp_lasso_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_int, _P(_c_int)]),
    'dcp_lasso_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_int, _P(_c_int)]),
    'dcp_lasso_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_int, _P(_c_int)]),
    'dcp_lasso_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_int, _P(_c_int)]),
    'dcp_lasso_pcd_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_vp, _c_i64, _P(_c_int)]),
    'dcp_lasso_admm_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_int)]),
    'dcp_lasso_pcd_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_vp, _c_i64, _P(_c_int)]),
    'dcp_lasso_admm_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_int)]),
    'dcp_lasso_pcd_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_vp, _c_i64, _P(_c_int)]),
    'dcp_lasso_admm_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_int)]),
    'dcp_lasso_pcd_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_vp, _c_i64, _P(_c_int)]),
    'dcp_lasso_admm_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_int)]),
    'dcp_dict_stats_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_dict_update_f32': (_c_int, [_c_vp, _c_vp, _c_f64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_step_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_step_async_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_gather_rows_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_stats_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_dict_update_f64': (_c_int, [_c_vp, _c_vp, _c_f64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_step_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_step_async_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_gather_rows_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_stats_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_dict_update_c64': (_c_int, [_c_vp, _c_vp, _c_f64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_step_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_step_async_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_gather_rows_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_stats_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_dict_update_c128': (_c_int, [_c_vp, _c_vp, _c_f64, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_step_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_step_async_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _c_vp, _P(_c_int)]),
    'dcp_gather_rows_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_dict_mask_step_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_mask_step_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_mask_step_c64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_dict_mask_step_c128': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _c_f64, _c_int, _c_int, _c_f64, _P(_c_f64), _P(_c_int)]),
    'dcp_nmf_grads_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp, _c_vp]),
    'dcp_nmf_grads_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp, _c_vp]),
    'dcp_nmf_grad_x_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_vp]),
    'dcp_nmf_grad_x_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_vp]),
    'dcp_nmf_gauss_logp_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _P(_c_f64)]),
    'dcp_nmf_gauss_logp_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_f64, _P(_c_f64)]),
    'dcp_nmf_apply_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_f64, _c_vp, _c_i64, _c_i64, _P(_c_f64)]),
    'dcp_nmf_apply_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_f64, _c_vp, _c_i64, _c_i64, _P(_c_f64)]),
    'dcp_axpby_f32': (_c_int, [_c_vp, _c_i64, _c_f64, _c_vp, _c_f64, _c_vp]),
    'dcp_axpby_f64': (_c_int, [_c_vp, _c_i64, _c_f64, _c_vp, _c_f64, _c_vp]),
    'dcp_mu_quotient_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_mu_quotient_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_l2_normalize_diff_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_int, _P(_c_f64)]),
    'dcp_l2_normalize_diff_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_int, _P(_c_f64)]),
    'dcp_gershgorin_f32': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_gershgorin_f64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_gershgorin_c64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_gershgorin_c128': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_inv_f32': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_inv_f64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_inv_c64': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_inv_c128': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    'dcp_debug_tn_plain': (_c_int, [_c_int]),
    'dcp_calib_read_f32': (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int, _c_vp]),
    'dcp_gemm_c64': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                              _c_int, _c_int]),
    'dcp_gemm_c128': (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                              _c_int, _c_int]),
    'dcp_nmf_mu_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                _c_int, _c_f32, _c_int, _P(_c_int), _P(_c_f32), _P(_c_f32)]),
    'dcp_nmf_mu_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                _c_int, _c_f64, _c_int, _P(_c_int), _P(_c_f64), _P(_c_f64)]),
    'dcp_nmf_mu_stats_width': (_c_i64, [_c_i64, _c_i64, _c_int, _c_int]),
    'dcp_nmf_mu_stats_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64,
                                      _c_i64, _c_int, _c_vp]),
    'dcp_nmf_mu_stats_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64,
                                      _c_i64, _c_int, _c_vp]),
    'dcp_nmf_mask_bits_words': (_c_i64, [_c_i64, _c_i64]),
    'dcp_nmf_mask_prepare_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _P(_c_int)]),
    'dcp_nmf_mask_prepare_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _P(_c_int)]),
    'dcp_nmf_mu_stats_prepared_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64,
                                               _c_i64, _c_i64, _c_int, _c_vp]),
    'dcp_nmf_mu_stats_prepared_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64,
                                               _c_i64, _c_i64, _c_int, _c_vp]),
    'dcp_nmf_mu_update_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_int,
                                       _c_int, _c_vp, _c_vp]),
    'dcp_nmf_mu_update_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_int,
                                       _c_int, _c_vp, _c_vp]),
    'dcp_nmf_residual_f32': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                      _P(_c_f64)]),
    'dcp_nmf_residual_f64': (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                      _P(_c_f64)]),
}

_lib = None
_lock = threading.Lock()
_handles = {}


class HipLibraryError(RuntimeError):
    """The HIP library is missing, failed to load, or a call into it failed."""


def load():
    """Load libdecomp_hip.so (once) and declare every entry point's signature."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                'libdecomp_hip.so not found at %s. Build it with `make` (or '
                '`python -c "import __graft_entry__ as g; g.build()"`). decomp_amd has no '
                'CPU fallback: the HIP library is the compute path.' % LIB_PATH)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def handle(device):
    """The per-device library handle (created on first use)."""
    lib = load()
    with _lock:
        h = _handles.get(device)
        if h is None:
            out = _c_vp()
            rc = lib.dcp_create(ctypes.byref(out), int(device))
            if rc != OK or not out.value:
                raise HipLibraryError(
                    'dcp_create(device=%d) failed (%s): no usable HIP device. decomp_amd '
                    'computes only on the GPU.' % (device, ERR_NAMES.get(rc, rc)))
            h = out
            _handles[device] = h
        return h


def check(h, rc, what):
    if rc != OK:
        msg = load().dcp_last_error_string(h)
        raise HipLibraryError('%s failed: %s (%s)' % (
            what, ERR_NAMES.get(rc, rc), msg.decode() if msg else ''))
