"""ctypes binding of the C ABI declared in include/dvins.h (libdvins_hip.so).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is present,
loading / dv_create fails loudly (DvinsError).  Nothing here imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVINS_HIP_LIB") or os.path.join(_HERE, "lib", "libdvins_hip.so")

import os as _os
DV_STATIC_REPORT_LAG = int(_os.environ.get("DVINS_STATIC_LAG", "2"))      # include/dvins.h (choice T1); the environment override is for experiments and must match the library's
DV_MEM_HOST, DV_MEM_DEVICE, DV_MEM_PINNED = 0, 1, 2      # PINNED: host memory pinned + mapped by the caller, read in place by the kernels
DV_FMT_BGR = 0x100
DV_MODE_RAW, DV_MODE_NAIVE, DV_MODE_SEMANTIC = 0, 1, 2
DV_MAX_FEATS = 1024


class DvinsError(RuntimeError):
    pass


class dv_cam(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")]


class dv_config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("max_cnt", C.c_int), ("min_dist", C.c_int),
                ("flow_back", C.c_int), ("stereo", C.c_int), ("cam0", dv_cam), ("cam1", dv_cam),
                ("device", C.c_int), ("mask_morphology_size", C.c_int), ("reserved", C.c_int * 6)]


class dv_feat(C.Structure):
    _fields_ = [("id", C.c_uint32), ("track_cnt", C.c_int32), ("has_right", C.c_int32), ("pad_", C.c_int32),
                ("left", C.c_double * 7), ("right", C.c_double * 7)]


_u8p = C.c_void_p      # image / status buffers: host ndarray pointer or device pointer (int)
_f32p = C.c_void_p
_ctx = C.c_void_p
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)      # dv_allgather_fn

# name -> (restype, argtypes).  tests/test_abi.py checks this table against include/dvins.h.
SIGNATURES = {
    "dv_create": (_ctx, [C.POINTER(dv_config)]),
    "dv_destroy": (None, [_ctx]),
    "dv_last_error": (C.c_char_p, [_ctx]),
    "dv_reset": (C.c_int, [_ctx]),
    "dv_sync": (C.c_int, [_ctx]),
    "dv_track_stereo": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.c_int, C.c_int,
                                  C.POINTER(dv_feat), C.POINTER(C.c_int)]),
    "dv_track_stereo_enqueue": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.c_int, C.c_int]),
    "dv_track_stereo_collect": (C.c_int, [_ctx, C.POINTER(dv_feat), C.POINTER(C.c_int)]),
    "dv_lk": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                        _f32p, _u8p, C.c_int]),
    "dv_track_by_lk": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_float,
                                 _f32p, _u8p, C.c_int]),
    "dv_lk_cuda": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _u8p, C.c_int]),
    "dv_track_by_lk_gpu": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, _u8p, C.c_int]),
    "dv_pyr_down_cuda": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
    "dv_gftt": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _f32p,
                          C.POINTER(C.c_int), C.c_int]),
    "dv_min_eigen": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]),
    "dv_gftt_cuda": (C.c_int, [_ctx, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _f32p,
                               C.POINTER(C.c_int), C.c_int]),
    "dv_min_eigen_cuda": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]),
    "dv_viode_mask": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, _u8p, _u8p, C.c_void_p, C.c_void_p]),
    "dv_bgr2gray": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
    "dv_remap": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _u8p, C.c_int]),
    "dv_set_undistort_maps": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "dv_pyr_down": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
    "dv_circle_mask": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int]),
    "dv_erode": (C.c_int, [_ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]),
    "dv_lift_projective": (C.c_int, [_ctx, C.POINTER(dv_cam), _f32p, C.c_int, _f32p, C.c_int]),
    "dv_lift_projective_offset": (C.c_int, [_ctx, C.POINTER(dv_cam), _f32p, C.c_int, C.c_double, C.c_double, _f32p, C.c_int]),
    "dv_ba_solve": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "dv_ba_eval": (C.c_int, [_ctx, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_void_p]),
    "dv_marginalize": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_proj_eval": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_imu_eval": (C.c_int, [_ctx, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_line_eval": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_line_plus": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dv_box_enclose_eval": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dv_box_dims_eval": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dv_box_orientation_eval": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dv_inst_proj_eval": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_obj_solve": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "dv_line_solve": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "dv_est_create": (C.c_int, [_ctx, C.c_void_p]),
    "dv_est_reset": (C.c_int, [_ctx]),
    "dv_est_input_imu": (C.c_int, [_ctx, C.c_double, C.c_void_p, C.c_void_p]),
    "dv_est_process": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "dv_est_process_begin": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double]),
    "dv_est_process_end": (C.c_int, [_ctx, C.c_void_p]),
    "dv_est_imu_available": (C.c_int, [_ctx, C.c_double]),
    "dv_est_process_dynamic": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_est_process_dynamic_begin": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dv_est_process_dynamic_begin_ego": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_double]),
    "dv_est_process_dynamic_attach": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dv_est_set_lines": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "dv_est_get_lines": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_undistort_lines": (C.c_int, [_ctx, C.POINTER(dv_cam), C.c_void_p, C.c_int, C.c_void_p]),
    "dv_est_change_sensor_type": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "dv_est_get_latest": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_est_get_extrinsics": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dv_est_get_landmarks": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_inst_config": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int]),
    "dv_inst_reset": (C.c_int, [_ctx]),
    "dv_inst_track_enqueue": (C.c_int, [_ctx, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "dv_inst_set_disparity": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_double]),
    "dv_inst_set_right_keys": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int]),
    "dv_pinned_alloc": (C.c_void_p, [C.c_size_t]),
    "dv_pinned_free": (None, [C.c_void_p]),
    "dv_track_unmask_static": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "dv_est_get_static_instances": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_extra_points": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_inst_track_collect": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_est_get_instances": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "dv_timing_enable": (C.c_int, [_ctx, C.c_int]),
    "dv_timing_reset": (C.c_int, [_ctx]),
    "dv_timing_get": (C.c_int, [_ctx, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "dv_debug_set": (C.c_int, [_ctx, C.c_char_p, C.c_int]),
    "dv_batch_create": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_int]),
    "dv_batch_destroy": (None, [C.c_void_p]),
    "dv_batch_enqueue": (C.c_int, [C.c_void_p]),
    "dv_batch_track_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "dv_batch_track_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dv_batch_arrive": (C.c_int, [C.c_void_p]),
    "dv_batch_abort": (C.c_int, [C.c_void_p]),
    "dv_batch_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dv_runner_create": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dv_runner_destroy": (None, [C.c_void_p]),
    "dv_runner_run": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "dv_runner_get": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dv_runner_get_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_ba_debug_slot_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dv_ba_debug_dev_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_est_debug_hash_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_runner_get_row_log": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_runner_get_frame_clock": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dv_runner_batch_rounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dv_runner_batch_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dv_runner_set_dynamic": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "dv_runner_set_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "dv_runner_dynamic_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dv_runner_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "dv_runner_track_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dv_runner_error": (C.c_char_p, [C.c_void_p]),
    "dv_est_get_marg_health": (C.c_int, [_ctx, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_void_p]),
    "dv_batch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dv_dist_unique_id": (C.c_int, [C.c_void_p]),
    "dv_dist_init_rccl": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "dv_dist_init_host": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dv_dist_peer_prepare": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p]),
    "dv_dist_init_peer": (C.c_int, [_ctx, C.c_void_p]),
    "dv_dist_rccl_ranks": (C.c_int, [_ctx, C.POINTER(C.c_int)]),
    "dv_dist_shutdown": (C.c_int, [_ctx]),
    "dv_dist_exchange_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dv_dist_info": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "dv_allreduce_reduced_system": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
}

_lib = None


def load():
    """Loads libdvins_hip.so; raises DvinsError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DvinsError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(the HIP path has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
