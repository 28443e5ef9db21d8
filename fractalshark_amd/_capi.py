"""ctypes declarations for the two native libraries.  No torch types cross these boundaries."""
import ctypes as C
import os

from . import _build

vp = C.c_void_p
u32 = C.c_uint32
u64 = C.c_uint64
i32 = C.c_int32


class RealHdr32(C.Structure):
    _fields_ = [("m", C.c_float), ("e", C.c_int32)]


class CplxHdr32(C.Structure):
    _fields_ = [("re", C.c_float), ("im", C.c_float), ("e", C.c_int32)]


class AtHdr32(C.Structure):
    _fields_ = [("StepLength", u32), ("ThresholdC", RealHdr32), ("SqrEscapeRadius", RealHdr32),
                ("RefC", CplxHdr32), ("ZCoeff", CplxHdr32), ("CCoeff", CplxHdr32), ("InvZCoeff", CplxHdr32),
                ("CCoeffSqrInvZCoeff", CplxHdr32), ("CCoeffInvZCoeff", CplxHdr32),
                ("CCoeffNormSqr", RealHdr32), ("RefCNormSqr", RealHdr32), ("factor", RealHdr32)]


class Reduction(C.Structure):
    _fields_ = [("Min", u64), ("Max", u64), ("Sum", u64)]


class RealHdr64(C.Structure):
    _fields_ = [("m", C.c_double), ("e", C.c_int32), ("pad_", C.c_int32)]


class CplxHdr64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double), ("e", C.c_int32), ("pad_", C.c_int32)]


class AtHdr64(C.Structure):
    _fields_ = [("StepLength", u32), ("pad_", u32), ("ThresholdC", RealHdr64), ("SqrEscapeRadius", RealHdr64),
                ("RefC", CplxHdr64), ("ZCoeff", CplxHdr64), ("CCoeff", CplxHdr64), ("InvZCoeff", CplxHdr64),
                ("CCoeffSqrInvZCoeff", CplxHdr64), ("CCoeffInvZCoeff", CplxHdr64),
                ("CCoeffNormSqr", RealHdr64), ("RefCNormSqr", RealHdr64), ("factor", RealHdr64)]


class Real2x32(C.Structure):
    _fields_ = [("head", C.c_float), ("tail", C.c_float), ("e", C.c_int32)]


class Cplx2x32(C.Structure):
    _fields_ = [("re_head", C.c_float), ("re_tail", C.c_float), ("im_head", C.c_float), ("im_tail", C.c_float),
                ("e", C.c_int32)]


class At2x32(C.Structure):
    _fields_ = [("StepLength", u32), ("ThresholdC", Real2x32), ("SqrEscapeRadius", Real2x32),
                ("RefC", Cplx2x32), ("ZCoeff", Cplx2x32), ("CCoeff", Cplx2x32), ("InvZCoeff", Cplx2x32),
                ("CCoeffSqrInvZCoeff", Cplx2x32), ("CCoeffInvZCoeff", Cplx2x32),
                ("CCoeffNormSqr", Real2x32), ("RefCNormSqr", Real2x32), ("factor", Real2x32)]


assert C.sizeof(AtHdr32) == 116 and C.sizeof(AtHdr64) == 232 and C.sizeof(At2x32) == 184

DONE_CB = C.CFUNCTYPE(None, vp)

_render = None
_inputs = None


def _decl(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def render_lib():
    """libfsmi355.so.  Raises if the HIP library has not been built: there is no fallback."""
    global _render
    if _render is not None:
        return _render
    # FSMI355_LIB: another build of the same library (A/B measurements against other build flags; tools/c2_ab.py)
    path = os.environ.get("FSMI355_LIB") or _build.LIB_RENDER
    if os.environ.get("FSMI355_LIB"):
        # (advisor, round 5: an override has no stamp check -- at least say which build is being measured)
        import sys
        print("fractalshark_amd: FSMI355_LIB overrides the in-tree library: %s" % path, file=sys.stderr)
    if not os.path.exists(path):
        raise RuntimeError("libfsmi355.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % path)
    lib = C.CDLL(path)
    _decl(lib, "fs_create", vp, [C.c_int])
    _decl(lib, "fs_destroy", None, [vp])
    _decl(lib, "fs_test_device_is_working", u32, [])
    _decl(lib, "fs_device_count", C.c_int, [])
    _decl(lib, "fs_host_fallback_bytes", u64, [vp])
    _decl(lib, "fs_idle_device_bytes", u64, [vp])
    _decl(lib, "fs_release_idle_device_memory", u64, [C.c_int])
    _decl(lib, "fs_set_compressed_orbit_mode", u32, [vp, C.c_int])
    _decl(lib, "fs_orbit_device_bytes", u64, [vp])
    _decl(lib, "fs_error_string", C.c_char_p, [u32])
    _decl(lib, "fs_init_memory", u32, [vp, u32, u32, u32, u32, vp, u32, u32, u64, C.c_int])
    _decl(lib, "fs_set_row_bands", u32, [vp, u32, u32, u32])
    _decl(lib, "fs_local_rows", u32, [vp])
    _decl(lib, "fs_set_external_iter_buffer", u32, [vp, vp, u64])
    _decl(lib, "fs_device_iter_buffer", vp, [vp])
    _decl(lib, "fs_rounded_width", u32, [vp])
    _decl(lib, "fs_upload_orbit", u32, [vp, u64, C.c_int, u32, vp, u64, u64, u64])
    _decl(lib, "fs_upload_orbit_compressed", u32, [vp, u64, C.c_int, u32, vp, u64, u64, u64, vp, vp])
    _decl(lib, "fs_upload_la", u32, [vp, u64, C.c_int, u32, vp, u32, vp, u32, C.c_int, C.c_int, vp])
    _decl(lib, "fs_upload_bla", u32, [vp, C.c_int, vp, vp, i32, i32])
    _decl(lib, "fs_render_lav2", u32, [vp, C.c_int, C.c_int, C.c_int, vp, u64])
    _decl(lib, "fs_render_bla", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_render_direct", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_upload_orbit_scaled", u32, [vp, C.c_int, u32, vp, vp, u64, u64])
    _decl(lib, "fs_render_scaled", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_build_bla", u32, [vp, C.c_int, vp])
    _decl(lib, "fs_bla_num_levels", i32, [vp])
    _decl(lib, "fs_bla_lm2", i32, [vp])
    _decl(lib, "fs_bla_level_size", u64, [vp, i32])
    _decl(lib, "fs_read_bla_level", u32, [vp, i32, vp, u64])
    _decl(lib, "fs_render_direct_lp", u32, [vp, C.c_int, vp, u64, C.c_int])
    _decl(lib, "fs_clear", u32, [vp])
    _decl(lib, "fs_render_current", u32, [vp, u64, vp, vp, vp, C.c_int])
    _decl(lib, "fs_sync_compute", u32, [vp])
    _decl(lib, "fs_compute_stream", vp, [vp])
    _decl(lib, "fs_sync_display", u32, [vp])
    _decl(lib, "fs_query_compute", u32, [vp])
    _decl(lib, "fs_enqueue_done_callback", u32, [vp, DONE_CB, vp])
    _decl(lib, "fs_get_width", u32, [vp])
    _decl(lib, "fs_get_height", u32, [vp])
    _decl(lib, "fs_last_kernel_ms", C.c_float, [vp])
    _decl(lib, "fs_set_kernel_variant", u32, [vp, C.c_int])
    _decl(lib, "fs_kernel_ms_history", u32, [vp, vp, u32])
    _decl(lib, "fs_kernel_ms_split_history", u32, [vp, vp, vp, u32])
    _decl(lib, "fs_forget_tile_costs", u32, [vp])
    _decl(lib, "fs_last_frame_tile_ordered", C.c_int, [vp])
    _decl(lib, "fs_last_frame_sampled_tile_order", C.c_int, [vp])
    _decl(lib, "fs_read_tile_costs", u32, [vp, vp, u64, vp])
    _decl(lib, "fs_read_tile_order", u32, [vp, vp, u64])
    _decl(lib, "fs_seq_cursor_probe", u32, [vp, C.c_int, u64, u32, vp])
    _decl(lib, "fs_enable_step_count", u32, [vp, C.c_int])
    _decl(lib, "fs_read_step_count", u32, [vp, vp])
    _decl(lib, "fs_time_render_current", u32, [vp, u64, u32, vp])
    _decl(lib, "fs_read_stats_raw", u32, [vp, vp, u64])
    _decl(lib, "fs_test_block_threshold", u32, [vp, vp, vp, vp, vp, u32])
    _decl(lib, "fs_build_la", u32, [vp, C.c_int, vp, C.c_int])
    _decl(lib, "fs_build_la_mt", u32, [vp, C.c_int, vp, C.c_int, C.c_int])
    _decl(lib, "fs_la_counts", u32, [vp, vp, vp, vp, vp])
    _decl(lib, "fs_read_la", u32, [vp, vp, u32, vp, u32, vp])
    _decl(lib, "fs_group_create", vp, [vp, C.c_int, C.c_int])
    _decl(lib, "fs_group_destroy", None, [vp])
    _decl(lib, "fs_group_size", C.c_int, [vp])
    _decl(lib, "fs_group_transport", C.c_int, [vp])
    _decl(lib, "fs_group_renderer", vp, [vp, C.c_int])
    _decl(lib, "fs_group_init_memory", u32, [vp, u32, u32, u32, u32, vp, u32, u32, u64])
    _decl(lib, "fs_group_upload_orbit", u32, [vp, u64, C.c_int, u32, vp, u64, u64, u64])
    _decl(lib, "fs_group_upload_orbit_compressed", u32, [vp, u64, C.c_int, u32, vp, u64, u64, u64, vp, vp])
    _decl(lib, "fs_group_upload_la", u32, [vp, u64, C.c_int, u32, vp, u32, vp, u32, C.c_int, C.c_int, vp])
    _decl(lib, "fs_group_upload_bla", u32, [vp, C.c_int, vp, vp, i32, i32])
    _decl(lib, "fs_group_upload_orbit_scaled", u32, [vp, C.c_int, u32, vp, vp, u64, u64])
    _decl(lib, "fs_group_render_lav2", u32, [vp, C.c_int, C.c_int, C.c_int, vp, u64])
    _decl(lib, "fs_group_render_bla", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_group_render_scaled", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_group_render_direct", u32, [vp, C.c_int, vp, u64])
    _decl(lib, "fs_group_clear", u32, [vp])
    _decl(lib, "fs_group_render_current", u32, [vp, u64, vp, vp])
    _decl(lib, "fs_group_render_current_colors", u32, [vp, u64, vp, vp, vp, C.c_int])
    _decl(lib, "fs_group_sync_display", u32, [vp])
    _decl(lib, "fs_display_stream", vp, [vp])
    _decl(lib, "fs_colorize_frame", u32, [vp, vp, u64, vp, vp, vp])
    _decl(lib, "fs_color_buffer_elements", u64, [vp])
    _decl(lib, "fs_group_sync", u32, [vp])
    _decl(lib, "fs_group_wait_current", u32, [vp, u32])
    _decl(lib, "fs_group_gather_ms", C.c_float, [vp])
    _decl(lib, "fs_group_plan", None, [u32, u32, u32, u32, vp, vp, vp])
    _decl(lib, "fs_group_set_host_path", u32, [vp, C.c_int])
    _decl(lib, "fs_group_host_path", C.c_int, [vp])
    _decl(lib, "fs_copy_bands_to_host", u32, [vp, vp, vp, vp])
    _decl(lib, "fs_host_register", u32, [vp, u64])
    _decl(lib, "fs_host_unregister", u32, [vp])
    _render = lib
    return lib


RENDER_SYMBOLS = [
    "fs_create", "fs_destroy", "fs_test_device_is_working", "fs_device_count", "fs_error_string", "fs_init_memory", "fs_set_row_bands",
    "fs_local_rows", "fs_set_external_iter_buffer", "fs_device_iter_buffer", "fs_rounded_width", "fs_upload_orbit", "fs_upload_orbit_compressed",
    "fs_upload_la", "fs_upload_bla", "fs_render_lav2", "fs_render_bla", "fs_render_direct", "fs_upload_orbit_scaled",
    "fs_render_scaled", "fs_build_bla", "fs_bla_num_levels", "fs_bla_lm2", "fs_bla_level_size", "fs_read_bla_level",
    "fs_render_direct_lp", "fs_clear",
    "fs_render_current", "fs_sync_compute", "fs_compute_stream", "fs_sync_display", "fs_query_compute", "fs_enqueue_done_callback",
    "fs_host_fallback_bytes", "fs_idle_device_bytes", "fs_release_idle_device_memory", "fs_set_compressed_orbit_mode", "fs_orbit_device_bytes", "fs_get_width", "fs_get_height", "fs_last_kernel_ms", "fs_set_kernel_variant", "fs_kernel_ms_history", "fs_kernel_ms_split_history", "fs_forget_tile_costs", "fs_last_frame_tile_ordered", "fs_last_frame_sampled_tile_order", "fs_read_tile_costs", "fs_read_tile_order", "fs_seq_cursor_probe", "fs_enable_step_count", "fs_read_step_count",
    "fs_time_render_current", "fs_read_stats_raw", "fs_test_block_threshold", "fs_build_la", "fs_build_la_mt", "fs_la_counts", "fs_read_la",
    "fs_group_create", "fs_group_destroy", "fs_group_size", "fs_group_transport", "fs_group_renderer", "fs_group_init_memory",
    "fs_group_upload_orbit", "fs_group_upload_orbit_compressed", "fs_group_upload_la", "fs_group_upload_bla",
    "fs_group_upload_orbit_scaled", "fs_group_render_lav2", "fs_group_render_bla", "fs_group_render_scaled",
    "fs_group_render_direct", "fs_group_clear", "fs_group_render_current", "fs_group_render_current_colors", "fs_group_sync_display", "fs_display_stream", "fs_colorize_frame",
    "fs_color_buffer_elements", "fs_group_sync", "fs_group_wait_current", "fs_group_gather_ms",
    "fs_group_plan", "fs_group_set_host_path", "fs_group_host_path", "fs_copy_bands_to_host", "fs_host_register", "fs_host_unregister",
]


def inputs_lib():
    """libfsinputs.so (GMP host builders)."""
    global _inputs
    if _inputs is not None:
        return _inputs
    path = _build.LIB_INPUTS
    if not os.path.exists(path):
        raise RuntimeError("libfsinputs.so is missing (%s): run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    _decl(lib, "fsh_view_create", vp, [C.c_char_p] * 4 + [u32, u32])
    _decl(lib, "fsh_view_destroy", None, [vp])
    _decl(lib, "fsh_view_save_im", C.c_int, [vp, u64, C.c_char_p, C.c_int])
    _decl(lib, "fsh_view_load_im", vp, [C.c_char_p, u32, u32, C.POINTER(u64), C.POINTER(C.c_int), C.POINTER(C.c_int)])
    _decl(lib, "fsh_view_precision_bits", u64, [vp])
    _decl(lib, "fsh_view_bbox_str", C.c_int, [vp, C.c_int, C.c_char_p, C.c_size_t])
    _decl(lib, "fsh_view_coords_direct_f64", None, [vp, u32, u32, vp])
    _decl(lib, "fsh_view_coords_direct_hdr32", None, [vp, u32, u32, vp])
    _decl(lib, "fsh_view_coords_direct_hdr64", None, [vp, u32, u32, vp])
    _decl(lib, "fsh_orbit_create", vp, [vp, C.c_int, u64, C.c_int])
    _decl(lib, "fsh_orbit_create_ex", vp, [vp, C.c_int, u64, C.c_int, C.c_int])
    _decl(lib, "fsh_orbit_is_compressed", C.c_int, [vp])
    _decl(lib, "fsh_orbit_compressed_count", u64, [vp])
    _decl(lib, "fsh_orbit_compressed_data_hdr32", vp, [vp])
    _decl(lib, "fsh_orbit_low_hdr32", None, [vp, vp])
    _decl(lib, "fsh_orbit_compressed_data_hdr64", vp, [vp])
    _decl(lib, "fsh_orbit_low_hdr64", None, [vp, vp])
    _decl(lib, "fsh_la_default_params", None, [C.POINTER(C.c_int32)])
    _decl(lib, "fsh_mpz_raw_write", C.c_size_t, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t])
    _decl(lib, "fsh_mpz_raw_read", C.c_size_t, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t])
    _decl(lib, "fsh_orbit_save_im", C.c_int, [vp, u64, C.c_int, C.c_char_p, C.c_int])
    _decl(lib, "fsh_orbit_load_im", vp, [C.c_char_p, C.POINTER(u64)])
    _decl(lib, "fsh_orbit_is64", C.c_int, [vp])
    _decl(lib, "fsh_orbit_destroy", None, [vp])
    _decl(lib, "fsh_orbit_count", u64, [vp])
    _decl(lib, "fsh_orbit_scale_entries", u64, [vp, vp, vp, u64])
    _decl(lib, "fsh_orbit_period", u64, [vp])
    _decl(lib, "fsh_orbit_data_hdr32", vp, [vp])
    _decl(lib, "fsh_orbit_data_hdr64", vp, [vp])
    _decl(lib, "fsh_orbit_max_radius_hdr32", None, [vp, vp])
    _decl(lib, "fsh_orbit_max_radius_hdr64", None, [vp, vp])
    _decl(lib, "fsh_orbit_data_hdr32_bad", vp, [vp])
    _decl(lib, "fsh_orbit_data_f32_bad", vp, [vp])
    _decl(lib, "fsh_orbit_bad_count", u64, [vp])
    _decl(lib, "fsh_view_coords_direct_lp", None, [vp, u32, u32, C.c_int, vp])
    _decl(lib, "fsh_view_coords_perturb_hdr32", None, [vp, vp, u32, u32, vp])
    _decl(lib, "fsh_view_coords_perturb_hdr64", None, [vp, vp, u32, u32, vp])
    _decl(lib, "fsh_df32_op", None, [C.c_int, vp, vp, vp])
    _decl(lib, "fsh_hr2_reduce", None, [vp])
    _decl(lib, "fsh_hr2_add", None, [vp, vp, C.c_int, vp])
    _decl(lib, "fsh_hc2_reduce", None, [vp])
    _decl(lib, "fsh_convert_orbit_hdr64_to_2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_convert_la_hdr64_to_2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_convert_at_hdr64_to_2x32", None, [vp, vp])
    _decl(lib, "fsh_view_coords_perturb_2x32", None, [vp, vp, u32, u32, vp])
    _decl(lib, "fsh_la_create_hdr32", vp, [vp, C.c_int])
    _decl(lib, "fsh_la_create", vp, [vp, C.c_int])
    _decl(lib, "fsh_la_create_ex", vp, [vp, C.c_int, C.c_int])
    _decl(lib, "fsh_la_is64", C.c_int, [vp])
    _decl(lib, "fsh_la_destroy", None, [vp])
    _decl(lib, "fsh_la_count", u32, [vp])
    _decl(lib, "fsh_la_data", vp, [vp])
    _decl(lib, "fsh_la_stage_count", u32, [vp])
    _decl(lib, "fsh_la_stages", vp, [vp])
    _decl(lib, "fsh_la_is_valid", C.c_int, [vp])
    _decl(lib, "fsh_la_use_at", C.c_int, [vp])
    _decl(lib, "fsh_la_at", None, [vp, vp])
    _decl(lib, "fsh_bla_create_hdr32", vp, [vp])
    _decl(lib, "fsh_bla_create", vp, [vp])
    _decl(lib, "fsh_bla_destroy", None, [vp])
    _decl(lib, "fsh_bla_num_levels", i32, [vp])
    _decl(lib, "fsh_bla_lm2", i32, [vp])
    _decl(lib, "fsh_bla_level_ptrs", vp, [vp])
    _decl(lib, "fsh_bla_level_sizes", vp, [vp])
    _decl(lib, "fsh_plain_create", vp, [vp, C.c_int, u64, C.c_int, C.c_int])
    _decl(lib, "fsh_plain_create_ex", vp, [vp, C.c_int, u64, C.c_int, C.c_int, C.c_int])
    _decl(lib, "fsh_plain_save_im", C.c_int, [vp, u64, C.c_int, C.c_char_p, C.c_int])
    _decl(lib, "fsh_plain_load_im", vp, [C.c_char_p, C.POINTER(u64), C.c_int])
    _decl(lib, "fsh_plain_is_compressed", C.c_int, [vp])
    _decl(lib, "fsh_plain_compressed_count", u64, [vp])
    _decl(lib, "fsh_plain_compressed_data", vp, [vp])
    _decl(lib, "fsh_plain_orbit_low", None, [vp, vp])
    _decl(lib, "fsh_convert_orbit_rc_f64_to_p2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_convert_orbit_rc_hdr64_to_2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_orbit_low_2x32", None, [vp, vp])
    _decl(lib, "fsh_plain_destroy", None, [vp])
    _decl(lib, "fsh_plain_kind", C.c_int, [vp])
    _decl(lib, "fsh_plain_orbit_count", u64, [vp])
    _decl(lib, "fsh_plain_orbit_period", u64, [vp])
    _decl(lib, "fsh_plain_orbit_data", vp, [vp])
    _decl(lib, "fsh_plain_la_count", u32, [vp])
    _decl(lib, "fsh_plain_la_data", vp, [vp])
    _decl(lib, "fsh_plain_la_stage_count", u32, [vp])
    _decl(lib, "fsh_plain_la_stages", vp, [vp])
    _decl(lib, "fsh_plain_la_is_valid", C.c_int, [vp])
    _decl(lib, "fsh_plain_la_use_at", C.c_int, [vp])
    _decl(lib, "fsh_plain_la_at", None, [vp, vp])
    _decl(lib, "fsh_plain_coords", None, [vp, vp, u32, u32, vp])
    _decl(lib, "fsh_convert_orbit_f64_to_p2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_convert_la_f64_to_p2x32", None, [vp, u64, vp])
    _decl(lib, "fsh_convert_at_f64_to_p2x32", None, [vp, vp])
    _decl(lib, "fsh_convert_coords_f64_to_p2x32", None, [vp, vp])
    _decl(lib, "fsh_orbit_f64_create", vp, [vp, u64, C.c_int])
    _decl(lib, "fsh_orbit_f64_destroy", None, [vp])
    _decl(lib, "fsh_orbit_f64_count", u64, [vp])
    _decl(lib, "fsh_orbit_f64_period", u64, [vp])
    _decl(lib, "fsh_orbit_f64_data", vp, [vp])
    _decl(lib, "fsh_orbit_f64_data_bad", vp, [vp])
    _decl(lib, "fsh_orbit_f64_data_f32_bad", vp, [vp])
    _decl(lib, "fsh_orbit_f64_bla_num_levels", i32, [vp])
    _decl(lib, "fsh_orbit_f64_bla_lm2", i32, [vp])
    _decl(lib, "fsh_orbit_f64_bla_level_ptrs", vp, [vp])
    _decl(lib, "fsh_orbit_f64_bla_level_sizes", vp, [vp])
    _decl(lib, "fsh_view_coords_perturb_f64", None, [vp, vp, u32, u32, vp])
    _inputs = lib
    return lib
