"""Loader and thin Python face of libansel_hip.so (the C-ABI of include/ansel_hip.h).

There is no CPU fallback: if the library is missing or no MI355X is visible, calls raise."""
import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
# ANSEL_HIP_LIB: another build of the same library -- tools/ load the measuring build (ansel_amd/build.py --measuring), whose
# A/B switches the product library does not have
LIB_PATH = os.environ.get("ANSEL_HIP_LIB") or os.path.join(HERE, "libansel_hip.so")

_lib = None


class AnselHipError(RuntimeError):
    pass


def load():
    """dlopen libansel_hip.so and declare prototypes.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if os.environ.get("ANSEL_HIP_LIB"):
            raise AnselHipError(
                "ANSEL_HIP_LIB=%s does not exist: the measuring build is made by `python -m ansel_amd.build --measuring` "
                "(tools/ default to it; unset the variable for the product library)" % LIB_PATH)
        raise AnselHipError(
            "%s not found: build it with `python -m ansel_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime; a process that uses both (bench.py, the row-band driver) must have
    # torch's loaded first, or torch.cuda finds no device afterwards.  C callers never get here.
    if os.environ.get("ANSEL_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # no torch (a plain ctypes user), or one that does not load: the library does not need it
            pass
    lib = C.CDLL(LIB_PATH)
    vp, i, u, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    P = C.POINTER
    protos = {
        "dt_hip_init": (i, []),
        "dt_hip_cleanup": (None, []),
        "dt_hip_is_inited": (i, []),
        "dt_hip_get_num_devices": (i, []),
        "dt_hip_get_device_name": (C.c_char_p, [i]),
        "dt_hip_get_device_available": (sz, [i]),
        "dt_hip_get_device_memalloc": (sz, [i]),
        "dt_hip_reserve_device_for_pipe": (i, [i]),
        "dt_hip_reserve_device_by_id": (None, [i]),
        "dt_hip_try_reserve_device_by_id": (i, [i]),
        "dt_hip_release_device": (None, [i]),
        "dt_hip_get_device_max_image_size": (i, [i, P(i), P(i)]),
        "dt_hip_get_device_max_global_mem": (sz, [i]),
        "dt_hip_report_pipe_error": (i, []),
        "dt_hip_is_enabled": (i, []),
        "dt_hip_update_settings": (i, []),
        "dt_hip_check_tuning": (None, [i]),
        "dt_hip_avoid_atomics": (i, [i]),
        "dt_hip_micro_nap": (i, [i]),
        "dt_hip_use_pinned_memory": (i, [i]),
        "dt_hip_dev_roundup_width": (i, [i, i]),
        "dt_hip_dev_roundup_height": (i, [i, i]),
        "dt_hip_image_fits_device_reason": (i, [i, sz, sz, C.c_uint, C.c_float, sz, P(sz), P(sz)]),
        "dt_hip_get_image_width": (i, [vp]),
        "dt_hip_get_image_height": (i, [vp]),
        "dt_hip_get_image_element_size": (i, [vp]),
        "dt_hip_get_mem_context_id": (i, [vp]),
        "dt_hip_alloc_device_use_host_pointer": (vp, [i, i, i, i, vp, i]),
        "dt_hip_map_buffer": (vp, [i, vp, i, i, sz, sz]),
        "dt_hip_map_image": (vp, [i, vp, i, i, sz, sz, i]),
        "dt_hip_unmap_mem_object": (i, [i, vp, vp]),
        "dt_hip_copy_host_to_device": (vp, [i, vp, i, i, i]),
        "dt_hip_copy_host_to_device_rowpitch": (vp, [i, vp, i, i, i, i]),
        "dt_hip_copy_host_to_device_constant": (vp, [i, sz, vp]),
        "dt_hip_copy_device_to_host": (i, [i, vp, vp, i, i, i]),
        "dt_hip_read_buffer_from_device": (i, [i, vp, vp, sz, sz, i]),
        "dt_hip_write_buffer_to_device": (i, [i, vp, vp, sz, sz, i]),
        "dt_hip_enqueue_barrier": (i, [i]),
        "dt_hip_events_wait_for": (None, [i]),
        "dt_hip_events_flush": (i, [i, i]),
        "dt_hip_image_fits_device": (i, [i, sz, sz, C.c_uint, C.c_float, sz]),
        "dt_hip_get_stream": (vp, [i]),
        "dt_hip_set_stream": (i, [i, vp]),
        "dt_hip_alloc_device": (vp, [i, i, i, i]),
        "dt_hip_alloc_device_buffer": (vp, [i, sz]),
        "dt_hip_release_mem_object": (None, [vp]),
        "dt_hip_get_mem_object_size": (sz, [vp]),
        "dt_hip_memory_statistics": (None, [i, P(sz), P(sz)]),
        "dt_hip_write_host_to_device": (i, [i, vp, vp, i, i, i]),
        "dt_hip_write_host_to_device_rowpitch": (i, [i, vp, vp, i, i, i, sz, i]),
        "dt_hip_read_host_from_device": (i, [i, vp, vp, i, i, i]),
        "dt_hip_read_host_from_device_rowpitch": (i, [i, vp, vp, i, i, i, sz, i]),
        "dt_hip_enqueue_copy_buffer_to_buffer": (i, [i, vp, vp, sz, sz, sz]),
        "dt_hip_read_host_from_device_raw": (i, [i, vp, vp, P(sz), P(sz), i, i]),
        "dt_hip_write_host_to_device_raw": (i, [i, vp, vp, P(sz), P(sz), i, i]),
        "dt_hip_enqueue_copy_image": (i, [i, vp, vp, P(sz), P(sz), P(sz)]),
        "dt_hip_enqueue_copy_region": (i, [i, vp, i, i, i, vp, i, i, i, i, i, i]),
        "dt_hip_finish": (i, [i]),
        "dt_hip_events_enable": (None, [i, i]),
        "dt_hip_events_reset": (None, [i]),
        "dt_hip_events_profiling": (i, [i, P(C.c_char_p), P(C.c_float), P(i), i]),
        "dt_hip_last_error": (C.c_char_p, []),
        "dt_hip_abi_sizeof": (sz, [C.c_char_p]),
        "dt_hip_crop_dcraw_filters": (u, [u, u, u]),
        "dt_hip_iop_rawprepare_process": (i, [i, P(abi.Piece), P(abi.RawprepareData), vp, vp]),
        "dt_hip_iop_temperature_process": (i, [i, P(abi.Piece), P(abi.TemperatureData), vp, vp]),
        "dt_hip_iop_highlights_process": (i, [i, P(abi.Piece), P(abi.HighlightsData), vp, vp]),
        "dt_hip_iop_demosaic_process": (i, [i, P(abi.Piece), P(abi.DemosaicData), vp, vp]),
        "dt_hip_iop_demosaic_tiling": (None, [P(abi.Piece), P(abi.DemosaicData), P(abi.Tiling)]),
        "dt_hip_iop_exposure_process": (i, [i, P(abi.Piece), P(abi.ExposureData), vp, vp]),
        "dt_hip_iop_colorin_process": (i, [i, P(abi.Piece), P(abi.Conversion), vp, vp]),
        "dt_hip_iop_colorout_process": (i, [i, P(abi.Piece), P(abi.Conversion), vp, vp]),
        "dt_hip_iop_channelmixerrgb_process": (i, [i, P(abi.Piece), P(abi.ChannelmixerrgbData), vp, vp]),
        "dt_hip_iop_filmicrgb_process": (i, [i, P(abi.Piece), P(abi.FilmicrgbData), vp, vp]),
        "dt_hip_iop_diffuse_process": (i, [i, P(abi.Piece), P(abi.DiffuseData), vp, vp]),
        "dt_hip_iop_denoiseprofile_process": (i, [i, P(abi.Piece), P(abi.DenoiseprofileData), vp, vp]),
        "dt_hip_iop_nlmeans_process": (i, [i, P(abi.Piece), P(abi.NlmeansData), vp, vp]),
        "dt_hip_iop_detailmask_process": (i, [i, P(abi.Piece), P(abi.DetailmaskData), vp, vp]),
        "dt_hip_transform_rgb_to_lab": (i, [i, P(abi.Piece), P(abi.LabData), vp, vp]),
        "dt_hip_transform_lab_to_rgb": (i, [i, P(abi.Piece), P(abi.LabData), vp, vp]),
        "dt_hip_iop_bilat_process": (i, [i, P(abi.Piece), P(abi.BilatData), vp, vp]),
        "dt_hip_iop_finalscale_process": (i, [i, P(abi.Piece), P(abi.FinalscaleData), vp, vp]),
        "dt_hip_iop_initialscale_process": (i, [i, P(abi.Piece), P(abi.FinalscaleData), vp, vp]),
        "dt_hip_raw_unpack": (i, [i, vp, i, i, C.c_size_t, i, i, vp]),
        "dt_hip_develop_blend_process": (i, [i, P(abi.Piece), P(abi.BlendData), vp, vp]),
        "dt_hip_iop_basebuffer_process": (i, [i, P(abi.Piece), i, i, i, vp, vp]),
        "dt_hip_alloc_host_pinned": (vp, [sz]),
        "dt_hip_free_host_pinned": (None, [vp]),
        "dt_hip_is_pinned_memory": (i, [vp]),
        "dt_hip_plan_tiles_ptp": (i, [i, i, i, i, P(abi.Tiling), C.c_uint, sz, sz, i, i, P(abi.TilePlan)]),
        "dt_hip_default_process_tiling_ptp": (i, [i, C.c_char_p, P(abi.Piece), vp, sz, P(abi.Tiling), vp, vp, i, i, sz]),
        "dt_hip_default_tiling": (None, [P(abi.Piece), i, P(abi.Tiling)]),
        "dt_hip_plan_tiles_roi": (i, [P(abi.Roi), P(abi.Roi), i, i, P(abi.Tiling), C.c_uint, sz, sz, i, i, P(abi.TilePlanRoi)]),
        "dt_hip_tile_rois_finalscale": (i, [P(abi.TilePlanRoi), P(abi.Roi), P(abi.Roi), i, i, P(abi.Roi), P(abi.Roi), P(abi.Roi)]),
        "dt_hip_default_process_tiling_roi": (i, [i, C.c_char_p, P(abi.Piece), vp, sz, P(abi.Tiling), vp, vp, i, i, sz]),
        "dt_hip_iop_denoiseprofile_tiling": (None, [P(abi.Piece), P(abi.DenoiseprofileData), P(abi.Tiling)]),
        "dt_hip_iop_nlmeans_tiling": (None, [P(abi.Piece), P(abi.NlmeansData), P(abi.Tiling)]),
        "dt_hip_iop_bilat_tiling": (None, [P(abi.Piece), P(abi.BilatData), P(abi.Tiling)]),
        "dt_hip_iop_diffuse_tiling": (None, [P(abi.Piece), P(abi.DiffuseData), P(abi.Tiling)]),
        "dt_hip_export_convert_u16": (i, [i, i, i, vp, vp]),
        "dt_hip_export_convert_u8": (i, [i, i, i, vp, vp]),
        "dt_hip_export_pack_rows": (i, [i, i, i, i, i, vp, vp]),
        "dt_hip_pipe_new": (vp, [i]),
        "dt_hip_pipe_free": (None, [vp]),
        "dt_hip_pipe_add_node": (i, [vp, C.c_char_p, P(abi.Piece), vp, sz]),
        "dt_hip_pipe_set_fusion": (None, [vp, i]),
        "dt_hip_pipe_num_groups": (i, [vp]),
        "dt_hip_pipe_process": (i, [vp, vp, vp]),
        "dt_hip_batch_new": (vp, [vp, i, sz, sz]),
        "dt_hip_batch_free": (None, [vp]),
        "dt_hip_batch_submit": (i, [vp, vp, vp]),
        "dt_hip_batch_wait": (i, [vp, i]),
        "dt_hip_batch_drain": (i, [vp]),
        "dt_hip_batch_set_writer": (i, [vp, vp, vp]),
        "dt_hip_plan_bands": (i, [i, i, i, i, P(abi.Band)]),
        "dt_hip_band_halo_rows": (i, [C.c_char_p, P(abi.Piece), vp, sz]),
        "dt_hip_pipe_band_begin": (i, [vp, P(abi.Band), vp, P(abi.BandState)]),
        "dt_hip_pipe_band_resolve": (i, [vp, P(abi.Band), P(abi.BandState)]),
        "dt_hip_pipe_band_finish": (i, [vp, P(abi.Band), P(abi.BandState), vp]),
        "dt_hip_pipe_band_relay": (i, [vp, P(abi.Band), P(abi.BandState)]),
        "dt_hip_pipe_band_abort": (None, [vp, P(abi.BandState)]),
        "dt_hip_pipe_process_bands": (i, [P(vp), i, P(abi.Band), P(vp), P(vp)]),
        "dt_hip_pipe_bands_stats": (None, [P(abi.BandStats)]),
        "dt_hip_peer_selftest": (i, [P(i), i]),
        "dt_hip_iop_highlights_process_deferred": (i, [i, P(abi.Piece), P(abi.HighlightsData), vp, vp, vp]),
        "dt_hip_iop_highlights_resolve": (i, [i, vp, vp]),
        "dt_hip_test_dispatch": (i, [C.c_char_p, i]),
    }
    missing = []
    for name, (res, args) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    lib._ansel_missing = missing
    lib._ansel_protos = protos
    _lib = lib
    return lib


def test_dispatch(key, value):
    """dt_hip_test_dispatch() (csrc/testhooks.hip): send launches to a fallback kernel the library carries anyway -- "nlm_v2",
    "nlm_fused", "amaze_unfused", "amaze_slab", "amaze_blocks" -- on frames where the primary kernel applies; 0 clears"""
    check(load().dt_hip_test_dispatch(key.encode(), int(value)), "dt_hip_test_dispatch(%s)" % key)


def check(rc, what=""):
    if rc != abi.DT_HIP_SUCCESS:
        msg = load().dt_hip_last_error()
        raise AnselHipError("%s failed with %d: %s" % (what, rc, msg.decode() if msg else ""))


def init():
    """dt_hip_init(): raises when no GPU is visible -- the product path fails loudly."""
    lib = load()
    if lib._ansel_missing:
        raise AnselHipError("libansel_hip.so lacks symbols declared in include/ansel_hip.h: %s"
                            % ", ".join(lib._ansel_missing))
    check(lib.dt_hip_init(), "dt_hip_init")
    return lib


class DeviceBuffer:
    """A dt_hip_alloc_device_buffer() allocation with numpy upload/download helpers."""

    def __init__(self, devid, nbytes):
        self.lib = load()
        self.devid = devid
        self.nbytes = int(nbytes)
        self.ptr = self.lib.dt_hip_alloc_device_buffer(devid, self.nbytes)
        if not self.ptr:
            raise AnselHipError("device allocation of %d bytes failed: %s"
                                % (self.nbytes, self.lib.dt_hip_last_error().decode()))

    @classmethod
    def from_numpy(cls, devid, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        b = cls(devid, arr.nbytes)
        # the size_t entry point: a 201 MP float4 plane is 3.2 GB, more than the int width of the image calls holds
        check(b.lib.dt_hip_write_buffer_to_device(devid, arr.ctypes.data_as(C.c_void_p), b.ptr, 0, arr.nbytes, 1),
              "write_buffer_to_device")
        return b

    def upload(self, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.lib.dt_hip_write_buffer_to_device(self.devid, arr.ctypes.data_as(C.c_void_p), self.ptr, 0, arr.nbytes, 1),
              "write_buffer_to_device")

    def to_numpy(self, shape, dtype):
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self.lib.dt_hip_read_buffer_from_device(self.devid, out.ctypes.data_as(C.c_void_p), self.ptr, 0,
                                                      out.nbytes, 1), "read_buffer_from_device")
        return out

    def release(self):
        if self.ptr:
            self.lib.dt_hip_release_mem_object(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
