"""ctypes mirror of include/ansel_hip.h (the C-ABI of libansel_hip).

Field order and types follow the header one to one; tests/test_abi.py checks the
sizes against the compiled library (dt_hip_abi_sizeof)."""
import ctypes as C

DT_HIP_SUCCESS = 0
DT_HIP_DEFAULT_ERROR = -999
DT_HIP_SYSMEM_ALLOCATION = -998
DT_HIP_INVALID_ARG = -997

DT_HIP_TYPE_FLOAT = 1
DT_HIP_TYPE_UINT16 = 2

DT_HIP_HIGHLIGHTS_CLIP = 0

DT_HIP_DEMOSAIC_PPG = 0
DT_HIP_DEMOSAIC_AMAZE = 1
DT_HIP_DEMOSAIC_RCD = 5
DT_HIP_DEMOSAIC_VNG4 = 2
DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME = 3
DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR = 4
DT_HIP_DEMOSAIC_DUAL = 2048

DT_HIP_ADAPTATION_LINEAR_BRADFORD = 0
DT_HIP_ADAPTATION_CAT16 = 1
DT_HIP_ADAPTATION_FULL_BRADFORD = 2
DT_HIP_ADAPTATION_XYZ = 3
DT_HIP_ADAPTATION_RGB = 4

DT_HIP_LUT_SAMPLES = 0x10000

f4 = C.c_float * 4
f3 = C.c_float * 3
f5 = C.c_float * 5
m34 = (C.c_float * 4) * 3
m33 = (C.c_float * 3) * 3


class Roi(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("scale", C.c_double)]

    @classmethod
    def make(cls, x, y, w, h, scale=1.0):
        return cls(int(x), int(y), int(w), int(h), float(scale))


class Piece(C.Structure):
    _fields_ = [("roi_in", Roi), ("roi_out", Roi), ("filters", C.c_uint32), ("channels", C.c_uint32),
                ("datatype", C.c_uint32), ("_pad", C.c_uint32), ("processed_maximum", f4)]

    @classmethod
    def make(cls, width, height, filters=0, channels=4, datatype=DT_HIP_TYPE_FLOAT,
             processed_maximum=(1.0, 1.0, 1.0, 1.0), roi_in=None, roi_out=None):
        p = cls()
        p.roi_in = roi_in if roi_in is not None else Roi.make(0, 0, width, height)
        p.roi_out = roi_out if roi_out is not None else Roi.make(0, 0, width, height)
        p.filters = filters
        p.channels = channels
        p.datatype = datatype
        p.processed_maximum = f4(*processed_maximum)
        return p


class Tiling(C.Structure):
    _fields_ = [("factor", C.c_float), ("factor_cl", C.c_float), ("maxbuf", C.c_float),
                ("maxbuf_cl", C.c_float), ("overhead", C.c_uint), ("overlap", C.c_uint),
                ("xalign", C.c_uint), ("yalign", C.c_uint)]


class RawprepareData(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("sub", f4), ("div", f4)]


class TemperatureData(C.Structure):
    _fields_ = [("coeffs", f4)]


class HighlightsData(C.Structure):
    _fields_ = [("mode", C.c_int), ("clip", C.c_float)]


class DemosaicData(C.Structure):
    _fields_ = [("green_eq", C.c_uint32), ("color_smoothing", C.c_uint32),
                ("demosaicing_method", C.c_uint32), ("median_thrs", C.c_float), ("green_eq_threshold", C.c_float),
                ("dual_thrs", C.c_float), ("wb_coeffs", C.c_float * 4)]


class ExposureData(C.Structure):
    _fields_ = [("black", C.c_float), ("scale", C.c_float)]


class DiffuseData(C.Structure):
    """dt_hip_diffuse_data_t == dt_iop_diffuse_params_t (src/iop/diffuse.c:76-105) + pipe->iscale"""
    _fields_ = [("iterations", C.c_int), ("sharpness", C.c_float), ("radius", C.c_int),
                ("regularization", C.c_float), ("variance_threshold", C.c_float),
                ("anisotropy_first", C.c_float), ("anisotropy_second", C.c_float),
                ("anisotropy_third", C.c_float), ("anisotropy_fourth", C.c_float), ("threshold", C.c_float),
                ("first", C.c_float), ("second", C.c_float), ("third", C.c_float), ("fourth", C.c_float),
                ("radius_center", C.c_int), ("iscale", C.c_float)]


class DenoiseprofileData(C.Structure):
    """dt_hip_denoiseprofile_data_t: the fields of dt_iop_denoiseprofile_data_t (src/iop/denoiseprofile.c:352-371)
    the wavelets path reads + the white-balance coefficients of the input buffer descriptor"""
    _fields_ = [("radius", C.c_float), ("nbhood", C.c_float), ("strength", C.c_float), ("shadows", C.c_float),
                ("bias", C.c_float), ("scattering", C.c_float), ("central_pixel_weight", C.c_float),
                ("overshooting", C.c_float), ("a", f3), ("b", f3), ("mode", C.c_int),
                ("force", (C.c_float * 7) * 6), ("wb_adaptive_anscombe", C.c_int),
                ("fix_anscombe_and_nlmeans_norm", C.c_int), ("use_new_vst", C.c_int),
                ("wavelet_color_mode", C.c_int), ("wb_coeffs", f4)]


DT_HIP_DENOISEPROFILE_WAVELETS = 1
DT_HIP_DENOISEPROFILE_RGB = 0
DT_HIP_DENOISEPROFILE_Y0U0V0 = 1


class BilatData(C.Structure):
    """dt_hip_bilat_data_t == dt_iop_bilat_params_t (src/iop/bilat.c:78-86) + pipe->iscale"""
    _fields_ = [("mode", C.c_int), ("sigma_r", C.c_float), ("sigma_s", C.c_float), ("detail", C.c_float),
                ("midtone", C.c_float), ("iscale", C.c_float)]

    @classmethod
    def bilateral(cls, sigma_s=50.0, sigma_r=25.0, detail=0.33, iscale=1.0):
        return cls(0, sigma_r, sigma_s, detail, 0.5, iscale)

    @classmethod
    def local_laplacian(cls, highlights=0.5, shadows=0.5, detail=0.25, midtone=0.5):
        """the module's default mode and $DEFAULT values (src/iop/bilat.c:78-86): sigma_r carries the
        highlights slider, sigma_s the shadows slider"""
        return cls(1, highlights, shadows, detail, midtone, 1.0)


class FinalscaleData(C.Structure):
    """dt_hip_finalscale_data_t: the export interpolator (0 bilinear, 1 bicubic, 2 Mitchell = default)"""
    _fields_ = [("interpolation", C.c_int)]


class LabData(C.Structure):
    """dt_hip_lab_data_t: the 3x3 (rows padded to 4) of the RGB <-> Lab glue, and the tone curves of a work profile
    that has them"""
    _fields_ = [("matrix", m34), ("nonlinearlut", C.c_int), ("unbounded_coeffs", (C.c_float * 3) * 3),
                ("lut", C.c_void_p * 3), ("lut_first", C.c_float * 3)]

    @classmethod
    def make(cls, m, luts=None):
        """luts: three (pointer or None, lut[0], (a, b, c) of the fitted power law) per channel"""
        d = cls()
        set_m34(d.matrix, m)
        for c in range(3):
            d.lut_first[c] = -1.0
        if luts:
            for c, (ptr, first, coeffs) in enumerate(luts):
                d.lut[c] = ptr
                d.lut_first[c] = first if ptr else -1.0
                for k in range(3):
                    d.unbounded_coeffs[c][k] = coeffs[k]
                if ptr and first >= 0.0:
                    d.nonlinearlut += 1
        return d


class NlmeansData(C.Structure):
    """dt_hip_nlmeans_data_t == dt_iop_nlmeans_params_t (src/iop/nlmeans.c:81-88)"""
    _fields_ = [("radius", C.c_float), ("strength", C.c_float), ("luma", C.c_float), ("chroma", C.c_float)]


DT_HIP_DENOISEPROFILE_NLMEANS = 0
DT_HIP_DENOISEPROFILE_NLMEANS_AUTO = 3
DT_HIP_DENOISEPROFILE_WAVELETS_AUTO = 4


class Conversion(C.Structure):
    _fields_ = [("matrix", m34), ("clip_matrix", m34), ("has_clipping", C.c_int),
                ("nonlinear_source", C.c_int), ("nonlinear_target", C.c_int), ("blue_mapping", C.c_int),
                ("coeffs_source", m33), ("coeffs_target", m33),
                ("lut_source", C.c_void_p * 3), ("lut_target", C.c_void_p * 3),
                ("lut_source_first", f3), ("lut_target_first", f3)]


class ChannelmixerrgbData(C.Structure):
    _fields_ = [("XYZ_to_RGB", m34), ("RGB_to_XYZ", m34), ("MIX", m34), ("illuminant", f4),
                ("saturation", f4), ("lightness", f4), ("grey", f4), ("p", C.c_float),
                ("gamut", C.c_float), ("clip", C.c_int), ("apply_grey", C.c_int),
                ("adaptation", C.c_int), ("version", C.c_int)]


class FilmicSpline(C.Structure):
    _fields_ = [("M1", f4), ("M2", f4), ("M3", f4), ("M4", f4), ("M5", f4),
                ("latitude_min", C.c_float), ("latitude_max", C.c_float), ("y", f5), ("x", f5),
                ("type", C.c_int * 2)]


class FilmicrgbData(C.Structure):
    _fields_ = [("white_source", C.c_float), ("grey_source", C.c_float), ("black_source", C.c_float),
                ("dynamic_range", C.c_float), ("saturation", C.c_float), ("output_power", C.c_float),
                ("agx_beta_hue", C.c_float), ("preserve_color", C.c_int), ("version", C.c_int),
                ("use_output_profile", C.c_int), ("spline", FilmicSpline),
                ("work_matrix_in", m34), ("work_matrix_out", m34),
                ("export_matrix_in", m34), ("export_matrix_out", m34)]


def set_m34(dst, rows):
    """fill a float[3][4] from a 3x3 (or 3x4) nested sequence"""
    for r in range(3):
        for c in range(4):
            dst[r][c] = float(rows[r][c]) if c < len(rows[r]) else 0.0


def set_vec(dst, vals):
    for i, v in enumerate(vals):
        dst[i] = float(v)


class TilePlan(C.Structure):
    """dt_hip_tile_plan_t: the tile plan of _default_process_tiling_cl_ptp() (src/develop/tiling.c:868-979)"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tile_wd", C.c_int32), ("tile_ht", C.c_int32),
                ("tiles_x", C.c_int32), ("tiles_y", C.c_int32), ("overlap", C.c_int32)]


class TilePlanRoi(C.Structure):
    """dt_hip_tile_plan_roi_t: the tile grid of _default_process_tiling_cl_roi() (src/develop/tiling.c:1100-1220)"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tile_wd", C.c_int32), ("tile_ht", C.c_int32),
                ("tiles_x", C.c_int32), ("tiles_y", C.c_int32), ("overlap_in", C.c_int32), ("overlap_out", C.c_int32),
                ("delta", C.c_int32), ("xyalign", C.c_int32)]


DT_HIP_TILE_EMPTY = 2
RAW_PACK_MSB, RAW_PACK_LSB = 0, 1  # dt_hip_raw_unpack(): bit order of the packed stream


class ExportRowsData(C.Structure):
    """dt_hip_export_rows_t: the scanline packing of the format writers (tiff.c:293-360)"""
    _fields_ = [("bpp", C.c_int32), ("layers", C.c_int32)]


class Band(C.Structure):
    """dt_hip_band_t: a row band of a frame split over several devices"""
    _fields_ = [("row0", C.c_int32), ("rows", C.c_int32), ("halo_top", C.c_int32), ("halo_bottom", C.c_int32),
                ("tile_row0", C.c_int32), ("tile_row1", C.c_int32)]


class BandState(C.Structure):
    """dt_hip_band_state_t"""
    _fields_ = [("halo_buf", C.c_void_p), ("row_bytes", C.c_size_t), ("clipped_count", C.c_void_p),
                ("priv", C.c_void_p), ("halo_rows", C.c_int32), ("sum_planes", C.c_int32), ("sum_buf", C.c_void_p),
                ("sum_count", C.c_size_t), ("relay_buf", C.c_void_p), ("relay_bytes", C.c_size_t)]


class BandStats(C.Structure):
    """dt_hip_band_stats_t: what the last dt_hip_pipe_process_bands() moved between the devices"""
    _fields_ = [("bands", C.c_int32), ("devices", C.c_int32), ("exchange_stops", C.c_int32),
                ("pairs_without_peer_access", C.c_int32), ("peer_copies", C.c_uint64), ("peer_bytes", C.c_uint64),
                ("host_wait_ns", C.c_uint64)]


DT_HIP_BAND_EXCHANGE = 1
DT_HIP_HIGHLIGHTS_JOURNAL_BYTES = 320


# dt_develop_blend_mode_t (src/develop/blend.h:61-107): the operators of the "RGB (scene)" colourspace
BLEND_NORMAL = 0x18
BLEND_MULTIPLY = 0x04
BLEND_AVERAGE = 0x05
BLEND_ADD = 0x06
BLEND_SUBTRACT = 0x07
BLEND_DIFFERENCE = 0x17
BLEND_LIGHTNESS = 0x10
BLEND_CHROMATICITY = 0x11
BLEND_RGB_R = 0x21
BLEND_RGB_G = 0x22
BLEND_RGB_B = 0x23
BLEND_SUBTRACT_INVERSE = 0x25
BLEND_DIVIDE = 0x26
BLEND_DIVIDE_INVERSE = 0x27
BLEND_GEOMETRIC_MEAN = 0x28
BLEND_HARMONIC_MEAN = 0x29
BLEND_REVERSE = 0x80000000
BLEND_RGB_SCENE_MODES = (BLEND_NORMAL, BLEND_MULTIPLY, BLEND_AVERAGE, BLEND_ADD, BLEND_SUBTRACT, BLEND_DIFFERENCE,
                         BLEND_LIGHTNESS, BLEND_CHROMATICITY, BLEND_RGB_R, BLEND_RGB_G, BLEND_RGB_B,
                         BLEND_SUBTRACT_INVERSE, BLEND_DIVIDE, BLEND_DIVIDE_INVERSE, BLEND_GEOMETRIC_MEAN,
                         BLEND_HARMONIC_MEAN)
# dt_develop_blendif_channels_t (blend.h:141-197), RGB (scene) names
BLENDIF_GRAY_in, BLENDIF_RED_in, BLENDIF_GREEN_in, BLENDIF_BLUE_in = 0, 1, 2, 3
BLENDIF_GRAY_out, BLENDIF_RED_out, BLENDIF_GREEN_out, BLENDIF_BLUE_out = 4, 5, 6, 7
BLENDIF_Jz_in, BLENDIF_Cz_in, BLENDIF_hz_in = 8, 9, 10
BLENDIF_Jz_out, BLENDIF_Cz_out, BLENDIF_hz_out = 12, 13, 14
MASK_ENABLED, MASK_SHAPE, MASK_PARAMETRIC, MASK_RASTER = 1, 2, 4, 8
COMBINE_INV, COMBINE_INCL = 1, 2
BLEND_CS_RGB_SCENE = 4
BLEND_CS_LAB = 2
BLEND_CS_RAW = 1
BLEND_CS_RGB_DISPLAY = 3
# the operators of the "RGB (display)" colourspace (src/develop/blends/blendif_rgb_hsl.c:915-1008), all thirty
BLEND_DISPLAY_MODES = (0x18, 0x19, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x17, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x0E, 0x0F,
                       0x10, 0x11, 0x12, 0x13, 0x16, 0x1C, 0x1D, 0x21, 0x22, 0x23)
BLENDIF_H_in, BLENDIF_S_in, BLENDIF_l_in, BLENDIF_H_out, BLENDIF_S_out, BLENDIF_l_out = 8, 9, 10, 12, 13, 14
# the operators of the "raw" colourspace (src/develop/blends/blendif_raw.c:290-353)
BLEND_RAW_MODES = (0x18, 0x19, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x17, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x0E, 0x0F)
# the operators of the "Lab" colourspace (src/develop/blends/blendif_lab.c:1070-1160), all twenty-seven
BLEND_LAB_MODES = (0x18, 0x19, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x17, 0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x0E, 0x0F,
                   0x10, 0x11, 0x12, 0x13, 0x16, 0x1A, 0x1B, 0x1E, 0x1F, 0x20)
BLEND_LAB_REFUSED = ()
# dt_develop_blendif_channels_t, Lab names
BLENDIF_L_in, BLENDIF_A_in, BLENDIF_B_in, BLENDIF_C_in, BLENDIF_h_in = 0, 1, 2, 8, 9
BLENDIF_L_out, BLENDIF_A_out, BLENDIF_B_out, BLENDIF_C_out, BLENDIF_h_out = 4, 5, 6, 12, 13


MASK_GUIDE_IN_BEFORE_BLUR, MASK_GUIDE_OUT_BEFORE_BLUR, MASK_GUIDE_IN_AFTER_BLUR, MASK_GUIDE_OUT_AFTER_BLUR = 1, 2, 5, 6


class DetailmaskData(C.Structure):
    """dt_hip_detailmask_data_t: the white-balance coefficients the hidden "detailmask" stage normalises by and the plane
    (roi_out floats) it leaves the raw detail mask in"""
    _fields_ = [("wb", C.c_float * 4), ("mask", C.c_void_p)]

    @classmethod
    def make(cls, wb, mask_ptr):
        d = cls()
        for k in range(3):
            d.wb[k] = wb[k]
        d.wb[3] = 1.0
        d.mask = mask_ptr
        return d


class BlendData(C.Structure):
    """dt_hip_blend_data_t: the fields of dt_develop_blend_params_t (src/develop/blend.h:199-244) the
    uniform / parametric RGB (scene) blend reads + the work profile's RGB -> XYZ(D50) matrix"""
    _fields_ = [("mask_mode", C.c_uint32), ("blend_cst", C.c_int32), ("blend_mode", C.c_uint32),
                ("blend_parameter", C.c_float), ("opacity", C.c_float), ("mask_combine", C.c_uint32),
                ("blendif", C.c_uint32), ("feathering_radius", C.c_float), ("blur_radius", C.c_float),
                ("details", C.c_float), ("contrast", C.c_float), ("brightness", C.c_float),
                ("blendif_parameters", C.c_float * 64), ("blendif_boost_factors", C.c_float * 16), ("matrix_in", m34),
                ("form_mask", C.c_void_p), ("feathering_guide", C.c_uint32), ("detail_mask", C.c_void_p)]

    @classmethod
    def uniform(cls, matrix_in, opacity=100.0, blend_mode=BLEND_NORMAL, blend_parameter=0.0, blend_cst=BLEND_CS_RGB_SCENE):
        """dt_develop_blend_init_blend_parameters() (blend.c:173-212) with a uniform mask: every channel's
        trapezoid is the whole range {0, 0, 1, 1}"""
        d = cls()
        d.mask_mode = MASK_ENABLED
        d.blend_cst = blend_cst
        d.blend_mode = blend_mode
        d.blend_parameter = blend_parameter
        d.opacity = opacity
        for ch in range(16):
            d.blendif_parameters[4 * ch + 2] = 1.0
            d.blendif_parameters[4 * ch + 3] = 1.0
        set_m34(d.matrix_in, matrix_in)
        return d

    def channel(self, ch, lo0, lo1, hi0, hi1, invert=False, boost=0.0):
        """switch on one parametric channel with its trapezoid (and the parametric mask with it)"""
        self.mask_mode |= MASK_PARAMETRIC
        self.blendif |= 1 << ch
        if invert:
            self.blendif |= 1 << (16 + ch)
        for k, v in enumerate((lo0, lo1, hi0, hi1)):
            self.blendif_parameters[4 * ch + k] = v
        self.blendif_boost_factors[ch] = boost
        return self
