"""Degenerate frame sizes for every module on the path (shared by the CPU and the GPU edge tests): one pixel, one
row, one column, smaller than any tile / chunk / pyramid level / grid cell."""
import numpy as np

from ansel_amd import abi, filmic, params, synth

SIZES = [(1, 1), (2, 2), (3, 5), (8, 6), (17, 9), (36, 34), (64, 2), (2, 64), (37, 1)]
MODULES = ["rawprepare", "temperature", "highlights", "demosaic_rcd", "demosaic_ppg", "demosaic_amaze", "exposure",
           "colorin", "channelmixerrgb", "filmicrgb", "colorout", "denoiseprofile", "denoiseprofile_nlm", "nlmeans",
           "bilat", "bilat_ll", "diffuse", "finalscale", "blend", "blend_lab", "rgb_to_lab", "lab_to_rgb"]

_KEEP = []


def case(module, w, h, lut_ptr=None):
    """-> (op, piece, data, input, output shape, pre-filled output or None); lut_ptr: where the tone curve of
    colorout lives for the callee (a device pointer for the HIP path; default: host memory)"""
    rng = np.random.default_rng(1000 * w + h)
    rgba = (rng.random((h, w, 4), dtype=np.float32) * 1.2 - 0.05).astype(np.float32)
    lab = (rgba * np.float32([100, 60, 60, 1]) - np.float32([0, 30, 30, 0])).astype(np.float32)
    cfa = rng.random((h, w), dtype=np.float32)
    rgb = abi.Piece.make(w, h, processed_maximum=synth.WB_COEFFS)
    one = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    if module == "rawprepare":
        u16 = (rng.random((h, w)) * 16000).astype(np.uint16)
        piece = abi.Piece.make(w, h, filters=synth.FILTERS_RGGB, channels=1, datatype=abi.DT_HIP_TYPE_UINT16)
        return "rawprepare", piece, abi.RawprepareData(0, 0, 0, 0, abi.f4(512, 510, 514, 512), abi.f4(15871, 15873, 15869, 15871)), u16, (h, w), None
    if module == "temperature":
        return "temperature", one, abi.TemperatureData(abi.f4(*synth.WB_COEFFS)), cfa, (h, w), None
    if module == "highlights":
        return "highlights", one, abi.HighlightsData(0, 0.4), cfa, (h, w), None
    if module.startswith("demosaic"):
        m = {"demosaic_rcd": abi.DT_HIP_DEMOSAIC_RCD, "demosaic_ppg": abi.DT_HIP_DEMOSAIC_PPG,
             "demosaic_amaze": abi.DT_HIP_DEMOSAIC_AMAZE}[module]
        return "demosaic", one, abi.DemosaicData(0, 0, m, 0.0), cfa, (h, w, 4), np.full((h, w, 4), -7.0, np.float32)
    if module == "exposure":
        return "exposure", rgb, abi.ExposureData(-0.0002, 1.7), rgba, rgba.shape, None
    if module in ("colorin", "colorout"):
        enc = params.srgb_encode_lut()
        _KEEP.append(enc)
        lt = [(lut_ptr if lut_ptr is not None else enc.ctypes.data, float(enc[0]), params.unbounded_coeffs(enc))] * 3
        d = params.conversion(params.WORK_OUT @ params.CAMERA_TO_XYZ) if module == "colorin" else \
            params.conversion(params.SRGB_OUT @ params.WORK_IN, lut_target=lt)
        return module, rgb, d, rgba, rgba.shape, None
    if module == "channelmixerrgb":
        return module, rgb, params.channelmixerrgb(), rgba, rgba.shape, None
    if module == "filmicrgb":
        return module, rgb, filmic.default_data(), rgba, rgba.shape, None
    if module == "denoiseprofile":
        return "denoiseprofile", rgb, params.denoiseprofile(), rgba, rgba.shape, None
    if module == "denoiseprofile_nlm":
        return "denoiseprofile", rgb, params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS), rgba, rgba.shape, None
    if module == "nlmeans":
        return "nlmeans", rgb, abi.NlmeansData(2.0, 50.0, 0.5, 1.0), lab, lab.shape, None
    if module == "bilat":
        return "bilat", rgb, abi.BilatData.bilateral(12.0, 10.0, 0.5), lab, lab.shape, None
    if module == "bilat_ll":
        return "bilat", rgb, abi.BilatData.local_laplacian(), lab, lab.shape, None
    if module == "diffuse":
        return "diffuse", rgb, params.diffuse("lens_deblur_soft"), rgba, rgba.shape, None
    if module == "finalscale":
        ow, oh = max(w // 2, 1), max(h // 2, 1)
        p = abi.Piece.make(ow, oh, roi_in=abi.Roi.make(0, 0, w, h, 1.0), roi_out=abi.Roi.make(0, 0, ow, oh, 0.5))
        return "finalscale", p, abi.FinalscaleData(2), rgba, (oh, ow, 4), None
    if module in ("blend", "blend_lab"):
        import blend_cases
        d = dict(blend_cases.cases())["multi-c0-0.4--0.3"] if module == "blend" else dict(blend_cases.lab_cases())["lab-multi-c0-0.4--0.3"]
        src = rgba if module == "blend" else lab
        return "develop_blend", rgb, d, src, src.shape, np.ascontiguousarray(src[::-1, ::-1])
    if module == "rgb_to_lab":
        return "rgb_to_lab", rgb, abi.LabData.make(params.WORK_IN), rgba, rgba.shape, None
    if module == "lab_to_rgb":
        return "lab_to_rgb", rgb, abi.LabData.make(params.WORK_OUT), lab, lab.shape, None
    raise KeyError(module)


def undefined_in_reference(module, w, h):
    """sizes on which the reference itself reads or writes outside its buffers (it crashes on the CPU): the oracle
    and the device refuse them"""
    if module == "bilat_ll":
        # a side of 2 indexes the pyramid array at -1 (locallaplacian.c:405), a side of 3 pads for two levels
        # but builds one: out-of-bounds reads either way.  A side of 1 returns before touching anything (:366)
        return min(w, h) in (2, 3)
    if module == "demosaic_amaze":
        return w < 34 or h < 34  # the border mirroring reads frame rows / columns up to 32 (amaze.cc:330-420)
    if module == "denoiseprofile":
        # no wavelet band fits (max_scale == 0): the reference then evaluates 1u << -1 (denoiseprofile.c:1322)
        m = max(w, h) * 0.2
        supp0 = min(2 * (2 << 6) + 1, m)
        if not supp0 > 1.0:
            return False  # log2 of a non-positive number: NaN, every comparison false, 7 bands, copy-through
        i0 = np.log2((supp0 - 1.0) * 0.5)
        return 1.0 - (np.log2((5.0 - 1.0) * 0.5) - 1.0 + 0.5) / i0 < 0.0
    if module == "bilat":
        import ctypes as C
        # a grid line shorter than the four entries blur_line() touches unconditionally
        sigma_s = max(12.0, 0.5)
        _x = min(max(int(round(w / sigma_s)), 4), 6000)
        _y = min(max(int(round(h / sigma_s)), 4), 6000)
        s = max(w / _x, h / _y)
        return int(np.ceil(w / s)) + 1 < 4 or int(np.ceil(h / s)) + 1 < 4
    return False


STENCIL_MODULES = ["demosaic_rcd", "demosaic_ppg", "demosaic_amaze", "denoiseprofile", "denoiseprofile_nlm", "nlmeans", "bilat",
                   "bilat_ll", "diffuse", "finalscale"]


def adversarial(module, w=200, h=150):
    """a stencil module's case with non-finite, denormal, huge and negative samples sprinkled over the input: what a
    neighbourhood does with them (spreads a NaN, clamps it, divides by it) must be the reference's doing"""
    op, piece, data, inp, shape, pre = case(module, w, h)
    inp = np.array(inp, copy=True)
    rng = np.random.default_rng(3)
    specials = [np.nan, np.inf, -np.inf, 1e-41, -1e-41, 1e30, -1e30, -0.0, -5.0]
    if module == "demosaic_amaze":
        # finite samples only: a NaN born inside AMaZE is 0xFFC00000 on x86 and 0x7FC00000 on the GPU, and the
        # reference reads those bytes back as Nyquist flags through its aliased planes (amaze.cc:300-327, :830) --
        # a mosaic is finite by construction (u16 -> rawprepare), so that path has no input that reaches it
        specials = specials[3:]
        # ... and no sample whose square overflows (+-1e30: its inf - inf makes a third of THIS frame NaN): where a NaN's sign
        # then turns into a number through the exponent tricks of amaze.cc:77-121, what comes out depends on which operand of an
        # addition the compiler put first and on x - NaN keeping the NaN's sign (x86) or flipping it (v_sub_f32) -- the compiler's
        # and the machine's choice, not the algorithm's.  +-1e18 has the magnitude without the overflow
        specials = [1e18 if v == 1e30 else (-1e18 if v == -1e30 else v) for v in specials]
    flat = inp.reshape(-1)
    for k, i in enumerate(rng.choice(flat.size, size=60, replace=False)):
        flat[i] = specials[k % len(specials)]
    return op, piece, data, inp, shape, pre
