"""The golden-vector cases: small seeded inputs and module parameters, shared by the generator
(make_golden.py, run where /root/reference exists) and the tests that replay them."""
import numpy as np

from ansel_amd import abi, filmic, params, synth

W, H = 96, 64          # pointwise modules
DW, DH = 224, 160      # demosaic: 3 x 2 RCD tiles, partial last column


def _cfa_u16():
    return synth.bayer_mosaic(W + 8, H + 6, seed=21)


def _cfa_f32(w, h, seed):
    cfa = synth.bayer_mosaic(w, h, seed=seed).astype(np.float32)
    out = (cfa - 512.0) / np.float32(synth.WHITE - 512)
    rows = np.arange(h)[:, None]
    cols = np.arange(w)[None, :]
    return (out * np.asarray(synth.WB_COEFFS, dtype=np.float32)[synth.fc(rows, cols)]).astype(np.float32)


def _rgba():
    a = synth.rgba_image(W, H // 2, seed=5, lo=-0.02, hi=4.0)
    b = synth.adversarial_rgba(W, H - H // 2, seed=9)
    return np.ascontiguousarray(np.concatenate([a, b], axis=0))


SINGLE_THREAD_OPS = ("denoiseprofile", "bilat")
SINGLE_THREAD_NAMES = ("demosaic_amaze",)
DNW, DNH = 288, 176    # denoiseprofile: 5 wavelet bands


def _noisy_rgba():
    rng = np.random.default_rng(31)
    img = synth.rgba_image(DNW, DNH, seed=8, lo=0.0, hi=0.9)
    img[..., :3] += rng.normal(0.0, 0.01, size=(DNH, DNW, 3)).astype(np.float32) * np.sqrt(np.maximum(img[..., :3], 0.01))
    return np.ascontiguousarray(img.astype(np.float32))


def _lab():
    rng = np.random.default_rng(77)
    rgb = synth.rgba_image(W, H, seed=12, lo=0.0, hi=1.0)
    lab = np.zeros((H, W, 4), np.float32)
    lab[..., 0] = 100.0 * rgb[..., 1] + rng.normal(0, 1.5, (H, W))
    lab[..., 1] = 80.0 * (rgb[..., 0] - rgb[..., 1]) + rng.normal(0, 2.0, (H, W))
    lab[..., 2] = 80.0 * (rgb[..., 1] - rgb[..., 2]) + rng.normal(0, 2.0, (H, W))
    return np.ascontiguousarray(lab.astype(np.float32))


_LUTS = []


def luts():
    # module-lifetime arrays: the conversion structs carry raw host pointers into them
    if not _LUTS:
        enc, dec = params.srgb_encode_lut(), params.srgb_decode_lut()
        _LUTS.append((enc, dec, params.unbounded_coeffs(enc), params.unbounded_coeffs(dec)))
    return _LUTS[0]


def cases(lut_ptrs=None):
    """yield (name, op, piece, data, input array, output shape).  lut_ptrs = (enc_ptr, dec_ptr) of
    wherever the curves live for the implementation under test (host pointers by default)."""
    enc, dec, ce, cd = luts()
    ep, dp = lut_ptrs if lut_ptrs is not None else (enc.ctypes.data, dec.ctypes.data)
    keep = (enc, dec)  # keep the host arrays alive for host-pointer users
    raw = _cfa_u16()
    yield ("rawprepare", "rawprepare",
           abi.Piece.make(W, H, filters=synth.FILTERS_RGGB, channels=1, datatype=abi.DT_HIP_TYPE_UINT16,
                          roi_in=abi.Roi.make(0, 0, W + 8, H + 6), roi_out=abi.Roi.make(0, 0, W, H)),
           abi.RawprepareData(3, 1, 5, 5, abi.f4(512, 510, 514, 512), abi.f4(15871, 15873, 15869, 15871)), raw, (H, W))
    cfa = _cfa_f32(W, H, 22)
    bay = abi.Piece.make(W, H, filters=synth.FILTERS_RGGB, channels=1, roi_in=abi.Roi.make(1, 1, W, H),
                         roi_out=abi.Roi.make(1, 1, W, H), processed_maximum=synth.WB_COEFFS)
    yield ("temperature", "temperature", bay, abi.TemperatureData(abi.f4(*synth.WB_COEFFS)), cfa, (H, W))
    yield ("highlights_clip", "highlights", bay, abi.HighlightsData(0, 0.6), cfa, (H, W))
    yield ("highlights_bypass", "highlights", bay, abi.HighlightsData(0, 50.0), cfa, (H, W))
    dcfa = _cfa_f32(DW, DH, 23)
    dp_ = abi.Piece.make(DW, DH, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    yield ("demosaic_rcd", "demosaic", dp_, abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_RCD, 0.0), dcfa, (DH, DW, 4))
    yield ("demosaic_ppg", "demosaic", dp_, abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_PPG, 0.0), dcfa, (DH, DW, 4))
    # AMaZE: recorded single-threaded (see SINGLE_THREAD_NAMES); even frame size, so only the tile-row seam is masked
    yield ("demosaic_amaze", "demosaic", dp_, abi.DemosaicData(0, 0, abi.DT_HIP_DEMOSAIC_AMAZE, 0.0), dcfa, (DH, DW, 4))
    img = _rgba()
    rgb = abi.Piece.make(W, H)
    yield ("exposure", "exposure", rgb, abi.ExposureData(-0.000244140625, 1.6245048), img, img.shape)
    cam = params.WORK_OUT @ params.CAMERA_TO_XYZ
    out = params.SRGB_OUT @ params.WORK_IN
    lt = [(ep, float(enc[0]), ce)] * 3
    ls = [(dp, float(dec[0]), cd)] * 3
    yield ("colorin_matrix", "colorin", rgb, params.conversion(cam), img, img.shape)
    yield ("colorin_blue", "colorin", rgb, params.conversion(cam, blue_mapping=True), img, img.shape)
    yield ("colorout_srgb", "colorout", rgb, params.conversion(out, lut_target=lt), img, img.shape)
    yield ("colorout_full", "colorout", rgb, params.conversion(out, clip_matrix=np.eye(3), lut_source=ls, lut_target=lt), img, img.shape)
    for ad in range(5):
        yield ("channelmixerrgb_a%d" % ad, "channelmixerrgb", rgb,
               params.channelmixerrgb(adaptation=ad, version=2, saturation=(0.1, -0.2, 0.05), lightness=(0.05, 0.0, -0.1)), img, img.shape)
    yield ("channelmixerrgb_grey", "channelmixerrgb", rgb, params.channelmixerrgb(grey=(0.3, 0.5, 0.2), clip=False, gamut=2.0, version=0), img, img.shape)
    for ver in (3, 4, 5, 7, 9):
        for curves in ((3, 3), (0, 1), (2, 2)):
            p = filmic.UserParams.defaults(version=ver, shadows=curves[0], highlights=curves[1], preserve_color=3 if ver == 3 else 1,
                                           saturation=10.0 if ver < 5 else 25.0)
            yield ("filmic_v%d_%d%d" % (ver, curves[0], curves[1]), "filmicrgb", rgb, filmic.commit(p), img, img.shape)
    yield ("filmic_split_v4", "filmicrgb", rgb, filmic.commit(filmic.UserParams.defaults(version=3, preserve_color=0, saturation=-15.0)), img, img.shape)
    yield ("filmic_no_export_profile", "filmicrgb", rgb, filmic.commit(filmic.UserParams.defaults(), use_output_profile=False), img, img.shape)
    # diffuse or sharpen: module defaults, two presets (several scales and iterations), mixed anisotropies
    yield ("diffuse_default", "diffuse", rgb, params.diffuse(), img, img.shape)
    yield ("diffuse_deblur", "diffuse", rgb, params.diffuse("lens_deblur_soft", iterations=3), img, img.shape)
    yield ("diffuse_contrast", "diffuse", rgb, params.diffuse("fast_local_contrast", radius=24, radius_center=12), img, img.shape)
    yield ("diffuse_mixed", "diffuse", rgb,
           params.diffuse("lens_deblur_soft", iterations=2, anisotropy_first=-2.0, anisotropy_second=1.5,
                          anisotropy_fourth=-3.0, variance_threshold=-0.5, regularization=2.5, sharpness=0.2),
           img, img.shape)
    # denoise (profiled), wavelets: the golden outputs come from the reference run on ONE thread (its
    # band statistics depend on the thread count, see oracle/src/denoiseprofile.c)
    dimg = _noisy_rgba()
    dpiece = abi.Piece.make(DNW, DNH, processed_maximum=synth.WB_COEFFS)
    yield ("denoiseprofile_y0u0v0", "denoiseprofile", dpiece, params.denoiseprofile(), dimg, dimg.shape)
    yield ("denoiseprofile_rgb", "denoiseprofile", dpiece,
           params.denoiseprofile(color_mode=abi.DT_HIP_DENOISEPROFILE_RGB, strength=1.4, shadows=0.7), dimg, dimg.shape)
    yield ("denoiseprofile_legacy", "denoiseprofile", dpiece, params.denoiseprofile(use_new_vst=False), dimg, dimg.shape)
    # RGB <-> Lab glue and denoise (non-local means)
    yield ("rgb_to_lab", "rgb_to_lab", rgb, abi.LabData.make(params.WORK_IN), img, img.shape)
    lab = _lab()
    yield ("lab_to_rgb", "lab_to_rgb", rgb, abi.LabData.make(params.WORK_OUT), lab, lab.shape)
    yield ("nlmeans", "nlmeans", rgb, abi.NlmeansData(2.0, 50.0, 0.5, 1.0), lab, lab.shape)
    fpiece = abi.Piece.make(60, 40, roi_in=abi.Roi.make(0, 0, W, H, 1.0), roi_out=abi.Roi.make(0, 0, 60, 40, 0.625))
    yield ("finalscale_down", "finalscale", fpiece, abi.FinalscaleData(2), img, (40, 60, 4))
    upiece = abi.Piece.make(144, 96, roi_in=abi.Roi.make(0, 0, W, H, 1.0), roi_out=abi.Roi.make(0, 0, 144, 96, 1.5))
    yield ("finalscale_up", "finalscale", upiece, abi.FinalscaleData(1), img, (96, 144, 4))
    yield ("bilat_local_laplacian", "bilat", rgb, abi.BilatData.local_laplacian(0.7, 0.4, 0.8, 0.4), lab, lab.shape)
    yield ("bilat", "bilat", rgb, abi.BilatData.bilateral(12.0, 10.0, 0.5), lab, lab.shape)
    yield ("denoiseprofile_nlmeans", "denoiseprofile", dpiece,
           params.denoiseprofile(mode=abi.DT_HIP_DENOISEPROFILE_NLMEANS), dimg, dimg.shape)
    # the blend stage: `inp` carries (module input, module output); the output is blended in place
    import blend_cases
    ba, bb = blend_cases.images(W, H, 41)
    pair = np.ascontiguousarray(np.stack([ba, bb]))
    bc = dict(blend_cases.cases())
    for nm in ("uniform-18", "uniform-reverse-multiply", "param-ch0", "param-ch10-inv", "multi-c1-0.4--0.3", "multi-c2--0.5-0.6"):
        yield ("blend_" + nm, "develop_blend", rgb, bc[nm], pair, (H, W, 4))
    del keep
