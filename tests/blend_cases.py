"""Parameter sets for the blend-stage tests (CPU: oracle vs oracle/_ref; GPU: HIP vs oracle)."""
import numpy as np

from ansel_amd import abi, params, synth

M = params.WORK_IN  # work profile RGB -> XYZ(D50)


def images(w, h, seed, iw=None, ih=None):
    """(module input, module output): two related scene-referred frames with negatives, values above 1,
    exact zeros and a few non-finite pixels"""
    iw, ih = iw or w, ih or h
    a = synth.rgba_image(iw, ih, seed=seed, lo=-0.02, hi=2.5)
    rng = np.random.default_rng(seed + 100)
    b = synth.rgba_image(w, h, seed=seed + 1, lo=-0.02, hi=2.5)
    b[..., :3] = 0.6 * b[..., :3] + 0.4 * a[:h, :w, :3] * rng.uniform(0.5, 1.8, size=(h, w, 1)).astype(np.float32)
    a[3, 5, :3] = 0.0
    b[4, 6, :3] = 0.0
    b[7, 2, 1] = 0.0
    a[9, 9, 0] = np.inf
    b[11, 3, 2] = np.nan
    a[..., 3] = 0.25
    b[..., 3] = 0.75
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def cases():
    out = []
    for mode in abi.BLEND_RGB_SCENE_MODES:
        out.append(("uniform-%02x" % mode, abi.BlendData.uniform(M, opacity=63.0, blend_mode=mode, blend_parameter=0.7)))
    out.append(("uniform-reverse-multiply", abi.BlendData.uniform(M, 40.0, abi.BLEND_MULTIPLY | abi.BLEND_REVERSE, -1.3)))
    out.append(("uniform-reverse-normal", abi.BlendData.uniform(M, 100.0, abi.BLEND_NORMAL | abi.BLEND_REVERSE)))
    out.append(("uniform-full", abi.BlendData.uniform(M, 100.0)))
    out.append(("uniform-zero", abi.BlendData.uniform(M, 0.0)))
    out.append(("uniform-over", abi.BlendData.uniform(M, 130.0, abi.BLEND_ADD, 1.0)))
    # parametric: one channel at a time, input and output side
    for ch, tr in ((abi.BLENDIF_GRAY_in, (0.05, 0.2, 0.6, 0.9)), (abi.BLENDIF_RED_in, (0.0, 0.0, 0.5, 0.8)),
                   (abi.BLENDIF_GREEN_in, (0.1, 0.3, 1.0, 1.0)), (abi.BLENDIF_BLUE_in, (0.2, 0.2001, 0.7, 0.7002)),
                   (abi.BLENDIF_GRAY_out, (0.1, 0.4, 0.8, 1.0)), (abi.BLENDIF_RED_out, (0.3, 0.5, 1.0, 1.0)),
                   (abi.BLENDIF_GREEN_out, (0.0, 0.0, 0.4, 0.6)), (abi.BLENDIF_BLUE_out, (0.05, 0.1, 0.9, 0.95)),
                   (abi.BLENDIF_Jz_in, (0.1, 0.3, 0.7, 0.9)), (abi.BLENDIF_Cz_in, (0.0, 0.0, 0.3, 0.6)),
                   (abi.BLENDIF_hz_in, (0.2, 0.35, 0.6, 0.8)), (abi.BLENDIF_Jz_out, (0.2, 0.4, 1.0, 1.0)),
                   (abi.BLENDIF_Cz_out, (0.1, 0.2, 0.5, 0.7)), (abi.BLENDIF_hz_out, (0.0, 0.0, 0.5, 0.55))):
        boost = {abi.BLENDIF_GRAY_in: 1.0, abi.BLENDIF_Jz_in: -5.0, abi.BLENDIF_Cz_in: -6.5, abi.BLENDIF_Jz_out: -5.0,
                 abi.BLENDIF_Cz_out: -6.5}.get(ch, 0.0)
        out.append(("param-ch%d" % ch, abi.BlendData.uniform(M, 85.0).channel(ch, *tr, boost=boost)))
        out.append(("param-ch%d-inv" % ch, abi.BlendData.uniform(M, 85.0, abi.BLEND_AVERAGE).channel(ch, *tr, invert=True, boost=boost)))
    # several channels, every combine mode, with and without the tone curve
    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL, abi.COMBINE_INV | abi.COMBINE_INCL):
        for contrast, brightness in ((0.0, 0.0), (0.4, -0.3), (-0.5, 0.6), (0.2, 1.0), (0.0, -1.0)):
            d = abi.BlendData.uniform(M, 72.0, abi.BLEND_NORMAL)
            d.channel(abi.BLENDIF_GRAY_in, 0.05, 0.25, 0.7, 1.0, boost=1.5)
            d.channel(abi.BLENDIF_BLUE_out, 0.0, 0.0, 0.6, 0.9, invert=True)
            d.channel(abi.BLENDIF_Jz_in, 0.05, 0.2, 1.0, 1.0, boost=-4.0)
            d.channel(abi.BLENDIF_hz_out, 0.1, 0.3, 0.8, 0.95)
            d.mask_combine = combine
            d.contrast = contrast
            d.brightness = brightness
            out.append(("multi-c%d-%g-%g" % (combine, contrast, brightness), d))
    # a channel that is switched on with the full range and inverted cancels the mask (make_mask's second case)
    d = abi.BlendData.uniform(M, 55.0).channel(abi.BLENDIF_GRAY_in, 0.1, 0.2, 0.8, 0.9)
    d.blendif |= 1 << (16 + abi.BLENDIF_RED_in)
    out.append(("canceling", d))
    d = abi.BlendData.uniform(M, 55.0).channel(abi.BLENDIF_GRAY_in, 0.1, 0.2, 0.8, 0.9)
    d.blendif |= 1 << (16 + abi.BLENDIF_RED_in)
    d.mask_combine = abi.COMBINE_INCL
    out.append(("canceling-incl", d))
    # the mask is disabled: nothing happens
    d = abi.BlendData.uniform(M, 50.0)
    d.mask_mode = 0
    out.append(("disabled", d))
    return out


def lab_images(w, h, seed):
    """(module input, module output) in Lab: L 0..100 (a little outside), a/b +-90, exact zeros, greys, non-finite"""
    rng = np.random.default_rng(seed)

    def one(sd):
        rgb = synth.rgba_image(w, h, seed=sd, lo=0.0, hi=1.0)
        lab = np.zeros((h, w, 4), np.float32)
        lab[..., 0] = 104.0 * rgb[..., 1] - 2.0
        lab[..., 1] = 110.0 * (rgb[..., 0] - rgb[..., 1]) + rng.normal(0, 3.0, (h, w))
        lab[..., 2] = 110.0 * (rgb[..., 1] - rgb[..., 2]) + rng.normal(0, 3.0, (h, w))
        return lab.astype(np.float32)

    a, b = one(seed), one(seed + 1)
    b[..., :3] = 0.5 * b[..., :3] + 0.5 * a[..., :3] * rng.uniform(0.6, 1.4, size=(h, w, 1)).astype(np.float32)
    a[2, 3, 1:3] = 0.0
    b[4, 5, 1:3] = 0.0
    a[6, 7, :3] = 0.0
    b[8, 9, 0] = 0.0
    a[10, 11, 1] = np.inf
    b[12, 13, 2] = np.nan
    a[..., 3] = 0.25
    b[..., 3] = 0.75
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def lab_cases():
    out = []
    L = abi.BLEND_CS_LAB
    for mode in abi.BLEND_LAB_MODES:
        out.append(("lab-uniform-%02x" % mode, abi.BlendData.uniform(M, 57.0, mode, blend_cst=L)))
    out.append(("lab-uniform-reverse-overlay", abi.BlendData.uniform(M, 80.0, 0x0A | abi.BLEND_REVERSE, blend_cst=L)))
    out.append(("lab-uniform-full-vivid", abi.BlendData.uniform(M, 100.0, 0x0D, blend_cst=L)))
    for ch, tr in ((abi.BLENDIF_L_in, (0.1, 0.3, 0.7, 0.9)), (abi.BLENDIF_A_in, (0.4, 0.45, 0.55, 0.7)),
                   (abi.BLENDIF_B_in, (0.0, 0.0, 0.5, 0.6)), (abi.BLENDIF_C_in, (0.05, 0.15, 1.0, 1.0)),
                   (abi.BLENDIF_h_in, (0.1, 0.3, 0.6, 0.8)), (abi.BLENDIF_L_out, (0.2, 0.4, 1.0, 1.0)),
                   (abi.BLENDIF_A_out, (0.3, 0.5, 0.6, 0.8)), (abi.BLENDIF_B_out, (0.45, 0.5, 1.0, 1.0)),
                   (abi.BLENDIF_C_out, (0.0, 0.0, 0.3, 0.5)), (abi.BLENDIF_h_out, (0.5, 0.6, 0.9, 0.95))):
        boost = {abi.BLENDIF_A_in: 1.0, abi.BLENDIF_C_out: 0.5}.get(ch, 0.0)
        out.append(("lab-param-ch%d" % ch, abi.BlendData.uniform(M, 85.0, blend_cst=L).channel(ch, *tr, boost=boost)))
        out.append(("lab-param-ch%d-inv" % ch, abi.BlendData.uniform(M, 85.0, 0x05, blend_cst=L).channel(ch, *tr, invert=True, boost=boost)))
    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL, abi.COMBINE_INV | abi.COMBINE_INCL):
        for contrast, brightness in ((0.0, 0.0), (0.4, -0.3)):
            d = abi.BlendData.uniform(M, 72.0, 0x0B, blend_cst=L)
            d.channel(abi.BLENDIF_L_in, 0.05, 0.25, 0.7, 1.0)
            d.channel(abi.BLENDIF_B_out, 0.3, 0.45, 0.6, 0.9, invert=True)
            d.channel(abi.BLENDIF_C_in, 0.02, 0.1, 1.0, 1.0)
            d.channel(abi.BLENDIF_h_out, 0.1, 0.3, 0.8, 0.95)
            d.mask_combine = combine
            d.contrast = contrast
            d.brightness = brightness
            out.append(("lab-multi-c%d-%g-%g" % (combine, contrast, brightness), d))
    return out


def raw_images(w, h, seed):
    """(module input, module output): one-channel mosaics around [0, 1] with excursions, zeros, non-finite"""
    rng = np.random.default_rng(seed)
    a = (rng.random((h, w), dtype=np.float32) * 1.3 - 0.1).astype(np.float32)
    b = (0.6 * a + 0.4 * rng.random((h, w), dtype=np.float32) * 1.2).astype(np.float32)
    a[1, 2] = 0.0
    b[3, 4] = 0.0
    b[5, 6] = 1.0
    a[7, 8] = np.inf
    b[9, 10] = np.nan
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def raw_cases():
    out = []
    R = abi.BLEND_CS_RAW
    for mode in abi.BLEND_RAW_MODES:
        out.append(("raw-uniform-%02x" % mode, abi.BlendData.uniform(M, 61.0, mode, blend_cst=R)))
    out.append(("raw-reverse-softlight", abi.BlendData.uniform(M, 90.0, 0x0B | abi.BLEND_REVERSE, blend_cst=R)))
    # a "parametric" raw mask has no channel to test: the opacity, optionally inverted, through the tone curve
    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL, abi.COMBINE_INV | abi.COMBINE_INCL):
        d = abi.BlendData.uniform(M, 70.0, 0x18, blend_cst=R).channel(abi.BLENDIF_GRAY_in, 0.1, 0.2, 0.8, 0.9)
        d.mask_combine = combine
        d.contrast, d.brightness = 0.3, 0.2
        out.append(("raw-parametric-c%d" % combine, d))
    return out


def display_images(w, h, seed):
    """(module input, module output), display-referred: mostly inside [0, 1], some outside, greys, zeros, non-finite"""
    a = synth.rgba_image(w, h, seed=seed, lo=-0.05, hi=1.15)
    rng = np.random.default_rng(seed + 200)
    b = synth.rgba_image(w, h, seed=seed + 1, lo=-0.05, hi=1.15)
    b[..., :3] = 0.5 * b[..., :3] + 0.5 * a[..., :3] * rng.uniform(0.6, 1.5, size=(h, w, 1)).astype(np.float32)
    a[2, 3, :3] = 0.4          # grey: delta 0
    b[4, 5, :3] = 0.0
    a[6, 7, :3] = (1.0, 1.0, 0.2)
    b[8, 9, :3] = (0.3, 0.9, 0.9)
    a[10, 11, 0] = np.inf
    b[12, 13, 2] = np.nan
    a[..., 3] = 0.25
    b[..., 3] = 0.75
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def display_cases():
    out = []
    D = abi.BLEND_CS_RGB_DISPLAY
    for mode in abi.BLEND_DISPLAY_MODES:
        out.append(("dsp-uniform-%02x" % mode, abi.BlendData.uniform(M, 64.0, mode, blend_cst=D)))
    out.append(("dsp-reverse-color", abi.BlendData.uniform(M, 75.0, 0x13 | abi.BLEND_REVERSE, blend_cst=D)))
    for ch, tr in ((abi.BLENDIF_GRAY_in, (0.05, 0.2, 0.6, 0.9)), (abi.BLENDIF_RED_out, (0.3, 0.5, 1.0, 1.0)),
                   (abi.BLENDIF_H_in, (0.1, 0.3, 0.6, 0.8)), (abi.BLENDIF_S_in, (0.0, 0.0, 0.4, 0.7)),
                   (abi.BLENDIF_l_in, (0.2, 0.4, 1.0, 1.0)), (abi.BLENDIF_H_out, (0.5, 0.6, 0.9, 0.95)),
                   (abi.BLENDIF_S_out, (0.1, 0.3, 1.0, 1.0)), (abi.BLENDIF_l_out, (0.0, 0.0, 0.5, 0.8))):
        out.append(("dsp-param-ch%d" % ch, abi.BlendData.uniform(M, 85.0, blend_cst=D).channel(ch, *tr)))
        out.append(("dsp-param-ch%d-inv" % ch, abi.BlendData.uniform(M, 85.0, 0x0A, blend_cst=D).channel(ch, *tr, invert=True)))
    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL, abi.COMBINE_INV | abi.COMBINE_INCL):
        d = abi.BlendData.uniform(M, 72.0, 0x12, blend_cst=D)
        d.channel(abi.BLENDIF_GRAY_in, 0.05, 0.25, 0.7, 1.0)
        d.channel(abi.BLENDIF_H_in, 0.1, 0.3, 0.8, 0.95)
        d.channel(abi.BLENDIF_S_out, 0.05, 0.2, 1.0, 1.0)
        d.channel(abi.BLENDIF_l_out, 0.0, 0.0, 0.6, 0.9, invert=True)
        d.mask_combine = combine
        d.contrast, d.brightness = 0.3, -0.2
        out.append(("dsp-multi-c%d" % combine, d))
    return out


def blur_cases():
    """(name, data, which images): a parametric mask through the recursive gaussian (blend.c:869-881) and the tone curve"""
    out = []
    for radius in (0.5, 3.0, 25.0):
        d = abi.BlendData.uniform(M, 80.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.2, 0.6, 0.9, boost=1.0)
        d.blur_radius = radius
        out.append(("blur-scene-%g" % radius, d, "scene"))
    d = abi.BlendData.uniform(M, 70.0, abi.BLEND_MULTIPLY, 0.5).channel(abi.BLENDIF_Jz_in, 0.05, 0.2, 1.0, 1.0, boost=-4.0)
    d.blur_radius, d.contrast, d.brightness = 6.0, 0.4, -0.2
    d.mask_combine = abi.COMBINE_INV
    out.append(("blur-scene-tone-inv", d, "scene"))
    d = abi.BlendData.uniform(M, 90.0, 0x0B, blend_cst=abi.BLEND_CS_LAB).channel(abi.BLENDIF_L_in, 0.2, 0.4, 0.7, 0.9)
    d.blur_radius = 4.0
    out.append(("blur-lab", d, "lab"))
    d = abi.BlendData.uniform(M, 60.0, 0x12, blend_cst=abi.BLEND_CS_RGB_DISPLAY).channel(abi.BLENDIF_S_in, 0.1, 0.3, 1.0, 1.0)
    d.blur_radius, d.contrast = 2.0, -0.3
    out.append(("blur-display", d, "display"))
    d = abi.BlendData.uniform(M, 70.0, 0x18, blend_cst=abi.BLEND_CS_RAW).channel(abi.BLENDIF_GRAY_in, 0.1, 0.2, 0.8, 0.9)
    d.blur_radius = 5.0
    out.append(("blur-raw", d, "raw"))
    # a blur radius on a uniform mask does nothing (post operations run on parametric masks only)
    d = abi.BlendData.uniform(M, 45.0)
    d.blur_radius = 9.0
    out.append(("blur-uniform-ignored", d, "scene"))
    return out


def feather_cases():
    """(name, data, which images): the guided filter over the mask (blend.c:603-623, :815-852), guided by the module's input
    or output, before or after the blur, alone or with the tone curve"""
    out = []
    for guide in (abi.MASK_GUIDE_IN_BEFORE_BLUR, abi.MASK_GUIDE_OUT_BEFORE_BLUR, abi.MASK_GUIDE_IN_AFTER_BLUR,
                  abi.MASK_GUIDE_OUT_AFTER_BLUR):
        d = abi.BlendData.uniform(M, 80.0).channel(abi.BLENDIF_GRAY_in, 0.05, 0.2, 0.6, 0.9, boost=1.0)
        d.feathering_radius, d.feathering_guide = 3.0, guide
        out.append(("feather-scene-g%d" % guide, d, "scene"))
        d = abi.BlendData.uniform(M, 65.0, abi.BLEND_MULTIPLY, 0.5).channel(abi.BLENDIF_Jz_in, 0.05, 0.2, 1.0, 1.0, boost=-4.0)
        d.feathering_radius, d.feathering_guide, d.blur_radius = 1.5, guide, 2.5
        d.contrast, d.brightness = 0.3, 0.1
        out.append(("feather-blur-tone-scene-g%d" % guide, d, "scene"))
    d = abi.BlendData.uniform(M, 90.0, 0x0B, blend_cst=abi.BLEND_CS_LAB).channel(abi.BLENDIF_L_in, 0.2, 0.4, 0.7, 0.9)
    d.feathering_radius, d.feathering_guide = 4.0, abi.MASK_GUIDE_IN_AFTER_BLUR
    out.append(("feather-lab", d, "lab"))  # guide weight 1 instead of 100
    d = abi.BlendData.uniform(M, 60.0, 0x12, blend_cst=abi.BLEND_CS_RGB_DISPLAY).channel(abi.BLENDIF_S_in, 0.1, 0.3, 1.0, 1.0)
    d.feathering_radius, d.feathering_guide, d.blur_radius = 0.3, abi.MASK_GUIDE_OUT_BEFORE_BLUR, 1.0
    out.append(("feather-display-w1", d, "display"))  # window 1
    d = abi.BlendData.uniform(M, 75.0).channel(abi.BLENDIF_GRAY_out, 0.1, 0.3, 0.7, 0.8)
    d.feathering_radius, d.feathering_guide = 40.0, abi.MASK_GUIDE_OUT_AFTER_BLUR
    out.append(("feather-scene-wide", d, "scene"))  # window 80: wider than the small frames
    # a feathering radius on a uniform mask does nothing (post operations run behind make_mask() only)
    d = abi.BlendData.uniform(M, 45.0)
    d.feathering_radius, d.feathering_guide = 9.0, abi.MASK_GUIDE_IN_BEFORE_BLUR
    out.append(("feather-uniform-ignored", d, "scene"))
    # one-channel buffers are never feathered (blend.c:431)
    d = abi.BlendData.uniform(M, 70.0, 0x18, blend_cst=abi.BLEND_CS_RAW).channel(abi.BLENDIF_GRAY_in, 0.1, 0.2, 0.8, 0.9)
    d.feathering_radius, d.feathering_guide = 5.0, abi.MASK_GUIDE_IN_BEFORE_BLUR
    out.append(("feather-raw-ignored", d, "raw"))
    return out


def images_for(kind, w, h, seed):
    return {"scene": images, "lab": lab_images, "display": display_images, "raw": raw_images}[kind](w, h, seed)


def form_plane(w, h, seed=5):
    """a host-rendered form mask as the reference's blend receives it (blend.c:740-790): a feathered ellipse (a drawn
    form), times a smooth raster mask, with exact 0 and 1 regions, values slightly outside [0, 1] (the detail refinement
    can leave those) and one NaN"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    r = np.sqrt(((xx - 0.45 * w) / (0.35 * w)) ** 2 + ((yy - 0.55 * h) / (0.3 * h)) ** 2)
    m = np.clip((1.2 - r) / 0.4, 0.0, 1.0).astype(np.float32)
    m *= (0.6 + 0.4 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.float32)
    m[: h // 10] = 0.0
    m[-h // 12:, : w // 3] = 1.0
    m[h // 2, w // 2] = 1.02
    m[h // 2 + 1, w // 2] = -0.01
    m[5, 7] = np.nan
    m += (rng.random((h, w), dtype=np.float32) - 0.5) * np.float32(1e-3) * (m > 0) * (m < 1)
    return np.ascontiguousarray(m.astype(np.float32))


def form_cases(blend_cst=abi.BLEND_CS_RGB_SCENE):
    """(name, BlendData without the plane): drawn / raster masks alone, with parametric channels, every combine mode,
    with the post operations"""
    lab = blend_cst == abi.BLEND_CS_LAB
    m_in = M
    ch_in, ch_out = (abi.BLENDIF_L_in, abi.BLENDIF_C_out) if lab else (abi.BLENDIF_GRAY_in, abi.BLENDIF_Jz_out)
    tr_in = (0.05, 0.3, 0.7, 0.95)
    out = []

    def base(opacity=80.0, mode=abi.BLEND_NORMAL):
        return abi.BlendData.uniform(m_in, opacity, mode, blend_cst=blend_cst)

    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL, abi.COMBINE_INV | abi.COMBINE_INCL):
        d = base()
        d.mask_mode |= abi.MASK_SHAPE
        d.mask_combine = combine
        out.append(("drawn-c%d" % combine, d))
        d = base(65.0)
        d.mask_mode |= abi.MASK_SHAPE | abi.MASK_RASTER
        d.channel(ch_in, *tr_in, boost=0.0 if lab else 1.0)
        d.channel(ch_out, 0.0, 0.0, 0.6, 0.9, invert=True, boost=0.0 if lab else -5.0)
        d.mask_combine = combine
        d.contrast, d.brightness = 0.3, -0.2
        out.append(("drawn+raster+parametric-c%d" % combine, d))
    d = base(55.0)
    d.mask_mode |= abi.MASK_RASTER
    d.contrast = 0.5  # a raster mask alone gets no post operation
    out.append(("raster-only", d))
    d = base(90.0)
    d.mask_mode |= abi.MASK_SHAPE
    d.blur_radius = 4.0
    d.brightness = 0.25
    out.append(("drawn-blurred-toned", d))
    d = base(70.0)
    d.mask_mode |= abi.MASK_SHAPE | abi.MASK_PARAMETRIC  # the conditional mode switched on, no channel away from its range
    d.details = 0.4  # applied by the host to the plane
    out.append(("drawn-details-conditional-idle", d))
    d = base(70.0)
    d.channel(ch_in, *tr_in)
    d.details = -0.3  # a parametric-only mask refined by the detail mask: the host supplies the refined fill
    out.append(("parametric-details", d))
    return out


def detail_cases(blend_cst=abi.BLEND_CS_RGB_SCENE):
    """(name, BlendData, needs a form plane): a details threshold refined on the spot from the raw detail mask
    (_refine_with_detail_mask(), blend.c:361-425): positive (keep detail) and negative (keep flat areas) levels, on a
    drawn mask, on a parametric-only mask with either combine mode, with post operations behind it"""
    lab = blend_cst == abi.BLEND_CS_LAB
    ch_in = abi.BLENDIF_L_in if lab else abi.BLENDIF_GRAY_in
    out = []

    def base(opacity=80.0):
        return abi.BlendData.uniform(M, opacity, abi.BLEND_NORMAL, blend_cst=blend_cst)

    for level in (0.35, -0.5, 1.0, -1.0):
        d = base()
        d.mask_mode |= abi.MASK_SHAPE
        d.details = level
        out.append(("drawn-details%+.2f" % level, d, True))
    for combine in (0, abi.COMBINE_INV, abi.COMBINE_INCL):
        d = base(70.0)
        d.channel(ch_in, 0.05, 0.3, 0.7, 0.95)
        d.mask_combine = combine
        d.details = 0.25
        out.append(("parametric-details-c%d" % combine, d, False))
    d = base(90.0)
    d.mask_mode |= abi.MASK_SHAPE
    d.channel(ch_in, 0.0, 0.2, 0.8, 1.0)
    d.details = -0.2
    d.blur_radius = 3.0
    d.contrast = 0.3
    out.append(("drawn+parametric-details-blurred-toned", d, True))
    d = base(60.0)
    d.mask_mode |= abi.MASK_RASTER  # a raster mask alone is form * opacity: no refinement (blend.c:740-745)
    d.details = 0.5
    out.append(("raster-only-details-ignored", d, True))
    d = base(60.0)
    d.details = 0.5  # no mask at all: uniform opacity, no refinement (blend.c:735-739)
    out.append(("uniform-details-ignored", d, False))
    return out
