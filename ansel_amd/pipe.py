"""Host-side mirror of the export pixelpipe for the hot path: an ordered list of nodes, each a
module with its committed `data` and its piece view, executed on one device through the C-ABI --
what dt_dev_pixelpipe_process_rec() + pixelpipe_process_on_GPU() do in the reference
(src/develop/pixelpipe_hb.c:881-1282, src/develop/pixelpipe_gpu.c:191-744), reduced to the
device-resident chain: every module output stays in HBM and is the next module's input.

Buffers are caller-provided device pointers (torch tensors in bench.py, DeviceBuffer in tests).
"""
import ctypes as C

import numpy as np

from . import abi, lib, params, synth

# bytes per pixel of each module's input / output (algorithmic traffic)
MODULE_BPP = {
    "rawprepare": (2, 4), "temperature": (4, 4), "highlights": (4, 4), "demosaic": (4, 16),
    "exposure": (16, 16), "colorin": (16, 16), "channelmixerrgb": (16, 16), "filmicrgb": (16, 16),
    "colorout": (16, 16), "export_u16": (16, 8), "rgb_to_lab": (16, 16), "lab_to_rgb": (16, 16), "nlmeans": (16, 16),
    "bilat": (16 + 16, 16),  # splat reads L, slice reads + writes the plane (48 B/px)
}


class Node:
    def __init__(self, op, data, piece):
        self.op = op
        self.data = data
        self.piece = piece


def light_pipe_nodes(width, height, lut_target_ptr, lut_first, lut_coeffs, with_filmic=True, filmic=None,
                     demosaic_method=abi.DT_HIP_DEMOSAIC_RCD):
    """config 2 of BASELINE.json: rawprepare -> temperature -> highlights(clip) -> demosaic ->
    exposure -> colorin -> color calibration -> filmic -> colorout -> u16, module defaults."""
    raw = abi.Piece.make(width, height, filters=synth.FILTERS_RGGB, channels=1, datatype=abi.DT_HIP_TYPE_UINT16)
    cfa1 = abi.Piece.make(width, height, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=(1, 1, 1, 1))
    cfa2 = abi.Piece.make(width, height, filters=synth.FILTERS_RGGB, channels=1, processed_maximum=synth.WB_COEFFS)
    rgb = abi.Piece.make(width, height, channels=4, processed_maximum=synth.WB_COEFFS)
    rng = float(synth.WHITE - synth.BLACK)
    nodes = [
        Node("rawprepare", abi.RawprepareData(0, 0, 0, 0, abi.f4(*[synth.BLACK] * 4), abi.f4(*[rng] * 4)), raw),
        Node("temperature", abi.TemperatureData(abi.f4(*synth.WB_COEFFS)), cfa1),
        Node("highlights", abi.HighlightsData(abi.DT_HIP_HIGHLIGHTS_CLIP, 1.0), cfa2),
        Node("demosaic", abi.DemosaicData(0, 0, demosaic_method, 0.0), cfa2),
        # exposure +0.7 EV, black -0.000244 (module defaults for scene-referred workflow)
        Node("exposure", abi.ExposureData(-0.000244140625, float(np.float32(2.0) ** np.float32(0.7))), rgb),
        Node("colorin", params.conversion(params.WORK_OUT @ params.CAMERA_TO_XYZ), rgb),
        Node("channelmixerrgb", params.channelmixerrgb(), rgb),
    ]
    if with_filmic:
        nodes.append(Node("filmicrgb", filmic, rgb))
    lt = [(lut_target_ptr, lut_first, lut_coeffs)] * 3
    nodes.append(Node("colorout", params.conversion(params.SRGB_OUT @ params.WORK_IN, lut_target=lt), rgb))
    nodes.append(Node("export_u16", None, rgb))
    return nodes


def denoise_pipe_nodes(width, height, lut_target_ptr, lut_first, lut_coeffs, filmic=None,
                       diffuse_preset="lens_deblur_soft", diffuse_iterations=2, with_nlmeans=False, with_bilat=None):
    """config 3 of BASELINE.json, as far as it runs on device: the light pipe + denoise (profiled)
    wavelets after demosaic and diffuse-or-sharpen after color calibration, in the reference's module
    order (src/develop/iop_order.c:196-232)."""
    nodes = light_pipe_nodes(width, height, lut_target_ptr, lut_first, lut_coeffs, with_filmic=filmic is not None,
                             filmic=filmic)
    rgb = abi.Piece.make(width, height, channels=4, processed_maximum=synth.WB_COEFFS)
    out = []
    for n in nodes:
        out.append(n)
        if n.op == "demosaic":
            out.append(Node("denoiseprofile", params.denoiseprofile(), rgb))
        if n.op == "channelmixerrgb":
            out.append(Node("diffuse", params.diffuse(diffuse_preset, iterations=diffuse_iterations), rgb))
            if with_nlmeans:
                # a Lab module: the pipe converts work RGB -> Lab before and back after it (pixelpipe_cpu.c:59-75)
                out.append(Node("rgb_to_lab", abi.LabData.make(params.WORK_IN), rgb))
                out.append(Node("nlmeans", abi.NlmeansData(2.0, 50.0, 0.5, 1.0), rgb))
                if with_bilat is None or with_bilat:
                    out.append(Node("bilat", abi.BilatData.bilateral(), rgb))
                out.append(Node("lab_to_rgb", abi.LabData.make(params.WORK_OUT), rgb))
    return out


def node_bytes_per_px(n):
    """algorithmic bytes per pixel of one node"""
    if n.op == "denoiseprofile":
        from . import modinfo
        return 112 + 96 * modinfo.denoiseprofile_bands(n.piece)
    if n.op == "diffuse":
        from . import modinfo
        return 96 * max(int(n.data.iterations), 1) * modinfo.diffuse_scales(n.piece, n.data)
    return sum(MODULE_BPP[n.op])


def algorithmic_bytes_per_pixel(nodes):
    return sum(node_bytes_per_px(n) for n in nodes)


class DevicePipe:
    """dt_hip_pipe_t: the C++ executor of libansel_hip (pipe.cpp) loaded with a node list"""

    def __init__(self, devid, nodes, fusion=True):
        self.lib = lib.load()
        self.nodes = nodes  # keeps the ctypes structs alive
        self.handle = self.lib.dt_hip_pipe_new(devid)
        if not self.handle:
            raise lib.AnselHipError("dt_hip_pipe_new failed: %s" % self.lib.dt_hip_last_error().decode())
        for n in nodes:
            if n.data is None:
                rc = self.lib.dt_hip_pipe_add_node(self.handle, n.op.encode(), C.byref(n.piece), None, 0)
            else:
                rc = self.lib.dt_hip_pipe_add_node(self.handle, n.op.encode(), C.byref(n.piece),
                                                   C.cast(C.byref(n.data), C.c_void_p), C.sizeof(n.data))
            lib.check(rc, "dt_hip_pipe_add_node(%s)" % n.op)
        self.lib.dt_hip_pipe_set_fusion(self.handle, 1 if fusion else 0)

    @property
    def num_groups(self):
        return self.lib.dt_hip_pipe_num_groups(self.handle)

    def process(self, dev_in, dev_out):
        lib.check(self.lib.dt_hip_pipe_process(self.handle, dev_in, dev_out), "dt_hip_pipe_process")

    def close(self):
        if self.handle:
            self.lib.dt_hip_pipe_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_nodes(devid, nodes, buffers):
    """buffers: list of len(nodes)+1 device pointers; node i reads buffers[i], writes buffers[i+1]"""
    l = lib.load()
    for i, n in enumerate(nodes):
        src, dst = buffers[i], buffers[i + 1]
        if n.op == "export_u16":
            rc = l.dt_hip_export_convert_u16(devid, n.piece.roi_out.width, n.piece.roi_out.height, src, dst)
        elif n.op in ("rgb_to_lab", "lab_to_rgb"):
            rc = getattr(l, "dt_hip_transform_" + n.op)(devid, C.byref(n.piece), C.byref(n.data), src, dst)
        else:
            fn = getattr(l, "dt_hip_iop_%s_process" % n.op)
            rc = fn(devid, C.byref(n.piece), C.byref(n.data), src, dst)
        lib.check(rc, n.op)
