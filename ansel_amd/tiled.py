"""One frame over several GPUs: row bands with one halo exchange (BASELINE.json config 4).

Replaces default_process_tiling_cl() (src/develop/tiling.c:1394) for the multi-GPU case.  Each
rank owns a band of output rows (include/ansel_hip.h section 3b), cut on the frame's RCD tile rows so
the assembled result is bit-identical to the unsplit frame.  Per frame and rank:

    engine.begin    CFA stages on the band's own rows                      (device)
    all-reduce      clipped-photosite count of the highlights bypass       (8 bytes, RCCL)
    engine.resolve  bypass decision on the frame-wide count                (device)
    send/recv       `halo` = 9 mosaic rows with the band above and below   (RCCL over xGMI)
    engine.finish   demosaic on the frame's tile rows + RGBA stages        (device)

There is no other data-path communication: every module after demosaic on this path is pointwise.
The communication layer is torch.distributed ("nccl" is RCCL on ROCm; "gloo" in the CPU tests);
the compute engine is the C++ executor of libansel_hip.so -- HipBandEngine below.  tests/ plug an
oracle-backed engine into the same driver to cover the N > 1 logic without a GPU.
"""
import ctypes as C

from . import abi, lib


def plan_bands(width, height, n_bands, demosaic_method=abi.DT_HIP_DEMOSAIC_RCD):
    """dt_hip_plan_bands(): pure host function, no device needed"""
    l = lib.load()
    bands = (abi.Band * n_bands)()
    lib.check(l.dt_hip_plan_bands(width, height, demosaic_method, n_bands, bands), "dt_hip_plan_bands")
    return list(bands)


def pipe_demosaic_method(nodes):
    for n in nodes:
        if n.op == "demosaic":
            return int(n.data.demosaicing_method)
    return -1


class BandWork:
    """what engine.begin() hands to the communication step: torch tensors viewing the band's
    mosaic buffer ([halo_top + rows + halo_bottom, width] float32) and its clipped count ([1] int64);
    either may be None"""

    def __init__(self, halo, count, token):
        self.halo = halo
        self.count = count
        self.token = token


class _DevicePtr:
    """__cuda_array_interface__ carrier so torch can view memory owned by the dt_hip runtime"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def device_view(ptr, shape, typestr, device):
    import torch
    return torch.as_tensor(_DevicePtr(ptr, shape, typestr), device=device)


class HipBandEngine:
    """dt_hip_pipe_band_begin / _finish of one DevicePipe"""

    def __init__(self, device_pipe, device):
        import torch
        self.pipe = device_pipe
        self.device = torch.device(device)
        self.lib = lib.load()
        # the collectives are torch work on torch's current stream; the kernels must be ordered with
        # them, so the runtime is put on that stream (dt_hip_set_stream)
        stream = torch.cuda.current_stream(self.device)
        lib.check(self.lib.dt_hip_set_stream(self.device.index or 0, C.c_void_p(stream.cuda_stream)),
                  "dt_hip_set_stream")

    def begin(self, band, dev_in_band, width):
        st = abi.BandState()
        lib.check(self.lib.dt_hip_pipe_band_begin(self.pipe.handle, C.byref(band), dev_in_band, C.byref(st)),
                  "dt_hip_pipe_band_begin")
        halo = count = None
        if st.halo_buf:
            rows = band.halo_top + band.rows + band.halo_bottom
            halo = device_view(st.halo_buf, (rows, st.row_bytes // 4), "<f4", self.device)
        if st.clipped_count:
            count = device_view(st.clipped_count, (1,), "<i8", self.device)
        return BandWork(halo, count, st)

    def resolve(self, band, work):
        lib.check(self.lib.dt_hip_pipe_band_resolve(self.pipe.handle, C.byref(band), C.byref(work.token)),
                  "dt_hip_pipe_band_resolve")

    def finish(self, band, work, dev_out_band):
        lib.check(self.lib.dt_hip_pipe_band_finish(self.pipe.handle, C.byref(band), C.byref(work.token), dev_out_band),
                  "dt_hip_pipe_band_finish")


def sum_clipped(work, bands, dist=None, group=None):
    """collective 1: the frame-wide clipped count (no-op for one band)"""
    if len(bands) > 1 and dist is not None and work.count is not None:
        dist.all_reduce(work.count, op=dist.ReduceOp.SUM, group=group)


def exchange_halo(work, bands, rank, dist=None, group=None):
    """collective 2: swap halo rows with the bands above and below (no-op for one band)"""
    n = len(bands)
    if n == 1 or dist is None or work.halo is None:
        return
    b = bands[rank]
    ops = []
    own0 = b.halo_top  # first own row inside the halo layout
    if rank > 0:
        need = bands[rank - 1].halo_bottom  # rows the band above reads from me
        if need:
            ops.append(dist.P2POp(dist.isend, work.halo[own0:own0 + need], rank - 1, group=group))
        if b.halo_top:
            ops.append(dist.P2POp(dist.irecv, work.halo[0:b.halo_top], rank - 1, group=group))
    if rank + 1 < n:
        need = bands[rank + 1].halo_top
        if need:
            ops.append(dist.P2POp(dist.isend, work.halo[own0 + b.rows - need:own0 + b.rows], rank + 1, group=group))
        if b.halo_bottom:
            ops.append(dist.P2POp(dist.irecv, work.halo[own0 + b.rows:own0 + b.rows + b.halo_bottom], rank + 1,
                                  group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def process_band(engine, bands, rank, dev_in_band, dev_out_band, width, dist=None, group=None):
    """one frame, this rank's band"""
    band = bands[rank]
    work = engine.begin(band, dev_in_band, width)
    sum_clipped(work, bands, dist, group)
    engine.resolve(band, work)
    exchange_halo(work, bands, rank, dist, group)
    engine.finish(band, work, dev_out_band)


def process_bands_locally(engine, bands, ins, outs, width):
    """all bands of a frame in one process, one after the other (single-GPU test of the band path):
    the same protocol with the two collectives done as tensor copies"""
    works = [engine.begin(b, i, width) for b, i in zip(bands, ins)]
    counts = [w.count for w in works if w.count is not None]
    if counts:
        total = sum(int(c.item()) for c in counts)
        for c in counts:
            c.fill_(total)
    for b, w in zip(bands, works):
        engine.resolve(b, w)
    local_halo(works, bands)
    for b, w, o in zip(bands, works, outs):
        engine.finish(b, w, o)


def local_halo(works, bands):
    n = len(bands)
    for r in range(n):
        w, b = works[r], bands[r]
        if w.halo is None:
            continue
        if r > 0 and b.halo_top:
            up, ub = works[r - 1], bands[r - 1]
            src0 = ub.halo_top + ub.rows - b.halo_top
            w.halo[0:b.halo_top].copy_(up.halo[src0:src0 + b.halo_top])
        if r + 1 < n and b.halo_bottom:
            dn, db = works[r + 1], bands[r + 1]
            w.halo[b.halo_top + b.rows:b.halo_top + b.rows + b.halo_bottom].copy_(
                dn.halo[db.halo_top:db.halo_top + b.halo_bottom])
