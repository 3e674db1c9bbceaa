"""One frame over several GPUs: row bands with one halo exchange (BASELINE.json config 4).

Replaces default_process_tiling_cl() (src/develop/tiling.c:1394) for the multi-GPU case.  Each
rank owns a band of output rows (include/ansel_hip.h section 3b), cut on the frame's RCD tile rows so
the assembled result is bit-identical to the unsplit frame.  Per frame and rank:

    engine.begin    CFA stages on the band's own rows                      (device)
    all-reduce      clipped-photosite count of the highlights bypass       (8 bytes, RCCL)
    engine.resolve  bypass decision on the frame-wide count                (device)
    send/recv       `halo` = 9 mosaic rows with the band above and below   (RCCL over xGMI)
    engine.finish   demosaic on the frame's tile rows + RGBA stages        (device)

On the light pipe there is no other data-path communication: every module after the demosaic is pointwise.
The stencil modules of the full pipe (denoiseprofile, diffuse, nlmeans) make engine.finish() resumable: it
stops in front of such a module and hands back a BandRequest --

    request.halo    RGBA rows [min(h, row0)][rows][min(h, H - row1)]: send/recv of h rows with each neighbour
    request.sums    the frame-wide table of the profiled wavelets' partial sums, own entries filled, the rest
                    zero: all-reduce(SUM), exact in any order
    request.relay   local contrast's bilateral grid (one accumulation over the frame in pixel order): the bands take
                    turns -- recv from the band above, engine.relay() adds the own rows, send to the band below --
                    and the last band broadcasts the complete grid (a few hundred KB per hop)

-- the driver serves it and calls finish() again (serve_request / process_band below).
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


class BandRequest:
    """what a resumable engine.finish() asks of the other bands: `halo` = the [top][rows][bottom] rows of the
    module input whose first / last part the neighbours fill (`h` rows each, clipped at the frame), `sums` =
    a float64 vector every band must hold the frame-wide version of -- `sum_planes` tables of [frame rows][...] doubles in
    which a band has filled its own rows and left the rest zero: the bands all-gather their row segments (sum_planes == 0:
    no such layout, all-reduce) --, `relay` = a buffer the bands fill in turn (engine.relay()); each may be None"""

    def __init__(self, halo, h, sums, relay=None, sum_planes=0):
        self.halo = halo
        self.h = h
        self.sums = sums
        self.relay = relay
        self.sum_planes = sum_planes


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
        self.frame_height = int(device_pipe.nodes[0].piece.roi_out.height)

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
        """None when the band is done, else the BandRequest to serve before calling again"""
        st = work.token
        rc = self.lib.dt_hip_pipe_band_finish(self.pipe.handle, C.byref(band), C.byref(st), dev_out_band)
        if rc != abi.DT_HIP_BAND_EXCHANGE:
            lib.check(rc, "dt_hip_pipe_band_finish")
            return None
        halo = sums = relay = None
        if st.relay_buf:
            relay = device_view(st.relay_buf, (st.relay_bytes // 4,), "<f4", self.device)
        if st.halo_rows > 0:
            top = min(st.halo_rows, band.row0)
            bottom = min(st.halo_rows, self.frame_height - band.row0 - band.rows)
            halo = device_view(st.halo_buf, (top + band.rows + bottom, st.row_bytes // 4), "<f4", self.device)
        if st.sum_buf:
            sums = device_view(st.sum_buf, (st.sum_count,), "<f8", self.device)
        return BandRequest(halo, st.halo_rows, sums, relay, sum_planes=int(st.sum_planes))

    def relay(self, band, work):
        """this band's turn in a relay stop: its rows on top of what the relay buffer holds"""
        lib.check(self.lib.dt_hip_pipe_band_relay(self.pipe.handle, C.byref(band), C.byref(work.token)),
                  "dt_hip_pipe_band_relay")

    def abort(self, work):
        """free a band whose walk will not be finished"""
        self.lib.dt_hip_pipe_band_abort(self.pipe.handle, C.byref(work.token))


def sum_clipped(work, bands, dist=None, group=None):
    """collective 1: the frame-wide clipped count (no-op for one band)"""
    if len(bands) > 1 and dist is not None and work.count is not None:
        dist.all_reduce(work.count, op=dist.ReduceOp.SUM, group=group)
        _log("clipped_count_all_reduce", 8)


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
        _log("mosaic_halo_send_recv", sum(op.tensor.numel() * op.tensor.element_size() for op in ops if op.op is dist.isend))
        for req in dist.batch_isend_irecv(ops):
            req.wait()


# what the collectives of the last frame moved (bench.py --mode tiled reports it): name -> [calls, bytes sent by this rank]
EXCHANGE_LOG = {}


def _log(name, nbytes):
    e = EXCHANGE_LOG.setdefault(name, [0, 0])
    e[0] += 1
    e[1] += int(nbytes)


def gather_sums(req, bands, rank, dist, group=None):
    """the frame-wide table of partial sums from the bands' own row segments: every entry is non-zero in exactly one
    band's table, so nothing is added -- an all-gather of the segments (padded to the longest), the peer of the strided
    peer copies of dt_hip_pipe_process_bands().  (Round 3 all-reduced the whole table: twice the bytes on a ring, and a
    sum where a copy does.)"""
    n = len(bands)
    height = bands[-1].row0 + bands[-1].rows
    t = req.sums.view(req.sum_planes, -1)
    per_row = t.shape[1] // height
    offs = [b.row0 * per_row for b in bands]
    lens = [b.rows * per_row for b in bands]
    longest = max(lens)
    stage = t.new_zeros((req.sum_planes, longest))
    stage[:, :lens[rank]] = t[:, offs[rank]:offs[rank] + lens[rank]]
    parts = [t.new_empty((req.sum_planes, longest)) for _ in range(n)]
    dist.all_gather(parts, stage, group=group)
    for j in range(n):
        if j != rank:
            t[:, offs[j]:offs[j] + lens[j]] = parts[j][:, :lens[j]]
    _log("sums_all_gather", stage.numel() * stage.element_size())


def _halo_parts(bands, k, h, height):
    b = bands[k]
    return min(h, b.row0), min(h, height - b.row0 - b.rows)


def serve_request(req, bands, rank, dist=None, group=None, engine=None, work=None):
    """the collectives of one BandRequest (no-ops for one band, except the band's own turn of a relay)"""
    n = len(bands)
    if req.relay is not None:
        many = n > 1 and dist is not None
        if many and rank > 0:
            dist.recv(req.relay, src=rank - 1, group=group)
        engine.relay(bands[rank], work)
        if many and rank + 1 < n:
            dist.send(req.relay, dst=rank + 1, group=group)
            _log("grid_relay_send", req.relay.numel() * req.relay.element_size())
        if many:
            dist.broadcast(req.relay, src=n - 1, group=group)
            _log("grid_broadcast", req.relay.numel() * req.relay.element_size() if rank == n - 1 else 0)
    if n == 1 or dist is None:
        return
    if req.sums is not None:
        if req.sum_planes > 0:
            gather_sums(req, bands, rank, dist, group)
        else:
            dist.all_reduce(req.sums, op=dist.ReduceOp.SUM, group=group)
            _log("sums_all_reduce", req.sums.numel() * req.sums.element_size())
    if req.halo is None:
        return
    height = bands[-1].row0 + bands[-1].rows
    b, h = bands[rank], req.h
    top, bottom = _halo_parts(bands, rank, h, height)
    for k in (rank - 1, rank + 1):
        if 0 <= k < n and bands[k].rows < h:
            raise lib.AnselHipError("band %d owns %d rows, fewer than the %d halo rows its neighbour needs: "
                                    "use fewer bands" % (k, bands[k].rows, h))
    ops = []
    if rank > 0:
        need = _halo_parts(bands, rank - 1, h, height)[1]  # rows the band above reads from me
        ops.append(dist.P2POp(dist.isend, req.halo[top:top + need], rank - 1, group=group))
        ops.append(dist.P2POp(dist.irecv, req.halo[0:top], rank - 1, group=group))
    if rank + 1 < n:
        need = _halo_parts(bands, rank + 1, h, height)[0]
        ops.append(dist.P2POp(dist.isend, req.halo[top + b.rows - need:top + b.rows], rank + 1, group=group))
        ops.append(dist.P2POp(dist.irecv, req.halo[top + b.rows:top + b.rows + bottom], rank + 1, group=group))
    _log("rgba_halo_send_recv", sum(op.tensor.numel() * op.tensor.element_size() for op in ops if op.op is dist.isend))
    for r in dist.batch_isend_irecv(ops):
        r.wait()


def process_band(engine, bands, rank, dev_in_band, dev_out_band, width, dist=None, group=None):
    """one frame, this rank's band"""
    band = bands[rank]
    EXCHANGE_LOG.clear()
    work = engine.begin(band, dev_in_band, width)
    try:
        sum_clipped(work, bands, dist, group)
        engine.resolve(band, work)
        exchange_halo(work, bands, rank, dist, group)
        while True:
            req = engine.finish(band, work, dev_out_band)
            if req is None:
                break
            serve_request(req, bands, rank, dist, group, engine, work)
    except Exception:
        if hasattr(engine, "abort"):
            engine.abort(work)  # a finish() that failed has freed the state already: abort is then a no-op
        raise


def process_bands_locally(engine, bands, ins, outs, width):
    """all bands of a frame in one process, in lockstep (single-GPU test of the band path):
    the same protocol with the collectives done as tensor copies"""
    works = [engine.begin(b, i, width) for b, i in zip(bands, ins)]
    counts = [w.count for w in works if w.count is not None]
    if counts:
        total = sum(int(c.item()) for c in counts)
        for c in counts:
            c.fill_(total)
    for b, w in zip(bands, works):
        engine.resolve(b, w)
    local_halo(works, bands)
    height = bands[-1].row0 + bands[-1].rows
    while True:
        reqs = [engine.finish(b, w, o) for b, w, o in zip(bands, works, outs)]
        if all(r is None for r in reqs):
            return
        assert all(r is not None for r in reqs), "the bands of a frame stop at the same modules"
        if reqs[0].relay is not None:
            for k, (r, b, w) in enumerate(zip(reqs, bands, works)):
                if k:
                    r.relay.copy_(reqs[k - 1].relay)
                engine.relay(b, w)
            for r in reqs[:-1]:
                r.relay.copy_(reqs[-1].relay)
        if reqs[0].sums is not None:
            total = reqs[0].sums.clone()
            for r in reqs[1:]:
                total += r.sums
            for r in reqs:
                r.sums.copy_(total)
        if reqs[0].halo is not None:
            h = reqs[0].h
            for k, (r, b) in enumerate(zip(reqs, bands)):
                top, bottom = _halo_parts(bands, k, h, height)
                if top:
                    up, ub = reqs[k - 1], bands[k - 1]
                    assert ub.rows >= top
                    utop = _halo_parts(bands, k - 1, h, height)[0]
                    r.halo[0:top].copy_(up.halo[utop + ub.rows - top:utop + ub.rows])
                if bottom:
                    dn = reqs[k + 1]
                    assert bands[k + 1].rows >= bottom
                    dtop = _halo_parts(bands, k + 1, h, height)[0]
                    r.halo[top + b.rows:top + b.rows + bottom].copy_(dn.halo[dtop:dtop + bottom])


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
