#!/usr/bin/env python3
"""Which engine carries a frame's download, and what it costs the kernels beside it (round 6).  The full pipe on a stream of frames, the
download of frame n (a) enqueued behind a stream-wait on the frame's last kernel -- what dt_hip_batch_* did --, (b) enqueued by a helper
thread once the HOST has seen that event complete (the download stream is then idle and waits for nothing), (c) no download at all.

    python tools/batch_sdma_probe.py [--size 100MP] [--frames 8]
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="100MP")
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from ansel_amd import lib, params, pipe, synth
    l = lib.init()
    dev = torch.device("cuda", 0)
    w, h = bench.frame_size(args.size)
    lut = params.srgb_encode_lut()
    d_lut = lib.DeviceBuffer.from_numpy(0, lut)
    nodes = bench.build_pipe(w, h, d_lut.ptr, lut, bench.have_filmic(), "full")
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    lib.check(l.dt_hip_set_stream(0, C.c_void_p(A.cuda_stream)), "set_stream")
    ex = pipe.DevicePipe(0, nodes, fusion=True)
    raw = torch.from_numpy(synth.bayer_mosaic_tiled(w, h, seed=1).astype(np.int16)).to(dev)
    outs = [torch.empty((h, w, 4), dtype=torch.int16, device=dev) for _ in range(3)]
    pins = [torch.empty((h, w, 4), dtype=torch.int16).pin_memory() for _ in range(3)]
    U = torch.cuda.Stream(dev)
    raw_pin = raw.cpu().pin_memory()
    ins = [torch.empty_like(raw) for _ in range(3)]
    for mode in ("stream_wait", "stream_wait+upload", "stream_wait+upload_same_stream", "stream_wait+upload_host_sync", "no_download+upload"):
        done = [torch.cuda.Event() for _ in range(args.frames + 3)]
        down = [None] * (args.frames + 3)
        lock = threading.Condition()

        def downloader():
            for k in range(args.frames + 3):
                with lock:
                    lock.wait_for(lambda: down[k] == "go")
                done[k].synchronize()
                with torch.cuda.stream(B):
                    pins[k % 3].copy_(outs[k % 3], non_blocking=True)
                    e = torch.cuda.Event()
                    e.record(B)
                with lock:
                    down[k] = e
                    lock.notify_all()
        th = threading.Thread(target=downloader) if mode == "host_wait_thread" else None
        if th:
            th.start()
        t0 = None
        for k in range(args.frames + 3):
            if k == 3:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            if k >= 3 and not mode.startswith("no_download"):  # the slot's previous download
                if mode == "host_wait_thread":
                    with lock:
                        lock.wait_for(lambda: down[k - 3] not in (None, "go"))
                down[k - 3].synchronize()
            src = raw
            if mode.endswith("+upload"):
                with torch.cuda.stream(U):
                    ins[k % 3].copy_(raw_pin, non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(U)
                A.wait_event(up)
                src = ins[k % 3]
            elif mode.endswith("+upload_same_stream"):
                with torch.cuda.stream(A):
                    ins[k % 3].copy_(raw_pin, non_blocking=True)
                src = ins[k % 3]
            elif mode.endswith("+upload_host_sync"):
                with torch.cuda.stream(U):
                    ins[k % 3].copy_(raw_pin, non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(U)
                up.synchronize()
                src = ins[k % 3]
            ex.process(src.data_ptr(), outs[k % 3].data_ptr())
            done[k].record(A)
            if mode.startswith("stream_wait"):
                B.wait_event(done[k])
                with torch.cuda.stream(B):
                    pins[k % 3].copy_(outs[k % 3], non_blocking=True)
                    e = torch.cuda.Event()
                    e.record(B)
                down[k] = e
            elif mode == "host_wait_thread":
                with lock:
                    down[k] = "go"
                    lock.notify_all()
        if th:
            th.join()
        torch.cuda.synchronize(dev)
        print("%-18s %.2f ms per frame" % (mode, (time.perf_counter() - t0) / args.frames * 1e3), flush=True)
    ex.close()


if __name__ == "__main__":
    main()
