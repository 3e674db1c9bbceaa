"""Host-side bookkeeping of the stencil modules: how many wavelet scales a frame gets.  Pure
integer/float arithmetic restated from the reference for reporting (bench.py); the device
code computes the same numbers itself (diffuse.hip scales_of(), denoiseprofile.hip setup())."""
import math

import numpy as np

B_SPLINE_SIGMA = np.float32(1.0553651328015339)  # src/pixel/bspline.h:39


def diffuse_scales(piece, d):
    """num_steps_to_reach_equivalent_sigma() clamped to [1, 10]: src/pixel/bspline.h:68-80, src/iop/diffuse.c:1183-1184"""
    f = np.float32
    zoom = f(d.iscale / piece.roi_in.scale)
    final_radius = f(f(d.radius + d.radius_center) * f(2.0) / zoom)
    s = 0
    radius = B_SPLINE_SIGMA
    while radius < final_radius:
        s += 1
        radius = f(np.sqrt(f(radius * radius + f(f(1 << s) * B_SPLINE_SIGMA) ** 2)))
    return min(max(s + 1, 1), 10)


def denoiseprofile_bands(piece):
    """max_scale of process_wavelets(), src/iop/denoiseprofile.c:1301-1317"""
    in_scale = min(piece.roi_in.scale, 1.0)
    big = max(piece.roi_in.width, piece.roi_in.height)
    supp0 = min(2 * (2 << 6) + 1, big * 0.2)
    i0 = math.log2((supp0 - 1.0) * 0.5)
    n = 0
    while n < 7:
        supp = 2 * (2 << n) + 1
        i_in = math.log2((supp / in_scale - 1) * 0.5) - 1.0
        if 1.0 - (i_in + 0.5) / i0 < 0.0:
            break
        n += 1
    return n
