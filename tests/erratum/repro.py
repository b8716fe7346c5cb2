#!/usr/bin/env python3
"""Reproducer for ERRATUM.md: the corner-triangle defect of the reference (Source.cpp:1055-1062, inherited by 1083-1086).

Runs the UNMODIFIED reference (oracle/_ref, built from /root/reference by oracle/Makefile) where it is present, else the
CPU restatement (oracle/aai_oracle.c, bit-identical to the reference on every golden vector), next to the restatement's
geometrically exact policy, and prints the numbers quoted in ERRATUM.md.  Test infrastructure: never imported by the package.

    python tests/erratum/repro.py            # ~1 minute on one core
"""
import ctypes
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402


def reference(mode, src, sr, dr, iso, ang):
    if po.have_ref():
        return po.ref_run(mode, src, sr, dr, iso, ang).dst
    return po.oracle_run(mode, src, sr, dr, iso, ang, policy=po.POLICY_REFERENCE).dst


def exact(src, sr, dr, iso, ang):
    return po.oracle_run(po.MODE_EXACT, src, sr, dr, iso, ang, policy=po.POLICY_EXACT).dst


def pair_areas(policy, W, H, sr, dr, iso, ang, dx, dy):
    lib = po._load_oracle()
    lib.aai_oracle_pixel_pairs.restype = ctypes.c_int
    lib.aai_oracle_pixel_pairs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_double] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    cap = 4096
    xs, ys, ar = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float64)
    n = lib.aai_oracle_pixel_pairs(policy, W, H, sr, dr, iso[0], iso[1], ang, dx, dy, cap, xs.ctypes.data, ys.ctypes.data, ar.ctypes.data)
    assert 0 <= n <= cap
    return xs[:n], ys[:n], ar[:n]


def interior(a, margin):
    return a[margin:-margin, margin:-margin]


def main():
    if not po.have_oracle():
        po.build()
    print("reference =", "unmodified Source.cpp (oracle/_ref)" if po.have_ref() else "CPU restatement, policy REFERENCE (oracle/_ref absent)")
    geoms = [("config 3 geometry (8192:2731, 17.5 deg) on 384 x 384", 384, 8192.0, 2731.0, 17.5),
             ("config 5 geometry (1:4 up-sampling, 45 deg) on 96 x 96", 96, 1.0, 4.0, 45.0),
             ("reference's own default (150 -> 25.4 dpi, 1.5 deg) on 455 x 455", 455, 150.0, 25.4, 1.5)]
    for (name, W, sr, dr, ang) in geoms:
        iso = ((W - 1) / 2.0, (W - 1) / 2.0)
        print("\n==", name)
        # 1. constant image: the normalisation by the (wrong) area sum hides the defect completely
        const = np.full((W, W), 7.25)
        ref_c = reference(po.MODE_EXACT, const, sr, dr, iso, ang)
        live = ref_c != 0
        print("constant image 7.25: reference output min %.17g max %.17g over %d non-empty pixels" % (ref_c[live].min(), ref_c[live].max(), int(live.sum())))
        # 2. linear ramp f(x, y) = x: an exact area average over a square returns f at the square's centre
        ramp = np.tile(np.arange(W, dtype=np.float64), (W, 1))
        ref_r, ex_r = reference(po.MODE_EXACT, ramp, sr, dr, iso, ang), exact(ramp, sr, dr, iso, ang)
        m = max(4, int(math.ceil(0.15 * ref_r.shape[0])))
        d = np.abs(interior(ref_r, m) - interior(ex_r, m))
        live = interior(ex_r, m) != 0
        print("ramp f = x, interior pixels: |reference - exact| max %.4f mean %.4f source pixels (exact policy reproduces the centre to 1e-12)" % (d[live].max(), d[live].mean()))
        # 3. noise image: how many output pixels change, and by how much
        noise = po.synth_image(W, W, 1).astype(np.float64)
        ref_n, ex_n = reference(po.MODE_EXACT, noise, sr, dr, iso, ang), exact(noise, sr, dr, iso, ang)
        nz = ex_n != 0
        rel = np.abs(ref_n - ex_n)[nz] / np.abs(ex_n[nz])
        print("uniform noise [0,1): %.1f %% of the %d output pixels differ by more than 1e-6 relative (%.1f %% of the non-empty ones); max %.3f, median of the changed %.4f" %
              (100.0 * (rel > 1e-6).sum() / ref_n.size, ref_n.size, 100.0 * (rel > 1e-6).mean(), rel.max(), np.median(rel[rel > 1e-6]) if (rel > 1e-6).any() else 0.0))
        # 4. the area sums themselves (restatement, bit-identical to the reference): interior dst pixels should see exactly L^2
        L2 = None
        ratios = []
        dH, dW = ref_n.shape
        rng = np.random.default_rng(3)
        for _ in range(300):
            dx, dy = int(rng.integers(m, dW - m)), int(rng.integers(m, dH - m))
            if ex_n[dy, dx] == 0:
                continue
            _, _, a_ref = pair_areas(0, W, W, sr, dr, iso, ang, dx, dy)
            _, _, a_ex = pair_areas(1, W, W, sr, dr, iso, ang, dx, dy)
            if L2 is None:
                L2 = a_ex.sum()
            if abs(a_ex.sum() - L2) < 1e-9:           # wholly inside the image
                ratios.append(a_ref.sum() / a_ex.sum())
        ratios = np.array(ratios)
        print("area sums of %d interior dst pixels: exact policy = L^2 = %.6f each; reference / L^2 in [%.4f, %.4f], standard deviation %.4f" %
              (len(ratios), L2, ratios.min(), ratios.max(), ratios.std()))
    # 5. one pair: a right edge cutting off the top-right corner of a source pixel
    W, sr, dr, ang = 64, 3.0, 1.0, 17.5
    iso = (31.5, 31.5)
    worst = None
    for dx in range(6, 16):
        for dy in range(6, 16):
            xs, ys, a_ref = pair_areas(0, W, W, sr, dr, iso, ang, dx, dy)
            _, _, a_ex = pair_areas(1, W, W, sr, dr, iso, ang, dx, dy)
            k = int(np.argmax(np.abs(a_ref - a_ex)))
            if worst is None or abs(a_ref[k] - a_ex[k]) > abs(worst[4] - worst[5]):
                worst = (dx, dy, int(xs[k]), int(ys[k]), float(a_ref[k]), float(a_ex[k]))
    print("\n== one pair (64 x 64 image, 3:1, 17.5 deg, isocenter (31.5, 31.5)): dst pixel (%d, %d) x source pixel (%d, %d): reference area %.6f, true overlap %.6f" % worst)


if __name__ == "__main__":
    main()
