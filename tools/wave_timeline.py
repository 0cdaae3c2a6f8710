#!/usr/bin/env python3
"""Where and when the waves of fastconv1024_pairs_kernel ran (diagnostic build; GPU box only):

    tools/build_variant.sh stamp -DJDSP_STAMP=1
    JDSP_LIB=build/variants/stamp.so python tools/wave_timeline.py

Every wave stamps s_memrealtime (100 MHz) at entry and exit plus HW_ID / XCC_ID.  Printed: the launch's span, how many
waves were alive over time, per-CU and per-SIMD wave counts, and the spread of start / end times.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jeicyboodsp_amd  # noqa: E402
from jeicyboodsp_amd._lib import lib as L  # noqa: E402


def main():
    eng = jeicyboodsp_amd.Engine(0)
    rng = np.random.default_rng(0)
    h = rng.normal(size=(2, 256)) * np.exp(-np.arange(256) / 40.0)[None, :] * 0.1
    nb = 65536
    pcm = torch.from_numpy(np.clip(np.rint(rng.normal(0, 2000, nb * 769)), -32768, 32767).astype(np.int16)).cuda()
    fc = eng.fastconv(h, 1024)
    for _ in range(30):
        fc.process(pcm)
    torch.cuda.synchronize()
    n = 3072
    buf = np.zeros(3 * n, np.uint64)
    L.jdsp_debug_wave_stamps.argtypes = [C.c_void_p, C.c_int]
    assert L.jdsp_debug_wave_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
    t0, t1, hw = buf[0::3].astype(np.int64), buf[1::3].astype(np.int64), buf[2::3]
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0                     # microseconds
    print("launch span %.1f us; wave lifetime mean %.1f us (min %.1f, max %.1f)" % (t1.max(), (t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max()))
    print("starts: 50%% by %.1f us, 90%% by %.1f, 99%% by %.1f, last %.1f" % tuple(np.percentile(t0, [50, 90, 99, 100])))
    print("ends  : 1%% by %.1f us, 10%% by %.1f, 50%% by %.1f, 90%% by %.1f, last %.1f" % tuple(np.percentile(t1, [1, 10, 50, 90, 100])))
    for t in np.linspace(0, t1.max(), 13):
        print("  t = %6.1f us: %4d waves alive" % (t, int(((t0 <= t) & (t1 > t)).sum())))
    hwl = (hw & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    simd, cu, sh, se = (hwl >> 4) & 3, (hwl >> 8) & 15, (hwl >> 12) & 1, (hwl >> 13) & 7
    place = xcc * 100000 + se * 1000 + sh * 100 + cu
    cus, per_cu = np.unique(place, return_counts=True)
    print("distinct (xcc, se, sh, cu): %d; waves per CU: min %d max %d; histogram %s" % (cus.size, per_cu.min(), per_cu.max(),
          dict(zip(*np.unique(per_cu, return_counts=True)))))
    ps, per_simd = np.unique(place * 10 + simd, return_counts=True)
    print("waves per SIMD histogram %s" % dict(zip(*np.unique(per_simd, return_counts=True))))
    print("per-XCC wave counts %s" % dict(zip(*np.unique(xcc, return_counts=True))))
    fc.close()
    eng.close()


if __name__ == "__main__":
    main()
