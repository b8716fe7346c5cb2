"""TEST INFRASTRUCTURE -- replays the small and knife-edge golden cases through sanitizer builds of the CPU side.

Run by tests/test_sanitizers.py in a child process with the AddressSanitizer runtime preloaded:
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/sanitize_replay.py <oracle.so> <hostemu.so>
Both libraries are built with -fsanitize=address,undefined -fno-sanitize-recover=all, so any out-of-bounds access or
undefined arithmetic aborts the process; the results are also compared with the goldens (bit-exact for the oracle).
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(oracle_so, emu_so):
    from oracle import pyoracle as po
    import area_average_interpolation_amd as aai
    from area_average_interpolation_amd import _lib as L
    po._ORACLE_SO = oracle_so                     # the sanitizer build instead of oracle/liboracle.so
    po._orc = None
    emu = ctypes.CDLL(emu_so)
    emu.aai_emu_resample.restype = ctypes.c_int
    emu.aai_emu_resample.argtypes = [ctypes.POINTER(L.Request), ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    emu.aai_emu_use_quad.argtypes = [ctypes.c_int]

    def replay(rq, src):
        src = np.ascontiguousarray(src, dtype=np.float32)
        dW, dH, ax = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert emu.aai_emu_resample(ctypes.byref(rq), src.ctypes.data, None, ctypes.byref(dW), ctypes.byref(dH), ctypes.byref(ax)) == 0
        out = np.empty((dH.value, dW.value), np.float32)
        emu.aai_emu_resample(ctypes.byref(rq), src.ctypes.data, out.ctypes.data, ctypes.byref(dW), ctypes.byref(dH), ctypes.byref(ax))
        return out

    checked = 0
    for name, prefix, stride in (("small_cases.npz", "c", 1), ("knife_cases.npz", "k", 3)):
        z = np.load(os.path.join(ROOT, "tests", "golden", name))
        manifest = json.loads(bytes(z["manifest"]).decode())
        for i in range(0, len(manifest), stride):
            c = manifest[i]
            src = po.synth_image(c["W"], c["H"], c["seed"])
            for mode, tag in ((1, "exact"), (2, "fast")):
                gold = z["%s%03d_%s" % (prefix, i, tag)]
                r = po.oracle_run(mode, src.astype(np.float64), c["src_res"], c["dst_res"], c["iso"], c["angle"])
                assert r.ok and np.array_equal(r.dst, gold), (name, i, tag)
                rq = aai.make_request(c["W"], c["H"], c["src_res"], c["dst_res"], c["iso"], c["angle"], mode=mode)
                for quad in (0, 1):
                    emu.aai_emu_use_quad(quad)
                    out = replay(rq, src)
                    err = np.abs(out.astype(np.float64) - gold) / np.maximum(np.abs(gold), 1e-3)
                    assert out.shape == gold.shape and err.max() <= 1e-5, (name, i, tag, quad, float(err.max()))
                checked += 1
    print("sanitize_replay: %d golden outputs replayed, clean" % checked)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
