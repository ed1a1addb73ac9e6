#!/usr/bin/env python3
"""Generate tests/golden/{golden.npz,golden.json,frames/} from the UNTOUCHED reference.

Runs only in the build container (needs /root/reference and oracle/_ref/libpapof_ref.so, built by
`make -C oracle ref` from the reference sources where they lie).  What is committed is data only:

* frames/<W>/frame_0000{1,2}.jpg  -- the first two frames of the reference's benchmark sets
  images_New/HoChiMinhTraffic_10FPS_<W> (data files; decoded-pixel SHA-256 recorded in golden.json
  so a differing JPEG decoder is detected instead of silently changing the inputs);
* golden.json -- per case/output: shape + SHA-256 of the float64 bytes the reference produced
  (pins bit-for-bit equality);
* golden.npz  -- per case/output: a strided subsample (<= 20000 values, float64) for
  tolerance-based comparison and for diagnosing a SHA mismatch.

Usage:  python tests/golden/make_golden.py [case ...]
"""
import json
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cases  # noqa: E402
from _libs import RefLib  # noqa: E402

REF_IMAGES = "/root/reference/images_New"


def main():
    if not RefLib.available():
        raise SystemExit("oracle/_ref/libpapof_ref.so missing: run `make -C oracle ref` first")
    for res in cases.SIZES:
        dst = os.path.join(cases.FRAMES, res)
        os.makedirs(dst, exist_ok=True)
        for idx in (1, 2):
            name = "frame_%05d.jpg" % idx
            if not os.path.exists(os.path.join(dst, name)):
                shutil.copyfile(os.path.join(REF_IMAGES, "HoChiMinhTraffic_10FPS_" + res, name),
                                os.path.join(dst, name))
    jpath, npath = os.path.join(HERE, "golden.json"), os.path.join(HERE, "golden.npz")
    manifest = json.load(open(jpath)) if os.path.exists(jpath) else {"frames": {}, "cases": {}}
    arrays = dict(np.load(npath)) if os.path.exists(npath) else {}
    for res in cases.SIZES:
        for idx in (1, 2):
            px = cases.load_frame_u8(res, idx)
            manifest["frames"]["%s/%d" % (res, idx)] = {"shape": list(px.shape),
                                                        "sha": cases.sha(px.astype(np.float64))}
    lib = RefLib()
    todo = sys.argv[1:] or list(cases.CASES)
    for name in todo:
        t0 = time.time()
        out = cases.CASES[name](lib)
        manifest["cases"][name] = {}
        for k, a in out.items():
            manifest["cases"][name][k] = {"shape": list(a.shape), "sha": cases.sha(a)}
            arrays["%s|%s" % (name, k)] = cases.subsample(a)
        print("%-20s %6.1fs  %s" % (name, time.time() - t0, ", ".join(out)))
    json.dump(manifest, open(jpath, "w"), indent=1, sort_keys=True)
    np.savez_compressed(npath, **arrays)
    print("wrote", jpath, npath, "%.1f MB" % (os.path.getsize(npath) / 1e6))


if __name__ == "__main__":
    main()
