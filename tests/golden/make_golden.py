#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory from the UNMODIFIED reference.

Needs oracle/_ref/ref_dump (`make -C oracle ref`, which compiles /root/reference/src in place).
Each <name>.npz holds the reference's own per-pixel output for one bundled scene at a reduced
resolution: ids (primary-hit triangle index, -1 = none), rgb (raw float colour), minmax, gamma and
the SHA-256 of the 24-bpp BMP the reference writes.  known_answers.json holds SHA-256 digests of the
id buffer and BMP at each scene's native resolution.  These are data (inputs = the bundled scene
files, outputs = what the reference computed); no reference source is stored here.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (prefix, overrides, (w, h), needs textures/brickwall.bmp in cwd)
CASES = {
    "test": ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, (160, 120), False),
    "s2": ("scenes/s2/s2", {}, (160, 120), False),
    "s3": ("scenes/s3/s3", {}, (160, 120), False),
    "s5": ("scenes/s5/s5", {}, (160, 120), False),
    "pokoj": ("scenes/pokoj/s2", {}, (160, 120), False),
    "pokoj_cegly": ("scenes/pokoj/s2", {"cfg": "scenes/cegly.cfg"}, (160, 120), True),
    "pokoj_kafelki": ("scenes/pokoj/s2", {"cfg": "scenes/kafelki.cfg"}, (96, 72), True),
    "temple": ("scenes/temple/temple", {}, (160, 120), False),
    "ulica": ("scenes/ulica/ulica", {}, (48, 64), False),
    # planar light of s2 with pokoj's surfaces (kt > 0 and kr > 0): rand() is also drawn at secondary hits
    "s2_glass": ("scenes/s2/s2", {"atr": "scenes/pokoj/s2.atr"}, (120, 90), False),
}

NATIVE = {  # native-resolution known answers (ids + BMP digests)
    "test": ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}),
    "s2": ("scenes/s2/s2", {}),
    "s3": ("scenes/s3/s3", {}),
    "s5": ("scenes/s5/s5", {}),
    "pokoj": ("scenes/pokoj/s2", {}),
    "temple": ("scenes/temple/temple", {}),
    "ulica": ("scenes/ulica/ulica", {}),
}


# larger frames kept as id buffer + 8-bit image (the reference's own BMP pixels, top row first, R G B)
LARGE = {
    "ulica_384x216": ("scenes/ulica/ulica", {}, (384, 216)),   # ~3.5 minutes of reference time
    "ulica_768x432": ("scenes/ulica/ulica", {}, (768, 432)),   # ~13 minutes: 64 lights, depth 5, several chunks of tiles on the GPU
}


# gamma lists handed to the reference's rtVisualizedSceneToBitmap (its param1, src/raytrace.c:439-466) on the s3 frame of
# CASES: the 8-bit images it writes (top row first, R G B)
GAMMA_LISTS = [[2.5, 1.0], [1.0, 2.0, 3.0], [0.45, 2.2, 2.5, 4.0], [2.5, 2.5], [0.45], []]


def gamma_lists():
    prefix, over, (w, h), _ = CASES["s3"]
    out = {}
    for i, gl in enumerate(GAMMA_LISTS):
        r = O.run_reference(os.path.join(ROOT, prefix), width=w, height=h, cwd=os.path.join(ROOT, "scenes"), want_bmp=True, gammas=gl)
        bmp = r["bmp"]
        img = np.frombuffer(bmp[54:], np.uint8).reshape(h, (w * 3 + 3) // 4 * 4)[:, :w * 3][::-1].reshape(h, w, 3)[:, :, ::-1]
        out["img%d" % i] = np.ascontiguousarray(img)
        out["bmp_sha256_%d" % i] = hashlib.sha256(bmp).hexdigest()
        print("gamma list", gl, hashlib.sha256(bmp).hexdigest())
    np.savez_compressed(os.path.join(HERE, "gamma_lists.npz"), lists=json.dumps(GAMMA_LISTS), case="s3", **out)


def large(only=None):
    for name, (prefix, over, (w, h)) in LARGE.items():
        if only and name not in only:
            continue
        r = O.run_reference(os.path.join(ROOT, prefix), width=w, height=h, cwd=os.path.join(ROOT, "scenes"), want_bmp=True)
        bmp = r["bmp"]
        img = np.frombuffer(bmp[54:], np.uint8).reshape(h, (w * 3 + 3) // 4 * 4)[:, :w * 3][::-1].reshape(h, w, 3)[:, :, ::-1]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=r["ids"], img=np.ascontiguousarray(img), minmax=r["minmax"],
                            gamma=np.float32(r["gamma"]), bmp_sha256=hashlib.sha256(bmp).hexdigest(), prefix=prefix,
                            seconds=r["seconds"])
        print(name, r["ids"].shape, "hits", int((r["ids"] >= 0).sum()), "%.2fs" % r["seconds"], hashlib.sha256(bmp).hexdigest())


def main():
    os.chdir(ROOT)
    if not O.have_reference():
        sys.exit("build oracle/_ref first: make -C oracle ref")
    if len(sys.argv) > 1 and sys.argv[1] == "gammas":
        return gamma_lists()         # make_golden.py gammas
    if len(sys.argv) > 1 and sys.argv[1] == "large":
        return large(sys.argv[2:])   # make_golden.py large [<case> ...]
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "only" else None   # make_golden.py only <case> ...
    for name, (prefix, over, (w, h), tex) in CASES.items():
        if only is not None and name not in only:
            continue
        cwd = ROOT if tex else os.path.join(ROOT, "scenes")  # scenes/ has no textures/ directory
        over_abs = {k: os.path.join(ROOT, v) for k, v in over.items()}
        r = O.run_reference(os.path.join(ROOT, prefix), width=w, height=h, cwd=cwd, want_bmp=True, **over_abs)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ids=r["ids"], rgb=r["rgb"], minmax=r["minmax"],
                            gamma=np.float32(r["gamma"]), bmp_sha256=hashlib.sha256(r["bmp"]).hexdigest(),
                            prefix=prefix, overrides=json.dumps(over), texture=tex, seconds=r["seconds"])
        print(name, r["ids"].shape, "hits", int((r["ids"] >= 0).sum()), "%.2fs" % r["seconds"])
    if only is not None:
        return
    known = {}
    for name, (prefix, over) in NATIVE.items():
        over_abs = {k: os.path.join(ROOT, v) for k, v in over.items()}
        r = O.run_reference(os.path.join(ROOT, prefix), cwd=os.path.join(ROOT, "scenes"), want_bmp=True, **over_abs)
        known[name] = dict(prefix=prefix, overrides=over, width=int(r["ids"].shape[1]), height=int(r["ids"].shape[0]),
                           primary_hits=int((r["ids"] >= 0).sum()), distinct_ids=int(len(np.unique(r["ids"]))),
                           ids_sha256=hashlib.sha256(np.ascontiguousarray(r["ids"], np.int32).tobytes()).hexdigest(),
                           bmp_sha256=hashlib.sha256(r["bmp"]).hexdigest(),
                           minmax=[float(x) for x in r["minmax"]], seconds=r["seconds"])
        print(name, known[name])
    json.dump(known, open(os.path.join(HERE, "known_answers.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
