"""The drop-in boundary against the reference as it is: oracle/_ref/raytrace_dropin is the reference's UNMODIFIED
main.c, scene.c, bitmap.c, error.c, stringtools.c, preprocess.c, intersection.c (compiled where they lie by
oracle/Makefile) linked with integration/raytrace_gpu.c, compiled against the reference's own headers, and librtgpu.so
in place of raytrace.c, voxelize.c, texture.c.  Run like the reference's command line (src/main.c:36-104), it has to
write the BMP files the all-CPU reference writes (digests: tests/golden/known_answers.json, made by the unmodified
reference).  The binary is built in the build container only (it needs /root/reference); without it these tests skip."""
import hashlib
import json
import os
import re
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "raytrace_dropin")
KNOWN = json.load(open(os.path.join(ROOT, "tests/golden/known_answers.json")))

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/raytrace_dropin not built (make -C oracle dropin)")]


def run(args, cwd, timeout=300):
    t0 = time.time()
    p = subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    return p, time.time() - t0


@pytest.mark.parametrize("name", ["s3", "temple", "pokoj", "test"])
def test_reference_command_line_with_the_gpu_path_writes_the_reference_bmp(tmp_path, name):
    k = KNOWN[name]
    out = str(tmp_path / "o.bmp")
    args = ["-s", os.path.join(ROOT, k["prefix"]), "-o", out]
    for opt, key in (("-a", "atr"),):
        if key in k["overrides"]:
            args += [opt, os.path.join(ROOT, k["overrides"][key])]
    p, secs = run(args, cwd=os.path.join(ROOT, "scenes"))  # (no textures/ directory there: texture off, as for the digests)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == k["bmp_sha256"]
    m = re.search(r"Time taken: ([0-9.]+) seconds", p.stdout)
    print("%s %dx%d: process %.2f s, rtVisualizedSceneRaytrace %s s of CPU time (first frame: context, upload, queues)"
          % (name, k["width"], k["height"], secs, m.group(1) if m else "?"))


def test_reference_command_line_temple_4k(tmp_path):
    """The headline frame through the reference's own main(): resolution is the last line of the .cam file
    (src/scene.c:655-657), so the test writes a camera file with `3840 2160`."""
    cam = open(os.path.join(ROOT, "scenes/temple/temple.cam")).read().rstrip().splitlines()
    cam[-1] = "3840 2160"
    (tmp_path / "t4k.cam").write_text("\n".join(cam) + "\n")
    out = str(tmp_path / "o.bmp")
    p, secs = run(["-s", os.path.join(ROOT, "scenes/temple/temple"), "-c", str(tmp_path / "t4k.cam"), "-o", out], cwd=os.path.join(ROOT, "scenes"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == "47a5b58a51f53a8ff5cc73fc010f1738942488e14a0019eb62aa63bb5cb70f7a"
    print("temple 3840x2160 through the reference's main(): process %.2f s" % secs)


def test_reference_command_line_reports_gpu_errors_through_errno(tmp_path):
    """A voxel mode with non-positive coefficients is E_INVALID_PARAM_VALUE in the reference (src/voxelize.c:169-191):
    rtVisualizedSceneRaytrace returns NULL with errno = 7; the reference's main() turns that errno into a warning,
    clears it and leaves without an image (src/main.c:190-196), exit status 0."""
    cfg = tmp_path / "bad.cfg"
    cfg.write_text("voxmode FIXED\nvoxparams 0 0 0\n")
    p, _ = run(["-s", os.path.join(ROOT, "scenes/s3/s3"), "-C", str(cfg), "-o", str(tmp_path / "o.bmp")], cwd=os.path.join(ROOT, "scenes"))
    assert p.returncode == 0 and "parameter has invalid value" in p.stderr
    assert not os.path.exists(str(tmp_path / "o.bmp"))
