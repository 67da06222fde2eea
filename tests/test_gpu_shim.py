"""-m gpu: builds examples/shim_demo.cc (the reference-shaped C++ class surface of ptam_shim.hpp over
the C ABI) with g++, runs it, and checks what it prints against the CPU oracle on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

from ptam_cg_amd import host
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg_stream(seed):
    s = seed
    while True:
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        yield s


def demo_image():
    im = np.full((120, 160), 128, np.uint8)
    g = lcg_stream(12345)
    for _ in range(40):
        x0, y0 = next(g) % 150, next(g) % 110
        w, h, v = 8 + next(g) % 40, 8 + next(g) % 40, 30 + next(g) % 190
        im[y0:min(y0 + h, 120), x0:min(x0 + w, 160)] = v
    return im


def test_shim_demo_matches_oracle(oracle, tmp_path):
    exe = str(tmp_path / "shim_demo")
    lib_dir = os.path.join(ROOT, "ptam_cg_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "shim_demo.cc"), "-L" + lib_dir, "-lptam_hip",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe], text=True, timeout=120)
    lines = [l.split() for l in out.splitlines()]
    im = demo_image()
    lv = util.keyframe_levels(oracle, im)
    for l in range(4):
        row = next(x for x in lines if x[0] == "LEVEL" and int(x[1]) == l)
        c = lv[l]["corners"]
        assert [int(v) for v in row[2:]] == [lv[l]["im"].shape[1], lv[l]["im"].shape[0], len(c), int(c[:, 0].sum()),
                                              int(c[:, 1].sum()), int(lv[l]["rowlut"].sum())]
    assert int(next(x for x in lines if x[0] == "CLONE")[1]) == len(lv[0]["corners"])
    # patch: template taken at the corner itself -> found, ZMSSD 0 at that corner
    p = next(x for x in lines if x[0] == "PATCH")
    assert int(p[3]) == 1 and int(p[6]) == 0
    # warped template: identity warp = the raw window (ZMSSD 0 at the corner), reused for the nearby warp, and the
    # rotated warp gives exactly what the oracle's template scores at that corner
    wline = next(x for x in lines if x[0] == "WARP")
    cx, cy, z0, z1, z2, bad = [int(v) for v in wline[1:]]
    assert z0 == 0 and z1 == 0 and bad == 0
    octx = host.Context(lib=oracle, size=(im.shape[1], im.shape[0]))
    okf = host.KeyFrame(octx).MakeKeyFrame_Lite(im)
    opf = host.PatchFinder(octx)
    tm, tr = opf.MakeTemplateCoarseCont(okf, [0], [[cx, cy]], [0], [[0.8, -0.6, 0.6, 0.8]])
    assert tr["bad"][0] == 0 and z2 == int(opf.ZMSSDAtPoint(okf, 0, [[cx, cy]], tm[0])[0]) and z2 > 0
    # the resident frame tracker: same measurement list, same pose (bit for bit) and outlier count as the host-vector path
    r = [int(v) for v in next(x for x in lines if x[0] == "RESIDENT")[1:]]
    assert r[0] > 20 and r[1] == r[2] and r[1] > 10 and r[3] == 1 and r[4] == r[5]
    # resident TrackMap + batched ReFind_Common over the toy map: every point is in the PVS at level 0, nearly all are found
    # again in their own source image, and the iteration set agrees with the result block
    t = [int(v) for v in next(x for x in lines if x[0] == "TRACKMAP")[1:]]
    assert t[0] >= 5 and t[1] == t[0] and t[2] == t[0] and t[3] == t[4] and t[3] >= 0.8 * t[0] and t[5] >= 0.8 * t[0]
    # two trackers in two contexts, one chain of launches: both frames come back exactly as the single call's
    bt = [int(v) for v in next(x for x in lines if x[0] == "BATCH")[1:]]
    assert bt[0] >= 0.8 * t[0] and bt[1] == bt[0] and bt[2] == bt[0] and bt[3] == 1
    # the motion-model frame call and the map update: a camera that stands still keeps (nearly) zero velocity and every template;
    # UpdateMap with all points persisting keeps every finder, SetMap starts them afresh
    mo = next(x for x in lines if x[0] == "MOTION")
    n_meas, reused_f, reused_u, reused_s, did_coarse = [int(v) for v in mo[1:6]]
    assert n_meas >= 0.8 * t[0] and reused_f == reused_u and reused_f >= t[0] - 1 and reused_s == 0 and did_coarse == 0
    assert float(mo[6]) < 1e-3 and float(mo[7]) < 1e-2
    # bundle: replicate the toy problem through the oracle
    ctx = host.Context(lib=oracle)
    ba = host.Bundle(ctx)
    for j in range(3):
        T = np.concatenate([np.eye(3).ravel(), [-0.2 * j, 0, 0]])
        ba.AddCamera(T, j == 0)
    for i in range(12):
        ba.AddPoint([-0.3 + 0.2 * (i % 4), -0.2 + 0.2 * (i // 4), 2.0 + 0.05 * (i % 3)])
    g = lcg_stream(777)
    for j in range(3):
        for i in range(12):
            X, Y, Z = -0.3 + 0.2 * (i % 4) + 0.2 * j, -0.2 + 0.2 * (i // 4), 2.0 + 0.05 * (i % 3)
            u = 332.3 + 691.4 * X / Z + (next(g) % 1000) / 1000.0 - 0.5
            v = 262.9 + 691.1 * Y / Z + (next(g) % 1000) / 1000.0 - 0.5
            ba.AddMeas(j, i, [u, v], 1.0)
    acc = ba.Compute()
    b = next(x for x in lines if x[0] == "BUNDLE")
    assert [int(b[1]), int(b[2]), int(b[3])] == [acc, int(ba.Converged()), len(ba.GetOutlierMeasurements())]
    for j in range(3):
        row = next(x for x in lines if x[0] == "CAM" and int(x[1]) == j)
        assert np.allclose([float(v) for v in row[2:]], ba.GetCamera(j)[9:], atol=1e-8)
    assert np.allclose([float(v) for v in next(x for x in lines if x[0] == "PT0")[1:]], ba.GetPoint(0), atol=1e-8)
