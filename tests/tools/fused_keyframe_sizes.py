"""the tracked frame's fused keyframe launch (pyramid pixels inside the FAST tiles) against MakeKeyFrame_Lite over image sizes:
tiny, odd, word-path (width a multiple of 32) and large ones, both halfSample roundings"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, _abi
from ptam_cg_amd._lib import load
hip = load()
rng = np.random.default_rng(5)
bad = 0
for (w, h) in [(8, 8), (9, 11), (16, 16), (31, 17), (32, 32), (40, 30), (64, 48), (72, 56), (96, 33), (128, 9), (160, 120), (352, 288), (640, 481), (641, 480), (1280, 720), (1920, 1080), (2048, 1536)]:
    for variant in (_abi.HALFSAMPLE_R, _abi.HALFSAMPLE_T):
        # smooth blobs + noise: corners on every level
        yy, xx = np.mgrid[0:h, 0:w]
        im = (127 + 90 * np.sin(xx / 3.1) * np.cos(yy / 2.3) + rng.integers(-30, 30, (h, w))).clip(0, 255).astype(np.uint8)
        cx = host.Context(lib=hip, size=(w, h), halfsample=variant)
        tr = host.Tracker(cx, 8)
        ka = host.KeyFrame(cx).MakeKeyFrame_Lite(im)
        tr.set_map(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), ka, np.zeros(0, np.int32), np.zeros((0, 2), np.int32))
        kb = host.KeyFrame(cx)
        pose = np.concatenate([np.eye(3).reshape(9), [0.0, 0.0, 1.5]])
        tr.TrackFrame(kb, host.DevBuf(cx, im), pose)
        ok = all(np.array_equal(kb.level(l)[k], ka.level(l)[k]) for l in range(4) for k in ("im", "corners", "rowlut"))
        bad += not ok
        print((w, h), "R" if variant == _abi.HALFSAMPLE_R else "T", "OK" if ok else "MISMATCH", [len(kb.level(l)["corners"]) for l in range(4)])
        tr.close()
print("mismatches:", bad)
