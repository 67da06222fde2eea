import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
hip, oracle = load(), load_oracle()
a, b = synth.make_frame_pair()
for (w, h) in ((1280, 960), (1920, 1080), (2048, 1536)):
    rng = np.random.default_rng(w)
    im = np.tile(a, (h // 480 + 1, w // 640 + 1))[:h, :w].copy()
    im[::7, ::5] = rng.integers(0, 256, im[::7, ::5].shape)
    out = []
    for lib in (hip, oracle):
        ctx = host.Context(lib=lib, size=(w, h)); kf = host.KeyFrame(ctx).MakeKeyFrame_Lite(im)
        out.append([kf.level(l) for l in range(4)])
    ok = all(np.array_equal(x["im"], y["im"]) and np.array_equal(x["corners"], y["corners"]) and np.array_equal(x["rowlut"], y["rowlut"]) for x, y in zip(*out))
    print("KF", w, h, "OK" if ok else "MISMATCH", [len(x["corners"]) for x in out[0]])
