"""The persistent camera solve under contention: bundle adjustments (headline shape and a banded one: one chain, two chains)
run in a loop while other host threads keep the device busy with batched tracking on their own queues.  The solve's workgroups
wait for each other through flags; they must all get dispatched (a spin that gives up would surface as PTAM_E_HIP) and the
results must be those of the quiet runs.   usage: python tests/tools/stress_chain_contention.py [seconds]"""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
from tests import util

hip = load()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
probs = [synth.make_ba_problem(n_cams=50, n_pts=2000, seed=7), synth.make_ba_problem(n_cams=140, n_pts=3000, seed=8, window=12)]
quiet = [util.run_ba(hip, p, max_iterations=6, deterministic=1) for p in probs]
stop = False


def tracker_load():
    ctx = host.Context(lib=hip)
    a, b = synth.make_frame_pair()
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
    tr = host.Tracker(ctx, len(case["world"]))
    tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
    kfb = host.KeyFrame(ctx)
    d_im = host.DevBuf(ctx, b)
    opts = tr.opts()
    n = 0
    while not stop:
        tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
        tr.TrackFrame(kfb, d_im, case["pose_in"], opts)
        n += 1
    counts.append(n)


counts = []
threads = [threading.Thread(target=tracker_load) for _ in range(6)]
for t in threads:
    t.start()
t0 = time.time()
runs = bad = 0
while time.time() - t0 < secs:
    for p, q in zip(probs, quiet):
        r = util.run_ba(hip, p, max_iterations=6, deterministic=1)
        runs += 1
        same = len(r["trials"]) == len(q["trials"]) and np.array_equal(r["poses"], q["poses"]) and np.array_equal(r["points"], q["points"])
        bad += 0 if same else 1
stop = True
for t in threads:
    t.join()
print(f"{runs} adjustments beside {sum(counts)} tracked frames of 6 other contexts in {secs:.0f} s: {bad} differ from the quiet runs (deterministic mode: must be 0)")
sys.exit(1 if bad else 0)
