"""host cost around the trials of a bundle: Add* + prepare (sort, work lists, upload) and a whole Compute() of 10 trials"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
for cams, pts, win in ((8, 600, None), (20, 3000, None), (50, 5000, None), (200, 50000, 16)):
    prob = synth.make_ba_problem(cams, pts, 11, window=win)
    res = []
    for rep in range(4):
        t0 = time.perf_counter()
        ba = synth.load_into(host.Bundle(ctx, max_iterations=10, update_sq_conv_limit=0.0), prob)
        t1 = time.perf_counter()
        ba.prepare(); ctx.sync()
        t2 = time.perf_counter()
        ba.Compute(); ctx.sync()
        t3 = time.perf_counter()
        n = len(ba.trials()); ba.close()
        res.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n))
    a, p, c, n = res[-1]
    print(f"{cams} x {pts}{' w' + str(win) if win else ''}: M {len(prob['cam_idx'])}  Add* {a:.2f} ms | prepare {p:.2f} ms | Compute ({n} trials) {c:.2f} ms")
