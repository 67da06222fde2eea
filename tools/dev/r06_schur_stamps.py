import os, sys, re
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
ctx = host.Context(lib=load())
for cams, pts, win in ((200, 50000, 16), (50, 5000, None)):
    prob = synth.make_ba_problem(cams, pts, 11, window=win)
    ba = synth.load_into(host.Bundle(ctx, max_iterations=4, update_sq_conv_limit=0.0), prob)
    ba.Compute(); ba.close()
