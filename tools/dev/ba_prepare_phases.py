import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
prob = synth.make_ba_problem(50, 5000, 11)
for rep in range(3):
    ba = synth.load_into(host.Bundle(ctx, max_iterations=10, update_sq_conv_limit=0.0), prob)
    if rep == 2: os.environ["PTAM_DEBUG_PREPARE"] = "1"
    ba.prepare(); ctx.sync(); ba.close()
