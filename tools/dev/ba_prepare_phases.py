import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
import sys as _s
args = [int(a) for a in _s.argv[1:]]
prob = synth.make_ba_problem(args[0], args[1], 11, window=(args[2] if len(args) > 2 else None)) if args else synth.make_ba_problem(50, 5000, 11)
for rep in range(3):
    ba = synth.load_into(host.Bundle(ctx, max_iterations=10, update_sq_conv_limit=0.0), prob)
    if rep == 2: os.environ["PTAM_DEBUG_PREPARE"] = "1"
    ba.prepare(); ctx.sync(); ba.close()
