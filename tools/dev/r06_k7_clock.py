import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
ctx = host.Context(lib=load())
prob = synth.make_ba_problem(50, 5000, 11)
for rep in range(3):
    ba = synth.load_into(host.Bundle(ctx, max_iterations=12, update_sq_conv_limit=0.0), prob)
    ba.Compute(); ba.close()
ba = synth.load_into(host.Bundle(ctx), prob)
for reps in (50, 2000, 2000, 2000):
    ba.bench_jacobian(reps)
ba.close()
