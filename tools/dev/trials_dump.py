"""Prints the trial records of the bench problem (lambda, errors, accepted) — to read next to a PTAM_TIMELINE dump."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load()
ctx = host.Context(lib=hip)
prob = synth.make_ba_problem(50, 5000, synth.SEED_BA_HEADLINE)
ba = synth.load_into(host.Bundle(ctx, max_iterations=20, update_sq_conv_limit=0.0), prob)
ba.prepare()
ba.Compute()
for i, t in enumerate(ba.trials()):
    print(i, "lam %.3e sig2 %.6f old %.9f new %.9f diff %.3e nbad %d acc %d" % (t["lambda"], t["sigma_sq"], t["err_old"], t["err_new"], t["err_old"] - t["err_new"], t["n_bad"], t["accepted"]))
