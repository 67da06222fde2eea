"""Where do the 10-28 ms of a Compute()'s first trial go?  Build a bundle (prepare = sort + uploads), wait for the stream,
then Compute() — the wait absorbs whatever the uploads cost, PTAM_DEBUG_STALL=1 shows what is left for the first trial."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
prob = synth.make_ba_problem(50, 5000, synth.SEED_BA_HEADLINE)
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for i in range(8):
    ba = synth.load_into(host.Bundle(ctx, max_iterations=5, update_sq_conv_limit=0.0), prob)
    t0 = time.perf_counter(); ba.prepare(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    time.sleep(idle)
    t3 = time.perf_counter(); ba.Compute(); t4 = time.perf_counter()
    print(f"prepare {1e3*(t1-t0):7.2f} ms  sync {1e3*(t2-t1):7.2f} ms  idle {idle*1e3:.0f} ms  Compute(5 trials) {1e3*(t4-t3):7.2f} ms")
    ba.close()
