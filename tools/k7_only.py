#!/usr/bin/env python3
"""Runs only the K7 (jac_accum_kernel) leg of the bench on the headline problem — the command the
rocprofv3 --pmc passes wrap (tools/profile_k7.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (shared HIP runtime, see ptam_cg_amd/_lib.py)
from ptam_cg_amd import host, synth  # noqa: E402
from ptam_cg_amd._lib import load  # noqa: E402

cams = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pts = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
window = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hip = load()
ctx = host.Context(lib=hip)
prob = synth.make_ba_problem(cams, pts, synth.SEED_BA_HEADLINE, window=window or None)
ba = synth.load_into(host.Bundle(ctx), prob)
ms, by = ba.bench_jacobian(reps)
print(f"K7 {cams}x{pts}: M={len(prob['cam_idx'])} avg {ms*1e3:.2f} us  {by/ms/1e6:.1f} GB/s algorithmic ({by/1e6:.2f} MB)")
