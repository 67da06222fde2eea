#!/usr/bin/env python3
"""Runs only the K7 (jac_accum_wave_kernel) leg of the bench — the command the rocprofv3 --pmc passes wrap
(tools/profile_k7.sh) and the A/B harness of K7 experiments (PTAM_HIP_LIB selects the build).
  k7_only.py [cams pts reps window [cold_copies]]   cold_copies > 0: also the Infinity-Cache-cold rotation"""
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
cold = int(sys.argv[5]) if len(sys.argv) > 5 else 0
hip = load()
ctx = host.Context(lib=hip)
seed = synth.SEED_BA_GLOBAL if (cams, pts, window) == (200, 50000, 16) else synth.SEED_BA_HEADLINE
prob = synth.make_ba_problem(cams, pts, seed, window=window or None)
ba = synth.load_into(host.Bundle(ctx), prob)
ba.bench_jacobian(max(reps, 500))
ms, by = ba.bench_jacobian(reps)
tag = os.environ.get("PTAM_HIP_LIB", "product").split("/")[-2] if os.environ.get("PTAM_HIP_LIB") else "product"
line = f"K7[{tag}] {cams}x{pts}w{window}: M={len(prob['cam_idx'])} warm {ms*1e3:.2f} us {by/ms/1e6:.0f} GB/s ({by/ms/8e7:.1f} %)"
if cold > 0:
    rot = [ba] + [synth.load_into(host.Bundle(ctx), prob) for _ in range(cold - 1)]
    host.Bundle.bench_jacobian_rotating(rot, 10)
    cms = host.Bundle.bench_jacobian_rotating(rot, max(3, reps // cold))
    line += f" | cold({cold}) {cms*1e3:.2f} us {by/cms/1e6:.0f} GB/s ({by/cms/8e7:.1f} %)"
print(line)
