"""Cycle counts of the pose kernel's phases (timing build: PTAM_HIP_LIB=tools/_timing/libptam_hip.so), summed over 10 iterations."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch  # noqa
from ptam_cg_amd import host, synth
from ptam_cg_amd._lib import load
hip = load(); ctx = host.Context(lib=hip)
pc = synth.make_pose_case(n=int(sys.argv[1]) if len(sys.argv) > 1 else 1000); n = len(pc["world"])
meas = np.zeros(n, dtype=host.POSE_MEAS_DT); meas["world"], meas["found"], meas["sqrt_inv_noise"] = pc["world"], pc["found"], pc["sqrt_inv_noise"]
d_m, d_p, d_u = host.DevBuf(ctx, meas), host.DevBuf(ctx, pc["init_pose"].copy()), host.DevBuf(ctx, 32 * 6 * 8)
opts = ctx.gn_opts(nonlinear_mask=0x3ff, override_sigma_sq=1.0, mark_outliers_iter=-1) if len(sys.argv) > 2 else ctx.gn_opts()
for _ in range(3):
    d_p.upload(pc["init_pose"].copy())
    ctx._check(hip.pose_gn_dev(ctx.h, n, d_m.p, None, d_p.p, C.byref(opts), None, d_u.p), "pose")
    ctx.sync()
u = d_u.download(np.float64, 32 * 6)
names = ["errors+barrier", "select", "accumulate", "barrier", "reduce+barrier", "solve+barrier"]
print({k: int(v) for k, v in zip(names, u[120:126])}, "total", int(u[120:126].sum()))
print("select sub-phases (cycles over the run): histogram + barrier %d | bin scan %d | candidates + rank %d" % tuple(int(v) for v in u[126:129]))
