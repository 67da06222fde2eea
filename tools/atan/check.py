"""atan_cr on the device against mpmath (correct rounding) and against the host libm; OCML's atan beside it"""
import numpy as np, mpmath as mp, subprocess, os, sys
here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(3)
x = np.concatenate([rng.uniform(0, 2.0, 200000), rng.uniform(2.0, 50.0, 20000), -rng.uniform(0, 3.0, 20000), 10.0 ** rng.uniform(-12, 4, 20000),
                    np.arange(0, 65) / 32.0, [0.0, 1e-300, 1e300, 1e305, np.inf, -np.inf, 2.0, 2.0000000000000004, 0.015625, 0.984375]])
x.tofile("/tmp/atan_in.bin")
subprocess.check_call([os.path.join(here, "atan_cr_check"), "/tmp/atan_in.bin", "/tmp/atan_out.bin"])
o = np.fromfile("/tmp/atan_out.bin")
cr, ocml = o[:x.size], o[x.size:]
mp.mp.prec = 160
sub = rng.choice(x.size, 40000, replace=False)
exact = np.array([float(mp.atan(mp.mpf(float(v)))) for v in x[sub]])      # mpf -> float rounds to nearest
libm = np.arctan(x)
def ulps(a, b): return np.abs(a.view(np.int64) - b.view(np.int64))
print("atan_cr vs correctly rounded (mpmath, %d samples): %.4f %% differ, max %d ulp" % (sub.size, 100.0 * (ulps(cr[sub], exact) > 0).mean(), int(ulps(cr[sub], exact).max())))
print("host libm vs correctly rounded: %.4f %% differ" % (100.0 * (ulps(libm[sub], exact) > 0).mean()))
print("OCML atan vs correctly rounded: %.4f %% differ" % (100.0 * (ulps(ocml[sub], exact) > 0).mean()))
print("atan_cr vs host libm (all %d): %.4f %% differ, max %d ulp" % (x.size, 100.0 * (ulps(cr, libm) > 0).mean(), int(ulps(cr, libm).max())))
