"""how often does the device's fp64 atan (OCML, through torch) differ from the host libm's, and by how many ulp?  (The PVS warp
matrix goes through ATANCamera::Project's atan; one ulp can flip a grey level of a warped template: DESIGN section 2.)"""
import numpy as np, torch
rng = np.random.default_rng(1)
x = rng.uniform(0.0, 1.2, 2_000_000)          # r * 2 tan(w/2) for image radii of a 640 x 480 camera
h = np.arctan(x)
d = torch.atan(torch.from_numpy(x).cuda()).cpu().numpy()
ulp = np.abs(d.view(np.int64) - h.view(np.int64))
print("device atan != libm atan: %.3f %% of %d arguments; max %d ulp" % (100.0 * (ulp > 0).mean(), x.size, int(ulp.max())))
