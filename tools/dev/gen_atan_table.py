"""the double-double table of atan(k / 32), k = 0..64, and pi/2 for ptam_cg_amd/csrc/atan_cr.h (mpmath, 200 bits): prints the
arrays as hexadecimal floating-point literals"""
import mpmath as mp
mp.mp.prec = 200
for name, f in (("A_HI", lambda a: float(a)), ("A_LO", lambda a: float(a - mp.mpf(float(a))))):
    vals = [f(mp.atan(mp.mpf(k) / 32)) for k in range(65)]
    print("static const double %s[65] = {" % name)
    for i in range(0, 65, 4):
        print("    " + ", ".join(float.hex(v) for v in vals[i:i + 4]) + ",")
    print("};")
h = float(mp.pi / 2)
print("PI2_HI = %s, PI2_LO = %s" % (float.hex(h), float.hex(float(mp.pi / 2 - mp.mpf(h)))))
