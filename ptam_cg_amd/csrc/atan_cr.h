// atan_cr.h — fp64 arc tangent, correctly rounded for all practical purposes (error < 2^-66 relative before the final rounding:
// one result in ~10^4 may be the neighbour of the correctly rounded one), for ATANCamera::Project on the device.
// Why: OCML's atan is within an ulp and differs from the host libm's in 11 % of its results (tools/dev/atan_ulp.py); the image
// position feeds ir() and the projection derivatives the warp matrix, whose last bit decides grey levels of a warped template
// (CVD::transform truncates, src/PatchFinder.cc:116).  Method: x = c + d with c = k / 32 from a table of
// atan(c) in double-double (generated with mpmath at 200 bits: tools/dev/gen_atan_table.py), atan(x) = atan(c) + atan(z),
// z = (x - c) / (1 + x c) in double-double (|z| <= 2^-6), atan(z) = z + z^3 P(z^2) with P to z^13 in plain fp64 (its error is
// below 2^-66 of the result); x > 2 through pi/2 - atan(1/x).  No contraction: the error-free transformations need exactly
// the products and sums written.
// Used by cam_project (PVS pass, search stage: bit-class results).  NOT by the pose loops' re-projection (pose_device.h:
// small_project — reciprocals and a 2-ulp atan, tolerance-class; INTEGRATION.md says which is which).
#pragma once
__device__ __forceinline__ double atan_cr(double x_in) {
#pragma clang fp contract(off)
    static const double A_HI[65] = {
        0x0.0p+0, 0x1.ffd55bba97625p-6, 0x1.ff55bb72cfdeap-5, 0x1.7ee182602f10fp-4,
        0x1.fd5ba9aac2f6ep-4, 0x1.3d6eee8c6626cp-3, 0x1.7b97b4bce5b02p-3, 0x1.b90d7529260a2p-3,
        0x1.f5b75f92c80ddp-3, 0x1.18bf5a30bf178p-2, 0x1.362773707ebccp-2, 0x1.530ad9951cd4ap-2,
        0x1.6f61941e4def1p-2, 0x1.8b24d394a1b25p-2, 0x1.a64eec3cc23fdp-2, 0x1.c0db4c94ec9f0p-2,
        0x1.dac670561bb4fp-2, 0x1.f40dd0b541418p-2, 0x1.0657e94db30d0p-1, 0x1.1255d9bfbd2a9p-1,
        0x1.1e00babdefeb4p-1, 0x1.2958e59308e31p-1, 0x1.345f01cce37bbp-1, 0x1.3f13fb89e96f4p-1,
        0x1.4978fa3269ee1p-1, 0x1.538f57b89061fp-1, 0x1.5d58987169b18p-1, 0x1.66d663923e087p-1,
        0x1.700a7c5784634p-1, 0x1.78f6bbd5d315ep-1, 0x1.819d0b7158a4dp-1, 0x1.89ff5ff57f1f8p-1,
        0x1.921fb54442d18p-1, 0x1.9a000a935bd8ep-1, 0x1.a1a25f2c82506p-1, 0x1.a908afa5b1d4ap-1,
        0x1.b034f38649c88p-1, 0x1.b7291b4e25bdap-1, 0x1.bde70ed439fe7p-1, 0x1.c470abf2d3d01p-1,
        0x1.cac7c57846f9ep-1, 0x1.d0ee2253886a6p-1, 0x1.d6e57cf4f0acap-1, 0x1.dcaf82dc1a6f4p-1,
        0x1.e24dd44c855d1p-1, 0x1.e7c2042350f87p-1, 0x1.ed0d97c9041c9p-1, 0x1.f232073aeb172p-1,
        0x1.f730bd281f69bp-1, 0x1.fc0b171ec926cp-1, 0x1.006132e34d617p+0, 0x1.02abf692f6d0cp+0,
        0x1.04e67277a01d7p+0, 0x1.07113c6a93a21p+0, 0x1.092ce471853ccp+0, 0x1.0b39f4eca23aep+0,
        0x1.0d38f2c5ba09fp+0, 0x1.0f2a5d9fff026p+0, 0x1.110eb007f39f7p+0, 0x1.12e65fa32aaedp+0,
        0x1.14b1dd5f90ce1p+0, 0x1.167195a203265p+0, 0x1.1825f074030d9p+0, 0x1.19cf51b0603ddp+0,
        0x1.1b6e192ebbe44p+0,
    };
    static const double A_LO[65] = {
        0x0.0p+0, -0x1.5ec431444912cp-60, -0x1.c934d86d23f1dp-60, -0x1.cfb654c0c3d98p-58,
        -0x1.cd37686760c17p-59, 0x1.61a3b0ce9281bp-57, 0x1.347b0b4f881cap-58, 0x1.17b10d2e0e5abp-61,
        0x1.8ab6e3cf7afbdp-57, 0x1.30ca4748b1bf9p-57, -0x1.963a544b672d8p-57, -0x1.2566480884082p-57,
        -0x1.c63aae6f6e918p-56, 0x1.b6d0ba3748fa8p-56, -0x1.24dec1b50b7ffp-56, -0x1.cc1ce70934c34p-56,
        0x1.a2b7f222f65e2p-56, -0x1.a3992dc382a23p-57, -0x1.d5b495f6349e6p-56, -0x1.2bdaee1c0ee35p-58,
        -0x1.928df287a668fp-58, -0x1.09e73b0c6c087p-56, 0x1.1021137c71102p-55, 0x1.ecf8b492644f0p-56,
        0x1.2419a87f2a458p-56, -0x1.1bb74abda520cp-55, 0x1.0028e4bc5e7cap-57, -0x1.6ea6febe8bbbap-56,
        -0x1.8c34d25aadef6p-56, 0x1.406a089803740p-55, -0x1.bf76229d3b917p-56, -0x1.55b9a5e177a1bp-55,
        0x1.1a62633145c07p-55, 0x1.59411df0dccefp-56, -0x1.8b4c3611182fcp-57, -0x1.5d7be5d5f808bp-56,
        -0x1.be88d6936f833p-55, -0x1.c49cc26e63660p-56, -0x1.a2b56372c05efp-56, 0x1.6a61dbf199479p-56,
        0x1.0dae13ad18a6bp-55, 0x1.2c9f73793ddedp-55, -0x1.763b9456ae66ep-55, -0x1.f99cb3ddd4790p-55,
        0x1.f7ac612ab33d8p-55, -0x1.0e14d8d5a7dd8p-57, -0x1.2629e3b5da490p-58, -0x1.5f5b3a2cdfc2cp-55,
        0x1.007887af0cbbdp-56, -0x1.3337369af334fp-58, 0x1.b343dfa868d93p-54, -0x1.7e03a29351e05p-54,
        0x1.7115496c13eb6p-57, 0x1.c2bc4d3a3e69fp-56, 0x1.269f9b3e200c2p-55, 0x1.25934545c016cp-54,
        -0x1.bd0dc231bfd70p-54, 0x1.e6ac2e9161719p-55, -0x1.12b2ff85e5500p-54, -0x1.f25b08b14d8d6p-54,
        -0x1.212d570a63fa2p-56, 0x1.1a5aca105c6aep-54, -0x1.9523f0af0d3b5p-58, -0x1.4b79cf12e503dp-55,
        0x1.b1b466a88828ep-54,
    };
    const double PI2_HI = 0x1.921fb54442d18p+0, PI2_LO = 0x1.1a62633145c07p-54;
    const double ax = x_in < 0 ? -x_in : x_in;
    if (!(ax == ax)) return x_in;
    if (ax > 0x1p1000) return x_in < 0 ? -PI2_HI : PI2_HI;   // (1 / x underflows its correction term; the result is pi/2 rounded from 2^54 on anyway)
    const bool recip = ax > 2.0;
    // t = ax, or 1 / ax in double-double
    double th = ax, tl = 0.0;
    if (recip) {
        th = 1.0 / ax;
        tl = __builtin_fma(-th, ax, 1.0) / ax;
    }
    const int k = (int)__builtin_rint(th * 32.0);
    const double c = (double)k * 0.03125;
    // numerator t - c (exact in its high part: c is a multiple of 2^-5 next to t)
    const double nh = th - c, nl = tl;
    // denominator 1 + t c in double-double
    const double p = th * c, pe = __builtin_fma(th, c, -p) + tl * c;
    const double dh = 1.0 + p, dl = ((1.0 - dh) + p) + pe;
    // z = n / d
    const double zh = nh / dh;
    const double r = (__builtin_fma(-zh, dh, nh) - zh * dl) + nl;
    const double zl = r / dh;
    const double s = zh * zh;
    const double P = s * (-1.0 / 3 + s * (1.0 / 5 + s * (-1.0 / 7 + s * (1.0 / 9 + s * (-1.0 / 11 + s * (1.0 / 13))))));
    const double corr = zh * P;
    // atan(c) + z: two_sum of the high parts, everything small behind it
    const double a = A_HI[k], sh = a + zh, bb = sh - a;
    double sl = ((a - (sh - bb)) + (zh - bb)) + ((A_LO[k] + zl) + corr);
    double res;
    if (recip) {
        // pi/2 - (sh + sl)
        const double qh = PI2_HI - sh, qb = qh - PI2_HI;
        const double ql = ((PI2_HI - (qh - qb)) + (-sh - qb)) + (PI2_LO - sl);
        res = qh + ql;
    } else
        res = sh + sl;
    return __builtin_copysign(res, x_in);   // (the sign of the argument, of a zero too: atan(-0.0) = -0.0 as IEEE and libm have it)
}
