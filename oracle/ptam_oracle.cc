// ptam_oracle.cc — CPU restatement of the reference's (cggos/ptam_cg) tracking + bundle hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under ptam_cg_amd/ or include/ may link, import or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
// and as the timed CPU baseline ("port"), never as the product path.
//
// PARITY UNPINNED: the reference tree holds no tests, golden vectors or fixtures for this path
// (SURVEY.md §4), and it cannot be built here: every hot-path file includes TooN-2.2 / libcvd-20150407
// / gvars-3.0 headers (src/KeyFrame.cc:4-5, src/Bundle.cc:4-8, ...) which are fetched by
// install_deps.sh:36-60, are not vendored and are absent from this image (no network).  The oracle is
// therefore anchored on (i) the cited reference lines it follows one for one, (ii) the published
// semantics of the third-party calls restated below, (iii) an independent numpy restatement
// (oracle/np_oracle.py) it is cross-checked against, (iv) hand-derivable known answers (tests/).
//
// Third-party semantics restated (cannot be re-verified in this container):
//   libCVD halfSample: two upstream variants — T: (a+b+c+d)/4 truncating (generic template);
//     R: SSE2 byte specialisation = vertical pavgb then horizontal pavgw:
//     v1=(a+c+1)>>1, v2=(b+d+1)>>1, out=(v1+v2+1)>>1  (a,b top row; c,d bottom row).  Default R.
//   libCVD fast_corner_detect_10: pixel p is a corner iff >= 10 contiguous pixels of the 16-pixel
//     radius-3 ring are all > p+b or all < p-b (strict); y in [3,h-3), x in [3,w-3); raster order.
//   TooN SE3::exp, SE3*SE3, generator_field, Cholesky (unpivoted LDL^T, lower triangle), WLS
//     (normal equations; in TooN 2.x WLS<Size, Precision, Decomposition = Cholesky> solves them with
//     that same Cholesky class — an LDL^T, the solve used below; SQSVD is the optional alternative,
//     which src/Tracker.cc does not ask for).
//
// Each function cites the reference file:line it follows.  Loop and data-structure order mirror the
// reference (std::list of measurements, dense camera x point LUT, off-diagonal scripts, std::sort
// median) so that this file is also an honest single-thread CPU baseline.

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <list>
#include <set>
#include <utility>
#include <vector>

#include "../include/ptam_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// mini math (TooN stand-in; semantics of SURVEY.md §8c)
// ------------------------------------------------------------------------------------------------
struct SE3 {
    double R[9];   // row-major
    double t[3];
};

SE3 se3_from12(const double* p) {
    SE3 s;
    std::memcpy(s.R, p, 9 * sizeof(double));
    std::memcpy(s.t, p + 9, 3 * sizeof(double));
    return s;
}
void se3_to12(const SE3& s, double* p) {
    std::memcpy(p, s.R, 9 * sizeof(double));
    std::memcpy(p + 9, s.t, 3 * sizeof(double));
}

// TooN SE3<>::exp(mu), mu = (t, w)
SE3 se3_exp(const double mu[6]) {
    static const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double* tr = mu;
    const double* w = mu + 3;
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double theta = std::sqrt(theta_sq);
    double A, B;
    const double cr[3] = {w[1] * tr[2] - w[2] * tr[1], w[2] * tr[0] - w[0] * tr[2],
                          w[0] * tr[1] - w[1] * tr[0]};   // w x t
    SE3 out;
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
        for (int i = 0; i < 3; i++) out.t[i] = tr[i] + 0.5 * cr[i];
    } else {
        double C;
        if (theta_sq < 1e-6) {
            C = one_6th * (1.0 - one_20th * theta_sq);
            A = 1.0 - theta_sq * C;
            B = 0.5 - 0.25 * one_6th * theta_sq;
        } else {
            const double inv_theta = 1.0 / theta;
            A = std::sin(theta) * inv_theta;
            B = (1 - std::cos(theta)) * (inv_theta * inv_theta);
            C = (1 - A) * (inv_theta * inv_theta);
        }
        const double wcr[3] = {w[1] * cr[2] - w[2] * cr[1], w[2] * cr[0] - w[0] * cr[2],
                               w[0] * cr[1] - w[1] * cr[0]};   // w x (w x t)
        for (int i = 0; i < 3; i++) out.t[i] = tr[i] + B * cr[i] + C * wcr[i];
    }
    // Rodrigues
    {
        const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
        out.R[0] = 1.0 - B * (wy2 + wz2);
        out.R[4] = 1.0 - B * (wx2 + wz2);
        out.R[8] = 1.0 - B * (wx2 + wy2);
        double a = A * w[2], b = B * (w[0] * w[1]);
        out.R[1] = b - a;
        out.R[3] = b + a;
        a = A * w[1];
        b = B * (w[0] * w[2]);
        out.R[2] = b + a;
        out.R[6] = b - a;
        a = A * w[0];
        b = B * (w[1] * w[2]);
        out.R[5] = b - a;
        out.R[7] = b + a;
    }
    return out;
}

// SE3 * SE3: R = R1 R2, t = t1 + R1 t2 (no re-orthonormalisation)
SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 o;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a.R[r * 3 + k] * b.R[k * 3 + c];
            o.R[r * 3 + c] = s;
        }
        double s = 0;
        for (int k = 0; k < 3; k++) s += a.R[r * 3 + k] * b.t[k];
        o.t[r] = a.t[r] + s;
    }
    return o;
}

// TooN SO3<>::ln (TooN/so3.h): result = antisymmetric part / 2; cos_angle from the trace; three ranges of the angle
void so3_ln(const double* m, double result[3]) {
    const double cos_angle = (m[0] + m[4] + m[8] - 1.0) * 0.5;
    result[0] = (m[2 * 3 + 1] - m[1 * 3 + 2]) / 2;
    result[1] = (m[0 * 3 + 2] - m[2 * 3 + 0]) / 2;
    result[2] = (m[1 * 3 + 0] - m[0 * 3 + 1]) / 2;
    double sin_angle_abs = std::sqrt(result[0] * result[0] + result[1] * result[1] + result[2] * result[2]);
    if (cos_angle > M_SQRT1_2) {   // [0 - Pi/4[ use asin
        if (sin_angle_abs > 0) {
            const double f = std::asin(sin_angle_abs) / sin_angle_abs;
            for (int i = 0; i < 3; i++) result[i] *= f;
        }
    } else if (cos_angle > -M_SQRT1_2) {   // [Pi/4 - 3Pi/4[ use acos, but antisymmetric part
        const double angle = std::acos(cos_angle);
        const double f = angle / sin_angle_abs;
        for (int i = 0; i < 3; i++) result[i] *= f;
    } else {   // rest use symmetric part
        const double angle = M_PI - std::asin(sin_angle_abs);
        const double d0 = m[0] - cos_angle, d1 = m[4] - cos_angle, d2 = m[8] - cos_angle;
        double r2[3];
        if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) {   // first is largest, fill with first column
            r2[0] = d0;
            r2[1] = (m[1 * 3 + 0] + m[0 * 3 + 1]) / 2;
            r2[2] = (m[0 * 3 + 2] + m[2 * 3 + 0]) / 2;
        } else if (d1 * d1 > d2 * d2) {   // second is largest
            r2[0] = (m[1 * 3 + 0] + m[0 * 3 + 1]) / 2;
            r2[1] = d1;
            r2[2] = (m[2 * 3 + 1] + m[1 * 3 + 2]) / 2;
        } else {   // third is largest
            r2[0] = (m[0 * 3 + 2] + m[2 * 3 + 0]) / 2;
            r2[1] = (m[2 * 3 + 1] + m[1 * 3 + 2]) / 2;
            r2[2] = d2;
        }
        if (r2[0] * result[0] + r2[1] * result[1] + r2[2] * result[2] < 0)   // flip, if we point in the wrong direction
            for (int i = 0; i < 3; i++) r2[i] *= -1;
        const double n = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
        for (int i = 0; i < 3; i++) result[i] = angle * (r2[i] / n);
    }
}

// TooN SE3<>::ln (TooN/se3.h)
void se3_ln(const SE3& se3, double out[6]) {
    double rot[3];
    so3_ln(se3.R, rot);
    const double rr = rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2];
    const double theta = std::sqrt(rr);
    double shtot = 0.5;
    if (theta > 0.00001) shtot = std::sin(theta / 2) / theta;
    // now do the rotation: halfrotator = SO3::exp(rot * -0.5)  (SO3::exp = the rotation part of SE3::exp)
    const double hmu[6] = {0, 0, 0, rot[0] * -0.5, rot[1] * -0.5, rot[2] * -0.5};
    const SE3 half = se3_exp(hmu);
    double rottrans[3];
    for (int r = 0; r < 3; r++) rottrans[r] = half.R[r * 3 + 0] * se3.t[0] + half.R[r * 3 + 1] * se3.t[1] + half.R[r * 3 + 2] * se3.t[2];
    const double tr = se3.t[0] * rot[0] + se3.t[1] * rot[1] + se3.t[2] * rot[2];
    if (theta > 0.001) {
        const double f = tr * (1 - 2 * shtot) / rr;
        for (int i = 0; i < 3; i++) rottrans[i] -= rot[i] * f;
    } else {
        const double f = tr / 24;
        for (int i = 0; i < 3; i++) rottrans[i] -= rot[i] * f;
    }
    for (int i = 0; i < 3; i++) {
        out[i] = rottrans[i] / (2 * shtot);
        out[3 + i] = rot[i];
    }
}

// SE3<>::inverse(): (R^T, -(R^T t))
SE3 se3_inverse(const SE3& s) {
    SE3 o;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) o.R[r * 3 + c] = s.R[c * 3 + r];
        o.t[r] = -(s.R[0 * 3 + r] * s.t[0] + s.R[1 * 3 + r] * s.t[1] + s.R[2 * 3 + r] * s.t[2]);
    }
    return o;
}

void se3_apply(const SE3& s, const double x[3], double out[3]) {
    for (int r = 0; r < 3; r++)
        out[r] = s.t[r] + (s.R[r * 3 + 0] * x[0] + s.R[r * 3 + 1] * x[1] + s.R[r * 3 + 2] * x[2]);
}

// SE3<>::generator_field(m, (x,y,z,w))
void generator_field(int m, const double p[4], double out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (m < 3) {
        out[m] = p[3];
        return;
    }
    switch (m) {
        case 3: out[1] = -p[2]; out[2] = p[1]; break;
        case 4: out[0] = p[2]; out[2] = -p[0]; break;
        default: out[0] = -p[1]; out[1] = p[0]; break;
    }
}

// TooN Cholesky<>: unpivoted LDL^T reading only the lower triangle; the strict upper triangle is
// used as a cache of the undivided column.  backsub = L solve, D scale, L^T solve.
struct LDLT {
    int n;
    std::vector<double> a;   // n*n row-major
    LDLT(int n_, const double* m) : n(n_), a(m, m + (size_t)n_ * n_) { compute(); }
    double& at(int r, int c) { return a[(size_t)r * n + c]; }
    double at(int r, int c) const { return a[(size_t)r * n + c]; }
    void compute() {
        for (int col = 0; col < n; col++) {
            double inv_diag = 1;
            for (int row = col; row < n; row++) {
                double val = at(row, col);
                for (int c2 = 0; c2 < col; c2++) val -= at(c2, col) * at(row, c2);
                if (row == col) {
                    at(row, col) = val;
                    if (val == 0) return;   // rank deficient: TooN stops; backsub then yields inf/nan
                    inv_diag = 1 / val;
                } else {
                    at(col, row) = val;
                    at(row, col) = val * inv_diag;
                }
            }
        }
    }
    void backsub(const double* v, double* x) const {
        std::vector<double> y(n);
        for (int i = 0; i < n; i++) {
            double val = v[i];
            for (int j = 0; j < i; j++) val -= at(i, j) * y[j];
            y[i] = val;
        }
        for (int i = 0; i < n; i++) y[i] /= at(i, i);
        for (int i = n - 1; i >= 0; i--) {
            double val = y[i];
            for (int j = i + 1; j < n; j++) val -= at(j, i) * x[j];
            x[i] = val;
        }
    }
    void inverse(double* out) const {   // get_inverse(): backsub of identity columns
        std::vector<double> e(n), x(n);
        for (int c = 0; c < n; c++) {
            std::fill(e.begin(), e.end(), 0.0);
            e[c] = 1.0;
            backsub(e.data(), x.data());
            for (int r = 0; r < n; r++) out[(size_t)r * n + c] = x[r];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// M-estimators: include/Tools.h:128-228
// ------------------------------------------------------------------------------------------------
double median_sigma(std::vector<double>& v, double k) {   // Tools.h:152-162 / 180-190 / 218-228
    assert(!v.empty());
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    // NB size_t arithmetic: (2n - 6) wraps for n < 3, is 0 for n == 3
    double sigma = 1.4826 * (1 + 5.0 / (v.size() * 2 - 6)) * std::sqrt(med);
    sigma = k * sigma;
    return sigma * sigma;
}
struct Tukey {
    static double FindSigmaSquared(std::vector<double>& v) { return median_sigma(v, 4.6851); }
    static double SquareRootWeight(double e2, double s2) { return e2 > s2 ? 0.0 : 1.0 - (e2 / s2); }
    static double Weight(double e2, double s2) {
        const double r = SquareRootWeight(e2, s2);
        return r * r;
    }
    static double ObjectiveScore(double e2, double s2) {
        if (e2 > s2) return 1.0;
        const double d = 1.0 - e2 / s2;
        return 1.0 - d * d * d;
    }
};
struct Cauchy {
    static double FindSigmaSquared(std::vector<double>& v) { return median_sigma(v, 4.6851); }
    static double Weight(double e2, double s2) { return 1.0 / (1.0 + e2 / s2); }
    static double SquareRootWeight(double e2, double s2) { return std::sqrt(Weight(e2, s2)); }
    static double ObjectiveScore(double e2, double s2) { return std::log(1.0 + e2 / s2); }
};
struct Huber {
    static double FindSigmaSquared(std::vector<double>& v) { return median_sigma(v, 1.345); }
    static double Weight(double e2, double s2) { return e2 < s2 ? 1.0 : std::sqrt(s2 / e2); }
    static double SquareRootWeight(double e2, double s2) { return std::sqrt(Weight(e2, s2)); }
    static double ObjectiveScore(double e2, double s2) {
        if (e2 < s2) return 0.5 * e2;
        const double s = std::sqrt(s2), e = std::sqrt(e2);
        return s * (e - 0.5 * s);
    }
};

// ------------------------------------------------------------------------------------------------
// ATANCamera: src/ATANCamera.cc:27-66 (RefreshParams), :109-121 (Project), :179-209 (derivs),
// include/ATANCamera.h:143-157 (rtrans_factor / invrtrans).  Stateful like the reference.
// ------------------------------------------------------------------------------------------------
struct ATANCamera {
    double focal[2], centre[2], w, two_tan, one_over_two_tan, w_inv, dist_enabled;
    double largest_radius, max_r;
    double size[2];
    // cache of the last projection
    double last_cam[2], last_r, last_factor;
    bool invalid;

    explicit ATANCamera(const ptam_cam_params& p) {
        size[0] = p.width;
        size[1] = p.height;
        focal[0] = size[0] * p.fx;
        focal[1] = size[1] * p.fy;
        centre[0] = size[0] * p.cx - 0.5;
        centre[1] = size[1] * p.cy - 0.5;
        w = p.w;
        if (w != 0.0) {
            two_tan = 2.0 * std::tan(w / 2.0);
            one_over_two_tan = 1.0 / two_tan;
            w_inv = 1.0 / w;
            dist_enabled = 1.0;
        } else {
            w_inv = 0.0;
            two_tan = 0.0;
            one_over_two_tan = 0.0;
            dist_enabled = 0.0;
        }
        double v[2];
        v[0] = std::max(p.cx, 1.0 - p.cx) / p.fx;
        v[1] = std::max(p.cy, 1.0 - p.cy) / p.fy;
        largest_radius = invrtrans(std::sqrt(v[0] * v[0] + v[1] * v[1]));
        max_r = 1.5 * largest_radius;
        last_cam[0] = last_cam[1] = last_r = 0;
        last_factor = 1;
        invalid = false;
    }
    double rtrans_factor(double r) const {
        if (r < 0.001 || w == 0.0) return 1.0;
        return w_inv * std::atan(r * two_tan) / r;
    }
    double invrtrans(double r) const {
        if (w == 0.0) return r;
        return std::tan(r * w) * one_over_two_tan;
    }
    // ATANCamera::UnProject src/ATANCamera.cc:125-140 (mvInvFocal = 1 / mvFocal, :38-39)
    void UnProject(const double im[2], double cam[2]) const {
        const double inv_focal[2] = {1.0 / focal[0], 1.0 / focal[1]};
        const double dc[2] = {(im[0] - centre[0]) * inv_focal[0], (im[1] - centre[1]) * inv_focal[1]};
        const double dist_r = std::sqrt(dc[0] * dc[0] + dc[1] * dc[1]);
        const double r = invrtrans(dist_r);
        const double factor = dist_r > 0.01 ? r / dist_r : 1.0;
        cam[0] = factor * dc[0];
        cam[1] = factor * dc[1];
    }
    // mdOnePixelDist src/ATANCamera.cc:69-75
    double OnePixelDist() const {
        const double c0[2] = {size[0] / 2, size[1] / 2}, c1[2] = {size[0] / 2 + 1, size[1] / 2 + 1};
        double a[2], b[2];
        UnProject(c0, a);
        UnProject(c1, b);
        const double d[2] = {a[0] - b[0], a[1] - b[1]};
        return std::sqrt(d[0] * d[0] + d[1] * d[1]) / std::sqrt(2.0);
    }
    void Project(const double cam[2], double im[2]) {
        last_cam[0] = cam[0];
        last_cam[1] = cam[1];
        last_r = std::sqrt(cam[0] * cam[0] + cam[1] * cam[1]);
        invalid = last_r > max_r;
        last_factor = rtrans_factor(last_r);
        im[0] = centre[0] + focal[0] * (last_factor * last_cam[0]);
        im[1] = centre[1] + focal[1] * (last_factor * last_cam[1]);
    }
    void GetProjectionDerivs(double d[4]) const {
        const double k = two_tan, x = last_cam[0], y = last_cam[1];
        const double r = last_r * dist_enabled;
        double fx, fy;
        if (r < 0.01) {
            fx = fy = 0.0;
        } else {
            fx = w_inv * (k * x) / (r * r * (1 + k * k * r * r)) - x * last_factor / (r * r);
            fy = w_inv * (k * y) / (r * r * (1 + k * k * r * r)) - y * last_factor / (r * r);
        }
        d[0] = focal[0] * (fx * x + last_factor);
        d[2] = focal[1] * (fx * y);
        d[1] = focal[0] * (fy * x);
        d[3] = focal[1] * (fy * y + last_factor);
    }
};

// ------------------------------------------------------------------------------------------------
// KeyFrame / Level: include/KeyFrame.h:55-124, src/KeyFrame.cc:18-54
// ------------------------------------------------------------------------------------------------
struct Level {
    int w = 0, h = 0;
    std::vector<uint8_t> im;
    std::vector<ptam_int2> corners;
    std::vector<int> rowlut;
};

// CVD::halfSample; out size = in / 2 (src/KeyFrame.cc:26)
void half_sample(const uint8_t* in, int w, int h, uint8_t* out, int variant) {
    const int ow = w / 2, oh = h / 2;
    for (int y = 0; y < oh; y++) {
        const uint8_t* top = in + (size_t)(2 * y) * w;
        const uint8_t* bot = top + w;
        uint8_t* o = out + (size_t)y * ow;
        for (int x = 0; x < ow; x++) {
            const int a = top[2 * x], b = top[2 * x + 1], c = bot[2 * x], d = bot[2 * x + 1];
            if (variant == PTAM_HALFSAMPLE_T) {
                o[x] = (uint8_t)((a + b + c + d) / 4);
            } else {
                const int v1 = (a + c + 1) >> 1, v2 = (b + d + 1) >> 1;
                o[x] = (uint8_t)((v1 + v2 + 1) >> 1);
            }
        }
    }
}

// CVD::fast_corner_detect_10 (definition in the header comment)
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

inline bool has_run10(unsigned m16) {
    const unsigned d = m16 | (m16 << 16);
    unsigned a = d & (d >> 1);
    unsigned b = a & (a >> 2);
    unsigned c = b & (b >> 4);
    return (c & (a >> 8) & 0xffffu) != 0;
}

void fast10(const uint8_t* im, int w, int h, int thr, std::vector<ptam_int2>& out) {
    int off[16];
    for (int i = 0; i < 16; i++) off[i] = kRing[i][1] * w + kRing[i][0];
    for (int y = 3; y < h - 3; y++) {
        const uint8_t* row = im + (size_t)y * w;
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = row + x;
            const int hi = *p + thr, lo = *p - thr;
            // a 10-arc contains at least one pixel of every opposite pair: cheap rejects
            const int p0 = p[off[0]], p8 = p[off[8]];
            if (!(p0 > hi || p8 > hi || p0 < lo || p8 < lo)) continue;
            const int p4 = p[off[4]], p12 = p[off[12]];
            if (!(p4 > hi || p12 > hi || p4 < lo || p12 < lo)) continue;
            unsigned brighter = 0, darker = 0;
            for (int i = 0; i < 16; i++) {
                const int v = p[off[i]];
                brighter |= (unsigned)(v > hi) << i;
                darker |= (unsigned)(v < lo) << i;
            }
            if (has_run10(brighter) || has_run10(darker)) out.push_back({x, y});
        }
    }
}

// libCVD fast_corner_score_10 (restated): binary search for the largest threshold at which the
// pixel is still a FAST-10 corner, starting from the detection barrier.
bool is_corner10(const uint8_t* im, int w, int x, int y, int thr) {
    const uint8_t* p = im + (size_t)y * w + x;
    const int hi = *p + thr, lo = *p - thr;
    unsigned brighter = 0, darker = 0;
    for (int i = 0; i < 16; i++) {
        const int v = p[kRing[i][1] * w + kRing[i][0]];
        brighter |= (unsigned)(v > hi) << i;
        darker |= (unsigned)(v < lo) << i;
    }
    return has_run10(brighter) || has_run10(darker);
}
int fast_corner_score10(const uint8_t* im, int w, int x, int y, int barrier) {
    int bmin = barrier, bmax = 255, b = (bmax + bmin) / 2;
    for (;;) {
        if (is_corner10(im, w, x, y, b))
            bmin = b;
        else
            bmax = b;
        if (bmin == bmax - 1 || bmin == bmax) return bmin;
        b = (bmin + bmax) / 2;
    }
}
// ImageProcess::ShiTomasiScoreAtPoint src/ImageProcess.cc:20-47
double shi_tomasi(const Level& L, int nHalfBoxSize, int cx, int cy) {
    double dXX = 0, dYY = 0, dXY = 0;
    for (int y = cy - nHalfBoxSize; y <= cy + nHalfBoxSize; y++)
        for (int x = cx - nHalfBoxSize; x <= cx + nHalfBoxSize; x++) {
            const double dx = L.im[(size_t)y * L.w + x + 1] - L.im[(size_t)y * L.w + x - 1];
            const double dy = L.im[(size_t)(y + 1) * L.w + x] - L.im[(size_t)(y - 1) * L.w + x];
            dXX += dx * dx;
            dYY += dy * dy;
            dXY += dx * dy;
        }
    const int nPixels = (2 * nHalfBoxSize + 1) * (2 * nHalfBoxSize + 1);
    dXX = dXX / (2.0 * nPixels);
    dYY = dYY / (2.0 * nPixels);
    dXY = dXY / (2.0 * nPixels);
    return 0.5 * (dXX + dYY - std::sqrt((dXX + dYY) * (dXX + dYY) - 4 * (dXX * dYY - dXY * dXY)));
}

struct KeyFrame {
    Level lev[PTAM_LEVELS];
    std::vector<ptam_int2> max_corners[PTAM_LEVELS];
    std::vector<double> st_scores[PTAM_LEVELS];
    // KeyFrame::MakeKeyFrame_Rest src/KeyFrame.cc:61-82 (without the SmallBlurryImage)
    void MakeKeyFrame_Rest() {
        for (int l = 0; l < PTAM_LEVELS; l++) {
            Level& L = lev[l];
            // fast_nonmax(lev.im, lev.vCorners, 10, lev.vMaxCorners): scores, then non-strict 3x3
            // suppression among corners (kept unless an 8-neighbour corner scores strictly higher)
            std::vector<int> score(L.corners.size());
            std::vector<int> at((size_t)L.w * L.h, -1);
            for (size_t i = 0; i < L.corners.size(); i++) {
                score[i] = fast_corner_score10(L.im.data(), L.w, L.corners[i].x, L.corners[i].y, 10);
                at[(size_t)L.corners[i].y * L.w + L.corners[i].x] = (int)i;
            }
            max_corners[l].clear();
            st_scores[l].clear();
            for (size_t i = 0; i < L.corners.size(); i++) {
                const int x = L.corners[i].x, y = L.corners[i].y;
                bool keep = true;
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        const int j = at[(size_t)(y + dy) * L.w + x + dx];   // corners lie >= 3 px inside
                        if (j >= 0 && score[j] > score[i]) {
                            keep = false;
                            break;
                        }
                    }
                if (!keep) continue;
                max_corners[l].push_back(L.corners[i]);
                const bool inside = x >= 10 && y >= 10 && x < L.w - 10 && y < L.h - 10;   // :68
                st_scores[l].push_back(inside ? shi_tomasi(L, 3, x, y) : -1.0);
            }
        }
    }
    // src/KeyFrame.cc:18-54
    void MakeKeyFrame_Lite(const uint8_t* im, int w, int h, int stride, int variant) {
        static const int thr[PTAM_LEVELS] = {10, 15, 15, 10};   // :35-42
        lev[0].w = w;
        lev[0].h = h;
        lev[0].im.resize((size_t)w * h);
        for (int y = 0; y < h; y++) std::memcpy(&lev[0].im[(size_t)y * w], im + (size_t)y * stride, w);
        for (int i = 0; i < PTAM_LEVELS; i++) {
            Level& L = lev[i];
            if (i != 0) {
                L.w = lev[i - 1].w / 2;
                L.h = lev[i - 1].h / 2;
                L.im.resize((size_t)L.w * L.h);
                half_sample(lev[i - 1].im.data(), lev[i - 1].w, lev[i - 1].h, L.im.data(), variant);
            }
            L.corners.clear();
            fast10(L.im.data(), L.w, L.h, thr[i], L.corners);
            // row LUT :46-52
            unsigned v = 0;
            L.rowlut.clear();
            for (int y = 0; y < L.h; y++) {
                while (v < L.corners.size() && y > L.corners[v].y) v++;
                L.rowlut.push_back((int)v);
            }
        }
    }
};

// ImageProcess::ZMSSDAtPoint src/ImageProcess.cc:130-163 (8x8 template)
int zmssd_at_point(const Level& L, int x, int y, const uint8_t* tmpl, int tsum, int tsumsq, int max_ssd) {
    const int b = PTAM_PATCH / 2;
    if (!(x >= b && y >= b && x < L.w - b && y < L.h - b)) return max_ssd + 1;
    const int bx = x - b, by = y - b;
    int isumsq = 0, isum = 0, cross = 0;
    for (int r = 0; r < PTAM_PATCH; r++) {
        const uint8_t* ip = &L.im[(size_t)(by + r) * L.w + bx];
        const uint8_t* tp = tmpl + r * PTAM_PATCH;
        for (int c = 0; c < PTAM_PATCH; c++) {
            const int n = ip[c];
            isum += n;
            isumsq += n * n;
            cross += n * tp[c];
        }
    }
    const int SA = tsum, SB = isum, N = PTAM_PATCH * PTAM_PATCH;
    return ((2 * SA * SB - SA * SA - SB * SB) / N + isumsq + tsumsq - 2 * cross);
}

void template_sums(const uint8_t* t, int& sum, int& sumsq) {   // PatchFinder::MakeTemplateSums
    sum = sumsq = 0;
    for (int i = 0; i < 64; i++) {
        sum += t[i];
        sumsq += t[i] * t[i];
    }
}

// PatchFinder::FindPatchCoarse src/PatchFinder.cc:160-211
void find_patch_coarse(const KeyFrame& kf, const ptam_patch_query& q, const uint8_t* tmpl,
                       ptam_patch_result& res) {
    res.found = 0;
    res.best_ssd = PTAM_MAX_SSD + 1;
    res.best_x = res.best_y = -1;
    res.n_scored = 0;
    res.pad_ = 0;
    res.pos[0] = res.pos[1] = 0;
    if (q.level < 0 || q.level >= PTAM_LEVELS) return;   // template bad: not searched (Tracker.cc:874)
    int tsum, tsumsq;
    template_sums(tmpl, tsum, tsumsq);
    const int scale = 1 << q.level;
    int px = q.x / scale, py = q.y / scale;                  // ImageRef / int: C division
    unsigned nRange = (q.range + scale - 1) / scale;          // unsigned like the reference
    int nTop = py - nRange;
    int nBottomPlusOne = py + nRange + 1;
    int nLeft = px - nRange;
    int nRight = px + nRange;
    const Level& L = kf.lev[q.level];
    if (nTop < 0) nTop = 0;
    if (nTop >= L.h) return;
    if (nBottomPlusOne <= 0) return;
    int bx = -1, by = -1;
    int best = PTAM_MAX_SSD + 1;
    size_t i = (size_t)L.rowlut[nTop];
    size_t i_end = nBottomPlusOne >= L.h ? L.corners.size() : (size_t)L.rowlut[nBottomPlusOne];
    for (; i < i_end; i++) {
        const ptam_int2 c = L.corners[i];
        if (c.x < nLeft || c.x > nRight) continue;
        const int dx = px - c.x, dy = py - c.y;
        if ((unsigned)(dx * dx + dy * dy) > nRange * nRange) continue;
        const int ssd = zmssd_at_point(L, c.x, c.y, tmpl, tsum, tsumsq, PTAM_MAX_SSD);
        res.n_scored++;
        if (ssd < best) {
            bx = c.x;
            by = c.y;
            best = ssd;
        }
    }
    res.best_ssd = best;
    res.best_x = bx;
    res.best_y = by;
    if (best < PTAM_MAX_SSD) {
        res.found = 1;
        res.pos[0] = (bx + 0.5) * scale - 0.5;   // Level::LevelZeroPos include/KeyFrame.h:91-94
        res.pos[1] = (by + 0.5) * scale - 0.5;
    }
}

// PatchFinder::MakeSubPixTemplate + IterateSubPix(ToConvergence)  src/PatchFinder.cc:219-318.
// libCVD's ir_rounded (round half away from zero) and ir (truncation) are restated; the bilinear
// mix is written out in float exactly as the reference does (the oracle is built with
// -ffp-contract=off so that no FMA is formed, like upstream's -march=nocona build).
void subpix_refine(const KeyFrame& kf, const ptam_subpix_query& q, const uint8_t* tmpl, ptam_subpix_result& res) {
    res.converged = 0;
    res.iterations = 0;
    res.pos[0] = q.coarse_pos[0];
    res.pos[1] = q.coarse_pos[1];
    res.mean_diff = 0.0;
    if (q.level < 0 || q.level >= PTAM_LEVELS) return;
    const Level& L = kf.lev[q.level];
    const int P = PTAM_PATCH;
    // MakeSubPixTemplate :219-240  (ir.x outer, ir.y inner)
    float jx[6][6], jy[6][6];   // mimJacs[ir - (1,1)] indexed [y-1][x-1]
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 1; x < P - 1; x++)
        for (int y = 1; y < P - 1; y++) {
            const double gx = 0.5 * (tmpl[y * P + x + 1] - tmpl[y * P + x - 1]);
            const double gy = 0.5 * (tmpl[(y + 1) * P + x] - tmpl[(y - 1) * P + x]);
            jx[y - 1][x - 1] = static_cast<float>(gx);
            jy[y - 1][x - 1] = static_cast<float>(gy);
            const double g3[3] = {gx, gy, 1.0};
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) H[a * 3 + b] += g3[a] * g3[b];
        }
    double Hinv[9];
    {
        LDLT chol(3, H);
        chol.inverse(Hinv);
    }
    double pos[2] = {q.coarse_pos[0], q.coarse_pos[1]}, mean_diff = 0.0;
    const int scale = 1 << q.level;
    const double dConvLimit = 0.03;
    for (int it = 0; it < q.max_its; it++) {
        res.iterations = it + 1;
        // IterateSubPix :271-318
        const double cx = (pos[0] + 0.5) / scale - 0.5, cy = (pos[1] + 0.5) / scale - 0.5;   // LevelNPos
        const int rx = static_cast<int>(cx > 0.0 ? cx + 0.5 : cx - 0.5), ry = static_cast<int>(cy > 0.0 ? cy + 0.5 : cy - 0.5);
        const int bord = P / 2 + 1;
        if (!(rx >= bord && ry >= bord && rx < L.w - bord && ry < L.h - bord)) break;   // returns -1 -> false
        const double bx = cx - 4, by = cy - 4;
        const double dX = bx - std::floor(bx), dY = by - std::floor(by);
        const float fTL = static_cast<float>((1.0 - dX) * (1.0 - dY)), fTR = static_cast<float>(dX * (1.0 - dY));
        const float fBL = static_cast<float>((1.0 - dX) * dY), fBR = static_cast<float>(dX * dY);
        const int ibx = static_cast<int>(bx), iby = static_cast<int>(by);   // ::ir() truncation
        double acc[3] = {0, 0, 0};
        for (int y = 1; y < P - 1; y++) {
            const uint8_t* p = &L.im[(size_t)(iby + y) * L.w + ibx + 1];
            for (int x = 1; x < P - 1; x++) {
                const float fPixel = fTL * p[0] + fTR * p[1] + fBL * p[L.w] + fBR * p[L.w + 1];
                p++;
                const double dDiff = fPixel - tmpl[y * P + x] + mean_diff;
                acc[0] += dDiff * jx[y - 1][x - 1];
                acc[1] += dDiff * jy[y - 1][x - 1];
                acc[2] += dDiff;
            }
        }
        double upd[3];
        for (int a = 0; a < 3; a++) upd[a] = Hinv[a * 3] * acc[0] + Hinv[a * 3 + 1] * acc[1] + Hinv[a * 3 + 2] * acc[2];
        pos[0] -= upd[0] * scale;
        pos[1] -= upd[1] * scale;
        mean_diff -= upd[2];
        const double d2 = upd[0] * upd[0] + upd[1] * upd[1];
        if (d2 < dConvLimit * dConvLimit) {
            res.converged = 1;
            break;
        }
    }
    res.pos[0] = pos[0];
    res.pos[1] = pos[1];
    res.mean_diff = mean_diff;
}

// ------------------------------------------------------------------------------------------------
// TrackerData: include/Tracker.h:41-145
// ------------------------------------------------------------------------------------------------
struct TrackerData {
    double world[3];
    double v3Cam[3], v2ImPlane[2], v2Image[2], m2CamDerivs[4];
    bool bInImage = false, bFound = false;
    double v2Found[2], dSqrtInvNoise;
    double v2Error_CovScaled[2];
    double m26Jacobian[12];

    // :70-85; returns true if the camera model was reached (for the stale-cache accounting)
    bool Project(const SE3& T, ATANCamera& cam) {
        bInImage = false;
        se3_apply(T, world, v3Cam);
        if (v3Cam[2] < 0.001) return false;
        v2ImPlane[0] = v3Cam[0] / v3Cam[2];
        v2ImPlane[1] = v3Cam[1] / v3Cam[2];
        if (v2ImPlane[0] * v2ImPlane[0] + v2ImPlane[1] * v2ImPlane[1] >
            cam.largest_radius * cam.largest_radius)
            return false;
        cam.Project(v2ImPlane, v2Image);
        if (cam.invalid) return true;
        if (v2Image[0] < 0 || v2Image[1] < 0 || v2Image[0] > cam.size[0] || v2Image[1] > cam.size[1])
            return true;
        bInImage = true;
        return true;
    }
    void CalcJacobian() {   // :125-136
        const double inv_z = 1.0 / v3Cam[2];
        const double p4[4] = {v3Cam[0], v3Cam[1], v3Cam[2], 1.0};
        for (int m = 0; m < 6; m++) {
            double g[4];
            generator_field(m, p4, g);
            const double mx = (g[0] - v3Cam[0] * g[2] * inv_z) * inv_z;
            const double my = (g[1] - v3Cam[1] * g[2] * inv_z) * inv_z;
            m26Jacobian[m] = m2CamDerivs[0] * mx + m2CamDerivs[1] * my;
            m26Jacobian[6 + m] = m2CamDerivs[2] * mx + m2CamDerivs[3] * my;
        }
    }
    void LinearUpdate(const double v6[6]) {   // :139-142
        for (int r = 0; r < 2; r++) {
            double s = 0;
            for (int m = 0; m < 6; m++) s += m26Jacobian[r * 6 + m] * v6[m];
            v2Image[r] += s;
        }
    }
};

// Tracker::CalcPoseUpdate src/Tracker.cc:928-1005
void calc_pose_update(std::vector<TrackerData>& vTD, double overrideSigma, int estimator, double prior,
                      bool markOutliers, int32_t* outlier_flags, double mu[6]) {
    std::vector<double> vdErrorSquared;
    for (auto& TD : vTD) {
        if (!TD.bFound) continue;
        TD.v2Error_CovScaled[0] = TD.dSqrtInvNoise * (TD.v2Found[0] - TD.v2Image[0]);
        TD.v2Error_CovScaled[1] = TD.dSqrtInvNoise * (TD.v2Found[1] - TD.v2Image[1]);
        vdErrorSquared.push_back(TD.v2Error_CovScaled[0] * TD.v2Error_CovScaled[0] +
                                 TD.v2Error_CovScaled[1] * TD.v2Error_CovScaled[1]);
    }
    for (int i = 0; i < 6; i++) mu[i] = 0;
    if (vdErrorSquared.empty()) return;
    double dSigmaSquared;
    if (overrideSigma > 0)
        dSigmaSquared = overrideSigma;
    else if (estimator == PTAM_EST_TUKEY)
        dSigmaSquared = Tukey::FindSigmaSquared(vdErrorSquared);
    else if (estimator == PTAM_EST_CAUCHY)
        dSigmaSquared = Cauchy::FindSigmaSquared(vdErrorSquared);
    else
        dSigmaSquared = Huber::FindSigmaSquared(vdErrorSquared);

    // WLS<6>: add_prior, add_mJ, compute
    double C[36] = {0}, b[6] = {0};
    for (int i = 0; i < 6; i++) C[i * 6 + i] += prior;
    size_t idx = 0;
    for (auto& TD : vTD) {
        const size_t me = idx++;
        if (!TD.bFound) continue;
        const double* v2 = TD.v2Error_CovScaled;
        const double e2 = v2[0] * v2[0] + v2[1] * v2[1];
        double wgt;
        if (estimator == PTAM_EST_TUKEY)
            wgt = Tukey::Weight(e2, dSigmaSquared);
        else if (estimator == PTAM_EST_CAUCHY)
            wgt = Cauchy::Weight(e2, dSigmaSquared);
        else
            wgt = Huber::Weight(e2, dSigmaSquared);
        if (wgt == 0.0) {
            if (markOutliers && outlier_flags) outlier_flags[me] = 1;
            continue;
        }
        for (int r = 0; r < 2; r++) {
            double J[6], Jw[6];
            for (int m = 0; m < 6; m++) {
                J[m] = TD.dSqrtInvNoise * TD.m26Jacobian[r * 6 + m];
                Jw[m] = J[m] * wgt;
            }
            for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 6; j++) C[i * 6 + j] += Jw[i] * J[j];
                b[i] += v2[r] * Jw[i];
            }
        }
    }
    LDLT chol(6, C);
    chol.backsub(b, mu);
}

// ------------------------------------------------------------------------------------------------
// Bundle: include/Bundle.h:38-152, src/Bundle.cc (whole file)
// ------------------------------------------------------------------------------------------------
struct BCamera {
    bool bFixed;
    SE3 se3CfW, se3CfWNew;
    double m6U[36];
    double v6EpsilonA[6];
    int nStartRow;
};
struct OffDiagScriptEntry {
    int j, k;
};
struct BPoint {
    double v3Pos[3], v3PosNew[3];
    double m3V[9], v3EpsilonB[3], m3VStarInv[9];
    int nMeasurements = 0, nOutliers = 0;
    std::set<int> sCameras;
    std::vector<OffDiagScriptEntry> vOffDiagonalScript;
};
struct Meas {
    int p, c;
    bool bBad = false;
    double v2Found[2], v2Epsilon[2];
    double m26A[12], m23B[6], m63W[18];
    double dSqrtInvNoise;
    double v3Cam[3], dErrorSquared, m2CamDerivs[4];
};

typedef int (*allreduce_fn)(void* user, double* ptr, size_t count, void* stream);

struct Bundle {
    ATANCamera mCamera;
    ptam_ba_opts opts;
    std::vector<BPoint> mvPoints;
    std::vector<BCamera> mvCameras;
    std::list<Meas> mMeasList;
    std::vector<std::pair<int, int>> mvOutlierMeasurementIdx;
    std::vector<std::vector<Meas*>> mvMeasLUTs;
    int mnCamsToUpdate = 0, mnStartRow = 0;
    double mdSigmaSquared = 0, mdLambda = 0, mdLambdaFactor = 0;
    bool mbConverged = false, mbHitMaxIterations = false;
    int mnCounter = 0, mnAccepted = 0;
    std::vector<ptam_ba_trial> trials;
    // sharded mode (test infrastructure for the N>1 protocol)
    int rank = 0, world = 1;
    allreduce_fn comm = nullptr;
    void* comm_user = nullptr;

    Bundle(const ptam_cam_params& cam, const ptam_ba_opts& o) : mCamera(cam), opts(o) {}

    int AddCamera(const SE3& pose, bool bFixed) {   // src/Bundle.cc:46-63
        const int n = (int)mvCameras.size();
        BCamera c;
        c.bFixed = bFixed;
        c.se3CfW = pose;
        c.se3CfWNew = pose;
        if (!bFixed) {
            c.nStartRow = mnStartRow;
            mnStartRow += 6;
            mnCamsToUpdate++;
        } else
            c.nStartRow = -999999999;
        mvCameras.push_back(c);
        return n;
    }
    int AddPoint(const double* pos) {   // :66-79
        const int n = (int)mvPoints.size();
        BPoint p;
        double v[3] = {pos[0], pos[1], pos[2]};
        if (std::isnan(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])) v[0] = v[1] = v[2] = 0;
        std::memcpy(p.v3Pos, v, sizeof v);
        std::memcpy(p.v3PosNew, v, sizeof v);
        mvPoints.push_back(p);
        return n;
    }
    void AddMeas(int nCam, int nPoint, const double* v2Pos, double dSigmaSquared) {   // :82-93
        mvPoints[nPoint].nMeasurements++;
        mvPoints[nPoint].sCameras.insert(nCam);
        Meas m;
        m.p = nPoint;
        m.c = nCam;
        m.v2Found[0] = v2Pos[0];
        m.v2Found[1] = v2Pos[1];
        m.dSqrtInvNoise = std::sqrt(1.0 / dSigmaSquared);
        mMeasList.push_back(m);
    }
    void ClearAccumulators() {   // :96-108
        for (auto& p : mvPoints) {
            std::fill(p.m3V, p.m3V + 9, 0.0);
            std::fill(p.v3EpsilonB, p.v3EpsilonB + 3, 0.0);
        }
        for (auto& c : mvCameras) {
            std::fill(c.m6U, c.m6U + 36, 0.0);
            std::fill(c.v6EpsilonA, c.v6EpsilonA + 6, 0.0);
        }
    }
    void GenerateMeasLUTs() {   // :558-567
        mvMeasLUTs.clear();
        for (size_t c = 0; c < mvCameras.size(); c++)
            mvMeasLUTs.push_back(std::vector<Meas*>(mvPoints.size(), nullptr));
        for (auto& m : mMeasList) mvMeasLUTs[m.c][m.p] = &m;
    }
    void GenerateOffDiagScripts() {   // :572-599
        for (size_t i = 0; i < mvPoints.size(); i++) {
            BPoint& p = mvPoints[i];
            p.vOffDiagonalScript.clear();
            for (auto it_j = p.sCameras.begin(); it_j != p.sCameras.end(); ++it_j) {
                const int j = *it_j;
                if (mvCameras[j].bFixed) continue;
                for (auto it_k = p.sCameras.begin(); it_k != it_j; ++it_k) {
                    const int k = *it_k;
                    if (mvCameras[k].bFixed) continue;
                    p.vOffDiagonalScript.push_back({j, k});
                }
            }
        }
    }
    void ProjectAndFindSquaredError(Meas& meas) {   // :164-180
        BCamera& cam = mvCameras[meas.c];
        BPoint& point = mvPoints[meas.p];
        se3_apply(cam.se3CfW, point.v3Pos, meas.v3Cam);
        if (meas.v3Cam[2] <= 0) {
            meas.bBad = true;
            return;
        }
        meas.bBad = false;
        const double ip[2] = {meas.v3Cam[0] / meas.v3Cam[2], meas.v3Cam[1] / meas.v3Cam[2]};
        double im[2];
        mCamera.Project(ip, im);
        mCamera.GetProjectionDerivs(meas.m2CamDerivs);
        meas.v2Epsilon[0] = meas.dSqrtInvNoise * (meas.v2Found[0] - im[0]);
        meas.v2Epsilon[1] = meas.dSqrtInvNoise * (meas.v2Found[1] - im[1]);
        meas.dErrorSquared = meas.v2Epsilon[0] * meas.v2Epsilon[0] + meas.v2Epsilon[1] * meas.v2Epsilon[1];
    }
    template <class ME>
    double FindNewError() {   // :188-207
        double dNewError = 0;
        for (auto& meas : mMeasList) {
            double v3Cam[3];
            se3_apply(mvCameras[meas.c].se3CfWNew, mvPoints[meas.p].v3PosNew, v3Cam);
            if (v3Cam[2] <= 0) {
                dNewError += 1.0;
                continue;
            }
            const double ip[2] = {v3Cam[0] / v3Cam[2], v3Cam[1] / v3Cam[2]};
            double im[2];
            mCamera.Project(ip, im);
            const double ex = meas.dSqrtInvNoise * (meas.v2Found[0] - im[0]);
            const double ey = meas.dSqrtInvNoise * (meas.v2Found[1] - im[1]);
            dNewError += ME::ObjectiveScore(ex * ex + ey * ey, mdSigmaSquared);
        }
        return dNewError;
    }

    // sharded helpers: sum a scalar / a buffer over ranks (identity when no comm is attached)
    void allreduce(double* p, size_t n) {
        if (comm && world > 1) {
            const int rc = comm(comm_user, p, n, nullptr);
            if (rc != 0) std::fprintf(stderr, "oracle: allreduce hook failed (%d)\n", rc);
        }
    }
    // the abort flag (src/Bundle.cc:134,338): sharded, the decision has to be the same on every rank at the same point of
    // the control flow (every trial runs collectives), so the local flags are summed; identity without a communicator
    bool Aborted(const volatile unsigned char* pb) {
        double a = (pb && *pb) ? 1.0 : 0.0;
        allreduce(&a, 1);
        return a > 0.5;
    }
    // gather every rank's squared errors (all-gather built from two all-reduces)
    void gather_errors(std::vector<double>& v) {
        if (!(comm && world > 1)) return;
        std::vector<double> counts(world, 0.0);
        counts[rank] = (double)v.size();
        allreduce(counts.data(), counts.size());
        size_t total = 0, off = 0;
        for (int r = 0; r < world; r++) {
            if (r == rank) off = total;
            total += (size_t)counts[r];
        }
        std::vector<double> all(total, 0.0);
        std::copy(v.begin(), v.end(), all.begin() + off);
        allreduce(all.data(), all.size());
        v.swap(all);
    }

    template <class ME>
    bool Do_LM_Step(const volatile unsigned char* pbAbort) {   // :209-551
        ClearAccumulators();
        std::vector<double> vdErrorSquared;
        for (auto& meas : mMeasList) {   // pass 1 :219-225
            ProjectAndFindSquaredError(meas);
            if (!meas.bBad) vdErrorSquared.push_back(meas.dErrorSquared);
        }
        gather_errors(vdErrorSquared);
        mdSigmaSquared = ME::FindSigmaSquared(vdErrorSquared);   // :230
        const double dMinSigmaSquared = opts.min_sigma * opts.min_sigma;
        if (mdSigmaSquared < dMinSigmaSquared) mdSigmaSquared = dMinSigmaSquared;

        double dCurrentError = 0.0;
        for (auto& meas : mMeasList) {   // pass 2 :250-332
            BCamera& cam = mvCameras[meas.c];
            BPoint& point = mvPoints[meas.p];
            if (meas.bBad) {
                dCurrentError += 1.0;
                continue;
            }
            const double dWeight = ME::SquareRootWeight(meas.dErrorSquared, mdSigmaSquared);
            meas.v2Epsilon[0] = dWeight * meas.v2Epsilon[0];
            meas.v2Epsilon[1] = dWeight * meas.v2Epsilon[1];
            if (dWeight == 0) {
                meas.bBad = true;
                dCurrentError += 1.0;
                continue;
            }
            dCurrentError += ME::ObjectiveScore(meas.dErrorSquared, mdSigmaSquared);
            double D[4];
            for (int i = 0; i < 4; i++) D[i] = dWeight * meas.m2CamDerivs[i];
            const double inv_z = 1.0 / meas.v3Cam[2];
            const double v4Cam[4] = {meas.v3Cam[0], meas.v3Cam[1], meas.v3Cam[2], 1.0};
            if (cam.bFixed)
                std::fill(meas.m26A, meas.m26A + 12, 0.0);
            else
                for (int m = 0; m < 6; m++) {
                    double g[4];
                    generator_field(m, v4Cam, g);
                    const double mx = (g[0] - v4Cam[0] * g[2] * inv_z) * inv_z;
                    const double my = (g[1] - v4Cam[1] * g[2] * inv_z) * inv_z;
                    meas.m26A[m] = meas.dSqrtInvNoise * (D[0] * mx + D[1] * my);
                    meas.m26A[6 + m] = meas.dSqrtInvNoise * (D[2] * mx + D[3] * my);
                }
            for (int m = 0; m < 3; m++) {
                // m-th column of R_cw
                const double g[3] = {cam.se3CfW.R[0 * 3 + m], cam.se3CfW.R[1 * 3 + m], cam.se3CfW.R[2 * 3 + m]};
                const double mx = (g[0] - v4Cam[0] * g[2] * inv_z) * inv_z;
                const double my = (g[1] - v4Cam[1] * g[2] * inv_z) * inv_z;
                meas.m23B[m] = meas.dSqrtInvNoise * (D[0] * mx + D[1] * my);
                meas.m23B[3 + m] = meas.dSqrtInvNoise * (D[2] * mx + D[3] * my);
            }
            if (!cam.bFixed) {
                for (int r = 0; r < 6; r++)   // BundleTriangle_UpdateM6U_LL :21-26
                    for (int c = 0; c <= r; c++)
                        cam.m6U[r * 6 + c] += meas.m26A[r] * meas.m26A[c] + meas.m26A[6 + r] * meas.m26A[6 + c];
                for (int r = 0; r < 6; r++)
                    cam.v6EpsilonA[r] += meas.m26A[r] * meas.v2Epsilon[0] + meas.m26A[6 + r] * meas.v2Epsilon[1];
            }
            for (int r = 0; r < 3; r++)   // BundleTriangle_UpdateM3V_LL :27-32
                for (int c = 0; c <= r; c++)
                    point.m3V[r * 3 + c] += meas.m23B[r] * meas.m23B[c] + meas.m23B[3 + r] * meas.m23B[3 + c];
            for (int r = 0; r < 3; r++)
                point.v3EpsilonB[r] += meas.m23B[r] * meas.v2Epsilon[0] + meas.m23B[3 + r] * meas.v2Epsilon[1];
            if (cam.bFixed)
                std::fill(meas.m63W, meas.m63W + 18, 0.0);
            else
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 3; c++)
                        meas.m63W[r * 3 + c] = meas.m26A[r] * meas.m23B[c] + meas.m26A[6 + r] * meas.m23B[3 + c];
        }
        // sharded mode: U / epsA stay per-rank partial sums; (1+lambda)*diag(U) is linear, so the
        // partials fold into the single S/E all-reduce below
        allreduce(&dCurrentError, 1);

        const int n = mnCamsToUpdate * 6;
        double dNewError = dCurrentError + 9999;
        // NaN / inf current error: the reference never enters the trial loop below, returns true, and Compute() calls it
        // again for ever.  The restatement (and the product) give up instead, so that a test cannot hang.
        if (!(dNewError > dCurrentError)) mbHitMaxIterations = true;
        int nBadSoFar = 0;
        for (auto& m : mMeasList) nBadSoFar += m.bBad;
        {
            double nb = nBadSoFar;   // trial log only: global count in sharded mode
            allreduce(&nb, 1);
            nBadSoFar = (int)(nb + 0.5);
        }
        while (dNewError > dCurrentError && !mbConverged && !mbHitMaxIterations && !Aborted(pbAbort)) {
            for (auto& point : mvPoints) {   // V*^-1 :341-359
                double V[9];
                std::memcpy(V, point.m3V, sizeof V);
                if (V[0] * V[4] * V[8] == 0)
                    std::fill(point.m3VStarInv, point.m3VStarInv + 9, 0.0);
                else {
                    V[1] = V[3];
                    V[2] = V[6];
                    V[5] = V[7];
                    for (int i = 0; i < 3; i++) V[i * 3 + i] *= (1.0 + mdLambda);
                    LDLT chol(3, V);
                    chol.inverse(point.m3VStarInv);
                }
            }
            std::vector<double> mS((size_t)n * n, 0.0), vE(n, 0.0);
            double m6[36], v6[6];
            for (size_t j = 0; j < mvCameras.size(); j++) {   // diagonal blocks :374-406
                BCamera& cam_j = mvCameras[j];
                if (cam_j.bFixed) continue;
                const int row = cam_j.nStartRow;
                for (int r = 0; r < 6; r++) {
                    for (int c = 0; c < r; c++) m6[r * 6 + c] = m6[c * 6 + r] = cam_j.m6U[r * 6 + c];
                    m6[r * 6 + r] = cam_j.m6U[r * 6 + r];
                }
                for (int nn = 0; nn < 6; nn++) m6[nn * 6 + nn] *= (1.0 + mdLambda);
                std::memcpy(v6, cam_j.v6EpsilonA, sizeof v6);
                std::vector<Meas*>& lut = mvMeasLUTs[j];
                for (size_t i = 0; i < mvPoints.size(); i++) {
                    Meas* pm = lut[i];
                    if (pm == nullptr || pm->bBad) continue;
                    const double* Vi = mvPoints[i].m3VStarInv;
                    double WV[18];
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 3; c++)
                            WV[r * 3 + c] = pm->m63W[r * 3 + 0] * Vi[0 * 3 + c] + pm->m63W[r * 3 + 1] * Vi[1 * 3 + c] +
                                            pm->m63W[r * 3 + 2] * Vi[2 * 3 + c];
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++)
                            m6[r * 6 + c] -= WV[r * 3 + 0] * pm->m63W[c * 3 + 0] + WV[r * 3 + 1] * pm->m63W[c * 3 + 1] +
                                             WV[r * 3 + 2] * pm->m63W[c * 3 + 2];
                    double ve[3];
                    for (int r = 0; r < 3; r++)
                        ve[r] = Vi[r * 3 + 0] * mvPoints[i].v3EpsilonB[0] + Vi[r * 3 + 1] * mvPoints[i].v3EpsilonB[1] +
                                Vi[r * 3 + 2] * mvPoints[i].v3EpsilonB[2];
                    for (int r = 0; r < 6; r++)
                        v6[r] -= pm->m63W[r * 3 + 0] * ve[0] + pm->m63W[r * 3 + 1] * ve[1] + pm->m63W[r * 3 + 2] * ve[2];
                }
                for (int r = 0; r < 6; r++) {
                    for (int c = 0; c < 6; c++) mS[(size_t)(row + r) * n + row + c] = m6[r * 6 + c];
                    vE[row + r] = v6[r];
                }
            }
            for (size_t i = 0; i < mvPoints.size(); i++) {   // off-diagonal blocks :410-446
                BPoint& p = mvPoints[i];
                int nCurrentJ = -1, nJRow = -1;
                double WV[18];
                for (auto& e : p.vOffDiagonalScript) {
                    Meas* pMeas_ik = mvMeasLUTs[e.k][i];
                    if (pMeas_ik == nullptr || pMeas_ik->bBad) continue;
                    if (e.j != nCurrentJ) {
                        Meas* pMeas_ij = mvMeasLUTs[e.j][i];
                        if (pMeas_ij == nullptr || pMeas_ij->bBad) continue;
                        nCurrentJ = e.j;
                        nJRow = mvCameras[e.j].nStartRow;
                        for (int r = 0; r < 6; r++)
                            for (int c = 0; c < 3; c++)
                                WV[r * 3 + c] = pMeas_ij->m63W[r * 3 + 0] * p.m3VStarInv[0 * 3 + c] +
                                                pMeas_ij->m63W[r * 3 + 1] * p.m3VStarInv[1 * 3 + c] +
                                                pMeas_ij->m63W[r * 3 + 2] * p.m3VStarInv[2 * 3 + c];
                    }
                    const int nKRow = mvCameras[pMeas_ik->c].nStartRow;
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++)
                            mS[(size_t)(nJRow + r) * n + nKRow + c] -=
                                WV[r * 3 + 0] * pMeas_ik->m63W[c * 3 + 0] + WV[r * 3 + 1] * pMeas_ik->m63W[c * 3 + 1] +
                                WV[r * 3 + 2] * pMeas_ik->m63W[c * 3 + 2];
                    assert(nKRow < nJRow);
                }
            }
            if (comm && world > 1) {   // the one exchange step of the path: lower triangle of S, then E
                std::vector<double> pk((size_t)n * (n + 1) / 2 + n);
                size_t q = 0;
                for (int i = 0; i < n; i++)
                    for (int j = 0; j <= i; j++) pk[q++] = mS[(size_t)i * n + j];
                for (int i = 0; i < n; i++) pk[q++] = vE[i];
                allreduce(pk.data(), pk.size());
                q = 0;
                for (int i = 0; i < n; i++)
                    for (int j = 0; j <= i; j++) mS[(size_t)i * n + j] = pk[q++];
                for (int i = 0; i < n; i++) vE[i] = pk[q++];
            }
            for (int i = 0; i < n; i++)   // mirror :451-453
                for (int j = 0; j < i; j++) mS[(size_t)j * n + i] = mS[(size_t)i * n + j];

            std::vector<double> vCamerasUpdate(n, 0.0);   // :457-458
            if (n > 0) {
                LDLT chol(n, mS.data());
                chol.backsub(vE.data(), vCamerasUpdate.data());
            }
            std::vector<double> vMapUpdates(mvPoints.size() * 3);   // :461-483
            for (size_t i = 0; i < mvPoints.size(); i++) {
                double v3Sum[3] = {0, 0, 0};
                for (size_t j = 0; j < mvCameras.size(); j++) {
                    BCamera& cam = mvCameras[j];
                    if (cam.bFixed) continue;
                    Meas* pm = mvMeasLUTs[j][i];
                    if (pm == nullptr || pm->bBad) continue;
                    const double* da = &vCamerasUpdate[cam.nStartRow];
                    for (int c = 0; c < 3; c++) {
                        double s = 0;
                        for (int r = 0; r < 6; r++) s += pm->m63W[r * 3 + c] * da[r];
                        v3Sum[c] += s;
                    }
                }
                double v3[3];
                for (int c = 0; c < 3; c++) v3[c] = mvPoints[i].v3EpsilonB[c] - v3Sum[c];
                const double* Vi = mvPoints[i].m3VStarInv;
                for (int r = 0; r < 3; r++)
                    vMapUpdates[i * 3 + r] = Vi[r * 3 + 0] * v3[0] + Vi[r * 3 + 1] * v3[1] + Vi[r * 3 + 2] * v3[2];
            }
            double dMapSq = 0;
            for (double v : vMapUpdates) dMapSq += v * v;
            allreduce(&dMapSq, 1);
            double dSumSquaredUpdate = dMapSq;   // :488-490
            for (double v : vCamerasUpdate) dSumSquaredUpdate += v * v;
            if (dSumSquaredUpdate < opts.update_sq_conv_limit) mbConverged = true;

            for (auto& cam : mvCameras) {   // :496-504
                if (cam.bFixed)
                    cam.se3CfWNew = cam.se3CfW;
                else
                    cam.se3CfWNew = se3_mul(se3_exp(&vCamerasUpdate[cam.nStartRow]), cam.se3CfW);
            }
            for (size_t i = 0; i < mvPoints.size(); i++)
                for (int c = 0; c < 3; c++) mvPoints[i].v3PosNew[c] = mvPoints[i].v3Pos[c] + vMapUpdates[i * 3 + c];
            dNewError = FindNewError<ME>();   // :506
            allreduce(&dNewError, 1);

            ptam_ba_trial t;
            t.lambda = mdLambda;
            t.sigma_sq = mdSigmaSquared;
            t.err_old = dCurrentError;
            t.err_new = dNewError;
            t.sum_sq_update = dSumSquaredUpdate;
            t.n_bad = nBadSoFar;
            t.accepted = 0;
            if (opts.verbose)
                std::printf("L%.1e\tOld %.6f  New %.6f  Diff %.6f\n", mdLambda, dCurrentError, dNewError,
                            dCurrentError - dNewError);
            if (dNewError > dCurrentError) {   // ModifyLambda_BadStep :607-611
                mdLambda = mdLambda * mdLambdaFactor;
                mdLambdaFactor = mdLambdaFactor * 2;
            }
            mnCounter++;
            if (mnCounter >= opts.max_iterations) mbHitMaxIterations = true;
            trials.push_back(t);
        }
        if (dNewError < dCurrentError) {   // :523-533, ModifyLambda_GoodStep :601-605
            mdLambdaFactor = 2.0;
            mdLambda *= 0.3;
            for (auto& c : mvCameras) c.se3CfW = c.se3CfWNew;
            for (auto& p : mvPoints) std::memcpy(p.v3Pos, p.v3PosNew, sizeof p.v3Pos);
            mnAccepted++;
            if (!trials.empty()) trials.back().accepted = 1;
        }
        // ditch the outliers :536-547
        for (auto it = mMeasList.begin(); it != mMeasList.end();) {
            if (it->bBad) {
                mvOutlierMeasurementIdx.push_back(std::make_pair(it->p, it->c));
                mvPoints[it->p].nOutliers++;
                mvMeasLUTs[it->c][it->p] = nullptr;
                it = mMeasList.erase(it);
            } else
                ++it;
        }
        return true;
    }

    int Compute(const volatile unsigned char* pbAbort) {   // :116-158
        GenerateMeasLUTs();
        GenerateOffDiagScripts();
        mdLambda = 0.0001;
        mdLambdaFactor = 2.0;
        mbConverged = false;
        mbHitMaxIterations = false;
        mnCounter = 0;
        mnAccepted = 0;
        trials.clear();
        while (!mbConverged && !mbHitMaxIterations && !Aborted(pbAbort)) {
            bool ok;
            if (opts.estimator == PTAM_EST_CAUCHY)
                ok = Do_LM_Step<Cauchy>(pbAbort);
            else if (opts.estimator == PTAM_EST_HUBER)
                ok = Do_LM_Step<Huber>(pbAbort);
            else
                ok = Do_LM_Step<Tukey>(pbAbort);
            if (!ok) return -1;
        }
        return mnAccepted;
    }
};

struct OCtx {
    ptam_cam_params cam;
    int variant = PTAM_HALFSAMPLE_R;
    long cache_hazards = 0;   // ProjectAndDerivs calls that read another point's cached derivs
};

}   // namespace

// ================================================================================================
// MapMaker::AddPointEpipolar, lines 598-637: template without warp, in-plane corner table, band / segment test,
// ZMSSD of the survivors, first strict minimum in corner order.
static void implane_corners(const ATANCamera& cam, const Level& L, int level, std::vector<double>& out) {
    out.resize(L.corners.size() * 2);
    const int scale = 1 << level;
    for (size_t i = 0; i < L.corners.size(); i++) {
        // imUnProj[ir(Level::LevelZeroPos(vIR[i], nLevel))]  :611-612 — ir() truncates
        const double im[2] = {(double)(int)((L.corners[i].x + 0.5) * scale - 0.5), (double)(int)((L.corners[i].y + 0.5) * scale - 0.5)};
        cam.UnProject(im, &out[2 * i]);
    }
}
static void epipolar_search(const Level& src, const Level& tgt, const std::vector<double>& ip, const ptam_epipolar_query& q,
                            ptam_epipolar_result& r) {
    r.best = -1;
    r.best_zmssd = PTAM_MAX_SSD + 1;
    r.n_scored = 0;
    r.template_bad = 0;
    const int b = PTAM_PATCH / 2 + 1;   // MakeTemplateCoarseNoWarp src/PatchFinder.cc:141
    if (!(q.level_x >= b && q.level_y >= b && q.level_x < src.w - b && q.level_y < src.h - b)) {
        r.template_bad = 1;
        return;
    }
    uint8_t tmpl[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) tmpl[y * 8 + x] = src.im[(size_t)(q.level_y - 4 + y) * src.w + q.level_x - 4 + x];
    int tsum, tsumsq;
    template_sums(tmpl, tsum, tsumsq);
    for (size_t i = 0; i < tgt.corners.size(); i++) {
        const double vx = ip[2 * i], vy = ip[2 * i + 1];
        const double dd = q.norm_dist - (vx * q.normal[0] + vy * q.normal[1]);
        if (dd * dd > q.max_dist_sq) continue;
        const double al = vx * q.along[0] + vy * q.along[1];
        if (al < q.min_len) continue;
        if (al > q.max_len) continue;
        const int z = zmssd_at_point(tgt, tgt.corners[i].x, tgt.corners[i].y, tmpl, tsum, tsumsq, PTAM_MAX_SSD);
        r.n_scored++;
        if (z < r.best_zmssd) {
            r.best = (int)i;
            r.best_zmssd = z;
        }
    }
}

// PatchFinder::MakeTemplateCoarseCont (src/PatchFinder.cc:98-127) minus the host-side reuse test:
// m2 = M2Inverse(mm2WarpInverse) * LevelScale(mnSearchLevel) (include/Tools.h:54-65), CVD::transform of
// the source level into the 8x8 template with inOrig = irCenter, outOrig = (4,4), then MakeTemplateSums.
// CVD::transform / CVD::sample belong to libCVD (absent from the reference tree): restated from the
// library's published vision.h — UNPINNED like the rest of this file.  Statement by statement:
//   across = M.T()[0], down = M.T()[1], p0 = inOrig - M*outOrig; bounding box from p0 with w*across / h*down
//   added on the side of their sign; if the box lies in [0, iw-1) x [0, ih-1): sample every pixel, else test
//   each position (0 <= p && p < bound) and write byte() = 0 + count it; the position advances by
//   p += across per pixel and p += (down - w*across) per row; sample = bilinear in double,
//   (1-y)*((1-x)*a + x*b) + y*((1-x)*c + x*d), converted by static_cast<byte>.
static void make_template_coarse_cont(const Level& src, int cx, int cy, int search_level, const double wi[4], uint8_t out[64],
                                      ptam_template_result& r) {
    const double det = wi[0] * wi[3] - wi[2] * wi[1];
    const double inv = 1.0 / det;
    const double sc = (double)(1 << search_level);
    const double m00 = wi[3] * inv * sc, m11 = wi[0] * inv * sc, m10 = -wi[2] * inv * sc, m01 = -wi[1] * inv * sc;
    r.m2[0] = m00, r.m2[1] = m01, r.m2[2] = m10, r.m2[3] = m11;
    const int w = 8, h = 8, iw = src.w, ih = src.h;
    const double ax = m00, ay = m10, dx = m01, dy = m11;
    const double p0x = (double)cx - (m00 * 4.0 + m01 * 4.0), p0y = (double)cy - (m10 * 4.0 + m11 * 4.0);
    double min_x = p0x, min_y = p0y, max_x = p0x, max_y = p0y;
    if (ax < 0) min_x += w * ax; else max_x += w * ax;
    if (dx < 0) min_x += h * dx; else max_x += h * dx;
    if (ay < 0) min_y += w * ay; else max_y += w * ay;
    if (dy < 0) min_y += h * dy; else max_y += h * dy;
    const double crx = dx - w * ax, cry = dy - w * ay;
    const bool all_inside = min_x >= 0 && min_y >= 0 && max_x < iw - 1 && max_y < ih - 1;
    const double x_bound = iw - 1, y_bound = ih - 1;
    int count = 0;
    double px = p0x, py = p0y;
    for (int i = 0; i < h; ++i, px += crx, py += cry)
        for (int j = 0; j < w; ++j, px += ax, py += ay) {
            if (all_inside || (0 <= px && 0 <= py && px < x_bound && py < y_bound)) {
                const int lx = (int)px, ly = (int)py;
                const double x = px - lx, y = py - ly;
                const uint8_t* p = src.im.data() + (size_t)ly * iw + lx;
                const double v = (1 - y) * ((1 - x) * p[0] + x * p[1]) + y * ((1 - x) * p[iw] + x * p[iw + 1]);
                out[i * 8 + j] = static_cast<uint8_t>(v);
            } else {
                out[i * 8 + j] = 0;
                ++count;
            }
        }
    r.n_outside = count;
    r.bad = count != 0;
    int s1 = 0, s2 = 0;
    for (int k = 0; k < 64; k++) {
        s1 += out[k];
        s2 += out[k] * out[k];
    }
    r.sum = s1;
    r.sum_sq = s2;
}

// C entry points (ptamo_*): same structs and argument meaning as include/ptam_hip.h
// ================================================================================================
extern "C" {

struct ptamo_ctx {
    OCtx c;
};
struct ptamo_kf {
    KeyFrame kf;
    int w = 0, h = 0;
};
struct ptamo_ba {
    Bundle b;
    ptamo_ba(const ptam_cam_params& cam, const ptam_ba_opts& o) : b(cam, o) {}
};

const char* ptamo_last_error(void) { return ""; }
int ptamo_ctx_create(const ptam_cam_params* cam, int /*device*/, ptamo_ctx** out) {
    if (!cam || !out) return PTAM_E_ARG;
    *out = new ptamo_ctx();
    (*out)->c.cam = *cam;
    return PTAM_OK;
}
int ptamo_ctx_destroy(ptamo_ctx* c) {
    delete c;
    return PTAM_OK;
}
int ptamo_ctx_set_halfsample(ptamo_ctx* c, int v) {
    c->c.variant = v;
    return PTAM_OK;
}
// same signature as the product's ptam_ctx_cache_hazards (include/ptam_hip.h)
int ptamo_ctx_cache_hazards(ptamo_ctx* c, long long* out) {
    *out = c->c.cache_hazards;
    return PTAM_OK;
}
int ptamo_ctx_camera_constants(ptamo_ctx* c, double out[8]) {
    ATANCamera cam(c->c.cam);
    out[0] = cam.focal[0];
    out[1] = cam.focal[1];
    out[2] = cam.centre[0];
    out[3] = cam.centre[1];
    out[4] = cam.two_tan;
    out[5] = cam.w_inv;
    out[6] = cam.largest_radius;
    out[7] = cam.max_r;
    return PTAM_OK;
}

int ptamo_half_sample(const uint8_t* in, int w, int h, uint8_t* out, int variant) {
    half_sample(in, w, h, out, variant);
    return PTAM_OK;
}
int ptamo_fast10(const uint8_t* im, int w, int h, int thr, ptam_int2* out, int cap) {
    std::vector<ptam_int2> v;
    fast10(im, w, h, thr, v);
    const int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = v[i];
    return n;
}

int ptamo_ctx_sync(ptamo_ctx*) { return PTAM_OK; }
int ptamo_kf_create(ptamo_ctx*, int w, int h, ptamo_kf** out) {
    *out = new ptamo_kf();
    (*out)->w = w;
    (*out)->h = h;
    return PTAM_OK;
}
int ptamo_kf_destroy(ptamo_kf* k) {
    delete k;
    return PTAM_OK;
}
int ptamo_make_keyframe_lite(ptamo_ctx* c, ptamo_kf* k, const uint8_t* im, int stride) {
    k->kf.MakeKeyFrame_Lite(im, k->w, k->h, stride, c->c.variant);
    return PTAM_OK;
}
int ptamo_kf_level_info(ptamo_ctx*, const ptamo_kf* k, int l, int* w, int* h, int* n) {
    if (l < 0 || l >= PTAM_LEVELS) return PTAM_E_ARG;
    if (w) *w = k->kf.lev[l].w;
    if (h) *h = k->kf.lev[l].h;
    if (n) *n = (int)k->kf.lev[l].corners.size();
    return PTAM_OK;
}
int ptamo_kf_read_level(ptamo_ctx*, const ptamo_kf* k, int l, uint8_t* px, ptam_int2* corners, int32_t* lut) {
    if (l < 0 || l >= PTAM_LEVELS) return PTAM_E_ARG;
    const Level& L = k->kf.lev[l];
    if (px) std::memcpy(px, L.im.data(), L.im.size());
    if (corners) std::memcpy(corners, L.corners.data(), L.corners.size() * sizeof(ptam_int2));
    if (lut) std::memcpy(lut, L.rowlut.data(), L.rowlut.size() * sizeof(int));
    return PTAM_OK;
}

int ptamo_make_keyframe_rest(ptamo_ctx*, ptamo_kf* k) {
    k->kf.MakeKeyFrame_Rest();
    return PTAM_OK;
}
int ptamo_kf_rest_info(ptamo_ctx*, const ptamo_kf* k, int l, int* n) {
    if (l < 0 || l >= PTAM_LEVELS) return PTAM_E_ARG;
    *n = (int)k->kf.max_corners[l].size();
    return PTAM_OK;
}
int ptamo_kf_read_rest(ptamo_ctx*, const ptamo_kf* k, int l, ptam_int2* mc, double* st) {
    if (l < 0 || l >= PTAM_LEVELS) return PTAM_E_ARG;
    if (mc) std::memcpy(mc, k->kf.max_corners[l].data(), k->kf.max_corners[l].size() * sizeof(ptam_int2));
    if (st) std::memcpy(st, k->kf.st_scores[l].data(), k->kf.st_scores[l].size() * sizeof(double));
    return PTAM_OK;
}

int ptamo_find_patch_coarse_batch(ptamo_ctx*, const ptamo_kf* k, int n, const ptam_patch_query* q,
                                  const uint8_t* tmpl, ptam_patch_result* res) {
    for (int i = 0; i < n; i++) find_patch_coarse(k->kf, q[i], tmpl + (size_t)i * 64, res[i]);
    return PTAM_OK;
}
int ptamo_ctx_one_pixel_dist(ptamo_ctx* c, double* out) {
    *out = ATANCamera(c->c.cam).OnePixelDist();
    return PTAM_OK;
}
int ptamo_kf_implane_corners(ptamo_ctx* c, ptamo_kf* k, int level, double* out_xy, int cap, int* n_out) {
    if (!c || !k || level < 0 || level >= PTAM_LEVELS) return PTAM_E_ARG;
    std::vector<double> ip;
    implane_corners(ATANCamera(c->c.cam), k->kf.lev[level], level, ip);
    const int nc = (int)(ip.size() / 2);
    if (n_out) *n_out = nc;
    if (out_xy) {
        if (cap < nc) return PTAM_E_ARG;
        std::memcpy(out_xy, ip.data(), ip.size() * 8);
    }
    return PTAM_OK;
}
int ptamo_epipolar_search_batch(ptamo_ctx* c, const ptamo_kf* src, ptamo_kf* tgt, int level, int n, const ptam_epipolar_query* q,
                                ptam_epipolar_result* res) {
    if (!c || !src || !tgt || level < 0 || level >= PTAM_LEVELS) return PTAM_E_ARG;
    std::vector<double> ip;
    implane_corners(ATANCamera(c->c.cam), tgt->kf.lev[level], level, ip);
    for (int i = 0; i < n; i++) epipolar_search(src->kf.lev[level], tgt->kf.lev[level], ip, q[i], res[i]);
    return PTAM_OK;
}
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_make_templates_batch(ptamo_ctx*, int n, const ptam_template_query* q, uint8_t* tmpl, ptam_template_result* res) {
    for (int i = 0; i < n; i++) {
        ptam_template_result& r = res[i];
        std::memset(&r, 0, sizeof r);
        uint8_t* out = tmpl + (size_t)i * 64;
        if (q[i].search_level < 0) {
            r.bad = 1;
            std::memset(out, 0, 64);
            continue;
        }
        const ptamo_kf* k = reinterpret_cast<const ptamo_kf*>(q[i].src_kf);
        if (!k || q[i].src_level < 0 || q[i].src_level >= PTAM_LEVELS || q[i].search_level >= PTAM_LEVELS) return PTAM_E_ARG;
        make_template_coarse_cont(k->kf.lev[q[i].src_level], q[i].center_x, q[i].center_y, q[i].search_level, q[i].warp_inverse, out, r);
    }
    return PTAM_OK;
}
#endif
int ptamo_zmssd_at_points(ptamo_ctx*, const ptamo_kf* k, int level, int n, const ptam_int2* pts,
                          const uint8_t* tmpl, int32_t* out) {
    int s, ss;
    template_sums(tmpl, s, ss);
    for (int i = 0; i < n; i++) out[i] = zmssd_at_point(k->kf.lev[level], pts[i].x, pts[i].y, tmpl, s, ss, PTAM_MAX_SSD);
    return PTAM_OK;
}

#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_project_points(ptamo_ctx* c, int n, const double* world, const double pose[12], ptam_projection* out) {
    ATANCamera cam(c->c.cam);
    const SE3 T = se3_from12(pose);
    for (int i = 0; i < n; i++) {
        TrackerData td;
        std::memcpy(td.world, world + 3 * i, sizeof td.world);
        td.v2Image[0] = td.v2Image[1] = 0;
        const bool reached = td.Project(T, cam);
        std::memset(&out[i], 0, sizeof out[i]);
        std::memcpy(out[i].cam, td.v3Cam, sizeof td.v3Cam);
        if (reached) {
            std::memcpy(out[i].image, td.v2Image, sizeof td.v2Image);
            cam.GetProjectionDerivs(out[i].derivs);
        }
        out[i].in_image = td.bInImage;
    }
    return PTAM_OK;
}
#endif

int ptamo_subpix_batch(ptamo_ctx*, const ptamo_kf* k, int n, const ptam_subpix_query* q, const uint8_t* tmpl,
                       ptam_subpix_result* res) {
    for (int i = 0; i < n; i++) subpix_refine(k->kf, q[i], tmpl + (size_t)i * 64, res[i]);
    return PTAM_OK;
}

// Tracker::TrackMap PVS loop src/Tracker.cc:453-478 + PatchFinder::CalcSearchLevelAndWarpMatrix
// src/PatchFinder.cc:52-84
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_track_pvs(ptamo_ctx* c, int n, const ptam_pvs_point* pts, const double pose[12], ptam_pvs_result* out,
                    int32_t counts[4]) {
    ATANCamera cam(c->c.cam);
    const SE3 T = se3_from12(pose);
    if (counts) counts[0] = counts[1] = counts[2] = counts[3] = 0;
    for (int i = 0; i < n; i++) {
        std::memset(&out[i], 0, sizeof out[i]);
        out[i].level = -1;
        TrackerData td;
        std::memcpy(td.world, pts[i].world, sizeof td.world);
        td.v2Image[0] = td.v2Image[1] = 0;
        const bool reached = td.Project(T, cam);
        std::memcpy(out[i].proj.cam, td.v3Cam, sizeof td.v3Cam);
        if (reached) {
            std::memcpy(out[i].proj.image, td.v2Image, sizeof td.v2Image);
            cam.GetProjectionDerivs(out[i].proj.derivs);
        }
        out[i].proj.in_image = td.bInImage;
        if (!td.bInImage) continue;
        const double* D = out[i].proj.derivs;
        double v3Cam[3];
        se3_apply(T, pts[i].world, v3Cam);
        const double dOneOverCameraZ = 1.0 / v3Cam[2];
        double mr[3], md[3];
        for (int r = 0; r < 3; r++) {
            mr[r] = T.R[r * 3] * pts[i].pixel_right_w[0] + T.R[r * 3 + 1] * pts[i].pixel_right_w[1] + T.R[r * 3 + 2] * pts[i].pixel_right_w[2];
            md[r] = T.R[r * 3] * pts[i].pixel_down_w[0] + T.R[r * 3 + 1] * pts[i].pixel_down_w[1] + T.R[r * 3 + 2] * pts[i].pixel_down_w[2];
        }
        double W[4];   // mm2WarpInverse row-major; .T()[0] = column 0
        {
            const double ax = (mr[0] - v3Cam[0] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double ay = (mr[1] - v3Cam[1] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[0] = D[0] * ax + D[1] * ay;
            W[2] = D[2] * ax + D[3] * ay;
            const double bx = (md[0] - v3Cam[0] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double by = (md[1] - v3Cam[1] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[1] = D[0] * bx + D[1] * by;
            W[3] = D[2] * bx + D[3] * by;
        }
        std::memcpy(out[i].warp_inverse, W, sizeof W);
        double dDet = W[0] * W[3] - W[1] * W[2];
        int level = 0;
        while (dDet > 3 && level < PTAM_LEVELS - 1) {
            level++;
            dDet *= 0.25;
        }
        if (dDet > 3 || dDet < 0.25)
            out[i].level = -1;
        else {
            out[i].level = level;
            if (counts) counts[level]++;
        }
    }
    return PTAM_OK;
}
#endif

// MapMaker::ReFind_Common src/MapMaker.cc:943-1020 for a batch of map points against ONE keyframe (what
// ReFindInSingleKeyFrame :1027-1042 loops over).  Per point, statement by statement; the set bookkeeping (sMeasurementKFs,
// sNeverRetryKFs) is the caller's: `never_retry` says the point went into sNeverRetryKFs.  The static PatchFinder of the
// reference keeps its last template between calls (:100-111); a batch never repeats a point, so every template is made anew.
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_refind_batch(ptamo_ctx* c, const ptamo_kf* k, const double kf_pose[12], int n, const ptam_pvs_point* pts,
                       const ptam_template_query* src, ptam_refind_result* out) {
    ATANCamera cam(c->c.cam);
    const SE3 T = se3_from12(kf_pose);
    for (int i = 0; i < n; i++) {
        ptam_refind_result& r = out[i];
        std::memset(&r, 0, sizeof r);
        r.level = -1;
        r.never_retry = 1;
        double v3Cam[3];
        se3_apply(T, pts[i].world, v3Cam);                                        // :950
        if (v3Cam[2] < 0.001) continue;                                            // :951-955
        const double v2ImPlane[2] = {v3Cam[0] / v3Cam[2], v3Cam[1] / v3Cam[2]};
        if (v2ImPlane[0] * v2ImPlane[0] + v2ImPlane[1] * v2ImPlane[1] > cam.largest_radius * cam.largest_radius) continue;   // :957-961
        double v2Image[2];
        cam.Project(v2ImPlane, v2Image);                                           // :963
        if (cam.invalid) continue;                                                 // :964-968
        if (v2Image[0] < 0 || v2Image[1] < 0 || v2Image[0] > cam.size[0] || v2Image[1] > cam.size[1]) continue;   // :970-975
        double D[4];
        cam.GetProjectionDerivs(D);                                                // :978
        // Finder.CalcSearchLevelAndWarpMatrix(p, k.se3CfromW, m2CamDerivs)  :979 — its return value is NOT looked at: a warp
        // it calls inappropriate (-1) still gets its template made at the level the loop stopped at (src/PatchFinder.cc:52-84)
        const double dOneOverCameraZ = 1.0 / v3Cam[2];
        double mr[3], md[3];
        for (int q = 0; q < 3; q++) {
            mr[q] = T.R[q * 3] * pts[i].pixel_right_w[0] + T.R[q * 3 + 1] * pts[i].pixel_right_w[1] + T.R[q * 3 + 2] * pts[i].pixel_right_w[2];
            md[q] = T.R[q * 3] * pts[i].pixel_down_w[0] + T.R[q * 3 + 1] * pts[i].pixel_down_w[1] + T.R[q * 3 + 2] * pts[i].pixel_down_w[2];
        }
        double W[4];
        {
            const double ax = (mr[0] - v3Cam[0] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double ay = (mr[1] - v3Cam[1] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[0] = D[0] * ax + D[1] * ay;
            W[2] = D[2] * ax + D[3] * ay;
            const double bx = (md[0] - v3Cam[0] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double by = (md[1] - v3Cam[1] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[1] = D[0] * bx + D[1] * by;
            W[3] = D[2] * bx + D[3] * by;
        }
        double dDet = W[0] * W[3] - W[1] * W[2];
        int level = 0;
        while (dDet > 3 && level < PTAM_LEVELS - 1) {
            level++;
            dDet *= 0.25;
        }
        // Finder.MakeTemplateCoarseCont(p)  :980 ; mbTemplateBad = (bool)nOutside overwrites the verdict above
        const ptamo_kf* sk = reinterpret_cast<const ptamo_kf*>(src[i].src_kf);
        if (!sk || src[i].src_level < 0 || src[i].src_level >= PTAM_LEVELS) return PTAM_E_ARG;
        uint8_t tmpl[64];
        ptam_template_result tr;
        std::memset(&tr, 0, sizeof tr);
        make_template_coarse_cont(sk->kf.lev[src[i].src_level], src[i].center_x, src[i].center_y, level, W, tmpl, tr);
        r.level = level;
        if (tr.bad) continue;                                                      // :982-986
        ptam_patch_query q;
        q.x = (int)v2Image[0];                                                     // ir(v2Image) :988
        q.y = (int)v2Image[1];
        q.level = level;
        q.range = 4;
        ptam_patch_result pr;
        find_patch_coarse(k->kf, q, tmpl, pr);
        if (!pr.found) continue;                                                   // :989-993
        r.never_retry = 0;
        r.found = 1;
        if (level > 0) {                                                           // :1000-1006 (convergence is not looked at)
            ptam_subpix_query sq;
            sq.coarse_pos[0] = pr.pos[0];
            sq.coarse_pos[1] = pr.pos[1];
            sq.level = level;
            sq.max_its = 8;
            ptam_subpix_result sr;
            subpix_refine(k->kf, sq, tmpl, sr);
            r.root_pos[0] = sr.pos[0];
            r.root_pos[1] = sr.pos[1];
            r.sub_pix = 1;
        } else {                                                                   // :1007-1011
            r.root_pos[0] = pr.pos[0];
            r.root_pos[1] = pr.pos[1];
            r.sub_pix = 0;
        }
    }
    return PTAM_OK;
}
#endif

// The same through the reference's ONE finder (`static PatchFinder Finder`, src/MapMaker.cc:977), pair after pair in the
// caller's order — ReFindNewlyMade :1046-1066, ReFindFromFailureQueue :1070-1082: the finder's state that outlives a call.
struct ptamo_refinder {
    bool has_last = false;         // mpLastTemplateMapPoint != NULL
    long long last_point = 0;      // mpLastTemplateMapPoint
    double last_m2[4] = {0, 0, 0, 0};   // mm2LastWarpMatrix {m00, m01, m10, m11}
    uint8_t tmpl[64] = {0};        // mimTemplate
    bool bad = false;              // mbTemplateBad
};
int ptamo_refinder_create(ptamo_ctx*, ptamo_refinder** out) {
    if (!out) return PTAM_E_ARG;
    *out = new ptamo_refinder();
    return PTAM_OK;
}
int ptamo_refinder_destroy(ptamo_refinder* f) {
    delete f;
    return PTAM_OK;
}
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_refind_pairs(ptamo_ctx* c, ptamo_refinder* F, int n, const ptam_refind_pair* pairs, ptam_refind_result* out, int32_t* kept) {
    if (!c || !F || n < 0 || (n > 0 && (!pairs || !out))) return PTAM_E_ARG;
    ATANCamera cam(c->c.cam);
    for (int i = 0; i < n; i++) {
        const ptam_refind_pair& pr_ = pairs[i];
        ptam_refind_result& r = out[i];
        std::memset(&r, 0, sizeof r);
        r.level = -1;
        if (kept) kept[i] = 0;
        if (pr_.skip) continue;                                                    // :947-948 (neither found nor newly "never retry")
        r.never_retry = 1;
        const ptamo_kf* k = reinterpret_cast<const ptamo_kf*>(pr_.kf);
        if (!k) return PTAM_E_ARG;
        const SE3 T = se3_from12(pr_.kf_pose);
        const ptam_pvs_point& p = pr_.point;
        double v3Cam[3];
        se3_apply(T, p.world, v3Cam);                                              // :950
        if (v3Cam[2] < 0.001) continue;                                            // :951-955
        const double v2ImPlane[2] = {v3Cam[0] / v3Cam[2], v3Cam[1] / v3Cam[2]};
        if (v2ImPlane[0] * v2ImPlane[0] + v2ImPlane[1] * v2ImPlane[1] > cam.largest_radius * cam.largest_radius) continue;   // :957-961
        double v2Image[2];
        cam.Project(v2ImPlane, v2Image);                                           // :963
        if (cam.invalid) continue;                                                 // :964-968
        if (v2Image[0] < 0 || v2Image[1] < 0 || v2Image[0] > cam.size[0] || v2Image[1] > cam.size[1]) continue;   // :970-975
        double D[4];
        cam.GetProjectionDerivs(D);                                                // :978
        // Finder.CalcSearchLevelAndWarpMatrix(p, k.se3CfromW, m2CamDerivs)  :979, src/PatchFinder.cc:52-84
        const double dOneOverCameraZ = 1.0 / v3Cam[2];
        double mr[3], md[3];
        for (int q = 0; q < 3; q++) {
            mr[q] = T.R[q * 3] * p.pixel_right_w[0] + T.R[q * 3 + 1] * p.pixel_right_w[1] + T.R[q * 3 + 2] * p.pixel_right_w[2];
            md[q] = T.R[q * 3] * p.pixel_down_w[0] + T.R[q * 3 + 1] * p.pixel_down_w[1] + T.R[q * 3 + 2] * p.pixel_down_w[2];
        }
        double W[4];
        {
            const double ax = (mr[0] - v3Cam[0] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double ay = (mr[1] - v3Cam[1] * mr[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[0] = D[0] * ax + D[1] * ay;
            W[2] = D[2] * ax + D[3] * ay;
            const double bx = (md[0] - v3Cam[0] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            const double by = (md[1] - v3Cam[1] * md[2] * dOneOverCameraZ) * dOneOverCameraZ;
            W[1] = D[0] * bx + D[1] * by;
            W[3] = D[2] * bx + D[3] * by;
        }
        double dDet = W[0] * W[3] - W[1] * W[2];
        int level = 0;
        while (dDet > 3 && level < PTAM_LEVELS - 1) {
            level++;
            dDet *= 0.25;
        }
        if (dDet > 3 || dDet < 0.25) F->bad = true;                                // src/PatchFinder.cc:78-81 (the -1 itself is ignored)
        // Finder.MakeTemplateCoarseCont(p)  :980, src/PatchFinder.cc:98-127
        const double det = W[0] * W[3] - W[2] * W[1];
        const double inv = 1.0 / det;
        const double sc = (double)(1 << level);
        const double m2[4] = {W[3] * inv * sc, -W[1] * inv * sc, -W[2] * inv * sc, W[0] * inv * sc};   // {m00, m01, m10, m11}
        bool need = !F->has_last || F->last_point != (long long)pr_.point_id;       // :103
        for (int col = 0; !need && col < 2; col++) {                               // :105-110: columns m2.T()[col]
            const double dx = m2[col] - F->last_m2[col], dy = m2[2 + col] - F->last_m2[2 + col];
            if (dx * dx + dy * dy > 0.07 * 0.07) need = true;
        }
        if (need) {
            const ptamo_kf* sk = reinterpret_cast<const ptamo_kf*>(pr_.source.src_kf);
            if (!sk || pr_.source.src_level < 0 || pr_.source.src_level >= PTAM_LEVELS) return PTAM_E_ARG;
            ptam_template_result tr;
            std::memset(&tr, 0, sizeof tr);
            make_template_coarse_cont(sk->kf.lev[pr_.source.src_level], pr_.source.center_x, pr_.source.center_y, level, W, F->tmpl, tr);
            F->bad = tr.bad != 0;                                                  // :118
            F->has_last = true;                                                    // :121-122
            F->last_point = (long long)pr_.point_id;
            for (int q = 0; q < 4; q++) F->last_m2[q] = m2[q];
        } else if (kept)
            kept[i] = 1;
        r.level = level;
        if (F->bad) continue;                                                      // :982-986
        ptam_patch_query q;
        q.x = (int)v2Image[0];                                                     // ir(v2Image) :988
        q.y = (int)v2Image[1];
        q.level = level;
        q.range = 4;
        ptam_patch_result pr;
        find_patch_coarse(k->kf, q, F->tmpl, pr);
        if (!pr.found) continue;                                                   // :989-993
        r.never_retry = 0;
        r.found = 1;
        if (level > 0) {                                                           // :1000-1006 (convergence is not looked at)
            ptam_subpix_query sq;
            sq.coarse_pos[0] = pr.pos[0];
            sq.coarse_pos[1] = pr.pos[1];
            sq.level = level;
            sq.max_its = 8;
            ptam_subpix_result sr;
            subpix_refine(k->kf, sq, F->tmpl, sr);
            r.root_pos[0] = sr.pos[0];
            r.root_pos[1] = sr.pos[1];
            r.sub_pix = 1;
        } else {                                                                   // :1007-1011
            r.root_pos[0] = pr.pos[0];
            r.root_pos[1] = pr.pos[1];
            r.sub_pix = 0;
        }
    }
    return PTAM_OK;
}
#endif

void ptamo_gn_opts_default(ptam_gn_opts* o) {
    o->iterations = 10;
    o->nonlinear_mask = 0x211;
    o->override_after = 5;
    o->override_sigma_sq = 16.0;
    o->mark_outliers_iter = 9;
    o->estimator = PTAM_EST_TUKEY;
    o->prior = 100.0;
}

// src/Tracker.cc:613-643 (fine stage) / :552-568 (coarse stage) driven by opts
static int pose_gn_impl(ptamo_ctx* c, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                        double pose[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out,
                        ptam_projection* state_out) {
    ATANCamera cam(c->c.cam);
    SE3 T = se3_from12(pose);
    std::vector<TrackerData> vTD(n);
    for (int i = 0; i < n; i++) {
        TrackerData& td = vTD[i];
        std::memcpy(td.world, meas[i].world, sizeof td.world);
        std::memcpy(td.v2Found, meas[i].found, sizeof td.v2Found);
        td.dSqrtInvNoise = meas[i].sqrt_inv_noise;
        td.bFound = true;
        if (entry) {
            std::memcpy(td.v3Cam, entry[i].cam, sizeof td.v3Cam);
            std::memcpy(td.v2Image, entry[i].image, sizeof td.v2Image);
            std::memcpy(td.m2CamDerivs, entry[i].derivs, sizeof td.m2CamDerivs);
        } else {
            // TrackMap's PVS projection (src/Tracker.cc:454-462): points not in the image never
            // enter the set
            td.Project(T, cam);
            if (!td.bInImage)
                td.bFound = false;
            else
                cam.GetProjectionDerivs(td.m2CamDerivs);
        }
        if (outlier_flags) outlier_flags[i] = 0;
    }
    double last[6] = {0, 0, 0, 0, 0, 0};
    for (int iter = 0; iter < opts->iterations; iter++) {
        const bool nonlinear = (opts->nonlinear_mask >> iter) & 1u;
        if (iter != 0) {
            if (nonlinear) {
                for (auto& td : vTD)
                    if (td.bFound) {   // ProjectAndDerivs include/Tracker.h:89-94
                        const bool reached = td.Project(T, cam);
                        if (reached)
                            cam.GetProjectionDerivs(td.m2CamDerivs);
                        else
                            c->c.cache_hazards++;   // reference would read another point's cache;
                                                    // oracle and product both keep the old derivs
                    }
            } else {
                for (auto& td : vTD)
                    if (td.bFound) td.LinearUpdate(last);
            }
        }
        if (nonlinear)
            for (auto& td : vTD)
                if (td.bFound) td.CalcJacobian();
        const double ov = iter > opts->override_after ? opts->override_sigma_sq : 0.0;
        double mu[6];
        calc_pose_update(vTD, ov, opts->estimator, opts->prior, iter == opts->mark_outliers_iter, outlier_flags, mu);
        T = se3_mul(se3_exp(mu), T);
        std::memcpy(last, mu, sizeof last);
        if (updates_out) std::memcpy(updates_out + 6 * iter, mu, sizeof mu);
    }
    se3_to12(T, pose);
    if (state_out)   // the TrackerData a following loop starts from (its iteration 0 does not re-project, src/Tracker.cc:617)
        for (int i = 0; i < n; i++) {
            std::memset(&state_out[i], 0, sizeof state_out[i]);
            std::memcpy(state_out[i].cam, vTD[i].v3Cam, sizeof vTD[i].v3Cam);
            std::memcpy(state_out[i].image, vTD[i].v2Image, sizeof vTD[i].v2Image);
            std::memcpy(state_out[i].derivs, vTD[i].m2CamDerivs, sizeof vTD[i].m2CamDerivs);
        }
    return PTAM_OK;
}
int ptamo_pose_gn(ptamo_ctx* c, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                  double pose[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out) {
    return pose_gn_impl(c, n, meas, entry, pose, opts, outlier_flags, updates_out, nullptr);
}
int ptamo_pose_gn_state(ptamo_ctx* c, int n, const ptam_pose_meas* meas, const ptam_projection* entry,
                        double pose[12], const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out,
                        ptam_projection* state_out) {
    return pose_gn_impl(c, n, meas, entry, pose, opts, outlier_flags, updates_out, state_out);
}

// TrackerData::Project (include/Tracker.h:70-85) on EXISTING TrackerData: v3Cam always, v2Image only when the camera model
// is reached, m2CamDerivs never (ProjectAndDerivs refreshes them only if bFound, :89-94 — false for every point TrackMap
// re-projects before a search, src/Tracker.cc:470,573,607)
int ptamo_reproject_points(ptamo_ctx* c, int n, const double* world, const double pose[12], ptam_projection* inout) {
    ATANCamera cam(c->c.cam);
    const SE3 T = se3_from12(pose);
    for (int i = 0; i < n; i++) {
        TrackerData td;
        std::memcpy(td.world, world + 3 * i, sizeof td.world);
        std::memcpy(td.v2Image, inout[i].image, sizeof td.v2Image);
        td.Project(T, cam);
        std::memcpy(inout[i].cam, td.v3Cam, sizeof td.v3Cam);
        std::memcpy(inout[i].image, td.v2Image, sizeof td.v2Image);
        inout[i].in_image = td.bInImage;
    }
    return PTAM_OK;
}

// ---- Tracker::TrackMap src/Tracker.cc:442-696, the whole frame, statement by statement — the twin of the product's resident
// chain (ptam_tracker_* / ptam_track_map) behind the same entry points, so that tests and bench.py's cpu_baseline drive the
// SAME chain through either library.  The map's TrackerData (include/Tracker.h:42-67) live in the tracker object: what
// outlives a frame is every point's PatchFinder (template, sums, mbTemplateBad, last warp — src/PatchFinder.cc:98-127).
// std::random_shuffle (:483-484, :597-600) is replaced, as in the product's interface, by two caller-provided permutations.
struct ptamo_tracker {
    ptamo_ctx* ctx = nullptr;
    int cap = 0, n = 0;
    std::vector<ptam_pvs_point> pts;
    std::vector<ptam_template_query> src;
    std::vector<int> sh_levels, sh_fine;
    struct Finder {
        bool valid = false, bad = false;
        double m2[4] = {0, 0, 0, 0};
        uint8_t tmpl[64] = {0};
    };
    std::vector<Finder> finder;
    std::vector<ptam_trackmap_meas> iteration_set;
};
namespace {
struct TmTD {   // TrackerData of one map point, the fields TrackMap touches
    int idx = 0, level = 0, searched_level = 0;
    double wi[4] = {0, 0, 0, 0};
    ptam_projection st;
    bool found = false, did_subpix = false;
    double v2_found[2] = {0, 0};
};
}   // namespace
void ptamo_trackmap_opts_default(ptam_trackmap_opts* o) {
    if (!o) return;
    o->try_coarse = 1;
    o->coarse_min = 20;
    o->coarse_max = 60;
    o->coarse_range = 30;
    o->coarse_subpix_its = 8;
    o->max_patches = 1000;
    o->estimator = PTAM_EST_TUKEY;
    o->pad_ = 0;
}
int ptamo_tracker_create(ptamo_ctx* c, int max_points, ptamo_tracker** out) {
    if (!c || !out || max_points < 1) return PTAM_E_ARG;
    ptamo_tracker* t = new ptamo_tracker();
    t->ctx = c;
    t->cap = max_points;
    *out = t;
    return PTAM_OK;
}
int ptamo_tracker_destroy(ptamo_tracker* t) {
    delete t;
    return PTAM_OK;
}
int ptamo_tracker_set_map(ptamo_tracker* t, int n, const ptam_pvs_point* pts, const ptam_template_query* src) {
    if (!t || n < 0 || n > t->cap || (n > 0 && (!pts || !src))) return PTAM_E_ARG;
    t->pts.assign(pts, pts + n);
    t->src.assign(src, src + n);
    t->finder.assign((size_t)n, ptamo_tracker::Finder());   // new TrackerData, new PatchFinders
    if (n != t->n) {
        t->sh_levels.resize((size_t)n);
        t->sh_fine.resize((size_t)n);
        for (int i = 0; i < n; i++) t->sh_levels[(size_t)i] = t->sh_fine[(size_t)i] = i;
    }
    t->n = n;
    return PTAM_OK;
}
// the same with the TrackerData of the points that persist (include/Tracker.h:42-67: TrackerData — and its PatchFinder — is a
// member of the MapPoint's tracking data and lives as long as the point; the map's vector changing around it does not touch it)
int ptamo_tracker_update_map(ptamo_tracker* t, int n, const ptam_pvs_point* pts, const ptam_template_query* src, const int32_t* prev) {
    if (!t || n < 0 || n > t->cap || (n > 0 && (!pts || !src || !prev))) return PTAM_E_ARG;
    std::vector<ptamo_tracker::Finder> old = t->finder;
    const int n_old = t->n;
    std::vector<char> seen((size_t)std::max(n_old, 1), 0);
    for (int i = 0; i < n; i++) {   // (an old point is one new point at most)
        if (prev[i] < -1 || prev[i] >= n_old) return PTAM_E_ARG;
        if (prev[i] >= 0) {
            if (seen[(size_t)prev[i]]) return PTAM_E_ARG;
            seen[(size_t)prev[i]] = 1;
        }
    }
    const int rc = ptamo_tracker_set_map(t, n, pts, src);
    if (rc) return rc;
    for (int i = 0; i < n; i++)
        if (prev[i] >= 0) t->finder[(size_t)i] = old[(size_t)prev[i]];
    return PTAM_OK;
}
int ptamo_tracker_set_shuffle(ptamo_tracker* t, const int32_t* a, const int32_t* b) {
    if (!t || !a || !b) return PTAM_E_ARG;
    t->sh_levels.assign(a, a + t->n);
    t->sh_fine.assign(b, b + t->n);
    return PTAM_OK;
}
static int pose_gn_impl(ptamo_ctx* c, int n, const ptam_pose_meas* meas, const ptam_projection* entry, double pose[12],
                        const ptam_gn_opts* opts, int32_t* outlier_flags, double* updates_out, ptam_projection* state_out);
// Tracker::SearchForPoints src/Tracker.cc:867-912
static int tm_search_for_points(ptamo_tracker* t, const ptamo_kf* kf, std::vector<TmTD*>& set, unsigned range, int subpix_its,
                                int attempted[4], int found_cnt[4], int* n_kept) {
    int n_found = 0;
    for (TmTD* td : set) {
        ptamo_tracker::Finder& F = t->finder[(size_t)td->idx];
        const ptam_template_query& sq_ = t->src[(size_t)td->idx];
        // Finder.MakeTemplateCoarseCont(TD.Point)  :873, src/PatchFinder.cc:98-127
        const double* W = td->wi;
        const double det = W[0] * W[3] - W[2] * W[1];
        const double inv = 1.0 / det;
        const double sc = (double)(1 << td->level);
        const double m2[4] = {W[3] * inv * sc, -W[1] * inv * sc, -W[2] * inv * sc, W[0] * inv * sc};   // {m00, m01, m10, m11}
        bool need = !F.valid;
        for (int col = 0; !need && col < 2; col++) {
            const double dx = m2[col] - F.m2[col], dy = m2[2 + col] - F.m2[2 + col];
            if (dx * dx + dy * dy > 0.07 * 0.07) need = true;
        }
        if (need) {
            const ptamo_kf* sk = reinterpret_cast<const ptamo_kf*>(sq_.src_kf);
            ptam_template_result tr;
            std::memset(&tr, 0, sizeof tr);
            make_template_coarse_cont(sk->kf.lev[sq_.src_level], sq_.center_x, sq_.center_y, td->level, W, F.tmpl, tr);
            F.bad = tr.bad != 0;
            F.valid = true;
            for (int q = 0; q < 4; q++) F.m2[q] = m2[q];
        } else
            ++*n_kept;
        td->searched_level = F.bad ? -1 : td->level;
        if (F.bad) {                                                   // :874-878
            td->st.in_image = 0;
            td->found = false;
            continue;
        }
        attempted[td->level]++;                                        // :880
        ptam_patch_query q;
        q.x = (int)td->st.image[0];                                    // ir(TD.v2Image) :882
        q.y = (int)td->st.image[1];
        q.level = td->level;
        q.range = range;
        ptam_patch_result pr;
        find_patch_coarse(kf->kf, q, F.tmpl, pr);
        if (!pr.found) {                                               // :884-887
            td->found = false;
            continue;
        }
        td->found = true;
        found_cnt[td->level]++;
        n_found++;
        if (subpix_its > 0) {                                          // :896-906
            td->did_subpix = true;
            ptam_subpix_query sq;
            sq.coarse_pos[0] = pr.pos[0];
            sq.coarse_pos[1] = pr.pos[1];
            sq.level = td->level;
            sq.max_its = subpix_its;
            ptam_subpix_result sr;
            subpix_refine(kf->kf, sq, F.tmpl, sr);
            if (!sr.converged) {
                td->found = false;
                found_cnt[td->level]--;
                n_found--;
                continue;
            }
            td->v2_found[0] = sr.pos[0];
            td->v2_found[1] = sr.pos[1];
        } else {
            td->v2_found[0] = pr.pos[0];
            td->v2_found[1] = pr.pos[1];
            td->did_subpix = false;
        }
    }
    return n_found;
}
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_track_map(ptamo_tracker* t, const ptamo_kf* kf, const double pose_in[12], const ptam_trackmap_opts* opts, ptam_trackmap_result* out) {
    if (!t || !kf || !pose_in || !out) return PTAM_E_ARG;
    ptam_trackmap_opts o;
    if (opts)
        o = *opts;
    else
        ptamo_trackmap_opts_default(&o);
    ptamo_ctx* c = t->ctx;
    ATANCamera cam(c->c.cam);
    const int n = t->n;
    double pose[12];
    std::memcpy(pose, pose_in, sizeof pose);
    int attempted[4] = {0, 0, 0, 0}, found_cnt[4] = {0, 0, 0, 0}, n_kept = 0;
    // ---- PVS loop :453-478 ----
    std::vector<ptam_pvs_result> pvs((size_t)std::max(n, 1));
    ptamo_track_pvs(c, n, t->pts.data(), pose, pvs.data(), nullptr);
    std::vector<TmTD> tds((size_t)n);
    std::vector<char> in_pvs((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (pvs[(size_t)i].level < 0) {
            if (pvs[(size_t)i].proj.in_image) t->finder[(size_t)i].bad = true;   // CalcSearchLevelAndWarpMatrix's -1: mbTemplateBad = true
            continue;
        }
        TmTD& td = tds[(size_t)i];
        td.idx = i;
        td.level = td.searched_level = pvs[(size_t)i].level;
        std::memcpy(td.wi, pvs[(size_t)i].warp_inverse, sizeof td.wi);
        td.st = pvs[(size_t)i].proj;
        in_pvs[(size_t)i] = 1;
    }
    std::vector<TmTD*> av[4];   // avPVS[l], shuffled (:483-484) = in the order of the caller's permutation
    for (int k = 0; k < n; k++) {
        const int i = t->sh_levels[(size_t)k];
        if (i >= 0 && i < n && in_pvs[(size_t)i]) av[tds[(size_t)i].level].push_back(&tds[(size_t)i]);
    }
    int n_pvs[4];
    for (int l = 0; l < 4; l++) n_pvs[l] = (int)av[l].size();
    std::vector<TmTD*> next, iteration_set;
    bool did_coarse = false;
    const size_t cmax = o.coarse_max;
    if (o.try_coarse && av[3].size() + av[2].size() > o.coarse_min) {                  // :519
        if (av[3].size() <= cmax) {                                                     // :523-530
            next = av[3];
            av[3].clear();
        } else {
            next.assign(av[3].begin(), av[3].begin() + (long)cmax);
            av[3].erase(av[3].begin(), av[3].begin() + (long)cmax);
        }
        if (next.size() < cmax) {                                                       // :533-545
            const size_t more = cmax - next.size();
            if (av[2].size() <= more) {
                next = av[2];                                                           // :538 (an assignment in the reference)
                av[2].clear();
            } else {
                next.insert(next.end(), av[2].begin(), av[2].begin() + (long)more);
                av[2].erase(av[2].begin(), av[2].begin() + (long)more);
            }
        }
        const int n_found = tm_search_for_points(t, kf, next, o.coarse_range, o.coarse_subpix_its, attempted, found_cnt, &n_kept);
        iteration_set = next;                                                           // :550
        if ((unsigned)n_found >= o.coarse_min) {                                        // :551
            did_coarse = true;
            std::vector<TmTD*> f;
            for (TmTD* td : iteration_set)
                if (td->found) f.push_back(td);
            std::vector<ptam_pose_meas> meas(f.size());
            std::vector<ptam_projection> entry(f.size()), st(f.size());
            for (size_t k = 0; k < f.size(); k++) {
                std::memcpy(meas[k].world, t->pts[(size_t)f[k]->idx].world, sizeof meas[k].world);
                meas[k].found[0] = f[k]->v2_found[0];
                meas[k].found[1] = f[k]->v2_found[1];
                meas[k].sqrt_inv_noise = 1.0 / (1 << f[k]->level);
                entry[k] = f[k]->st;
            }
            ptam_gn_opts g;
            ptamo_gn_opts_default(&g);
            g.nonlinear_mask = 0x3ff;        // :552-568: ten iterations, every one non-linear
            g.override_sigma_sq = 1.0;       // :565
            g.mark_outliers_iter = -1;
            g.estimator = o.estimator;
            pose_gn_impl(c, (int)f.size(), meas.data(), entry.data(), pose, &g, nullptr, nullptr, st.data());
            for (size_t k = 0; k < f.size(); k++) {   // the TrackerData the loop leaves behind
                std::memcpy(f[k]->st.cam, st[k].cam, sizeof st[k].cam);
                std::memcpy(f[k]->st.image, st[k].image, sizeof st[k].image);
                std::memcpy(f[k]->st.derivs, st[k].derivs, sizeof st[k].derivs);
            }
        }
    }
    const int n_coarse = (int)iteration_set.size();
    const unsigned fine_range = did_coarse ? 5 : 10;                                    // :572
    const SE3 Tcur = se3_from12(pose);
    auto reproject = [&](std::vector<TmTD*>& lst) {   // TrackerData::Project with bFound == false: the derivatives stay
        const SE3 T = se3_from12(pose);
        for (TmTD* td : lst) {
            TrackerData x;
            std::memcpy(x.world, t->pts[(size_t)td->idx].world, sizeof x.world);
            std::memcpy(x.v2Image, td->st.image, sizeof x.v2Image);
            x.Project(T, cam);
            std::memcpy(td->st.cam, x.v3Cam, sizeof x.v3Cam);
            std::memcpy(td->st.image, x.v2Image, sizeof x.v2Image);
            td->st.in_image = x.bInImage;
        }
    };
    (void)Tcur;
    std::vector<TmTD*> top = av[3];                                                     // :574-581
    reproject(top);
    tm_search_for_points(t, kf, top, fine_range, 8, attempted, found_cnt, &n_kept);
    iteration_set.insert(iteration_set.end(), top.begin(), top.end());
    std::vector<TmTD*> fine;                                                            // :586-590
    for (int l = 2; l >= 0; l--) fine.insert(fine.end(), av[l].begin(), av[l].end());
    const int n_use = std::max(0, o.max_patches - (int)iteration_set.size());           // :593-596
    if ((int)fine.size() > n_use) {                                                     // :597-600: shuffle, then chop
        std::vector<char> member((size_t)n, 0);
        for (TmTD* td : fine) member[(size_t)td->idx] = 1;
        std::vector<TmTD*> chopped;
        for (int k = 0; k < n && (int)chopped.size() < n_use; k++) {
            const int i = t->sh_fine[(size_t)k];
            if (i >= 0 && i < n && member[(size_t)i]) chopped.push_back(&tds[(size_t)i]);
        }
        fine.swap(chopped);
    }
    if (did_coarse) reproject(fine);                                                    // :603-605
    tm_search_for_points(t, kf, fine, fine_range, 0, attempted, found_cnt, &n_kept);
    iteration_set.insert(iteration_set.end(), fine.begin(), fine.end());
    // ---- fine pose loop :613-643 ----
    std::vector<TmTD*> f;
    for (TmTD* td : iteration_set)
        if (td->found) f.push_back(td);
    std::vector<int32_t> outl(f.size(), 0);
    double dsum = 0, dsq = 0;
    if (!f.empty()) {
        std::vector<ptam_pose_meas> meas(f.size());
        std::vector<ptam_projection> entry(f.size()), st(f.size());
        for (size_t k = 0; k < f.size(); k++) {
            std::memcpy(meas[k].world, t->pts[(size_t)f[k]->idx].world, sizeof meas[k].world);
            meas[k].found[0] = f[k]->v2_found[0];
            meas[k].found[1] = f[k]->v2_found[1];
            meas[k].sqrt_inv_noise = 1.0 / (1 << f[k]->level);
            entry[k] = f[k]->st;
        }
        ptam_gn_opts g;
        ptamo_gn_opts_default(&g);
        g.estimator = o.estimator;
        pose_gn_impl(c, (int)f.size(), meas.data(), entry.data(), pose, &g, outl.data(), nullptr, st.data());
        for (size_t k = 0; k < f.size(); k++) {                                         // :680-690
            dsum += st[k].cam[2];
            dsq += st[k].cam[2] * st[k].cam[2];
        }
    }
    t->iteration_set.assign(iteration_set.size(), ptam_trackmap_meas());
    size_t kf_ = 0;
    for (size_t s_ = 0; s_ < iteration_set.size(); s_++) {
        const TmTD* td = iteration_set[s_];
        ptam_trackmap_meas& m = t->iteration_set[s_];
        std::memset(&m, 0, sizeof m);
        m.point = td->idx;
        m.level = td->searched_level;
        m.found = td->found;
        m.did_subpix = td->did_subpix;
        if (td->found) {
            m.v2_found[0] = td->v2_found[0];
            m.v2_found[1] = td->v2_found[1];
            m.outlier = outl[kf_++];
        }
    }
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->pose, pose, sizeof pose);
    out->did_coarse = did_coarse;
    for (int l = 0; l < 4; l++) {
        out->n_pvs[l] = n_pvs[l];
        out->attempted[l] = attempted[l];
        out->found[l] = found_cnt[l];
    }
    out->n_coarse = n_coarse;
    out->n_top = (int)top.size();
    out->n_fine = (int)fine.size();
    out->n_meas = (int)f.size();
    out->depth_n = (int)f.size();
    out->templates_reused = n_kept;
    out->depth_sum = dsum;
    out->depth_sum_sq = dsq;
    return PTAM_OK;
}
#endif
// (the product's frame pointer is a device address; here it is the host image, stride == width)
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_track_map_frame(ptamo_tracker* t, ptamo_kf* cur, const uint8_t* frame, const double pose_in[12], const ptam_trackmap_opts* opts,
                          ptam_trackmap_result* out) {
    if (!t || !cur || !frame) return PTAM_E_ARG;
    const int rc = ptamo_make_keyframe_lite(t->ctx, cur, frame, cur->w);
    return rc ? rc : ptamo_track_map(t, cur, pose_in, opts, out);
}
#endif
// ---- the motion model and the tracking branch of Tracker::TrackFrame (src/Tracker.cc:134-137), rotation estimator off ----
void ptamo_motion_reset(ptam_motion_model* m, const double pose[12]) {
    std::memset(m, 0, sizeof *m);
    std::memcpy(m->pose, pose, 96);                  // mse3CamFromWorld
    std::memcpy(m->start_pose, pose, 96);
    m->scene_depth_mean = 1.0;                       // :56
    m->coarse_min_velocity = 0.006;                  // gvdCoarseMinVel :496
    m->use_constant_velocity = 1;                    // gvnConstVel :1041
}
// Tracker::PredictPoseWithMotionModel :1013-1030
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
void ptamo_motion_predict(ptam_motion_model* m) {
    std::memcpy(m->start_pose, m->pose, 96);         // mse3StartPos = mse3CamFromWorld
    double v6Velocity[6];
    std::memcpy(v6Velocity, m->velocity, 48);
    se3_to12(se3_mul(se3_exp(v6Velocity), se3_from12(m->start_pose)), m->pose);   // :1029
}
#endif
// the scene-depth update that closes TrackMap (:678-697), then Tracker::UpdateMotionModel :1036-1056
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
void ptamo_motion_update(ptam_motion_model* m, const ptam_trackmap_result* r) {
    std::memcpy(m->pose, r->pose, 96);
    if (r->depth_n > 20) {
        m->scene_depth_mean = r->depth_sum / r->depth_n;
        m->scene_depth_sigma = std::sqrt((r->depth_sum_sq / r->depth_n) - (m->scene_depth_mean) * (m->scene_depth_mean));
    }
    const SE3 se3NewFromOld = se3_mul(se3_from12(m->pose), se3_inverse(se3_from12(m->start_pose)));
    double v6Motion[6];
    se3_ln(se3NewFromOld, v6Motion);
    if (m->use_constant_velocity) {
        std::memcpy(m->velocity, v6Motion, 48);
    } else {
        double v6OldVel[6];
        std::memcpy(v6OldVel, m->velocity, 48);
        for (int i = 0; i < 6; i++) m->velocity[i] = 0.9 * (0.5 * v6Motion[i] + 0.5 * v6OldVel[i]);
    }
    double v6[6];
    std::memcpy(v6, m->velocity, 48);
    for (int i = 0; i < 3; i++) v6[i] *= 1.0 / m->scene_depth_mean;
    double ss = 0;
    for (int i = 0; i < 6; i++) ss += v6[i] * v6[i];
    m->msd_scaled_velocity = std::sqrt(ss);
}
#endif
void ptamo_se3_ln(const double pose[12], double out6[6]) { se3_ln(se3_from12(pose), out6); }
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_track_frame(ptamo_tracker* t, ptamo_kf* cur, const uint8_t* frame, ptam_motion_model* m, const ptam_trackmap_opts* opts,
                      ptam_trackmap_result* out) {
    if (!t || !cur || !frame || !m || !out) return PTAM_E_ARG;
    ptam_trackmap_opts o;
    if (opts) o = *opts;
    else ptamo_trackmap_opts_default(&o);
    // :94 MakeKeyFrame_Lite, :136 PredictPoseWithMotionModel, :137 TrackMap with the heuristics of :503-514, :138 UpdateMotionModel
    int rc = ptamo_make_keyframe_lite(t->ctx, cur, frame, cur->w);
    if (rc) return rc;
    ptamo_motion_predict(m);
    bool bTryCoarse = true;
    if (m->disable_coarse || m->msd_scaled_velocity < m->coarse_min_velocity || o.coarse_max == 0) bTryCoarse = false;
    if (m->just_recovered) {
        bTryCoarse = true;
        o.coarse_max *= 2;
        o.coarse_range *= 2;
        m->just_recovered = 0;
    }
    o.try_coarse = bTryCoarse ? 1 : 0;
    rc = ptamo_track_map(t, cur, m->pose, &o, out);
    if (rc) return rc;
    ptamo_motion_update(m, out);
    return PTAM_OK;
}
#endif
int ptamo_tracker_read_iteration_set(ptamo_tracker* t, ptam_trackmap_meas* out, int cap, int* n) {
    if (!t || !n) return PTAM_E_ARG;
    *n = (int)t->iteration_set.size();
    if (out)
        for (int i = 0; i < *n && i < cap; i++) out[i] = t->iteration_set[(size_t)i];
    return PTAM_OK;
}

int ptamo_calc_pose_update(ptamo_ctx*, int n, const ptam_pose_update_meas* meas, double override_sigma_sq,
                           int estimator, double prior, double mu_out[6], int32_t* flags) {
    std::vector<TrackerData> vTD(n);
    for (int i = 0; i < n; i++) {
        vTD[i].bFound = true;
        std::memcpy(vTD[i].v2Found, meas[i].found, 16);
        std::memcpy(vTD[i].v2Image, meas[i].image, 16);
        vTD[i].dSqrtInvNoise = meas[i].sqrt_inv_noise;
        std::memcpy(vTD[i].m26Jacobian, meas[i].jac, sizeof meas[i].jac);
        if (flags) flags[i] = 0;
    }
    calc_pose_update(vTD, override_sigma_sq, estimator, prior, flags != nullptr, flags, mu_out);
    return PTAM_OK;
}

int ptamo_se3_exp(const double mu[6], double out12[12]) {
    se3_to12(se3_exp(mu), out12);
    return PTAM_OK;
}
int ptamo_se3_mul(const double a[12], const double b[12], double out[12]) {
    se3_to12(se3_mul(se3_from12(a), se3_from12(b)), out);
    return PTAM_OK;
}
double ptamo_tukey_sigma_sq(const double* e2, int n) {
    std::vector<double> v(e2, e2 + n);
    return Tukey::FindSigmaSquared(v);
}
int ptamo_ldlt_solve(int n, const double* A, const double* b, double* x) {
    LDLT c(n, A);
    c.backsub(b, x);
    return PTAM_OK;
}

void ptamo_ba_opts_default(ptam_ba_opts* o) {
    o->max_iterations = 20;
    o->update_sq_conv_limit = 1e-6;
    o->min_sigma = 0.4;
    o->estimator = PTAM_EST_TUKEY;
    o->verbose = 0;
    o->deterministic = 0;   // (the CPU loops are sequential: nothing to choose)
    o->pad_ = 0;
}
int ptamo_ba_create(ptamo_ctx* c, const ptam_ba_opts* opts, ptamo_ba** out) {
    ptam_ba_opts o;
    if (opts)
        o = *opts;
    else
        ptamo_ba_opts_default(&o);
    *out = new ptamo_ba(c->c.cam, o);
    return PTAM_OK;
}
int ptamo_ba_destroy(ptamo_ba* b) {
    delete b;
    return PTAM_OK;
}
int ptamo_ba_add_camera(ptamo_ba* b, const double pose[12], int fixed) {
    return b->b.AddCamera(se3_from12(pose), fixed != 0);
}
int ptamo_ba_add_point(ptamo_ba* b, const double pos[3]) { return b->b.AddPoint(pos); }
int ptamo_ba_add_meas(ptamo_ba* b, int cam, int point, const double found[2], double sigma_sq) {
    if (cam < 0 || cam >= (int)b->b.mvCameras.size() || point < 0 || point >= (int)b->b.mvPoints.size())
        return PTAM_E_ARG;
    b->b.AddMeas(cam, point, found, sigma_sq);
    return PTAM_OK;
}
int ptamo_ba_add_cameras(ptamo_ba* b, int n, const double* poses, const uint8_t* fixed) {
    for (int i = 0; i < n; i++) b->b.AddCamera(se3_from12(poses + 12 * i), fixed[i] != 0);
    return PTAM_OK;
}
int ptamo_ba_add_points(ptamo_ba* b, int n, const double* pos) {
    for (int i = 0; i < n; i++) b->b.AddPoint(pos + 3 * i);
    return PTAM_OK;
}
int ptamo_ba_add_measurements(ptamo_ba* b, int n, const int32_t* cam, const int32_t* point, const double* found,
                              const double* sigma_sq) {
    for (int i = 0; i < n; i++) {
        const int rc = ptamo_ba_add_meas(b, cam[i], point[i], found + 2 * i, sigma_sq[i]);
        if (rc) return rc;
    }
    return PTAM_OK;
}
int ptamo_ba_compute(ptamo_ba* b, const volatile unsigned char* abort_flag, int* accepted) {
    const int a = b->b.Compute(abort_flag);
    if (accepted) *accepted = a;
    return PTAM_OK;
}
int ptamo_ba_converged(const ptamo_ba* b) { return b->b.mbConverged; }
int ptamo_ba_counts(const ptamo_ba* b, int* nc, int* nf, int* np, int* nm) {
    if (nc) *nc = (int)b->b.mvCameras.size();
    if (nf) *nf = b->b.mnCamsToUpdate;
    if (np) *np = (int)b->b.mvPoints.size();
    if (nm) *nm = (int)b->b.mMeasList.size();
    return PTAM_OK;
}
int ptamo_ba_get_point(const ptamo_ba* b, int n, double pos[3]) {
    if (n < 0 || n >= (int)b->b.mvPoints.size()) return PTAM_E_ARG;
    std::memcpy(pos, b->b.mvPoints[n].v3Pos, 3 * sizeof(double));
    return PTAM_OK;
}
int ptamo_ba_get_camera(const ptamo_ba* b, int n, double pose[12]) {
    if (n < 0 || n >= (int)b->b.mvCameras.size()) return PTAM_E_ARG;
    se3_to12(b->b.mvCameras[n].se3CfW, pose);
    return PTAM_OK;
}
int ptamo_ba_get_all(const ptamo_ba* b, double* poses, double* points) {
    for (size_t i = 0; i < b->b.mvCameras.size(); i++) se3_to12(b->b.mvCameras[i].se3CfW, poses + 12 * i);
    for (size_t i = 0; i < b->b.mvPoints.size(); i++) std::memcpy(points + 3 * i, b->b.mvPoints[i].v3Pos, 3 * sizeof(double));
    return PTAM_OK;
}
int ptamo_ba_get_outliers(const ptamo_ba* b, int32_t* pairs, int cap) {
    const int n = (int)b->b.mvOutlierMeasurementIdx.size();
    for (int i = 0; i < n && i < cap; i++) {
        pairs[2 * i] = b->b.mvOutlierMeasurementIdx[i].first;
        pairs[2 * i + 1] = b->b.mvOutlierMeasurementIdx[i].second;
    }
    return n;
}
int ptamo_ba_get_trials(const ptamo_ba* b, ptam_ba_trial* out, int cap) {
    const int n = (int)b->b.trials.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = b->b.trials[i];
    return n;
}
#ifndef PTAMO_REFEREE   // (takes ABI arrays of doubles: not part of the extended-precision build, oracle/referee.cc)
int ptamo_ba_set_comm(ptamo_ba* b, int rank, int world, ptam_allreduce_f64_fn fn, void* user) {
    b->b.rank = rank;
    b->b.world = world;
    b->b.comm = fn;
    b->b.comm_user = user;
    return PTAM_OK;
}
#endif

}   // extern "C"
