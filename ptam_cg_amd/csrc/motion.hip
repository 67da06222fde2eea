// motion.hip — the tracker's motion model (src/Tracker.cc:1008-1056), the bTryCoarse heuristics (:505-516) and the tracking
// branch of Tracker::TrackFrame (:94, :134-137) over the resident TrackMap chain; plus the native driver that tracks a
// sequence of frames (ptam_hip_bench.h).  Host scalar code: nothing here launches a kernel of its own.
#include <chrono>
#include <cmath>
#include <cstring>

#include "common.h"
#include "../../include/ptam_hip_bench.h"

namespace {

// TooN SO3<>::ln: rotation vector of R (row-major).  Three ranges of the angle: asin of the antisymmetric part's norm up to
// pi/4, acos of the trace up to 3 pi/4, and beyond that the axis from the symmetric part (the antisymmetric one vanishes at pi).
void so3_ln(const double* R, double w[3]) {
    const double cos_angle = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    w[0] = (R[7] - R[5]) / 2;
    w[1] = (R[2] - R[6]) / 2;
    w[2] = (R[3] - R[1]) / 2;
    const double sin_angle_abs = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (cos_angle > M_SQRT1_2) {
        if (sin_angle_abs > 0) {
            const double s = std::asin(sin_angle_abs) / sin_angle_abs;
            for (int i = 0; i < 3; i++) w[i] *= s;
        }
    } else if (cos_angle > -M_SQRT1_2) {
        const double s = std::acos(cos_angle) / sin_angle_abs;
        for (int i = 0; i < 3; i++) w[i] *= s;
    } else {
        const double angle = M_PI - std::asin(sin_angle_abs);
        const double d0 = R[0] - cos_angle, d1 = R[4] - cos_angle, d2 = R[8] - cos_angle;
        double r2[3];
        if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) {
            r2[0] = d0;
            r2[1] = (R[3] + R[1]) / 2;
            r2[2] = (R[2] + R[6]) / 2;
        } else if (d1 * d1 > d2 * d2) {
            r2[0] = (R[3] + R[1]) / 2;
            r2[1] = d1;
            r2[2] = (R[7] + R[5]) / 2;
        } else {
            r2[0] = (R[2] + R[6]) / 2;
            r2[1] = (R[7] + R[5]) / 2;
            r2[2] = d2;
        }
        if (r2[0] * w[0] + r2[1] * w[1] + r2[2] * w[2] < 0)
            for (int i = 0; i < 3; i++) r2[i] = -r2[i];
        const double inv = 1.0 / std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
        for (int i = 0; i < 3; i++) w[i] = angle * (r2[i] * inv);
    }
}

const double kIdentity[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};

}   // namespace

extern "C" {

void ptam_se3_exp(const double mu[6], double pose_out[12]) {
    if (!mu || !pose_out) return;
    se3_exp_mul<false>(mu, kIdentity, pose_out);
}

// TooN SE3<>::ln: rotation vector first, then the translation part "un-rotated" by half the rotation and rescaled
void ptam_se3_ln(const double pose[12], double mu_out[6]) {
    if (!pose || !mu_out) return;
    double rot[3];
    so3_ln(pose, rot);
    const double theta_sq = rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2];
    const double theta = std::sqrt(theta_sq);
    double shtot = 0.5;
    if (theta > 0.00001) shtot = std::sin(theta / 2) / theta;
    // halfrotator = SO3::exp(-rot / 2)
    const double half[6] = {0, 0, 0, rot[0] * -0.5, rot[1] * -0.5, rot[2] * -0.5};
    double H[12];
    se3_exp_mul<false>(half, kIdentity, H);
    const double* t = pose + 9;
    double rt[3];
    for (int r = 0; r < 3; r++) rt[r] = H[r * 3 + 0] * t[0] + H[r * 3 + 1] * t[1] + H[r * 3 + 2] * t[2];
    const double tdot = t[0] * rot[0] + t[1] * rot[1] + t[2] * rot[2];
    if (theta > 0.001) {
        const double f = tdot * (1 - 2 * shtot) / theta_sq;
        for (int i = 0; i < 3; i++) rt[i] -= rot[i] * f;
    } else {
        const double f = tdot / 24;
        for (int i = 0; i < 3; i++) rt[i] -= rot[i] * f;
    }
    for (int i = 0; i < 3; i++) {
        mu_out[i] = rt[i] / (2 * shtot);
        mu_out[3 + i] = rot[i];
    }
}

void ptam_motion_reset(ptam_motion_model* m, const double pose[12]) {
    if (!m) return;
    std::memset(m, 0, sizeof *m);
    std::memcpy(m->pose, pose ? pose : kIdentity, 96);
    std::memcpy(m->start_pose, m->pose, 96);
    m->scene_depth_mean = 1.0;            // src/Tracker.cc:56
    m->coarse_min_velocity = 0.006;       // :496
    m->use_constant_velocity = 1;         // :1041
}

// src/Tracker.cc:1013-1030 (rotation estimator off)
void ptam_motion_predict(ptam_motion_model* m) {
    if (!m) return;
    std::memcpy(m->start_pose, m->pose, 96);
    se3_exp_mul<false>(m->velocity, m->start_pose, m->pose);
}

void ptam_motion_update(ptam_motion_model* m, const ptam_trackmap_result* r) {
    if (!m || !r) return;
    std::memcpy(m->pose, r->pose, 96);
    if (r->depth_n > 20) {                // :692-696
        m->scene_depth_mean = r->depth_sum / r->depth_n;
        m->scene_depth_sigma = std::sqrt(r->depth_sum_sq / r->depth_n - m->scene_depth_mean * m->scene_depth_mean);
    }
    // :1038-1039  se3NewFromOld = mse3CamFromWorld * mse3StartPos.inverse()
    const double* S = m->start_pose;
    double inv[12];
    for (int r_ = 0; r_ < 3; r_++) {
        for (int c = 0; c < 3; c++) inv[r_ * 3 + c] = S[c * 3 + r_];
        inv[9 + r_] = -(S[0 * 3 + r_] * S[9] + S[1 * 3 + r_] * S[10] + S[2 * 3 + r_] * S[11]);
    }
    const double* P = m->pose;
    double nfo[12];
    for (int r_ = 0; r_ < 3; r_++) {
        for (int c = 0; c < 3; c++) nfo[r_ * 3 + c] = P[r_ * 3 + 0] * inv[c] + P[r_ * 3 + 1] * inv[3 + c] + P[r_ * 3 + 2] * inv[6 + c];
        nfo[9 + r_] = P[9 + r_] + (P[r_ * 3 + 0] * inv[9] + P[r_ * 3 + 1] * inv[10] + P[r_ * 3 + 2] * inv[11]);
    }
    double motion[6];
    ptam_se3_ln(nfo, motion);
    if (m->use_constant_velocity) {
        for (int i = 0; i < 6; i++) m->velocity[i] = motion[i];
    } else {
        for (int i = 0; i < 6; i++) m->velocity[i] = 0.9 * (0.5 * motion[i] + 0.5 * m->velocity[i]);
    }
    double v[6], s = 0;
    const double inv_depth = 1.0 / m->scene_depth_mean;
    for (int i = 0; i < 6; i++) {
        v[i] = i < 3 ? m->velocity[i] * inv_depth : m->velocity[i];
        s += v[i] * v[i];
    }
    m->msd_scaled_velocity = std::sqrt(s);
}

int ptam_track_frame(ptam_tracker* t, ptam_kf* current, const uint8_t* d_frame, ptam_motion_model* m,
                     const ptam_trackmap_opts* opts, ptam_trackmap_result* out) {
    ARG_TRY(t && current && d_frame && m && out);
    ptam_trackmap_opts o;
    if (opts) o = *opts;
    else ptam_trackmap_opts_default(&o);
    // src/Tracker.cc:503-514
    o.try_coarse = 1;
    if (m->disable_coarse || m->msd_scaled_velocity < m->coarse_min_velocity || o.coarse_max == 0) o.try_coarse = 0;
    // (the model is advanced on a COPY and committed only when the frame was tracked: a call that fails — time-out, HIP error,
    //  PTAM_E_STATE — leaves the caller's model as it was, so that a retry neither predicts twice nor has lost the doubled
    //  coarse range of a recovery; the reference has no such partial state)
    ptam_motion_model tmp = *m;
    if (tmp.just_recovered) {
        o.try_coarse = 1;
        o.coarse_max *= 2;
        o.coarse_range *= 2;
        tmp.just_recovered = 0;
    }
    ptam_motion_predict(&tmp);
    const int rc = ptam_track_map_frame(t, current, d_frame, tmp.pose, &o, out);
    if (rc) return rc;
    ptam_motion_update(&tmp, out);
    *m = tmp;
    return PTAM_OK;
}

int ptam_bench_track_sequence(ptam_tracker* t, ptam_kf* current, int n_frames, const uint8_t* const* d_frames,
                              ptam_motion_model* m, const ptam_trackmap_opts* opts, const int32_t* shuffle_levels,
                              const int32_t* shuffle_fine, int passes, const double* poses_true, double* seconds_out,
                              double* stats_out) {
    ARG_TRY(t && current && n_frames >= 1 && d_frames && m && shuffle_levels && shuffle_fine && passes >= 1 && seconds_out);
    for (int i = 0; i < n_frames; i++) ARG_TRY(d_frames[i]);
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    ptam_trackmap_result res;
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < passes; p++)
        for (int f = 0; f < n_frames; f++) {
            const bool tried = !(m->disable_coarse || m->msd_scaled_velocity < m->coarse_min_velocity) || m->just_recovered;
            int rc = ptam_tracker_set_shuffle(t, shuffle_levels, shuffle_fine);
            if (!rc) rc = ptam_track_frame(t, current, d_frames[f], m, opts, &res);
            if (rc) return rc;
            st[0] += 1;
            st[1] += res.n_coarse + res.n_top + res.n_fine;
            st[2] += res.templates_reused;
            st[3] += res.n_meas;
            st[4] += res.did_coarse ? 1 : 0;
            st[5] += tried ? 1 : 0;
            if (poses_true) {
                const double* q = poses_true + (size_t)f * 12;
                // camera centres: -R^T t of the tracked and of the true pose
                double d2 = 0;
                for (int c = 0; c < 3; c++) {
                    const double a = -(res.pose[0 * 3 + c] * res.pose[9] + res.pose[1 * 3 + c] * res.pose[10] + res.pose[2 * 3 + c] * res.pose[11]);
                    const double b = -(q[0 * 3 + c] * q[9] + q[1 * 3 + c] * q[10] + q[2 * 3 + c] * q[11]);
                    d2 += (a - b) * (a - b);
                }
                st[6] = std::fmax(st[6], std::sqrt(d2));
            }
            st[7] += res.n_meas < 50 ? 1 : 0;
        }
    *seconds_out = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats_out) std::memcpy(stats_out, st, sizeof st);
    return PTAM_OK;
}

}   // extern "C"
