// keyframe.h — device keyframe (pyramid + FAST corners + row LUTs) shared by keyframe.hip / patch.hip
#pragma once
#include "common.h"

struct KfLevels {
    uint8_t* im[PTAM_LEVELS];                 // Level::im            include/KeyFrame.h:62
    int w[PTAM_LEVELS], h[PTAM_LEVELS];
    int thr[PTAM_LEVELS];                     // FAST thresholds 10,15,15,10  src/KeyFrame.cc:35-42
    int ntx[PTAM_LEVELS];                     // 64-pixel tiles per row
    int block_begin[PTAM_LEVELS];             // first detect-kernel block of each level
    unsigned long long* mask[PTAM_LEVELS];    // [h][ntx] corner bit masks (bit = x within tile)
    ptam_int2* corners[PTAM_LEVELS];          // Level::vCorners (raster order)
    int* rowlut[PTAM_LEVELS];                 // Level::vCornerRowLUT
    int* ncorners;                            // [4] device counters
    // MakeKeyFrame_Rest (src/KeyFrame.cc:61-82)
    uint8_t* score[PTAM_LEVELS];              // FAST score per pixel (0 = not a corner)
    unsigned long long* mmask[PTAM_LEVELS];   // [h][ntx] bit masks of the maximal corners
    ptam_int2* mcorners[PTAM_LEVELS];         // Level::vMaxCorners (raster order)
    double* st[PTAM_LEVELS];                  // Shi-Tomasi score per maximal corner (-1: within 10 px of the border)
    int* nmax;                                // [4] device counters
};

struct ptam_kf {
    int device;
    KfLevels L;
    void* base;
    size_t bytes_total, bytes_px;
    int n_blocks;
    int n_corners[PTAM_LEVELS];   // host copy, valid iff counts_valid
    int counts_valid;
    int n_max[PTAM_LEVELS];       // host copy, valid iff rest_valid
    int rest_valid;
    size_t off_rest_clear, bytes_rest_clear;   // score maps + maximal-corner masks: zeroed per call
    // Level::vImplaneCorners (src/MapMaker.cc:605-614), built on first use by the epipolar search
    double2* implane[PTAM_LEVELS];
    int implane_cap[PTAM_LEVELS];
    int implane_valid[PTAM_LEVELS];   // bImplaneCornersCached
};

int kf_fetch_counts(ptam_ctx* ctx, const ptam_kf* kf);
