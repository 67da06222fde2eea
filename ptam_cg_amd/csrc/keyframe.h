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
};

struct ptam_kf {
    int device;
    KfLevels L;
    void* base;
    size_t bytes_total, bytes_px;
    int n_blocks;
    int n_corners[PTAM_LEVELS];   // host copy, valid iff counts_valid
    int counts_valid;
};

int kf_fetch_counts(ptam_ctx* ctx, const ptam_kf* kf);
