// bundle.h — device data layout of the bundle adjuster (Bundle, include/Bundle.h:106-152).
//
// HBM layout (all fp64 unless noted; M = measurements, P = points, C = cameras, F = free cameras):
//   measurements are sorted POINT-MAJOR (point id, then camera id) at Compute() time — the CSR
//   `rowptr[P+1]` replaces the reference's dense C x P pointer LUT (src/Bundle.cc:558-567) and its
//   per-point off-diagonal scripts (:572-599).  Planar (SoA) arrays so that a wave's 64 lanes touch
//   64 consecutive elements:
//     m_cam[M] i32 | m_pt[M] i32 | m_found[M] double2 | m_s[M] f64 | m_state[M] u8 | m_orig[M] i32
//     m_e2[M]                       squared error of pass 1
//     W[3][M+1][6]                  W_ij = A^T B (6x3) COORDINATE-MAJOR: plane c holds, per measurement, column c of the
//                                   block (the six camera parameters) as 6 consecutive doubles — the k-major operand
//                                   layout of the Schur product (k = point coordinate, row = camera parameter), so an
//                                   MFMA fragment is a plain global load; slot M of every plane is a zero block
//   points:  pt[2][P][3] (current / trial), V[P][6] (lower 00,10,11,20,21,22), epsB[P][3],
//            Vinv[P][9]
//   cameras: pose[2][C][12] (current / trial), Usplit[16][F][27] (U lower triangle 21 + epsA 6)
//   camera system: SE = [ S | E (npad) | 3 scalars ] contiguous so that the sharded path all-reduces ONE
//            buffer.  S is stored BLOCK-BANDED: only the 32x32 blocks (bi, bj) of the lower triangle with
//            bi - band <= bj <= bi exist (band = block bandwidth of the covisibility, se_blk below), each
//            block 8 KB contiguous, block rows one after another — the upper half is redundant
//            (src/Bundle.cc:451-453 mirrors it) and outside the band S is zero, so neither is stored,
//            factored or exchanged.  L (same layout) + Dg (npad) hold the LDL^T factor.
#pragma once
#include <cstddef>

#include "common.h"

enum { MS_ALIVE = 0, MS_BAD = 1, MS_DEAD = 2 };

#define BA_CHUNK 256          // threads per measurement block; a block owns whole points
#define SCHUR_TC 8            // cameras per Schur tile (48 rows)
#define SOLVE_NB 32           // LDL^T block size (64 with 1024-thread workgroups measured 1.7x slower)
#define HIST_BINS 4096        // first-level histogram of the order-statistic select (bits 62..51)

struct BaChunk {
    int pt_begin, pt_end, m_begin, m_end;
};

// one (camera-tile pair, point) work item of the Schur build
struct SchurEntry {
    int pt;            // point id
    int ma;            // first measurement of the point inside tile a (a free camera's)
    int mb;            // first measurement of the point inside tile b
    int pad;
    unsigned offa[2];  // byte s: offset from ma of the measurement by the tile's camera slot s (0..7), 0xff = the point
    unsigned offb[2];  //         is not measured by that camera (fixed cameras in between have no slot)
};
struct SchurWG {    // one segment of a workgroup of the tile kernel
    int pair;       // tile pair index a*(a+1)/2 + b
    int e_begin, e_end;
    int slot;       // its partial tile in s_part (contiguous per pair)
};

// device scalars shared between kernels and read back once per trial
struct BaScalars {
    double sigma_sq;        // mdSigmaSquared
    double cur_err;         // dCurrentError
    double new_err;         // dNewError
    double sumsq_cam;       // |delta a|^2
    double sumsq_pt;        // |delta b|^2
    double median;          // selected order statistic
    long long n_valid;      // non-bad measurements in pass 1
    int n_bad;              // measurements flagged bad so far in this LM step
    int n_outliers;         // total entries in the outlier list
    int sel_bin;            // first-level bin of the order statistic
    int sel_k;              // residual rank inside that bin
    int n_cand;             // compacted candidates
    int sel_bin2;           // sharded select: second-level bin (-1: clamped first-level bin, all candidates count)
    int sel_k2;             //                 residual rank inside it
    int select_overflow;    //                 a rank had more last-stage candidates than its exchange slot holds
    int n_outliers_pub;     // n_outliers as the last select found it — behind every purge, never while one is running: what the host reads
                            // as "the outlier list's length when the previous step closed" from a trial whose decision and whose step-closing
                            // purge are ONE launch (purge_pass1_decide_kernel)
    int pad_;
    // the device-side decision the speculatively enqueued kernels wait for: FOUR words 16-byte aligned, read as one scalar load
    // (ba_guard_blocks: a second dependent load in a kernel's guard is ~2 us, and five guarded kernels follow every trial)
    int end_step;           // after a trial: the LM step is over (accepted, converged or out of trials)
    int spec_go;            //                and the next step will run with the trial state as current
    int spec_stay;          //                or: the step is over WITHOUT an accepted trial (new == current error) and the
                            //                next one starts from the unchanged state with the unchanged lambda
    int spec_seq;           // which trial's decision these are (its mailbox sequence number): a guarded kernel enqueued behind another
                            // trial — a leftover on the queue a rejected trial's continuation has left — does nothing
    int abort_any;          // sharded: some rank's abort flag was up when this trial was enqueued (summed with the trial's scalars)
    int solve_fault;        // the persistent factorisation gave up waiting for one of its workgroups (ldlt_chain.inc): the solve is void
};
static_assert(offsetof(BaScalars, end_step) % 16 == 0 && sizeof(BaScalars) % 8 == 0, "BaScalars: the decision words are read as one 16-byte load");

struct BaDev {
    int C, F, P, M;
    int guard;              // 0: run; 1: run only if sc->spec_go or sc->spec_stay; 2: only if sc->end_step (speculatively enqueued kernels)
    int guard_seq;          //    ... and only if sc->spec_seq is this (the trial the kernel was enqueued behind)
    int n, npad;            // camera system order 6F and its padding to SOLVE_NB
    int band;               // block bandwidth of S (in SOLVE_NB blocks): |block(row) - block(col)| <= band wherever two cameras share a point
    int n_chunks, grid_acc; // measurement chunks; persistent grid of the accumulate kernel
    int u_rows;             // rows of Upart the accumulate kernel leaves: grid_acc (one per workgroup), or 1 (many cameras: every wave adds to one row)
    int n_wchunks;          // wave chunks (<= 64 measurements, whole points); 0 = block variant only
    int n_tiles, n_pairs, n_schur_wg, n_schur_entries;
    // cameras
    double* pose[2];
    int* cam_free;          // [C] free index or -1
    // points
    double* pt[2];
    double* V;
    double* epsB;
    double* Vinv;
    double* cut;            // wave variant of K7: [2 * ceil(M/64)][9] V|epsB pieces of the points a 64-measurement chunk
                            // boundary cuts (slot 2c: the chunk's leading segment, 2c+1: its trailing one); null otherwise
    int* rowptr;
    // measurements
    int* m_cam;
    int* m_pt;
    double2* m_found;
    double* m_s;
    int* m_orig;
    int* m_fidx;            // free-camera index of the measurement's camera (-1 = fixed)
    uint8_t* m_state;
    double* m_e2;
    double* m_e2t;          // squared errors of the last trial state (adopted as pass 1 when the trial is accepted)
    uint8_t* m_zbad_t;      // its z <= 0 flags
    double* W;              // [3][M + 1][6], see above
    // accumulators
    double* Usplit;         // [16][F*27] : per camera 21 lower-triangle U sums + 6 epsA sums, in 16
                            // fixed-order row splits of the accumulate grid (consumers add the 16)
    double* Upart;          // [grid_acc][F*27]
    // deterministic mode (ptam_ba_opts.deterministic): K7 stores per measurement its weighted camera Jacobian and residual,
    // a camera-major pass (reduce_det_kernel) adds each camera's U and epsA in a fixed order into Upart row 0
    double* Adet;           // [14][M]: planes A0[0..5], A1[0..5], ex, ey (zeros for fixed cameras / rejected measurements); null otherwise
    int* cam_ptr;           // [tiles][F+1] per tile of DET_TILE consecutive measurements: where each free camera's entries of cam_meas begin
    int* cam_meas;          // [measurements by free cameras] measurement ids sorted by (tile, camera, id)
    double* err_part;       // [max(n_chunks, grid_acc)][2]
    int* bad_part;          // [grid_acc]
    BaChunk* chunks;
    BaChunk* wchunks;
    // select
    unsigned* hist;         // [HIST_BINS]
    double* cand;           // [M]
    // Schur
    SchurEntry* s_entries;
    SchurWG* s_segs;        // segments, a workgroup's one after another
    int* s_wg_seg;          // [n_schur_wg + 1] first segment of each workgroup (block 8 i + x = i-th workgroup of XCD x)
    int* s_pair_wg_begin;   // [n_pairs+1] first WG of each pair
    int* s_wg_head;         // [n_schur_wg][8] {first segment, end, then the first segment itself: pair, e_begin, e_end, slot, 0, 0}
    double* s_part;         // [partial-tile slots][48*48 + 48]
    unsigned* s_map;        // [6 variants of the tile body][256 threads][SCHUR_MAP_WPT] where the elements of the partial tile a thread
                            // stores sit in the accumulator layout of the cross-wave reduction (two 16-bit indices per word: value
                            // index * 64 + lane; 0xffff: a zero; 0xffffffff: no element) — schur_index_map_device()
    // camera system
    double* SE;             // S then E
    double* L;
    double* Dg;
    double* y;              // forward-substituted rhs
    double* da;             // [npad] camera update
    double *SE2, *L2, *Dg2, *y2;   // the second chain of a two-ended persistent elimination: the system's bottom end in mirrored coordinates (ldlt_chain.inc)
    unsigned* sflags2;      //   and its flag words
    unsigned* sflags;       // flag words of the persistent factorisation (ldlt_chain.inc); [0]: a spin gave up
    unsigned solve_seq;     // its sequence number: a flag is up when it holds the current solve's number (the host increments it)
    int chain_off;          // host: use the launch-per-block-column forms only (a persistent solve of this bundle timed out, or another
                            // bundle of this process is adjusting on the same device: the persistent form needs its workgroups co-resident)
    int chain_xcd;          // the XCD (blockIdx % 8) the persistent solve's working blocks sit on; a second chain takes the next one
    int spin_limit;         // looks a waiting workgroup of the persistent solve takes before it gives up (ldlt_chain.inc)
    double* bw_scratch;     // [2][6 npad] the backward substitution's vectors when they do not fit LDS (solve.hip)
    double* sumsq2;         // [2] |da|^2 in two parts (the two workgroups of the backward substitution; consumers add them)
    // outliers
    int* outliers;          // [M] original indices, in purge order
    BaScalars* sc;
    long long* dbg;         // 16 cycle stamps (only written by -DK7_TIMING builds)
};

// ---- W planes ----
__host__ __device__ __forceinline__ size_t w_plane(const BaDev& d) { return (size_t)(d.M + 1) * 6; }   // doubles per coordinate plane

// ---- block-banded packed storage of S and L (lower triangle, 32x32 blocks inside the band) ----
__host__ __device__ __forceinline__ size_t se_blocks_before_row(int bi, int band) {
    return bi <= band ? (size_t)bi * (bi + 1) / 2 : (size_t)(band + 1) * (band + 2) / 2 + (size_t)(bi - band - 1) * (band + 1);
}
// offset (in doubles) of block (bi, bj), bi - band <= bj <= bi; element (r, c) of the block sits at + r * SOLVE_NB + c
__host__ __device__ __forceinline__ size_t se_blk(int bi, int bj, int band) {
    const int j0 = bi - band > 0 ? bi - band : 0;
    return (se_blocks_before_row(bi, band) + (size_t)(bj - j0)) * (SOLVE_NB * SOLVE_NB);
}
// doubles of S (or L) for nblk block rows; E follows S at this offset
__host__ __device__ __forceinline__ size_t se_size(int nblk, int band) { return se_blocks_before_row(nblk, band) * (SOLVE_NB * SOLVE_NB); }
__host__ __device__ __forceinline__ int se_band(const BaDev& d) {
    const int nblk = d.npad / SOLVE_NB;
    return d.band < nblk - 1 ? d.band : (nblk > 0 ? nblk - 1 : 0);
}
__host__ __device__ __forceinline__ double* se_E(const BaDev& d) { return d.SE + se_size(d.npad / SOLVE_NB, se_band(d)); }

// solve.hip
int ba_solve(ptam_ctx* ctx, BaDev& d, int cur);   // also writes the trial poses pose[cur^1] and |da|^2
#define CH_SPIN_DEFAULT (1 << 18)   // BaDev::spin_limit unless PTAM_CH_SPIN_LIMIT says otherwise (ldlt_chain.inc)
size_t ba_solve_flag_bytes(int nblk);   // bytes of BaDev::sflags for a system of nblk blocks
int ba_solve_init();   // raises the dynamic-LDS limits of the solve kernels (once per process/device)

// -DK7_TIMING builds only (tools/): device-side timeline, one (kernel id, 100 MHz time stamp) pair per launch
#ifdef K7_TIMING
#define TL_BASE 4096
#define TL_MAX 1900
#define TL_MARK(d, kid)                                                                                   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                         \
        const unsigned long long q_ = atomicAdd((unsigned long long*)&(d).dbg[TL_BASE], 1ull);            \
        if (q_ < TL_MAX) {                                                                                \
            (d).dbg[TL_BASE + 2 + 2 * q_] = (kid);                                                        \
            (d).dbg[TL_BASE + 3 + 2 * q_] = (long long)__builtin_amdgcn_s_memrealtime();                  \
        }                                                                                                 \
    }
#else
#define TL_MARK(d, kid)
#endif
