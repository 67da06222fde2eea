// bundle.hip — Bundle::Compute (src/Bundle.cc:116-158) and Do_LM_Step (:209-551) on gfx950.
//
// Kernels (DESIGN.md has the byte counts):
//   K5  project_e2_kernel     pass 1 (:219-225, :164-180) + first-level histogram of e^2
//   K6  select_*_kernel       exact order statistic sorted[n/2] (include/Tools.h:152-162) by radix
//                             select on the fp64 bit pattern -> sigma^2 (floor MinTukeySigma^2 :234-237)
//   K7  jac_accum_kernel      pass 2 (:250-332) FUSED: weight, A(2x6), B(2x3), U += A^T A, epsA,
//                             V += B^T B, epsB, W = A^T B.  Roofline kernel (HBM-bound, ~185 B/meas).
//       reduce_partials_kernel  fixed-order sum of the per-workgroup camera partials
//   K8a vinv_kernel           V*^-1 (:341-359)
//   K8  schur_tile_mfma_kernel S = U* - sum_i (W V*^-1) W^T, E = epsA - sum_i W V*^-1 epsB (:374-446),
//                             output-stationary over 8x8-camera tiles; schur_reduce_kernel sums the
//                             tile partials in fixed order (deterministic) and writes S lower / E
//   K9  solve.hip             blocked LDL^T + substitutions (:457-458)
//   K10 pose_update_kernel, point_update_kernel, finalize_new_kernel
//                             delta b (:461-483), update norm (:488-490), trial poses/points
//                             (:496-504), new robust error (:188-207, :506)
//       purge_kernel          erase bad measurements, record outliers (:536-547)
// The lambda-trial control flow, convergence test and abort polling stay on the host, one readback of
// a 64-byte scalar block per trial.
#include <algorithm>
#include <array>
#include <chrono>
#include <climits>
#include <atomic>
#include <map>
#include <mutex>
#include <numeric>
#include <utility>

#include "bundle.h"
#include "../../include/ptam_hip_bench.h"

#include "ba_math.inc"
#include "ba_pass1.inc"
#include "ba_select.inc"
#include "ba_jacobian.inc"
#include "ba_schur.inc"
#include "ba_update.inc"
#include "ba_prepare.inc"

// =================================================================================================
// host
// =================================================================================================
// A growing array in PINNED host memory: the measurements are uploaded as they were added (the index structures are built on the
// device, ba_prepare.inc), and a copy out of pageable memory is staged by the runtime piece by piece (2.5 ms for the 8 MB of
// 250 000 measurements against 0.3 ms).  Released arrays go to the context's cache: MapMaker builds a new Bundle for every
// adjustment (src/MapMaker.cc:838-845), and pinning pages costs far more than using them.
template <class T>
struct PinVec {
    ptam_ctx* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0, cap = 0, cap_bytes = 0;
    size_t size() const { return n; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    int reserve(size_t want) {
        if (want <= cap) return PTAM_OK;
        const size_t bytes = std::max<size_t>(want * sizeof(T), 4096);
        void* np = nullptr;
        size_t nbytes = 0;
        if (!ctx_cache_take(ctx->pin_cache, CTX_NCACHE(ctx->pin_cache), bytes, &np, &nbytes)) {
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipHostMalloc(&np, bytes, hipHostMallocDefault));
            nbytes = bytes;
        }
        if (n) std::memcpy(np, p, n * sizeof(T));
        release();
        p = (T*)np;
        cap_bytes = nbytes;
        cap = nbytes / sizeof(T);
        return PTAM_OK;
    }
    int grow(size_t more) { return n + more <= cap ? PTAM_OK : reserve(std::max(n + more, 2 * cap)); }
    int push_back(const T& v) {
        if (n == cap) {
            const int rc = grow(1);
            if (rc) return rc;
        }
        p[n++] = v;
        return PTAM_OK;
    }
    int append(const T* src, size_t k) {
        const int rc = grow(k);
        if (rc) return rc;
        std::memcpy(p + n, src, k * sizeof(T));
        n += k;
        return PTAM_OK;
    }
    void release() {   // (keeps n: reserve() re-points the array)
        if (p) {
            void* drop = ctx_cache_give(ctx->pin_cache, CTX_NCACHE(ctx->pin_cache), p, cap_bytes);
            if (drop) hipHostFree(drop);
        }
        p = nullptr;
        cap = cap_bytes = 0;
    }
};

// The measurements as they are added (ba_prepare.inc: MS_CH, ms_cam ...): 1 MB chunks of [cam | point | found | sigma^2] in pinned
// host memory and, in the same layout, on the device — a chunk is uploaded as ONE copy the moment it is full, so the PCIe transfer
// (0.3 ms of a 0.5 ms prepare at 250 000 measurements) runs while the host is still adding; prepare sends the last, partial chunk.
struct MeasStore {
    ptam_ctx* ctx = nullptr;
    char* h = nullptr;   // pinned host chunks
    size_t h_chunks = 0, h_bytes = 0;
    size_t n = 0;        // measurements
    char* d = nullptr;   // device chunks
    size_t d_chunks = 0, d_bytes = 0;
    size_t up_chunks = 0;     // complete chunks already on their way to the device
    bool in_flight = false;   // a copy out of h may still be queued
    size_t size() const { return n; }
    int& cam(size_t i) { return const_cast<int&>(ms_cam(h, i)); }
    int& pt(size_t i) { return const_cast<int&>(ms_pt(h, i)); }
    double2& found(size_t i) { return const_cast<double2&>(ms_found(h, i)); }
    double& sig(size_t i) { return const_cast<double&>(ms_sig(h, i)); }
    int reserve_host(size_t chunks) {
        if (chunks <= h_chunks) return PTAM_OK;
        chunks = std::max(chunks, 2 * h_chunks);
        void* np = nullptr;
        size_t nbytes = 0;
        if (!ctx_cache_take(ctx->pin_cache, CTX_NCACHE(ctx->pin_cache), chunks * MS_CH_BYTES, &np, &nbytes)) {
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipHostMalloc(&np, chunks * MS_CH_BYTES, hipHostMallocDefault));
            nbytes = chunks * MS_CH_BYTES;
        }
        if (n) std::memcpy(np, h, ((n + MS_CH - 1) >> MS_LOG) * MS_CH_BYTES);
        if (in_flight) HIP_TRY(ptam_stream_wait(ctx->stream));   // (copies out of the old chunks)
        in_flight = false;
        release_host();
        h = (char*)np;
        h_bytes = nbytes;
        h_chunks = nbytes / MS_CH_BYTES;
        return PTAM_OK;
    }
    int reserve_dev(size_t chunks) {
        if (chunks <= d_chunks) return PTAM_OK;
        chunks = std::max(std::max(chunks, 2 * d_chunks), h_chunks);
        void* np = nullptr;
        size_t nbytes = 0;
        HIP_TRY(hipSetDevice(ctx->device));
        if (!ctx_cache_take(ctx->dev_cache, CTX_NCACHE(ctx->dev_cache), chunks * MS_CH_BYTES, &np, &nbytes)) {
            HIP_TRY(hipMalloc(&np, chunks * MS_CH_BYTES));
            nbytes = chunks * MS_CH_BYTES;
        }
        release_dev();   // (queued copies into the old block: its next owner only touches it through the same queue)
        d = (char*)np;
        d_bytes = nbytes;
        d_chunks = nbytes / MS_CH_BYTES;
        up_chunks = 0;   // everything goes up again
        return PTAM_OK;
    }
    // enqueue the upload of the complete chunks that have not gone yet (and of the partial last one: prepare)
    int flush(bool with_partial) {
        const size_t full = n >> MS_LOG, want = with_partial ? (n + MS_CH - 1) >> MS_LOG : full;
        if (want > d_chunks) {
            const int rc = reserve_dev(want);
            if (rc) return rc;
        }
        if (want > up_chunks) {
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(hipMemcpyAsync(d + up_chunks * MS_CH_BYTES, h + up_chunks * MS_CH_BYTES, (want - up_chunks) * MS_CH_BYTES, hipMemcpyHostToDevice,
                                   ctx->stream));
            in_flight = true;
        }
        up_chunks = full;
        return PTAM_OK;
    }
    void release_host() {
        if (h) {
            void* drop = ctx_cache_give(ctx->pin_cache, CTX_NCACHE(ctx->pin_cache), h, h_bytes);
            if (drop) hipHostFree(drop);
        }
        h = nullptr;
        h_chunks = h_bytes = 0;
    }
    void release_dev() {
        if (d) {
            void* drop = ctx_cache_give(ctx->dev_cache, CTX_NCACHE(ctx->dev_cache), d, d_bytes);
            if (drop) hipFree(drop);
        }
        d = nullptr;
        d_chunks = d_bytes = 0;
        up_chunks = 0;
    }
};

struct ptam_ba {
    ptam_ctx* ctx;
    ptam_ba_opts opts;
    // inputs (insertion order), copied at add_* time like the reference; the measurements and the points in pinned memory
    std::vector<double> cam_pose;
    std::vector<uint8_t> cam_fixed;
    PinVec<double> pts;
    MeasStore ms;                  // camera, point, found position, sigma^2 per measurement (dSqrtInvNoise is formed on the device)
    std::vector<uint8_t> m_dead;   // erased by an earlier Compute()
    int n_dead = 0;
    // results
    bool converged = false;
    int accepted = 0;
    int solve_fallbacks = 0;   // trials repeated with the launch-per-block-column solve after the persistent one timed out
    std::vector<ptam_ba_trial> trials;
    std::vector<std::pair<int, int>> outliers;   // (point, camera)
    std::vector<int> raw_out, raw_out_ends;      // measurement indices as purged + segment ends (per LM step), not yet digested
    int raw_out_base = 0, raw_out_done = 0;
    // device
    bool prepared = false;
    BaDev d;
    void* block = nullptr;
    size_t block_bytes = 0, block_cap = 0;
    void* sblock = nullptr;   // the Schur work lists (sized after the device has counted their entries)
    size_t sblock_bytes = 0, sblock_cap = 0;
    int n_schur_segs = 0;
    int dups_refused = 0;     // measurements of the last prepare that had a twin (same point, same camera)
    bool xchg_owned = false;   // d_xchg is an allocation of its own (a sharded bundle whose prepare failed)
    bool e2_is_current = false;   // m_e2 / m_state hold pass 1 of the CURRENT poses and points (the last step accepted nothing)
    int band_local = 0;     // block bandwidth of S needed by THIS process' points (ba->d.band: the one in force)
    int cur = 0;
    size_t smem_acc = 0;
    bool use_wave = false;
    bool det = false;        // ptam_ba_opts.deterministic: K7 stores A / epsilon per measurement, reduce_det_kernel sums per camera
    bool k7_big = false;     // camera partials + poses of K7 in global memory (more cameras than a workgroup's LDS holds)
    int per_wave = 1, extra_waves = 0;
    // host-mapped mailbox the device publishes BaScalars into (the LM loop's one host decision per trial)
    struct Mailbox {
        struct Slot {
            volatile unsigned long long v, seq;
        } slot[sizeof(BaScalars) / 8];
    };
    Mailbox* mbox = nullptr;       // host address
    Mailbox* mbox_dev = nullptr;   // device address of the same memory
    unsigned long long mbox_seq = 0;
    bool published_by_finalize = false;
    bool decide_pending = false;     // the trial's decision is left to the first launch of the speculative prologue (purge_pass1_decide_kernel)
    int decide_last_allowed = 0;
    bool trial_is_current = false;   // the last trial was accepted: its new-error pass == pass 1 of the next step
    int k7_threads = BA_CHUNK;
    bool k7_loop = false;
    std::vector<int> pt_orig;       // device point id -> original point id (the points with a live measurement, ascending)
    // gather buffers (sharded mode)
    double* d_gather = nullptr;
    size_t gather_cap = 0;
    double* d_xchg = nullptr;   // small exchange buffer (counts, scalars)
    double* d_sel = nullptr;    // sharded select: [4096 histogram words][world counts][world x XCAND_CAP keys][same again: list]
    int xcand_cap = XCAND_CAP_DEFAULT;
    bool cur_pending = false;   // sharded: (current error, bad count) are waiting behind S|E for the next all-reduce
    bool slow_select = false;   // sharded select fell back to gathering every key (after a select_overflow)
    // profiling
    bool prof = false;
    hipEvent_t ev[PTAM_K_COUNT][2];
    bool ev_ok = false;
    bool ev_used[PTAM_K_COUNT];
    double k_ms[PTAM_K_COUNT];
    int k_n[PTAM_K_COUNT];
    // communicator
    int rank = 0, world = 1;
    ptam_allreduce_f64_fn comm = nullptr;
    void* comm_user = nullptr;
};

static void ba_free_device(ptam_ba* ba) {
    for (int k = 0; k < 2; k++) {
        void* blk = k ? ba->sblock : ba->block;
        if (blk) {
            // the block's kernels may still be queued: the next owner only touches it through the same stream
            void* drop = ctx_cache_give(ba->ctx->dev_cache, CTX_NCACHE(ba->ctx->dev_cache), blk, k ? ba->sblock_cap : ba->block_cap);
            if (drop) hipFree(drop);
        }
    }
    if (ba->d_gather) hipFree(ba->d_gather);
    if (ba->d_xchg && ba->xchg_owned) hipFree(ba->d_xchg);   // (otherwise a piece of the block)
    ba->xchg_owned = false;
    if (ba->d_sel) hipFree(ba->d_sel);
    ba->d_sel = nullptr;
    ba->block = nullptr;
    ba->sblock = nullptr;
    ba->d_gather = nullptr;
    ba->d_xchg = nullptr;
    ba->gather_cap = 0;
    ba->prepared = false;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

static void ba_finish_outliers(ptam_ba* ba);
// Wait for a stamp a kernel writes into host-mapped memory behind its results (ptam_stream_wait sleeps on an interrupt: up to
// milliseconds to wake up); every now and then ask the runtime whether the stream died instead.
static int ba_wait_stamp(ptam_ctx* ctx, volatile unsigned long long* slot, unsigned long long seq) {
    unsigned spins = 0;
    while (*slot != seq) {
        if (++spins == 100000) {
            spins = 0;
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q != hipSuccess && q != hipErrorNotReady) {
                ptam_set_error("the device queue failed while the bundle was being prepared: %s", hipGetErrorString(q));
                return PTAM_E_HIP;
            }
            if (q == hipSuccess && *slot != seq) {
                ptam_set_error("bundle prepare: the queue drained without its stamp");
                return PTAM_E_HIP;
            }
        }
    }
    (void)hipGetLastError();   // (hipErrorNotReady is sticky in the last-error slot)
    std::atomic_thread_fence(std::memory_order_acquire);
    return PTAM_OK;
}

// launch shape of K7 for (threads, LDS bytes): asked of the runtime once per process and shape (each query is 10 - 40 us)
static int ba_k7_occupancy(const void* k7, int threads, size_t smem, int* per_cu) {
    struct Key {
        const void* f;
        int t;
        size_t s;
        bool operator<(const Key& o) const { return f != o.f ? f < o.f : (t != o.t ? t < o.t : s < o.s); }
    };
    static std::mutex mu;
    static std::map<Key, std::pair<int, int>> known;
    std::lock_guard<std::mutex> lock(mu);
    const Key key{k7, threads, smem};
    auto it = known.find(key);
    if (it == known.end()) {
        int rc = PTAM_OK, n = 0;
        if (smem > 64 * 1024 && hipFuncSetAttribute(k7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) rc = PTAM_E_HIP;
        if (!rc && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k7, threads, smem) != hipSuccess) rc = PTAM_E_HIP;
        if (rc) (void)hipGetLastError();
        it = known.emplace(key, std::make_pair(rc, n)).first;
    }
    *per_cu = it->second.second;
    return it->second.first;
}

static int ba_prepare_impl(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    // PTAM_DEBUG_PREPARE=1: host time of the phases of this function
    static const bool dbg_prep = getenv("PTAM_DEBUG_PREPARE") != nullptr;
    auto pt0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg_prep) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ptam] prepare: %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - pt0).count());
        pt0 = t;
    };
    ba_finish_outliers(ba);   // erased measurements of earlier Compute() calls leave the problem here
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ptam_stream_wait(ctx->stream));
    ba_free_device(ba);
    BaDev& d = ba->d;
    std::memset(&d, 0, sizeof d);
    const int C = (int)ba->cam_fixed.size(), P_all = (int)(ba->pts.size() / 3);
    const int Mall = (int)ba->ms.size();
    const int M = Mall - ba->n_dead;
    // free-camera indices in insertion order  (nStartRow, src/Bundle.cc:52-57)
    std::vector<int> cam_free(C, -1);
    int F = 0;
    for (int c = 0; c < C; c++)
        if (!ba->cam_fixed[c]) cam_free[c] = F++;
    const int n_tiles = (F + SCHUR_TC - 1) / SCHUR_TC;
    const int n_pairs = n_tiles * (n_tiles + 1) / 2;
    const bool lists = F > 0 && M > 0;   // Schur work lists exist
    d.C = C;
    d.F = F;
    d.M = M;
    d.n = 6 * F;
    d.npad = ((d.n + SOLVE_NB - 1) / SOLVE_NB) * SOLVE_NB;
    d.n_wchunks = M > 0 ? 1 : 0;
    ba->use_wave = M > 0;   // (false: no live measurement at all — K7 is then a memset of its outputs)
    d.n_tiles = n_tiles;
    d.n_pairs = n_pairs;
    // The tile kernel's work split (ba_split.h): cost model fitted to the kernel's per-workgroup stamps at 50 x 5000
    // (tools/dev/schur_fit.py, docs/LOG_r05.md) — a group of four entries takes 10 f + 36 units (f: the 16x16 fragment products of
    // its pattern; the 36 are its loads), a SEGMENT costs as much as ~10 full groups (pipeline fill: three dependent round trips;
    // cross-wave reduction; partial tile out), and the SECOND workgroup of a CU — its waves are the younger ones, the issue arbiter
    // prefers the older — ends 4.5 - 9 us behind the first for the same work, so it gets 7 us less.
    SplitCfg cfg;
    cfg.cost_model = 36;
    cfg.seg_cost = 4900;
    cfg.second_lag = 7000;
    cfg.min_seg = 16;
    cfg.slots = 256 * SCHUR_WG_PER_CU / 8;   // workgroups one XCD holds at once (block b runs on XCD b % 8)
    cfg.n_first = SCHUR_WG_PER_CU == 2 ? cfg.slots / 2 : cfg.slots;
    if (const char* e = ptam_ab_env("PTAM_SCHUR_COST")) cfg.cost_model = atoi(e);   // A/B runs
    if (const char* e = ptam_ab_env("PTAM_SCHUR_SEGCOST")) cfg.seg_cost = atoi(e);
    if (const char* e = ptam_ab_env("PTAM_SCHUR_LAG")) cfg.second_lag = atoi(e);
    cfg.greedy = ptam_ab_env("PTAM_SPLIT_GREEDY") ? 1 : 0;   // (A/B: the greedy fill + budget search, this round's first form)
    cfg.min_room = cfg.seg_cost;   // a workgroup begins another segment only for at least this much work
    if (const char* e = ptam_ab_env("PTAM_SCHUR_MINROOM")) cfg.min_room = atoi(e);
    // Order of the pairs in an XCD's list, both candidates (the device picks: by products only where both slots of the CUs are
    // in use).  By products = the fewest fragment products first: the list's first half becomes the FIRST workgroup of each CU,
    // whose (older) waves the issue arbiter prefers — a load-bound pair there (a last tile of one or two cameras: 3 products a step
    // against 21 loads) leaves the matrix pipe to the younger product-bound partner, while in the second slot it starved behind the
    // partner's products and then ran on alone for 12 us (stamps: docs/LOG_r05.md).
    std::vector<int> pair_order((size_t)2 * std::max(n_pairs, 1), 0);
    {
        std::vector<int> pa(n_pairs), pb(n_pairs);
        for (int a = 0, pr = 0; a < n_tiles; a++)
            for (int b = 0; b <= a; b++, pr++) pa[pr] = a, pb[pr] = b;
        for (int pr = 0; pr < n_pairs; pr++) pair_order[pr] = pair_order[(size_t)n_pairs + pr] = pr;
        std::stable_sort(pair_order.begin() + n_pairs, pair_order.begin() + 2 * (size_t)n_pairs,
                         [&](int p, int q) { return split_full_products(F, pa[p], pb[p]) < split_full_products(F, pa[q], pb[q]); });
    }
    // persistent grid of the accumulate kernel: bounded by LDS residency, 2 x 256 CUs by default
    // (wave variant: + one 3 KB W transposition buffer per wave, K7_WT_DOUBLES)
    ba->k7_big = false;
    ba->det = ba->opts.deterministic != 0 && M > 0;
    auto k7_smem = [&](int threads) {
        const size_t wt = (size_t)(threads / 64) * K7_WT_DOUBLES * sizeof(double);
        if (ba->k7_big) return wt;   // (camera partials and poses in global memory)
        if (ba->det) return ((size_t)C * 12 + 2) * sizeof(double) + wt;   // (no camera partials at all)
        return ((((size_t)F * 27 + 1) & ~(size_t)1) + (size_t)C * 12 + 2) * sizeof(double) + wt;
    };
    // wave variant, two shapes:
    //  - few chunks (every 64-measurement chunk can be resident at once: <= 24 waves per CU):
    //    straight-line kernel, ONE chunk per wave, 1024-thread workgroups (79 VGPRs);
    //  - many chunks: persistent 256-thread workgroups looping over `per_wave` consecutive chunks,
    //    which amortises the LDS prologue and the camera-partial flush.
    const int n64_all = (M + 63) / 64;
    ba->k7_loop = ba->use_wave && n64_all > 256 * 24;
    if (const char* e = ptam_ab_env("PTAM_K7_LOOP")) ba->k7_loop = ba->use_wave && atoi(e) != 0;   // shape sweeps (tools/k7_only.py)
    ba->k7_threads = ba->k7_loop ? 256 : 1024;   // (one chunk per wave: 1024-thread workgroups halve the
                                                                                 //  camera-partial flush — 12.0 vs 12.3 us at 50 x 5000)
    // (a small bundle in 1024-thread workgroups leaves most CUs idle — 20 x 3 000: 58 workgroups, four waves per SIMD on 58 CUs —
    //  and its waves share a SIMD for nothing: 512 threads, 13.5 instead of 14.6 us per launch there; not in deterministic mode,
    //  which has no such instantiation)
    if (!ba->k7_loop && ba->opts.deterministic == 0 && n64_all < 16 * 128) ba->k7_threads = 512;
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    auto k7_fn = [&](int threads) -> const void* { return k7_wave_fn(threads, ba->k7_loop, ba->opts.estimator, ba->k7_big, ba->det); };
    auto k7_occupancy = [&](int threads, int* per_cu) -> int { return ba_k7_occupancy(k7_fn(threads), threads, k7_smem(threads), per_cu); };
    int per_cu = 0;
    {
        // the workgroup's LDS — camera partials F * 216 B + poses C * 96 B + 3 KB per wave — must fit the CU's 160 KB: many
        // cameras take narrower workgroups (fewer transposition buffers), and beyond ~600 free cameras the BIG form, whose
        // partials and poses stay in global memory (ba_jacobian.inc)
        const size_t lds_max = 160 * 1024;
        if (k7_smem(ba->k7_threads) > lds_max) {
            for (int t : {512, 256})
                if (k7_smem(t) <= lds_max) {
                    ba->k7_threads = t;
                    break;
                }
            if (k7_smem(ba->k7_threads) > lds_max) {
                ba->k7_big = true;
                ba->k7_loop = true;
                ba->k7_threads = 256;
            }
        }
        // (the deterministic instantiations are the one-chunk-per-wave form at 1024 threads and the looping form at 256; where the
        //  former does not fit the CU's LDS — the reason a narrower workgroup was picked above — the looping form runs)
        if (ba->det && !ba->k7_big && ba->k7_threads != (ba->k7_loop ? 256 : 1024)) {
            ba->k7_loop = true;
            ba->k7_threads = 256;
        }
    }
    if (int rc = k7_occupancy(ba->k7_threads, &per_cu)) {
        ptam_set_error("the accumulation kernel cannot be launched with %zu bytes of LDS (%d free cameras)", k7_smem(ba->k7_threads), F);
        return rc;
    }
    if (ba->k7_loop && !ba->k7_big && !ba->det) {
        // many cameras: the LDS partials (F*27 + C*12 doubles per workgroup) bound the workgroups per CU,
        // so a wider workgroup keeps more waves resident (200 cameras: 62 KB + 3 KB per wave -> ONE workgroup per CU
        // whatever its width, and only the 1024-thread one fills the four waves per SIMD the loop needs)
        for (int t : {512, 1024}) {
            int per_cu_t = 0;
            if (int rc = k7_occupancy(t, &per_cu_t)) return rc;
            if (per_cu_t * t > per_cu * ba->k7_threads) {
                ba->k7_threads = t;
                per_cu = per_cu_t;
            }
        }
    }
    per_cu = std::max(1, std::min(per_cu, 8));
    if (ba->use_wave && ba->k7_loop && !ba->k7_big) {
        if (const char* e = ptam_ab_env("PTAM_K7_THREADS")) {
            const int t = atoi(e);
            // (A/B override; never 512 in deterministic mode, which has no such instantiation, and never a width whose LDS does not fit)
            if ((t == 256 || t == 512 || t == 1024) && !(ba->det && t != 256) && k7_smem(t) <= 160 * 1024) {
                ba->k7_threads = t;
                if (int rc = k7_occupancy(t, &per_cu)) return rc;
                per_cu = std::max(1, std::min(per_cu, 8));
            }
        }
        if (const char* e = ptam_ab_env("PTAM_K7_WG_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));
    }
    if (ba->use_wave && !ba->k7_loop) {
        if (const char* e = ptam_ab_env("PTAM_K7_THREADS1")) {   // (A/B of the one-chunk-per-wave shape: 512 / 1024-thread workgroups)
            const int t = atoi(e);
            if ((t == 512 || t == 1024) && !(ba->det && t != 1024) && k7_smem(t) <= 160 * 1024) {
                ba->k7_threads = t;
                if (int rc = k7_occupancy(t, &per_cu)) return rc;
            }
        }
    }
    ba->smem_acc = k7_smem(ba->k7_threads);
    if (ba->use_wave) {
        // every wave gets the same number of consecutive 64-measurement chunks
        // one 64-measurement chunk per wave (straight-line kernel body: 68 VGPRs instead of ~160 for the
        // looping form, i.e. every chunk of a 250 k-measurement problem is resident at once)
        const int n64 = (M + 63) / 64;
        const int wpb = ba->k7_threads / 64;
        if (ba->k7_loop) {
            // every resident slot gets a workgroup; chunks are dealt out evenly (q or q + 1 per wave)
            d.grid_acc = std::max(1, std::min(n_cu * per_cu, n64 / wpb));
            const int waves = d.grid_acc * wpb;
            ba->per_wave = n64 / waves;
            ba->extra_waves = n64 - ba->per_wave * waves;
        } else {
            ba->per_wave = 1;
            ba->extra_waves = 0;
            d.grid_acc = std::max(1, (n64 + wpb - 1) / wpb);
        }
    } else
        d.grid_acc = 1;
    const int n_dtiles = (M + DET_TILE - 1) / DET_TILE;
    if (ba->det) {
        static const hipError_t det_attr = hipFuncSetAttribute((const void*)reduce_det_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                               (int)((size_t)14 * (DET_TILE + 1) * sizeof(double)));
        HIP_TRY(det_attr);
    }
    d.u_rows = ba->det ? n_dtiles : (ba->k7_big || !ba->use_wave) ? 1 : d.grid_acc;
    {
        static const hipError_t schur_attr =
            hipFuncSetAttribute((const void*)schur_tile_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SCHUR_RED_BYTES);
        HIP_TRY(schur_attr);
    }
    lap("launch shape");

    // ---- carve the main device allocation (sizes the host knows: M exactly, the points by their upper bound) --------------------
    Carver cv;
    const size_t Mz = std::max(M, 1), Pz = std::max(P_all, 1), Cz = std::max(C, 1), Fz = std::max(F, 1), Maz = std::max(Mall, 1);
    // chunks: consecutive whole points, at most BA_CHUNK measurements — two neighbours together exceed BA_CHUNK measurements or
    // BA_CHUNK points, or one of them is a long point
    const size_t chunks_cap = std::min<size_t>(Pz, (size_t)M / 64 + (size_t)P_all / 128 + 8);
    const size_t o_pose0 = cv.take(Cz * 96), o_pose1 = cv.take(Cz * 96);
    const size_t o_pt0 = cv.take(Pz * 24), o_pt1 = cv.take(Pz * 24), o_V = cv.take(Pz * 48), o_epsB = cv.take(Pz * 24),
                 o_Vinv = cv.take(Pz * 72), o_rowptr = cv.take((Pz + 1) * 4),
                 o_cut = cv.take(ba->use_wave ? (size_t)((M + 63) / 64) * 2 * 72 : 8);
    const size_t o_mcam = cv.take(Mz * 4), o_mpt = cv.take(Mz * 4), o_mfound = cv.take(Mz * 16), o_ms = cv.take(Mz * 8),
                 o_morig = cv.take(Mz * 4), o_mfidx = cv.take(Mz * 4), o_mstate = cv.take(Mz), o_me2 = cv.take(Mz * 8), o_me2t = cv.take(Mz * 8), o_zbad = cv.take(Mz), o_W = cv.take((Mz + 1) * 144);
    const size_t o_U = cv.take(Fz * 27 * 8 * 16), o_Upart = cv.take((size_t)std::max(d.grid_acc, ba->det ? n_dtiles : 1) * Fz * 27 * 8);
    const size_t o_adet = cv.take(ba->det ? Mz * 14 * 8 : 8), o_camptr = cv.take(ba->det ? (size_t)n_dtiles * (Fz + 1) * 4 : 8), o_cammeas = cv.take(ba->det ? Mz * 4 : 8),
                 o_tilefree = cv.take(ba->det ? ((size_t)n_dtiles + 1) * 4 : 8);
    const size_t n_part = std::max<size_t>(chunks_cap, (size_t)d.grid_acc);
    const size_t o_errp = cv.take(n_part * 16 + 16), o_badp = cv.take((size_t)d.grid_acc * 4 + 16);
    const size_t o_chunks = cv.take(chunks_cap * sizeof(BaChunk));
    const size_t o_wchunks = cv.take(sizeof(BaChunk));
    const size_t o_hist = cv.take(2 * HIST_BINS * 4), o_cand = cv.take(Mz * 8);
    const size_t npad = std::max(d.npad, SOLVE_NB);
    // S and L block-banded (bundle.h: se_blk); sized for the full lower triangle, because a sharded bundle only learns the
    // bandwidth in force (the widest over all ranks) in Compute()'s first exchange
    const size_t se_full = se_size((int)(npad / SOLVE_NB), (int)(npad / SOLVE_NB) - 1);
    const size_t o_SE = cv.take((se_full + npad + 8) * 8), o_L = cv.take(se_full * 8), o_Dg = cv.take(npad * 8),
                 o_y = cv.take(npad * 8), o_da = cv.take(npad * 8), o_sq2 = cv.take(16),
                 o_bws = cv.take(npad * 8 * 12), o_sflags = cv.take(ba_solve_flag_bytes((int)(npad / SOLVE_NB))),
                 o_sflags2 = cv.take(ba_solve_flag_bytes((int)(npad / SOLVE_NB))), o_SE2 = cv.take((se_full + npad + 8) * 8),
                 o_L2 = cv.take(se_full * 8), o_Dg2 = cv.take(npad * 8), o_y2 = cv.take(npad * 8);
    const size_t o_out = cv.take(Mz * 4), o_sc = cv.take(sizeof(BaScalars)), o_dbg = cv.take(65536), o_xchg = cv.take(4096);
    // the builder's cleared tables: live measurements per point, tile codes, (XCD, pair, pattern) counts
    const int nw = std::max(1, (n_tiles + 31) / 32);
    const size_t hist_n = (size_t)8 * std::max(n_pairs, 1) * 16;
    const size_t o_cnt = cv.take(Pz * 4), o_tcode = cv.take(Pz * (size_t)nw * 8), o_phist = cv.take(hist_n * 4);
    const size_t clear_bytes = cv.off;   // ---- everything up to here is cleared; what follows is written before it is read ----
    // small tables the host fills, one upload: [initial PrepScalars | poses | free-camera indices | both pair orders]
    Carver sm;
    const size_t s_ps = sm.take(sizeof(PrepScalars)), s_pose = sm.take(Cz * 96), s_camfree = sm.take(Cz * 4),
                 s_porder = sm.take(pair_order.size() * 4);
    const size_t o_small = cv.take(sm.off);
    // raw input (insertion order) and the builder's temporaries
    const size_t o_rdead = cv.take(Maz), o_ptsraw = cv.take(Pz * 24);
    const size_t o_start = cv.take((Pz + 1) * 4), o_denseof = cv.take(Pz * 4), o_ptorig = cv.take(Pz * 4), o_arr = cv.take(Maz * 4),
                 o_tmpi = cv.take(Mz * 4), o_tmpkey = cv.take(Mz * 4), o_ptile = cv.take(Mz * 16), o_kp = cv.take(Pz * 4),
                 o_costp = cv.take(Pz * 4), o_entpre = cv.take((Pz + 1) * 8), o_costpre = cv.take((Pz + 1) * 8), o_ebase = cv.take(hist_n * 4);
    ba->block_bytes = cv.off;
    if (!ctx_cache_take(ctx->dev_cache, CTX_NCACHE(ctx->dev_cache), ba->block_bytes, &ba->block, &ba->block_cap)) {   // (a released bundle's block, if it fits)
        HIP_TRY(hipMalloc(&ba->block, ba->block_bytes));
        ba->block_cap = ba->block_bytes;
    }
    HIP_TRY(hipMemsetAsync(ba->block, 0, clear_bytes, ctx->stream));
    char* base = (char*)ba->block;
    d.pose[0] = (double*)(base + o_small + s_pose);   // (uploaded with the small tables)
    d.pose[1] = (double*)(base + o_pose1);
    (void)o_pose0;
    d.cam_free = (int*)(base + o_small + s_camfree);
    d.pt[0] = (double*)(base + o_pt0);
    d.pt[1] = (double*)(base + o_pt1);
    d.V = (double*)(base + o_V);
    d.epsB = (double*)(base + o_epsB);
    d.Vinv = (double*)(base + o_Vinv);
    d.cut = ba->use_wave ? (double*)(base + o_cut) : nullptr;
    d.rowptr = (int*)(base + o_rowptr);
    d.m_cam = (int*)(base + o_mcam);
    d.m_pt = (int*)(base + o_mpt);
    d.m_found = (double2*)(base + o_mfound);
    d.m_s = (double*)(base + o_ms);
    d.m_orig = (int*)(base + o_morig);
    d.m_fidx = (int*)(base + o_mfidx);
    d.m_state = (uint8_t*)(base + o_mstate);
    d.m_e2 = (double*)(base + o_me2);
    d.m_e2t = (double*)(base + o_me2t);
    d.m_zbad_t = (uint8_t*)(base + o_zbad);
    d.W = (double*)(base + o_W);   // (slot M of every plane stays zero: the block is cleared and nobody writes it)
    d.Usplit = (double*)(base + o_U);
    d.Upart = (double*)(base + o_Upart);
    d.Adet = ba->det ? (double*)(base + o_adet) : nullptr;
    d.cam_ptr = (int*)(base + o_camptr);
    d.cam_meas = (int*)(base + o_cammeas);
    d.err_part = (double*)(base + o_errp);
    d.bad_part = (int*)(base + o_badp);
    d.chunks = (BaChunk*)(base + o_chunks);
    d.wchunks = (BaChunk*)(base + o_wchunks);
    d.hist = (unsigned*)(base + o_hist);
    d.cand = (double*)(base + o_cand);
    d.SE = (double*)(base + o_SE);
    d.L = (double*)(base + o_L);
    d.Dg = (double*)(base + o_Dg);
    d.y = (double*)(base + o_y);
    d.da = (double*)(base + o_da);
    d.sumsq2 = (double*)(base + o_sq2);
    d.bw_scratch = (double*)(base + o_bws);
    d.sflags = (unsigned*)(base + o_sflags);   // (cleared with the block: sequence numbers start at 1)
    d.sflags2 = (unsigned*)(base + o_sflags2);
    d.SE2 = (double*)(base + o_SE2);
    d.L2 = (double*)(base + o_L2);
    d.Dg2 = (double*)(base + o_Dg2);
    d.y2 = (double*)(base + o_y2);
    d.solve_seq = 0;
    d.chain_off = 0;
    {
        // every bundle of the process takes the next XCD for its persistent solves: two bundles adjusting side by side (two
        // contexts / threads) do not compete for the same 32 CUs
        static std::atomic<unsigned> next_xcd{0};
        d.chain_xcd = (int)(next_xcd.fetch_add(1) & 7u);
        static const int lim = [] {
            const char* e = getenv("PTAM_CH_SPIN_LIMIT");
            return e ? std::max(1, atoi(e)) : CH_SPIN_DEFAULT;
        }();
        d.spin_limit = lim;
    }
    d.outliers = (int*)(base + o_out);
    d.sc = (BaScalars*)(base + o_sc);
    d.dbg = (long long*)(base + o_dbg);
    ba->d_xchg = (double*)(base + o_xchg);
    if (!ctx->d_smap) {   // the tile kernel's index maps: a constant of the build, one device copy per context
        static const std::vector<unsigned> s_map = schur_index_map_device();
        HIP_TRY(hipMalloc((void**)&ctx->d_smap, s_map.size() * 4));
        HIP_TRY(hipMemcpy(ctx->d_smap, s_map.data(), s_map.size() * 4, hipMemcpyHostToDevice));
    }
    d.s_map = ctx->d_smap;
    lap("alloc + clear");

    // ---- pinned staging: [stamp | scalars read back | rowptr | dense point ids | small tables | chunks] -------------------------
    Carver hs;
    const size_t h_stamp = hs.take(64), h_psrb = hs.take(sizeof(PrepScalars)), h_rowptr = hs.take((Pz + 1) * 4), h_ptorig = hs.take(Pz * 4),
                 h_small = hs.take(sm.off), h_chunks = hs.take(chunks_cap * sizeof(BaChunk)), h_dead = hs.take(ba->n_dead > 0 ? Maz : 1);
    (void)h_stamp;
    void* pin = nullptr;
    {
        // (also what Compute()'s read-back will need: growing the staging later would re-map host memory under queued kernels)
        const size_t later = 64 + (size_t)C * 96 + (size_t)P_all * 24 + (size_t)M * 4 + 64;
        const int rc_p = ctx_pinned(ctx, std::max(hs.off, later), &pin);
        if (rc_p) return rc_p;
    }
    char* hp = (char*)pin;
    char* dp = (char*)ctx->d_pinned;
    volatile unsigned long long* stamp = (volatile unsigned long long*)hp;
    PrepScalars* ps_host = (PrepScalars*)(hp + h_psrb);
    {
        PrepScalars init;
        std::memset(&init, 0, sizeof init);
        init.dup_key = ~0ull;
        std::memcpy(hp + h_small + s_ps, &init, sizeof init);
        std::memcpy(hp + h_small + s_pose, ba->cam_pose.data(), (size_t)C * 96);
        std::memcpy(hp + h_small + s_camfree, cam_free.data(), (size_t)C * 4);
        std::memcpy(hp + h_small + s_porder, pair_order.data(), pair_order.size() * 4);
    }
#define UP(dst, src, bytes)                                                                        \
    if ((bytes) > 0) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
    UP(base + o_small, hp + h_small, sm.off);
    if (int rc_f = ba->ms.flush(true)) return rc_f;   // (the complete chunks went up while they were added: only the last one is left)
    if (ba->n_dead > 0) {
        std::memcpy(hp + h_dead, ba->m_dead.data(), (size_t)Mall);
        UP(base + o_rdead, hp + h_dead, (size_t)Mall);
    }
    UP(base + o_ptsraw, ba->pts.data(), (size_t)P_all * 24);
    PrepDev q;
    std::memset(&q, 0, sizeof q);
    q.C = C, q.F = F, q.P_all = P_all, q.Mall = Mall, q.M = M;
    q.n_tiles = n_tiles, q.n_pairs = n_pairs, q.nw = nw;
    q.r_base = ba->ms.d;
    q.r_dead = ba->n_dead > 0 ? (const uint8_t*)(base + o_rdead) : nullptr;
    q.pts_raw = (const double*)(base + o_ptsraw);
    q.cnt = (int*)(base + o_cnt);
    q.start = (int*)(base + o_start);
    q.dense_of = (int*)(base + o_denseof);
    q.pt_orig = (int*)(base + o_ptorig);
    q.arr = (int*)(base + o_arr);
    q.tmp_i = (int*)(base + o_tmpi);
    q.tmp_key = (int*)(base + o_tmpkey);
    q.ptile = (int4*)(base + o_ptile);
    q.tcode = (unsigned long long*)(base + o_tcode);
    q.kp = (int*)(base + o_kp);
    q.costp = (int*)(base + o_costp);
    q.ent_pre = (long long*)(base + o_entpre);
    q.cost_pre = (long long*)(base + o_costpre);
    q.hist = (int*)(base + o_phist);
    q.ebase = (int*)(base + o_ebase);
    q.pair_order = (const int*)(base + o_small + s_porder);
    q.ps = (PrepScalars*)(base + o_small + s_ps);
    q.h_ps = (PrepScalars*)(dp + h_psrb);
    q.h_rowptr = (int*)(dp + h_rowptr);
    q.h_pt_orig = (int*)(dp + h_ptorig);
    // ---- phase 1: point-major sort, rowptr, per-point tiles, XCD ranges, pattern counts ------------------------------------------
    if (Mall > 0) hipLaunchKernelGGL(prep_count_kernel, dim3((Mall + 255) / 256), dim3(256), 0, ctx->stream, q);
    hipLaunchKernelGGL(prep_scan_points_kernel, dim3(1), dim3(1024), 0, ctx->stream, q, d);
    if (M > 0) {
        hipLaunchKernelGGL(prep_scatter_kernel, dim3((Mall + 255) / 256), dim3(256), 0, ctx->stream, q, (const int*)d.cam_free);
        hipLaunchKernelGGL(prep_rank_kernel, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, q, d);
    }
    if (lists) {
        hipLaunchKernelGGL(prep_tiles_kernel, dim3((P_all + 63) / 64), dim3(64), 0, ctx->stream, q, d, cfg.cost_model);
        hipLaunchKernelGGL(prep_scan_cost_kernel, dim3(1), dim3(1024), 0, ctx->stream, q, cfg.min_seg, cfg.slots);
        hipLaunchKernelGGL(prep_hist_kernel, dim3(n_pairs, 8), dim3(256), 0, ctx->stream, q);
    }
    unsigned long long seq = ++ctx->pose_seq;
    *stamp = 0;
    hipLaunchKernelGGL(prep_publish_kernel, dim3(1), dim3(64), 0, ctx->stream, q, (volatile unsigned long long*)dp, seq);
    HIP_TRY(hipGetLastError());
    lap("phase 1 enqueued");
    if (int rc = ba_wait_stamp(ctx, stamp, seq)) return rc;
    lap("phase 1 wait");
    PrepScalars ps = *ps_host;
    ba->dups_refused = ps.dup;
    if (ps.dup > 0) {
        ptam_set_error("duplicate measurement of point %d by camera %d", (int)(ps.dup_key >> 32), (int)(ps.dup_key & 0xffffffffu));
        return PTAM_E_ARG;
    }
    if (ps.M != M || ps.P > P_all || ps.n_entries >= (1ll << 31)) {
        ptam_set_error("bundle prepare: inconsistent counts (live measurements %d / %d, points %d / %d, Schur entries %lld)", ps.M, M, ps.P,
                       P_all, ps.n_entries);
        return PTAM_E_STATE;
    }
    const int P = ps.P;
    d.P = P;
    d.band = ba->band_local = ps.band;
    const int* rowptr = (const int*)(hp + h_rowptr);
    ba->pt_orig.assign((const int*)(hp + h_ptorig), (const int*)(hp + h_ptorig) + P);
    // chunks: consecutive whole points, at most BA_CHUNK measurements — or ONE point with more than that (a point seen by
    // more than 256 keyframes: the kernels that own whole points walk such a chunk BA_CHUNK measurements at a time;
    // src/Bundle.cc:75-93 puts no bound on the measurements of a point).  (The one walk over the points the host keeps: a chunk
    // begins where the previous one ends.)
    BaChunk* chunks = (BaChunk*)(hp + h_chunks);
    size_t n_chunks = 0;
    {
        int p = 0;
        while (p < P) {
            BaChunk ch;
            ch.pt_begin = p;
            ch.m_begin = rowptr[p];
            int cnt = 0, np = 0;
            while (p < P && cnt + (rowptr[p + 1] - rowptr[p]) <= BA_CHUNK && np < BA_CHUNK) {
                cnt += rowptr[p + 1] - rowptr[p];
                p++;
                np++;
            }
            if (np == 0) p++;   // a long point, alone in its chunk
            ch.pt_end = p;
            ch.m_end = rowptr[p];
            if (n_chunks >= chunks_cap) {
                ptam_set_error("bundle prepare: more chunks than their bound (%zu)", chunks_cap);
                return PTAM_E_STATE;
            }
            chunks[n_chunks++] = ch;
        }
    }
    d.n_chunks = (int)n_chunks;
    // K7's wave variant walks the point-major list 64 measurements at a time whatever the points' lengths: a point cut by a
    // chunk edge — or covering whole chunks, when more than 64 cameras measure it — leaves one piece per chunk, which K8a adds in
    // chunk order.  (Rounds 1-2 also had a block variant whose workgroups owned whole points: removed in round 3.)
    UP(d.chunks, chunks, n_chunks * sizeof(BaChunk));
    lap("chunks");
    // ---- phase 2: the Schur work lists ----------------------------------------------------------------------------------------
    d.n_schur_entries = (int)ps.n_entries;
    d.n_schur_wg = 0;
    if (lists && ps.n_entries > 0) {
        const int cap_pl = std::max(1, std::min(n_pairs, ps.n_xp));
        const int cap_cut = cfg.slots + cap_pl;
        const size_t cap_segs = (size_t)8 * cfg.slots + (size_t)ps.n_xp;
        const int nb_max = 8 * cfg.slots;
        Carver sv;
        const size_t o_sent = sv.take((size_t)ps.n_entries * sizeof(SchurEntry)), o_swg = sv.take(cap_segs * sizeof(SchurWG)),
                     o_swgseg = sv.take(((size_t)nb_max + 1) * 4), o_spw = sv.take((size_t)(n_pairs + 1) * 4),
                     o_swghead = sv.take((size_t)nb_max * 32), o_spart = sv.take(cap_segs * SCHUR_TILE_ELEMS * 8),
                     o_cpx = sv.take(((size_t)n_pairs * 8 + 1) * 4);
        const size_t sclear = sv.off;
        const size_t o_runcnt = sv.take((size_t)8 * 16 * cap_pl * 4), o_runcost = sv.take((size_t)8 * 16 * cap_pl * 4),
                     o_plpair = sv.take((size_t)8 * cap_pl * 4), o_pln = sv.take((size_t)8 * cap_pl * 4),
                     o_plrun0 = sv.take((size_t)8 * (cap_pl + 1) * 4), o_ple0 = sv.take((size_t)8 * cap_pl * 4),
                     o_ple = sv.take((size_t)8 * (cap_pl + 1) * 4), o_plh = sv.take((size_t)8 * (cap_pl + 1) * 8),
                     o_cutpair = sv.take((size_t)8 * cap_cut * 4), o_cute0 = sv.take((size_t)8 * cap_cut * 4),
                     o_cute1 = sv.take((size_t)8 * cap_cut * 4), o_cutwg = sv.take((size_t)8 * cap_cut * 4),
                     o_wgfirst = sv.take((size_t)8 * (cfg.slots + 1) * 4), o_pairfirst = sv.take((size_t)n_pairs * 8 * 4),
                     o_wsegtmp = sv.take(((size_t)nb_max + 1) * 4);
        // every lane of the budget search keeps the cut its budget makes, unless that would be more than 64 MB
        const size_t rec_bytes = (size_t)8 * 512 * cap_cut * sizeof(int4);
        const bool with_rec = cfg.greedy && rec_bytes <= ((size_t)64 << 20);   // (the greedy form's lanes only)
        const size_t o_rec = with_rec ? sv.take(rec_bytes) : 0;
        ba->sblock_bytes = sv.off;
        if (!ctx_cache_take(ctx->dev_cache, CTX_NCACHE(ctx->dev_cache), ba->sblock_bytes, &ba->sblock, &ba->sblock_cap)) {
            HIP_TRY(hipMalloc(&ba->sblock, ba->sblock_bytes));
            ba->sblock_cap = ba->sblock_bytes;
        }
        HIP_TRY(hipMemsetAsync(ba->sblock, 0, sclear, ctx->stream));
        char* sb = (char*)ba->sblock;
        d.s_entries = (SchurEntry*)(sb + o_sent);
        d.s_segs = (SchurWG*)(sb + o_swg);
        d.s_wg_seg = (int*)(sb + o_swgseg);
        d.s_pair_wg_begin = (int*)(sb + o_spw);
        d.s_wg_head = (int*)(sb + o_swghead);
        d.s_part = (double*)(sb + o_spart);
        q.cap_pl = cap_pl, q.cap_cut = cap_cut;
        q.cpx = (int*)(sb + o_cpx);
        q.run_cnt = (int*)(sb + o_runcnt);
        q.run_cost = (int*)(sb + o_runcost);
        q.pl_pair = (int*)(sb + o_plpair);
        q.pl_n = (int*)(sb + o_pln);
        q.pl_run0 = (int*)(sb + o_plrun0);
        q.pl_e0 = (int*)(sb + o_ple0);
        q.pl_e = (int*)(sb + o_ple);
        q.pl_h = (long long*)(sb + o_plh);
        q.cut_pair = (int*)(sb + o_cutpair);
        q.cut_e0 = (int*)(sb + o_cute0);
        q.cut_e1 = (int*)(sb + o_cute1);
        q.cut_wg = (int*)(sb + o_cutwg);
        q.wg_first = (int*)(sb + o_wgfirst);
        q.pair_first = (int*)(sb + o_pairfirst);
        q.wseg_tmp = (int*)(sb + o_wsegtmp);
        q.rec = with_rec ? (int4*)(sb + o_rec) : nullptr;
        hipLaunchKernelGGL(prep_split_kernel, dim3(8), dim3(512), 0, ctx->stream, q, cfg);
        hipLaunchKernelGGL(prep_entries_kernel, dim3(n_pairs, 8), dim3(256), 0, ctx->stream, q, d, cfg.cost_model);
        hipLaunchKernelGGL(prep_finish_kernel, dim3(1), dim3(1024), 0, ctx->stream, q, d, cfg.slots);
    } else if (F > 0) {
        // free cameras but no entry: the reduction still reads the pairs' (empty) slot ranges
        Carver sv;
        const size_t o_spw = sv.take((size_t)(n_pairs + 1) * 4), o_spart = sv.take(SCHUR_TILE_ELEMS * 8), o_one = sv.take(256);
        ba->sblock_bytes = sv.off;
        if (!ctx_cache_take(ctx->dev_cache, CTX_NCACHE(ctx->dev_cache), ba->sblock_bytes, &ba->sblock, &ba->sblock_cap)) {
            HIP_TRY(hipMalloc(&ba->sblock, ba->sblock_bytes));
            ba->sblock_cap = ba->sblock_bytes;
        }
        HIP_TRY(hipMemsetAsync(ba->sblock, 0, ba->sblock_bytes, ctx->stream));
        char* sb = (char*)ba->sblock;
        d.s_pair_wg_begin = (int*)(sb + o_spw);
        d.s_part = (double*)(sb + o_spart);
        d.s_entries = (SchurEntry*)(sb + o_one);
        d.s_segs = (SchurWG*)(sb + o_one);
        d.s_wg_seg = (int*)(sb + o_one);
        d.s_wg_head = (int*)(sb + o_one);
    }
    if (ba->det) {
        int* tile_free = (int*)(base + o_tilefree);
        hipLaunchKernelGGL(prep_det_count_kernel, dim3(n_dtiles), dim3(256), 0, ctx->stream, d, tile_free);
        hipLaunchKernelGGL(prep_det_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile_free, n_dtiles);
        hipLaunchKernelGGL(prep_det_lists_kernel, dim3(n_dtiles), dim3(256), 0, ctx->stream, d, (const int*)tile_free);
    }
#undef UP
    seq = ++ctx->pose_seq;
    *stamp = 0;
    hipLaunchKernelGGL(prep_publish_kernel, dim3(1), dim3(64), 0, ctx->stream, q, (volatile unsigned long long*)dp, seq);
    HIP_TRY(hipGetLastError());
    lap("phase 2 enqueued");
    if (int rc = ba_wait_stamp(ctx, stamp, seq)) return rc;
    lap("phase 2 wait");
    ps = *ps_host;
    if (ps.bad) {
        ptam_set_error("bundle prepare: a Schur work list outgrew its bound (%d; workgroups per list %d %d %d %d %d %d %d %d, budgets %lld %lld, segments %d %d)", ps.bad,
                       ps.n_wgs[0], ps.n_wgs[1], ps.n_wgs[2], ps.n_wgs[3], ps.n_wgs[4], ps.n_wgs[5], ps.n_wgs[6], ps.n_wgs[7], ps.t_cut[0], ps.t_cut[1],
                       ps.n_cuts[0], ps.n_cuts[1]);
        return PTAM_E_STATE;
    }
    d.n_schur_wg = ps.n_schur_wg;
    ba->n_schur_segs = ps.n_segs;
#ifdef PREP_STAMPS
    for (int x = 0; x < 8; x++) {
        std::fprintf(stderr, "[ptam] split kernel, list %d (us from its start): lists built %.1f | in LDS %.1f | round 1 done %.1f | search done %.1f | cut written %.1f\n", x,
                     (ps.stamp[x][1] - ps.stamp[x][0]) * 0.01, (ps.stamp[x][2] - ps.stamp[x][0]) * 0.01, (ps.stamp[x][3] - ps.stamp[x][0]) * 0.01,
                     (ps.stamp[x][4] - ps.stamp[x][0]) * 0.01, (ps.stamp[x][5] - ps.stamp[x][0]) * 0.01);
        std::fprintf(stderr, "[ptam]    the search: %lld shader cycles in %.1f us = %.2f GHz\n", ps.stamp[x][7] - ps.stamp[x][6], (ps.stamp[x][4] - ps.stamp[x][2]) * 0.01,
                     (double)(ps.stamp[x][7] - ps.stamp[x][6]) / ((ps.stamp[x][4] - ps.stamp[x][2]) * 10.0));
    }
#endif
    if (getenv("PTAM_DEBUG_SCHUR")) {
        std::fprintf(stderr, "[ptam] schur: %d segments, %d workgroups; %lld entries in %d (XCD, pair) lists, budgets", ps.n_segs, ps.n_schur_wg,
                     ps.n_entries, ps.n_xp);
        for (int x = 0; x < 8; x++) std::fprintf(stderr, " %lld", ps.t_cut[x]);
        std::fprintf(stderr, "; workgroups per XCD");
        for (int x = 0; x < 8; x++) std::fprintf(stderr, " %d", ps.n_wgs[x]);
        std::fprintf(stderr, "\n");
    }
    {
        const int rc_s = ba_solve_init();
        if (rc_s) return rc_s;
    }
    ba->cur = 0;
    ba->prepared = true;
    return PTAM_OK;
}

// ---- profiling --------------------------------------------------------------------------------
static void prof_begin(ptam_ba* ba, int k) {
    if (!ba->prof) return;
    hipEventRecord(ba->ev[k][0], ba->ctx->stream);
}
static void prof_end(ptam_ba* ba, int k) {
    if (!ba->prof) return;
    hipEventRecord(ba->ev[k][1], ba->ctx->stream);
    ba->ev_used[k] = true;
}
static void prof_collect(ptam_ba* ba) {   // call after a stream sync
    if (!ba->prof) return;
    for (int k = 0; k < PTAM_K_COUNT; k++)
        if (ba->ev_used[k]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ba->ev[k][0], ba->ev[k][1]) == hipSuccess) {
                ba->k_ms[k] += ms;
                ba->k_n[k]++;
            }
            ba->ev_used[k] = false;
        }
}

static int ba_allreduce(ptam_ba* ba, double* dptr, size_t count) {
    if (!(ba->comm && ba->world > 1)) return PTAM_OK;
    const int rc = ba->comm(ba->comm_user, dptr, count, (void*)ba->ctx->stream);
    if (rc != 0) {
        ptam_set_error("all-reduce hook failed (%d)", rc);
        return PTAM_E_COMM;
    }
    return PTAM_OK;
}

// pass 1 + sigma^2
static int ba_sel_blocks() {   // workgroup cap of select_compact_kernel (each scans the first-level histogram for itself)
    static const int n = [] {
        const char* e = ptam_ab_env("PTAM_SEL_BLOCKS");
        return e ? std::max(1, atoi(e)) : 256;
    }();
    return n;
}
static int ba_pass1_sigma(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    const bool sharded = ba->comm && ba->world > 1;
    const double min_s2 = ba->opts.min_sigma * ba->opts.min_sigma;
    const int build_hist = (sharded && ba->slow_select) ? 0 : 1;   // (the gather-everything path histograms the gathered keys)
    prof_begin(ba, PTAM_K_PROJECT);
    if (ba->trial_is_current && d.M > 0)
        hipLaunchKernelGGL(pass1_from_trial_kernel, dim3(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), 512))), dim3(256), 0, ctx->stream, d,
                           build_hist);
    else if (ba->e2_is_current && d.M > 0) {
        if (build_hist)
            hipLaunchKernelGGL(pass1_keep_kernel, dim3(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), 512))), dim3(256), 0, ctx->stream, d);
    } else if (d.n_chunks > 0)
        hipLaunchKernelGGL(project_e2_kernel, dim3(std::min(d.n_chunks, 512)), dim3(BA_CHUNK), 0, ctx->stream, ctx->cam, d,
                           ba->cur, build_hist);
    prof_end(ba, PTAM_K_PROJECT);
    prof_begin(ba, PTAM_K_SELECT);
    if (!sharded) {
        hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, ba_sel_blocks()))), dim3(256), 0,
                           ctx->stream, d, (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state, 0);
    } else if (!ba->slow_select) {
        if (!ba->d_sel) {   // sized by the world the communicator was set for (ptam_ba_set_comm drops it on a change)
            if (const char* e = ptam_ab_env("PTAM_XCAND_CAP")) ba->xcand_cap = std::max(1, std::min(atoi(e), 1 << 20));
            HIP_TRY(hipMalloc((void**)&ba->d_sel, (HIST_BINS + (size_t)ba->world * (1 + 2 * (size_t)ba->xcand_cap)) * sizeof(double)));
        }
        double* hx = ba->d_sel;                  // histogram exchange
        double* xc = ba->d_sel + HIST_BINS;      // candidate exchange
        const int cap = ba->xcand_cap;
        double* list = xc + (size_t)ba->world * (1 + (size_t)cap);
        const size_t n_xc = (size_t)ba->world * (1 + (size_t)cap);
        auto reduce_hist = [&](unsigned* h) -> int {
            hipLaunchKernelGGL(hist_to_f64_kernel, dim3(HIST_BINS / 256), dim3(256), 0, ctx->stream, (const unsigned*)h, hx, HIST_BINS);
            const int rc = ba_allreduce(ba, hx, HIST_BINS);
            if (rc) return rc;
            hipLaunchKernelGGL(f64_to_hist_kernel, dim3(HIST_BINS / 256), dim3(256), 0, ctx->stream, (const double*)hx, h, HIST_BINS);
            return PTAM_OK;
        };
        int rc = reduce_hist(d.hist);
        if (rc) return rc;
        hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, ba_sel_blocks()))), dim3(256), 0,
                           ctx->stream, d, (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state, 1);
        rc = reduce_hist(d.hist + HIST_BINS);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(xc, 0, n_xc * sizeof(double), ctx->stream));
        hipLaunchKernelGGL(select_stage_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, xc, ba->rank, ba->world, cap);
        rc = ba_allreduce(ba, xc, n_xc);
        if (rc) return rc;
        hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, (const double*)xc, list, ba->world, cap,
                           ba->opts.estimator, min_s2);
        prof_end(ba, PTAM_K_SELECT);
        HIP_TRY(hipGetLastError());
        return PTAM_OK;
    } else {
        // all-gather of the valid e^2 built from two all-reduces (counts, then a zero-padded vector)
        int* d_cnt = (int*)(ba->d_xchg + 256);
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
        hipLaunchKernelGGL(compact_valid_kernel, dim3(std::max(1, std::min((d.M + 255) / 256, 1024))), dim3(256), 0,
                           ctx->stream, d, d.cand, d_cnt);
        int n_local = 0;
        HIP_TRY(hipMemcpyAsync(&n_local, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        std::vector<double> counts(ba->world, 0.0);
        counts[ba->rank] = n_local;
        HIP_TRY(hipMemcpyAsync(ba->d_xchg, counts.data(), ba->world * 8, hipMemcpyHostToDevice, ctx->stream));
        int rc = ba_allreduce(ba, ba->d_xchg, ba->world);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(counts.data(), ba->d_xchg, ba->world * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        long long total = 0, off = 0;
        for (int r = 0; r < ba->world; r++) {
            if (r == ba->rank) off = total;
            total += (long long)(counts[r] + 0.5);
        }
        if ((size_t)total > ba->gather_cap) {
            if (ba->d_gather) HIP_TRY(hipFree(ba->d_gather));
            ba->gather_cap = (size_t)total + (size_t)total / 8 + 1024;
            HIP_TRY(hipMalloc((void**)&ba->d_gather, ba->gather_cap * 16));   // keys + candidate area
        }
        HIP_TRY(hipMemsetAsync(ba->d_gather, 0, (size_t)total * 8, ctx->stream));
        if (n_local > 0)
            hipLaunchKernelGGL(place_keys_kernel, dim3((n_local + 255) / 256), dim3(256), 0, ctx->stream,
                               (const double*)d.cand, n_local, ba->d_gather + off);
        rc = ba_allreduce(ba, ba->d_gather, (size_t)total);
        if (rc) return rc;
        BaDev dg = d;
        dg.cand = ba->d_gather + ba->gather_cap;   // candidates go behind the gathered keys
        if (total > 0)
            hipLaunchKernelGGL(hist_keys_kernel, dim3((int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 1024))),
                               dim3(256), 0, ctx->stream, (const double*)ba->d_gather, total, d.hist);
        hipLaunchKernelGGL(select_compact_kernel, dim3((int)std::max<long long>(1, std::min<long long>((total + 1023) / 1024, 256))),
                           dim3(256), 0, ctx->stream, dg, (const double*)ba->d_gather, total, (const uint8_t*)nullptr, 0);
        hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, dg, ba->opts.estimator, min_s2);
        prof_end(ba, PTAM_K_SELECT);
        HIP_TRY(hipGetLastError());
        return PTAM_OK;
    }
    hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, ba->opts.estimator, min_s2);
    prof_end(ba, PTAM_K_SELECT);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static void launch_k7(ptam_ba* ba, int guard = 0) {
    ptam_ctx* ctx = ba->ctx;
    BaDev d = ba->d;
    d.guard = guard;
    d.guard_seq = (int)ba->mbox_seq;   // (the trial just enqueued)
    int cur = guard ? (ba->cur ^ 1) : ba->cur;   // a guarded launch belongs to the next step: the trial state is current there
    if ((ba->k7_big && !ba->det) || !ba->use_wave)   // the one row of camera partials every wave adds to (or, without measurements, all there is)
        (void)hipMemsetAsync(d.Upart, 0, std::max<size_t>(1, (size_t)d.F * 27) * sizeof(double), ctx->stream);
    if (!ba->use_wave) {   // no live measurement: zero error, zero bad count
        (void)hipMemsetAsync(d.err_part, 0, 16, ctx->stream);
        (void)hipMemsetAsync(d.bad_part, 0, 4, ctx->stream);
        return;
    }
    int est = ba->opts.estimator;
    void* args[] = {&ctx->cam, &d, &cur, &est, &ba->per_wave, &ba->extra_waves};
    (void)hipLaunchKernel(k7_wave_fn(ba->k7_threads, ba->k7_loop, est, ba->k7_big, ba->det), dim3(d.grid_acc), dim3(ba->k7_threads), args,
                          ba->smem_acc, ctx->stream);
    if (ba->det && d.F > 0)
        hipLaunchKernelGGL(reduce_det_kernel, dim3(d.u_rows), dim3(256), (size_t)14 * (DET_TILE + 1) * sizeof(double), ctx->stream, d);
}

static int ba_pass2(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    prof_begin(ba, PTAM_K_JACOBIAN);
    launch_k7(ba);
    prof_end(ba, PTAM_K_JACOBIAN);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(std::max(1, (d.F * 27 + 63) / 64), RSPLIT), dim3(256), 0, ctx->stream,
                       d, d.grid_acc);
    HIP_TRY(hipGetLastError());
    if (ba->comm && ba->world > 1) {
        if (d.F > 0) {
            // current error / bad count ride behind S|E in the step's first camera-system all-reduce
            hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc, se_E(d) + d.npad, 0, 0);
            ba->cur_pending = true;
        } else {
            hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc, ba->d_xchg, 0, 0);
            int rc = ba_allreduce(ba, ba->d_xchg, 2);
            if (rc) return rc;
            hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)ba->d_xchg, 0);
        }
    }
    return PTAM_OK;
}

static int ba_ensure_mailbox(ptam_ba* ba) {
    if (ba->mbox) return PTAM_OK;
    void* h = nullptr;
    size_t cap = 0;
    if (!ctx_cache_take(ba->ctx->host_cache, CTX_NCACHE(ba->ctx->host_cache), sizeof(ptam_ba::Mailbox), &h, &cap))
        HIP_TRY(hipHostMalloc(&h, sizeof(ptam_ba::Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(h, 0, sizeof(ptam_ba::Mailbox));
    void* dv = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dv, h, 0));
    ba->mbox = (ptam_ba::Mailbox*)h;
    ba->mbox_dev = (ptam_ba::Mailbox*)dv;
    return PTAM_OK;
}

static int ba_trial(ptam_ba* ba, double lambda, bool skip_vinv, int last_allowed, int abort_local, bool decide_in_prologue = false) {
    ptam_ctx* ctx = ba->ctx;
    BaDev& d = ba->d;
    prof_begin(ba, PTAM_K_VINV);
    if (d.P > 0 && !skip_vinv) hipLaunchKernelGGL(vinv_kernel, dim3((d.P + 255) / 256), dim3(256), 0, ctx->stream, d, lambda);
    prof_end(ba, PTAM_K_VINV);
    if (d.F > 0) {
        prof_begin(ba, PTAM_K_SCHUR);
        if (d.n_schur_wg > 0)
            hipLaunchKernelGGL(schur_tile_mfma_kernel, dim3(d.n_schur_wg), dim3(64 * SCHUR_NW), SCHUR_RED_BYTES, ctx->stream, d);
        hipLaunchKernelGGL(schur_reduce_kernel, dim3(d.n_pairs, SRED_SLICES), dim3(256), 0, ctx->stream, d, lambda,
                           (ba->world > 1 && ba->rank != 0) ? 0 : 1);
        prof_end(ba, PTAM_K_SCHUR);
        HIP_TRY(hipGetLastError());
        // the path's one exchange step: the in-band lower-triangle blocks of S, then E (+ the step's two scalars) — one buffer
        const size_t n_se = se_size(d.npad / SOLVE_NB, se_band(d)) + d.npad;
        const bool sharded = ba->comm && ba->world > 1;
        if (sharded) prof_begin(ba, PTAM_K_EXCHANGE);
        int rc = ba_allreduce(ba, d.SE, n_se + (ba->cur_pending ? 2 : 0));
        if (sharded) prof_end(ba, PTAM_K_EXCHANGE);
        if (rc) return rc;
        if (ba->cur_pending) {
            hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)(d.SE + n_se), 0);
            ba->cur_pending = false;
        }
        prof_begin(ba, PTAM_K_SOLVE);
        rc = ba_solve(ctx, d, ba->cur);
        prof_end(ba, PTAM_K_SOLVE);
        if (rc) return rc;
    }
    prof_begin(ba, PTAM_K_UPDATE);
    if (d.F == 0)   // no free camera: nothing was solved, the trial poses are copies
        hipLaunchKernelGGL(pose_update_kernel, dim3(std::max(1, (d.C + 63) / 64)), dim3(64), 0, ctx->stream, d, ba->cur);
    if (d.n_chunks > 0)
        hipLaunchKernelGGL(point_update_kernel, dim3(d.n_chunks), dim3(BA_CHUNK), 0, ctx->stream, ctx->cam, d, ba->cur,
                           ba->opts.estimator);
    {
        // single device, no per-kernel events: finalize publishes the scalars itself
        const bool fuse = !(ba->comm && ba->world > 1) && !ba->prof;
        ulonglong2* slots = nullptr;
        unsigned long long seq = 0;
        if (fuse) {
            const int rcm = ba_ensure_mailbox(ba);
            if (rcm) return rcm;
            slots = (ulonglong2*)ba->mbox_dev;
            seq = ++ba->mbox_seq;
            ba->published_by_finalize = true;
        }
        // (with the next step's prologue enqueued behind every trial, the decision is the first thing ITS first launch works out:
        //  purge_pass1_decide_kernel, ba_enqueue_speculative — one dependent launch less per trial)
        ba->decide_pending = decide_in_prologue;
        ba->decide_last_allowed = last_allowed;
        if (!decide_in_prologue)
            hipLaunchKernelGGL(finalize_new_kernel, dim3(1), dim3(256), 0, ctx->stream, d, ba->opts.update_sq_conv_limit, last_allowed,
                               slots, seq);
    }
    prof_end(ba, PTAM_K_UPDATE);
    HIP_TRY(hipGetLastError());
    if (ba->comm && ba->world > 1) {
        hipLaunchKernelGGL(pack2_kernel, dim3(1), dim3(1), 0, ctx->stream, (const BaScalars*)d.sc, ba->d_xchg, 1, abort_local);
        int rc = ba_allreduce(ba, ba->d_xchg, 4);
        if (rc) return rc;
        hipLaunchKernelGGL(unpack2_kernel, dim3(1), dim3(1), 0, ctx->stream, d.sc, (const double*)ba->d_xchg, 1);
    }
    return PTAM_OK;
}

// The next LM step's prologue behind the device-side decision: purge (if the step ended), then — if the trial was
// accepted and the loop goes on — pass 1 from the trial's errors, the select, K7 with the trial state as current, the
// partial reduction and V*^-1 for lambda * 0.3.  Enqueued right after the scalars were published, i.e. while the host
// is still waiting for them.
static int ba_p1_blocks() {   // workgroup cap of the pass-1 kernels (each flushes its LDS histogram with global atomics)
    static const int n = [] {
        const char* e = ptam_ab_env("PTAM_P1_BLOCKS");
        return e ? std::max(1, atoi(e)) : 512;
    }();
    return n;
}
static int ba_enqueue_speculative(ptam_ba* ba, double lambda_next, double lambda_stay) {
    ptam_ctx* ctx = ba->ctx;
    BaDev d = ba->d;
    const double min_s2 = ba->opts.min_sigma * ba->opts.min_sigma;
    d.guard = 1;
    d.guard_seq = (int)ba->mbox_seq;   // (the trial just enqueued: its decision carries this number)
    const dim3 g_p1(std::max(1, std::min((d.M + 256 * P1_U - 1) / (256 * P1_U), ba_p1_blocks())));
    if (ba->decide_pending) {
        hipLaunchKernelGGL(purge_pass1_decide_kernel, g_p1, dim3(256), 0, ctx->stream, d, ba->opts.update_sq_conv_limit, ba->decide_last_allowed,
                           (ulonglong2*)ba->mbox_dev, ba->mbox_seq);
        ba->decide_pending = false;
    } else
        hipLaunchKernelGGL(purge_pass1_kernel, g_p1, dim3(256), 0, ctx->stream, d);
    hipLaunchKernelGGL(select_compact_kernel, dim3(std::max(1, std::min((d.M + 1023) / 1024, ba_sel_blocks()))), dim3(256), 0, ctx->stream, d,
                       (const double*)d.m_e2, (long long)d.M, (const uint8_t*)d.m_state, 0);
    hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(1024), 0, ctx->stream, d, ba->opts.estimator, min_s2);
    launch_k7(ba, 1);
    {
        const int nx = std::max(1, (d.F * 27 + 63) / 64);
        hipLaunchKernelGGL(reduce_vinv_kernel, dim3(nx * RSPLIT + (d.P + 255) / 256), dim3(256), 0, ctx->stream, d, d.grid_acc, nx,
                           lambda_next, lambda_stay);
    }
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static int ba_publish_scalars(ptam_ba* ba);
static int ba_wait_scalars(ptam_ba* ba, BaScalars* out);
static int ba_read_scalars(ptam_ba* ba, BaScalars* out) {
    const int rc = ba_publish_scalars(ba);
    return rc ? rc : ba_wait_scalars(ba, out);
}

static int ba_publish_scalars(ptam_ba* ba) {
    ptam_ctx* ctx = ba->ctx;
    int rc = ba_ensure_mailbox(ba);
    if (rc) return rc;
    if (ba->published_by_finalize) {   // the trial's finalize kernel already wrote sequence number mbox_seq
        ba->published_by_finalize = false;
        return PTAM_OK;
    }
    const unsigned long long seq = ++ba->mbox_seq;
    hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, ctx->stream, (const BaScalars*)ba->d.sc,
                       (ulonglong2*)ba->mbox_dev, seq);
    HIP_TRY(hipGetLastError());
    return PTAM_OK;
}

static int ba_wait_scalars(ptam_ba* ba, BaScalars* out) {
    ptam_ctx* ctx = ba->ctx;
    const unsigned long long seq = ba->mbox_seq;
    constexpr unsigned NW = sizeof(BaScalars) / 8;
    auto arrived = [&]() {
        for (unsigned i = 0; i < NW; i++)
            if (ba->mbox->slot[i].seq != seq) return false;
        return true;
    };
    if (ba->prof) {
        HIP_TRY(ptam_stream_wait(ctx->stream));   // the profiling events must have completed as well
    } else {
        // spin on the sequence words; every now and then ask the runtime whether the stream died instead
        unsigned spins = 0, idle_polls = 0, dbg_polls = 0;
        while (!arrived()) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
                if (getenv("PTAM_DEBUG_WAIT") && (++dbg_polls % 100) == 0) {
                    std::fprintf(stderr, "[ptam] waiting for seq %llu: stream %s, slots", seq, q == hipSuccess ? "drained" : "busy");
                    for (unsigned i = 0; i < NW; i++) std::fprintf(stderr, " %llu", (unsigned long long)ba->mbox->slot[i].seq);
                    std::fprintf(stderr, "\n");
                }
                if (q == hipSuccess && !arrived() && ++idle_polls > 50) {
                    // the stream has drained and nothing published this sequence number: a logic error, not a wait
                    ptam_set_error("scalar mailbox: sequence %llu never arrived (slot 0 holds %llu)", seq,
                                   (unsigned long long)ba->mbox->slot[0].seq);
                    return PTAM_E_STATE;
                }
            }
        }
    }
    unsigned long long words[NW];
    for (;;) {
        std::atomic_thread_fence(std::memory_order_acquire);
        for (unsigned i = 0; i < NW; i++) words[i] = ba->mbox->slot[i].v;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (arrived()) break;   // (cannot change again before the next publish; re-checked for torn 16-byte reads)
    }
    std::memcpy(out, words, sizeof(BaScalars));
    prof_collect(ba);
    return PTAM_OK;
}

// digest the purged-measurement indices of finished Compute() calls: per LM step sorted by insertion index, appended
// to the (point, camera) list, and the measurements marked erased for the next prepare
static void ba_finish_outliers(ptam_ba* ba) {
    int begin = ba->raw_out_done;
    for (size_t k = 0; k < ba->raw_out_ends.size(); k++) {
        const int end = ba->raw_out_ends[k];
        if (end <= begin) continue;
        std::sort(ba->raw_out.begin() + begin, ba->raw_out.begin() + end);
        for (int i = begin; i < end; i++) {
            const int o = ba->raw_out[i];
            ba->outliers.push_back(std::make_pair(ba->ms.pt((size_t)o), ba->ms.cam((size_t)o)));
            if (!ba->m_dead[o]) ba->n_dead++;
            ba->m_dead[o] = 1;
        }
        begin = end;
    }
    ba->raw_out_done = begin;
    ba->raw_out_ends.clear();
}

extern "C" {

void ptam_ba_opts_default(ptam_ba_opts* o) {
    if (!o) return;
    o->max_iterations = 20;
    o->update_sq_conv_limit = 1e-6;
    o->min_sigma = 0.4;
    o->estimator = PTAM_EST_TUKEY;
    o->verbose = 0;
    o->deterministic = 0;
    o->pad_ = 0;
}

int ptam_ba_create(ptam_ctx* ctx, const ptam_ba_opts* opts, ptam_ba** out) {
    ARG_TRY(ctx && out);
    ptam_ba* ba = new ptam_ba();
    ba->ctx = ctx;
    ba->pts.ctx = ba->ms.ctx = ctx;
    if (opts)
        ba->opts = *opts;
    else
        ptam_ba_opts_default(&ba->opts);
    std::memset(&ba->d, 0, sizeof ba->d);
    std::memset(ba->k_ms, 0, sizeof ba->k_ms);
    std::memset(ba->k_n, 0, sizeof ba->k_n);
    std::memset(ba->ev_used, 0, sizeof ba->ev_used);
    *out = ba;
    return PTAM_OK;
}

int ptam_ba_destroy(ptam_ba* ba) {
    if (!ba) return PTAM_OK;
    hipSetDevice(ba->ctx->device);
    ptam_stream_wait(ba->ctx->stream);
    ba_free_device(ba);
    if (ba->mbox) {
        void* drop = ctx_cache_give(ba->ctx->host_cache, CTX_NCACHE(ba->ctx->host_cache), ba->mbox, sizeof(ptam_ba::Mailbox));
        if (drop) hipHostFree(drop);
    }
    if (ba->ev_ok)
        for (int k = 0; k < PTAM_K_COUNT; k++) {
            hipEventDestroy(ba->ev[k][0]);
            hipEventDestroy(ba->ev[k][1]);
        }
    ba->pts.release();
    ba->ms.release_host();
    ba->ms.release_dev();
    delete ba;
    return PTAM_OK;
}

int ptam_ba_add_camera(ptam_ba* ba, const double pose[12], int fixed) {
    ARG_TRY(ba && pose);
    const int n = (int)ba->cam_fixed.size();
    ba->cam_pose.insert(ba->cam_pose.end(), pose, pose + 12);
    ba->cam_fixed.push_back(fixed ? 1 : 0);
    ba->prepared = false;
    return n;
}

int ptam_ba_add_point(ptam_ba* ba, const double pos[3]) {
    ARG_TRY(ba && pos);
    const int n = (int)(ba->pts.size() / 3);
    double v[3] = {pos[0], pos[1], pos[2]};
    if (std::isnan(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])) v[0] = v[1] = v[2] = 0;   // src/Bundle.cc:70-74
    if (int rc = ba->pts.append(v, 3)) return rc;
    ba->prepared = false;
    return n;
}

int ptam_ba_add_meas(ptam_ba* ba, int cam, int point, const double found[2], double sigma_sq) {
    ARG_TRY(ba && found);
    ARG_TRY(cam >= 0 && cam < (int)ba->cam_fixed.size());
    ARG_TRY(point >= 0 && point < (int)(ba->pts.size() / 3));
    MeasStore& ms = ba->ms;
    if (int rc = ms.reserve_host((ms.n >> MS_LOG) + 1)) return rc;
    const size_t i = ms.n;
    ms.cam(i) = cam;
    ms.pt(i) = point;
    ms.found(i) = make_double2(found[0], found[1]);
    ms.sig(i) = sigma_sq;   // (dSqrtInvNoise = sqrt(1 / sigma^2), src/Bundle.cc:91, is formed when the measurements are sorted: prep_rank_kernel)
    ms.n++;
    ba->m_dead.push_back(0);
    ba->prepared = false;
    if ((ms.n & (MS_CH - 1)) == 0) return ms.flush(false);   // a chunk is full: on its way while the caller adds the next
    return PTAM_OK;
}

int ptam_ba_add_cameras(ptam_ba* ba, int n, const double* poses12, const uint8_t* fixed) {
    ARG_TRY(ba && n >= 0 && (n == 0 || (poses12 && fixed)));
    for (int i = 0; i < n; i++) ptam_ba_add_camera(ba, poses12 + 12 * i, fixed[i]);
    return PTAM_OK;
}
int ptam_ba_add_points(ptam_ba* ba, int n, const double* pos3) {
    ARG_TRY(ba && n >= 0 && (n == 0 || pos3));
    for (int i = 0; i < n; i++) ptam_ba_add_point(ba, pos3 + 3 * i);
    return PTAM_OK;
}
int ptam_ba_add_measurements(ptam_ba* ba, int n, const int32_t* cam, const int32_t* point, const double* found2,
                             const double* sigma_sq) {
    ARG_TRY(ba && n >= 0 && (n == 0 || (cam && point && found2 && sigma_sq)));
    const int n_cams = (int)ba->cam_fixed.size(), n_pts = (int)(ba->pts.size() / 3);
    for (int i = 0; i < n; i++) {
        ARG_TRY(cam[i] >= 0 && cam[i] < n_cams);
        ARG_TRY(point[i] >= 0 && point[i] < n_pts);
    }
    if (n == 0) return PTAM_OK;
    MeasStore& ms = ba->ms;
    if (int rc = ms.reserve_host(((ms.n + (size_t)n - 1) >> MS_LOG) + 1)) return rc;
    ba->m_dead.resize(ba->m_dead.size() + (size_t)n, 0);
    ba->prepared = false;
    // chunk by chunk: a full chunk's upload runs while the next one is being filled
    for (size_t done = 0; done < (size_t)n;) {
        const size_t i = ms.n, room = MS_CH - (i & (MS_CH - 1)), k = std::min(room, (size_t)n - done);
        std::memcpy(&ms.cam(i), cam + done, k * 4);
        std::memcpy(&ms.pt(i), point + done, k * 4);
        std::memcpy(&ms.found(i), found2 + 2 * done, k * 16);
        std::memcpy(&ms.sig(i), sigma_sq + done, k * 8);
        ms.n += k;
        done += k;
        if ((ms.n & (MS_CH - 1)) == 0)
            if (int rc = ms.flush(false)) return rc;
    }
    return PTAM_OK;
}

static int ba_ensure_mailbox(ptam_ba* ba);
int ptam_ba_prepare(ptam_ba* ba) {
    ARG_TRY(ba);
    if (ba->prepared) return PTAM_OK;
    int rc = ba_prepare_impl(ba);
    if (rc) return rc;
    // Every allocation Compute() would otherwise make on first use happens here: the mailbox and the pinned read-back
    // staging.  Mapping new host memory into the GPU's address space while kernels are queued makes the driver evict and
    // restore the queues — measured as a 10-28 ms wait for the first trial's scalars in about every third Compute().
    rc = ba_ensure_mailbox(ba);
    if (rc) return rc;
    void* pin = nullptr;
    return ctx_pinned(ba->ctx, 64 + (size_t)ba->d.C * 96 + (size_t)ba->d.P * 24 + (size_t)ba->d.M * 4 + 64, &pin);
}

int ptam_ba_set_profiling(ptam_ba* ba, int on) {
    ARG_TRY(ba);
    HIP_TRY(hipSetDevice(ba->ctx->device));
    if (on && !ba->ev_ok) {
        for (int k = 0; k < PTAM_K_COUNT; k++) {
            HIP_TRY(hipEventCreate(&ba->ev[k][0]));
            HIP_TRY(hipEventCreate(&ba->ev[k][1]));
        }
        ba->ev_ok = true;
    }
    ba->prof = on != 0;
    return PTAM_OK;
}

int ptam_ba_kernel_time(const ptam_ba* ba, int kernel, double* total_ms, int* launches) {
    ARG_TRY(ba && kernel >= 0 && kernel < PTAM_K_COUNT);
    if (total_ms) *total_ms = ba->k_ms[kernel];
    if (launches) *launches = ba->k_n[kernel];
    return PTAM_OK;
}

int ptam_ba_set_comm(ptam_ba* ba, int rank, int world, ptam_allreduce_f64_fn fn, void* user) {
    ARG_TRY(ba && world >= 1 && rank >= 0 && rank < world);
    ARG_TRY(world == 1 || fn);
    ARG_TRY(world <= 32);
    if (ba->d_sel && world != ba->world) {   // the select's exchange buffer is sized by the world
        HIP_TRY(hipSetDevice(ba->ctx->device));
        HIP_TRY(ptam_stream_wait(ba->ctx->stream));
        HIP_TRY(hipFree(ba->d_sel));
        ba->d_sel = nullptr;
    }
    ba->rank = rank;
    ba->world = world;
    ba->comm = fn;
    ba->comm_user = user;
    return PTAM_OK;
}

// Bundle::Compute src/Bundle.cc:116-158
#define BA_DBG(...) do { if (dbg_) { std::fprintf(stderr, "[ptam] " __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)
int ptam_ba_compute(ptam_ba* ba, const volatile unsigned char* abort_flag, int* accepted_out) {
    ARG_TRY(ba);
    ptam_ctx* ctx = ba->ctx;
    const bool dbg_ = getenv("PTAM_DEBUG_WAIT") != nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    BA_DBG("compute: prepare");
    const bool sharded = ba->comm && ba->world > 1;
    int rc = ptam_ba_prepare(ba);
    if (rc && !sharded) return rc;
    const int rc_prepare = rc;   // sharded: a local failure is made unanimous by the first exchange below
    BA_DBG("compute: prepared M=%d P=%d F=%d chunks=%d wchunks=%d grid_acc=%d schur_wg=%d", ba->d.M, ba->d.P, ba->d.F, ba->d.n_chunks,
           ba->d.n_wchunks, ba->d.grid_acc, ba->d.n_schur_wg);
    BaDev& d = ba->d;
    // The persistent camera solve spins on flags between workgroups that must all be resident.  One bundle's launch is a dozen
    // workgroups on an idle XCD; two bundles of this process adjusting on one device at the same moment could each be granted
    // part of theirs.  So only the first one in uses that form; a second one takes the launch-per-block-column form for this call.
    struct ChainTurn {
        static std::atomic<int>& busy(int dev) {
            static std::atomic<int> b[64];
            return b[dev & 63];   // (device ids 64 apart would share a counter: one node has at most 8)
        }
        BaDev& d;
        int dev, was;
        bool forced;
        ChainTurn(BaDev& d_, int dev_) : d(d_), dev(dev_), was(d_.chain_off), forced(false) {
            if (busy(dev).fetch_add(1) > 0 && !d.chain_off) {
                d.chain_off = 1;
                forced = true;
            }
        }
        ~ChainTurn() {
            busy(dev).fetch_sub(1);
            if (forced && d.chain_off == 1) d.chain_off = was;   // (a fault during the call sets 2: that one stays)
        }
    } chain_turn(d, ctx->device);
    // (PTAM_TWO_QUEUES=1 only, see below.)  A REJECTED trial's continuation — the same step again with a larger lambda — does not
    // queue behind the guarded kernels that were enqueued for the other outcome (each leaves at once, but a kernel that only reads
    // its guard still costs its launch and its first memory access: ~4.4 us each): it goes to the context's second queue, which is empty.  Nothing on the queue left behind writes (the guards' sequence word also stops a leftover that is late), the
    // trial's own kernels were all complete when its finalize kernel published the verdict the host has just read.  The context's
    // queues are swapped for the rest of this call and put back at its end.
    struct QueueTurn {
        ptam_ctx* c;
        bool flipped = false, ever = false;
        void flip() {
            std::swap(c->stream, c->stream_alt);
            flipped = !flipped;
            ever = true;
        }
        ~QueueTurn() {
            // (whatever is left on the queue that is not the current one are guarded kernels of trials long decided: wait them out —
            //  a query in practice — so that none of them outlives the call and meets the block in another bundle's hands)
            if (ever) (void)ptam_stream_wait(c->stream_alt);
            if (flipped) std::swap(c->stream, c->stream_alt);
        }
    } queue_turn{ctx};
    double lambda = 0.0001, lambda_factor = 2.0;   // :125-126
    ba->converged = false;
    ba->published_by_finalize = false;
    ba->cur_pending = false;
    ba->slow_select = false;
    ba->trial_is_current = false;
    ba->e2_is_current = false;
    bool hit_max = false;
    int counter = 0;
    ba->accepted = 0;
    ba->trials.clear();
    std::vector<int> step_outlier_end;   // outlier-list length after every LM step
    int n_steps = 0;
    bool prev_end_pending = false;   // the previous step's outlier-list length has not been read yet
    // The abort flag (src/Bundle.cc:134,338).  Sharded, every trial runs collectives, so the decision must be the same on
    // every rank at the same trial: the local flag is sampled when a trial is enqueued, summed over the ranks with the
    // trial's scalars, and only the summed value (abort_all) is acted upon — one trial later than a local test would.
    bool abort_all = false;
    auto abort_local = [&]() { return abort_flag && *abort_flag; };
    auto aborted = [&]() { return sharded ? abort_all : abort_local(); };
    BaScalars sc;
    std::memset(&sc, 0, sizeof sc);
    double dbg_enq_ms = 0, dbg_wait_ms = 0;   // host time spent enqueueing trials / waiting for their scalars (PTAM_DEBUG_STALL)
    static const bool dbg_times = getenv("PTAM_DEBUG_TRIAL_TIMES") != nullptr;   // print when each trial's verdict reached the host
    std::vector<double> trial_read_us;
    const auto dbg_t0 = std::chrono::steady_clock::now();
#ifdef K7_TIMING
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ht0 = now_us();
    double ht_first = 0;
    (void)hipMemsetAsync(ba->d.dbg + TL_BASE, 0, 8, ba->ctx->stream);
#endif
    if (sharded) {
        // every rank must walk the same sequence of collectives: a rank without measurements, or one whose prepare failed
        // (a point with more than BA_CHUNK cameras, a duplicate measurement, an allocation failure — all of them local to a
        // shard), would return at once and the others would wait for it for ever.  One tiny all-reduce up front makes the
        // refusal unanimous and carries the first abort sample.
        // The same exchange settles the block bandwidth of the camera system: every rank marks the bandwidth its own
        // points need (one-hot, the collective only sums), the widest one wins — S is the sum of all ranks' parts.
        // (the layout of this vector only depends on the cameras, which every rank holds, never on the outcome of prepare)
        int n_free = 0;
        for (uint8_t x : ba->cam_fixed) n_free += x ? 0 : 1;
        const int nblk_x = (6 * n_free + SOLVE_NB - 1) / SOLVE_NB;
        // ... and the FORM of the replicated camera solve: a rank that may not use the persistent launch for this call (another
        // bundle of its process is adjusting on the same device — ranks that are threads of one process sharing a GPU —, or an
        // earlier solve of it timed out) says so, and then NO rank uses it: every rank solves with the same code, the pose copies
        // stay bit-identical (the two forms agree to 1e-6 only)
        constexpr int XF = 5;   // fixed slots in front of the bandwidth marks
        const bool band_fits = nblk_x + XF <= 512;   // d_xchg holds 512 doubles
        std::vector<double> mine(band_fits ? XF + nblk_x : XF, 0.0), all(mine.size(), 0.0);
        mine[0] = (!rc_prepare && d.M == 0) ? 1.0 : 0.0;
        mine[1] = (double)d.M;
        mine[2] = rc_prepare ? 1.0 : 0.0;
        mine[3] = abort_local() ? 1.0 : 0.0;
        mine[4] = d.chain_off ? 1.0 : 0.0;
        if (!rc_prepare && band_fits && nblk_x > 0) mine[XF + std::min(ba->band_local, nblk_x - 1)] = 1.0;
        if (!ba->d_xchg) {   // (a failed prepare has not carved it)
            HIP_TRY(hipMalloc((void**)&ba->d_xchg, 4096));
            ba->xchg_owned = true;
        }
        HIP_TRY(hipMemcpyAsync(ba->d_xchg, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = ba_allreduce(ba, ba->d_xchg, mine.size());
        if (rc) return rc_prepare ? rc_prepare : rc;
        HIP_TRY(hipMemcpyAsync(all.data(), ba->d_xchg, all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ptam_stream_wait(ctx->stream));
        if (all[2] > 0.5) {
            if (rc_prepare) return rc_prepare;   // (this rank's own error message stands)
            ptam_set_error("sharded bundle: %d of %d ranks could not prepare their shard; nothing was computed on any rank",
                           (int)(all[2] + 0.5), ba->world);
            return PTAM_E_STATE;
        }
        abort_all = all[3] > 0.5;
        if (all[4] > 0.5 && !d.chain_off) {   // (for this call only: chain_turn puts a forced value back)
            d.chain_off = 1;
            chain_turn.forced = true;
        }
        ba->d.band = std::max(0, nblk_x - 1);
        if (band_fits)
            for (int b = nblk_x - 1; b >= 0; b--)
                if (all[XF + b] > 0.5) {
                    ba->d.band = b;
                    break;
                }
        if (all[0] > 0.5) {
            ptam_set_error("sharded bundle: %d of %d ranks hold no measurement (shard the points so that every rank gets some)",
                           (int)(all[0] + 0.5), ba->world);
            return PTAM_E_STATE;
        }
    }
    const bool empty = d.M == 0;
    // speculative step prologue (ba_enqueue_speculative): single device, not while per-kernel events are being taken
    const bool spec = !sharded && !ba->prof && d.n_chunks > 0 && !ptam_ab_env("PTAM_NO_SPECULATION");
    bool spec_ready = false;   // pass 1 .. V*^-1 of the coming step are already running behind the device-side flag
    // Two ways to shorten the chain behind a trial, and they exclude each other.  (1) The trial's decision inside the next step's first
    // launch (default): one dependent launch less per trial, 2.3 us.  (2) PTAM_TWO_QUEUES=1: a rejected trial's continuation on the
    // context's second queue, 6 us per REJECTED trial — safe only while the kernel that publishes the verdict is over when the host
    // reads it (finalize_new_kernel: one workgroup, the mailbox is its last act).  purge_pass1_decide_kernel publishes from
    // workgroup 0 while its other workgroups may still be summing: a continuation on another queue could overwrite those sums
    // (the solve's |da|^2, the point update's per-chunk errors) under a workgroup that is late, which would then decide differently.
    // On one queue everything is in order.
    static const bool two_queues = getenv("PTAM_TWO_QUEUES") != nullptr;
    const bool fused_decision = !two_queues && !ptam_ab_env("PTAM_SEPARATE_FINALIZE");   // (A/B: the decision as a launch of its own, rounds 2-5)
    while (!empty && !ba->converged && !hit_max && !aborted()) {
        // ---- Do_LM_Step :209-551 ----
        bool skip_vinv = false;
        if (spec_ready) {
            spec_ready = false;
            skip_vinv = true;
        } else {
            rc = ba_pass1_sigma(ba);
            if (rc) return rc;
            rc = ba_pass2(ba);
            if (rc) return rc;
        }
        bool have_cur = false;
        double cur_err = 0, new_err = 0;
        bool ran_any = false, redo_step = false, nan_stop = false;
        // while(dNewError > dCurrentError && !converged && !hitmax && !abort)  :338
        for (;;) {
            if (have_cur && !(new_err > cur_err)) break;
            if (ba->converged || hit_max || aborted()) break;
            BA_DBG("trial %d lambda %g", counter, lambda);
            const auto q0 = std::chrono::steady_clock::now();
            rc = ba_trial(ba, lambda, skip_vinv, counter + 1 >= ba->opts.max_iterations ? 1 : 0, abort_local() ? 1 : 0, spec && fused_decision);
            {
                const double qms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - q0).count();
                dbg_enq_ms += qms;
                if (qms > 3.0 && getenv("PTAM_DEBUG_STALL"))
                    std::fprintf(stderr, "[ptam] stall: trial %d took %.2f ms to enqueue\n", counter, qms);
            }
            skip_vinv = false;
            if (rc) return rc;
            rc = ba_publish_scalars(ba);
            if (rc) return rc;
            if (spec) {   // what follows an accepted trial, queued while the host waits for the verdict
                rc = ba_enqueue_speculative(ba, lambda * 0.3, lambda);
                if (rc) return rc;
            }
            BA_DBG("trial %d enqueued, waiting", counter);
            {
                const auto w0 = std::chrono::steady_clock::now();
                rc = ba_wait_scalars(ba, &sc);
                const double wms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                dbg_wait_ms += wms;
                if (wms > 3.0 && getenv("PTAM_DEBUG_STALL"))
                    std::fprintf(stderr, "[ptam] stall: trial %d waited %.2f ms for its scalars\n", counter, wms);
            }
            if (rc) return rc;
            BA_DBG("trial %d read: cur %g new %g sigma2 %g median %g n_valid %lld n_bad %d sumsq %g %g", counter, sc.cur_err, sc.new_err,
                   sc.sigma_sq, sc.median, sc.n_valid, sc.n_bad, sc.sumsq_cam, sc.sumsq_pt);
#ifdef K7_TIMING
            if (ht_first == 0) ht_first = now_us();
#endif
            if (!have_cur) {
                // (every step runs at least one trial: this first read of step s also carries the outlier-list
                //  length left by the purge that closed step s-1 — no separate read-back for it)
                // (n_outliers_pub: the list's length as the last select found it — with the decision inside the step-closing launch the
                //  live counter may already hold part of THIS trial's purge)
                if (prev_end_pending) step_outlier_end.push_back(spec && fused_decision ? sc.n_outliers_pub : sc.n_outliers);
                prev_end_pending = false;
                if (sc.select_overflow && !ba->slow_select) {
                    // sharded select: a rank's exchange slot overflowed (thousands of bit-identical errors), so this
                    // step ran with a wrong sigma^2.  Nothing has been committed: repeat it with the gather path.
                    ba->slow_select = true;
                    if (getenv("PTAM_DEBUG_SELECT")) std::fprintf(stderr, "[ptam] rank %d: select overflow, repeating the step on the gather path\n", ba->rank);
                    redo_step = true;
                    break;
                }
                have_cur = true;
                cur_err = sc.cur_err;
                new_err = cur_err + 9999;   // :337
                if (!(new_err > cur_err)) {
                    // NaN / inf current error: the reference never enters its trial loop and then calls Do_LM_Step again
                    // for ever (src/Bundle.cc:118-123 has no exit for it).  A library must not hang: give up here.
                    nan_stop = true;
                    break;
                }
            }
            if (sc.solve_fault) {
                // A workgroup of the persistent factorisation gave up waiting for another one (ldlt_chain.inc): it cannot happen
                // while all of them are resident — a launch of a dozen workgroups on an otherwise idle XCD — but a device shared
                // with another process' persistent solve can leave each with half of its workgroups.  Nothing was committed (the
                // trial's finalize kernel saw the fault and armed none of the guarded kernels; a sharded bundle has summed the
                // fault over the ranks, so every rank is here): the rest of this adjustment uses the launch-per-block-column form,
                // which waits for nobody, and the trial is run again.
                if (d.chain_off) {   // (that form raises no fault: something else is wrong)
                    ptam_set_error("bundle adjustment: the camera solve reports a fault in its launch-per-block form");
                    return PTAM_E_HIP;
                }
                d.chain_off = 2;
                ba->solve_fallbacks++;
                HIP_TRY(hipMemsetAsync(&d.sc->solve_fault, 0, sizeof(int), ctx->stream));
                // (the launch's error word too: where the right-hand-side workgroup substitutes backwards inside the launch it takes
                //  the word down itself, and a workgroup that gives up AFTER that would leave it up for the per-column form's
                //  backward kernel to find; behind the launch, in stream order, nobody is left to raise it)
                if (d.sflags) HIP_TRY(hipMemsetAsync(d.sflags, 0, sizeof(unsigned), ctx->stream));
                if (getenv("PTAM_DEBUG_SOLVE")) std::fprintf(stderr, "[ptam] rank %d: persistent camera solve timed out at trial %d; repeating it with the launch-per-column form\n", ba->rank, counter);
                continue;
            }
            ran_any = true;
            if (dbg_times) trial_read_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
            if (sharded) abort_all = abort_all || sc.abort_any != 0;
            new_err = sc.new_err;
            const double sumsq = sc.sumsq_cam + sc.sumsq_pt;
            if (sumsq < ba->opts.update_sq_conv_limit) ba->converged = true;   // :488-490
            ptam_ba_trial t;
            t.lambda = lambda;
            t.sigma_sq = sc.sigma_sq;
            t.err_old = cur_err;
            t.err_new = new_err;
            t.sum_sq_update = sumsq;
            t.n_bad = sc.n_bad;
            t.accepted = 0;
            if (ba->opts.verbose)
                std::printf("L%.1e\tOld %.6f  New %.6f  Diff %.6f\n", lambda, cur_err, new_err, cur_err - new_err);
            if (new_err > cur_err) {   // ModifyLambda_BadStep :607-611
                lambda = lambda * lambda_factor;
                lambda_factor = lambda_factor * 2;
                if (spec && two_queues && ctx->stream_alt && !ba->converged && counter + 1 < ba->opts.max_iterations && !aborted())
                    queue_turn.flip();   // (the retry starts on the empty queue)
            }
            counter++;
            if (counter >= ba->opts.max_iterations) hit_max = true;   // :518-520
            ba->trials.push_back(t);
        }
        if (redo_step) continue;   // (keeps trial_is_current: pass 1 may still adopt the previous trial's errors)
        if (nan_stop) hit_max = true;
        ba->trial_is_current = false;
        ba->e2_is_current = ran_any && !(new_err < cur_err);   // nothing committed: the step's own pass 1 still describes the state
        if (ran_any && new_err < cur_err) {   // :523-533
            lambda_factor = 2.0;
            lambda *= 0.3;
            ba->cur ^= 1;   // commit: trial poses / points become current
            ba->accepted++;
            ba->trials.back().accepted = 1;
            ba->trial_is_current = true;
        }
        if (prev_end_pending) {   // a step that ran no trial (abort raised in between): read the length the slow way
            rc = ba_read_scalars(ba, &sc);
            if (rc) return rc;
            step_outlier_end.push_back(sc.n_outliers);
        }
        // the device took the same decision for the guarded kernels: accepted and the loop goes on -> they ran
        // (purge of this step included); otherwise they left at once and the step is closed here
        // ... or took the third way: the step ended with new == current error (nothing accepted, nothing rejected — the
        // update no longer moves a coordinate): the guarded kernels ran for the UNCHANGED state and lambda ("stay")
        const bool stayed = spec && ran_any && !(new_err < cur_err) && !(new_err > cur_err) && !ba->converged && !hit_max;
        spec_ready = (spec && ran_any && new_err < cur_err && !ba->converged && !hit_max) || stayed;
        if (!spec_ready && d.M > 0)
            hipLaunchKernelGGL(purge_kernel, dim3((d.M + 255) / 256), dim3(256), 0, ctx->stream, d);   // :536-547
        prev_end_pending = true;
        HIP_TRY(hipGetLastError());
        n_steps++;
    }
    BA_DBG("loop done, steps %d", n_steps);
    if (dbg_times && !trial_read_us.empty()) {
        // per trial: a(ccepted) / r(ejected) / s(tay: new == old error) and the time since the previous verdict (the first: since the call began)
        std::fprintf(stderr, "[ptam] trial times (us):");
        for (size_t i = 0; i < trial_read_us.size() && i < ba->trials.size(); i++) {
            const ptam_ba_trial& t = ba->trials[i];
            std::fprintf(stderr, " %c%.1f", t.accepted ? 'a' : (t.err_new > t.err_old ? 'r' : 's'), trial_read_us[i] - (i ? trial_read_us[i - 1] : 0.0));
        }
        std::fprintf(stderr, "\n");
    }
    if (getenv("PTAM_DEBUG_STALL"))
        std::fprintf(stderr, "[ptam] compute loop: %zu trials in %.3f ms; host enqueueing trials %.3f ms, waiting for scalars %.3f ms\n",
                     ba->trials.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count(),
                     dbg_enq_ms, dbg_wait_ms);
    if (n_steps > 0) {   // the last step's purge
        rc = ba_read_scalars(ba, &sc);
        if (rc) return rc;
        step_outlier_end.push_back(sc.n_outliers);
    }
#ifdef K7_TIMING
    const double ht_loop = now_us();
#endif
#ifdef K7_TIMING
    {
        long long h[16];
        HIP_TRY(hipMemcpy(h, d.dbg, sizeof h, hipMemcpyDeviceToHost));
        std::printf("LDLT step2 wg0: loop %lld tail %lld | last wg (role %lld): loop %lld tail %lld\n", h[1] - h[0], h[2] - h[1], h[7],
                    h[5] - h[4], h[6] - h[5]);
        long long q[8];
        HIP_TRY(hipMemcpy(q, d.dbg + 32, sizeof q, hipMemcpyDeviceToHost));
        std::printf("LDLT step2 wg0 iteration 3 (cycles): micro factor %lld | my rows %lld | next panel columns %lld | publish %lld | barrier %lld | "
                    "operand reads issued %lld | off-chain updates %lld\n", q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4], q[6] - q[5], q[7] - q[6]);
    }
#endif
    // ---- read back results: one pinned staging buffer, one synchronisation (pageable destinations cost ~100 us each) ----
    const int n_out = step_outlier_end.empty() ? 0 : step_outlier_end.back();
    std::vector<int> out_idx(std::max(n_out, 1));
    {
        const size_t b_pose = (size_t)d.C * 96, b_pts = (size_t)d.P * 24, b_out = (size_t)n_out * 4;
        void* pin = nullptr;
        rc = ctx_pinned(ctx, 64 + b_pose + b_pts + b_out + 64, &pin);
        if (rc) return rc;
        char* hp = (char*)pin + 64;   // [sequence word | poses | points | outlier indices]
        char* dp = (char*)ctx->d_pinned + 64;
        volatile unsigned long long* slot = (volatile unsigned long long*)pin;
        const unsigned long long seq = ++ctx->pose_seq;
        *slot = 0;
        const size_t n_all = (size_t)d.C * 12 + (size_t)d.P * 3 + (size_t)n_out;
        hipLaunchKernelGGL(readback_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n_all + 255) / 256, 1024))), dim3(256), 0,
                           ctx->stream, (const double*)d.pose[ba->cur], (size_t)d.C * 12, (const double*)d.pt[ba->cur], (size_t)d.P * 3,
                           (const int*)d.outliers, (size_t)n_out, (double*)dp, (double*)(dp + b_pose), (int*)(dp + b_pose + b_pts));
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ctx->stream, (volatile unsigned long long*)ctx->d_pinned, seq);
        HIP_TRY(hipGetLastError());
        unsigned spins = 0;
        while (*slot != seq) {
            if (++spins == 100000) {
                spins = 0;
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return PTAM_E_HIP;
                if (q == hipSuccess && *slot != seq) return PTAM_E_HIP;   // drained without the stamp: something failed
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(ba->cam_pose.data(), hp, b_pose);
        for (int q = 0; q < d.P; q++)   // (device point q is original point pt_orig[q]; unobserved points keep their position)
            std::memcpy(&ba->pts[(size_t)3 * ba->pt_orig[(size_t)q]], hp + b_pose + (size_t)q * 24, 24);
        if (n_out > 0) std::memcpy(out_idx.data(), hp + b_pose + b_pts, b_out);
    }
    // the outlier list (reference order: LM step, then insertion order) is put together when somebody asks for it
    // (ba_finish_outliers): sorting ~10 k indices costs as much as a whole lambda trial
    ba->raw_out.insert(ba->raw_out.end(), out_idx.begin(), out_idx.begin() + n_out);
    for (int end : step_outlier_end) ba->raw_out_ends.push_back(ba->raw_out_base + end);
    ba->raw_out_base += n_out;
    // the device copy stays valid for another Compute(): poses/points are current in pose[cur]; the
    // outlier counter restarts
    if (n_out > 0) {
        ba->prepared = false;   // rebuild without the erased measurements on the next Compute
    }
#ifdef K7_TIMING
    std::printf("HOST Compute: first trial read at %.1f us, loop end %.1f us, total %.1f us (%zu trials)\n", ht_first - ht0, ht_loop - ht0,
                now_us() - ht0, ba->trials.size());
    {
        long long c[4];
        HIP_TRY(hipMemcpy(c, d.dbg + 5000, sizeof c, hipMemcpyDeviceToHost));
        if (c[3] > c[1])
            std::printf("K7 INSIDE Compute() (its last launch, block 100): %lld shader cycles in %.2f us = %.2f GHz\n", c[2] - c[0], (c[3] - c[1]) * 0.01,
                        (double)(c[2] - c[0]) / ((c[3] - c[1]) * 10.0));
    }
    if (getenv("PTAM_TIMELINE")) {
        std::vector<long long> tl(2 + 2 * TL_MAX);
        HIP_TRY(hipMemcpy(tl.data(), d.dbg + TL_BASE, tl.size() * 8, hipMemcpyDeviceToHost));
        const long long n_tl = std::min<long long>(tl[0], TL_MAX);
        static const char* nm[] = {"?", "purge_pass1", "select_compact", "select_final", "K7", "reduce_vinv", "vinv", "schur_tile", "schur_reduce",
                                   "ldlt_step0", "backward", "point_update", "finalize", "project_e2", "pass1_trial", "purge", "reduce_partials", "publish"};
        for (long long i = 0; i < n_tl; i++)
            std::printf("TL %4lld %-16s start %9.2f us  (+%.2f)\n", i, nm[tl[2 + 2 * i] < 18 ? tl[2 + 2 * i] : 0], (tl[3 + 2 * i] - tl[3]) * 0.01,
                        i ? (tl[3 + 2 * i] - tl[1 + 2 * i]) * 0.01 : 0.0);
    }
#endif
#ifdef SCHUR_STAMPS
    {
        std::vector<long long> wt(1024);
        HIP_TRY(hipMemcpy(wt.data(), d.dbg + 3072, wt.size() * 8, hipMemcpyDeviceToHost));
        const int nw = std::min(512, d.n_schur_wg);
        const long long M40 = (1ll << 40) - 1, M56 = (1ll << 56) - 1;
        long long e0 = wt[0] & M40, x1 = 0;
        for (int i = 0; i < nw; i++) e0 = std::min(e0, wt[2 * i] & M40), x1 = std::max(x1, wt[2 * i + 1] & M56 & M40);
        std::printf("SCHUR workgroups %d: makespan %.2f us; per workgroup (entry, exit in us from the first entry, segments, hw id):\n", nw, (x1 - e0) * 0.01);
        for (int i = 0; i < nw; i++)
            std::printf("%s[%d %.1f %.1f %d %llx]", i % 8 ? " " : "\n  ", i, ((wt[2 * i] & M40) - e0) * 0.01, ((wt[2 * i + 1] & M40) - e0) * 0.01, (int)(wt[2 * i + 1] >> 56),
                        (unsigned long long)(wt[2 * i] >> 40));
        std::printf("\n");
        // the schedule: per workgroup its segments as pair:groups
        std::vector<int> wseg((size_t)d.n_schur_wg + 1);
        HIP_TRY(hipMemcpy(wseg.data(), d.s_wg_seg, wseg.size() * 4, hipMemcpyDeviceToHost));
        std::vector<SchurWG> segs((size_t)wseg.back());
        HIP_TRY(hipMemcpy(segs.data(), d.s_segs, segs.size() * sizeof(SchurWG), hipMemcpyDeviceToHost));
        std::vector<SchurEntry> ents((size_t)std::max(1, d.n_schur_entries));
        HIP_TRY(hipMemcpy(ents.data(), d.s_entries, (size_t)d.n_schur_entries * sizeof(SchurEntry), hipMemcpyDeviceToHost));
        std::printf("SCHUR schedule:");   // per segment  pair : groups : model cost of its entries (the split's units)
        for (int i = 0; i < nw; i++) {
            std::printf("%s{%d", i % 8 ? " " : "\n  ", i);
            for (int sg = wseg[i]; sg < wseg[i + 1]; sg++) {
                long long cost = 0;
                for (int e = segs[sg].e_begin; e < segs[sg].e_end; e++) cost += ents[(size_t)e].pad & 0xffff;
                std::printf(" %d:%d:%lld", segs[sg].pair, (segs[sg].e_end - segs[sg].e_begin + 3) / 4, cost);
            }
            std::printf("}");
        }
        std::printf("\n");
    }
    {
        std::vector<long long> st(1024);
        HIP_TRY(hipMemcpy(st.data(), d.dbg + 2048, st.size() * 8, hipMemcpyDeviceToHost));
        for (int sel = 0; sel < 2; sel++) {
            const long long* b = st.data() + sel * 512;
            long long t0 = b[0];
            for (int w = 0; w < 4; w++) if (b[w * 64] && b[w * 64] < t0) t0 = b[w * 64];
            std::printf("SCHUR stamps wg %d (groups %lld): per wave, per group: top, loads issued, data there, MFMAs issued (cycles from the first top)\n", sel ? 150 : 0, b[260]);
            for (int w = 0; w < 4; w++) {
                std::printf("  w%d:", w);
                for (int i = 0; i < 16 && b[(w * 16 + i) * 4]; i++)
                    std::printf(" [%lld %lld %lld %lld]", b[(w * 16 + i) * 4] - t0, b[(w * 16 + i) * 4 + 1] - t0, b[(w * 16 + i) * 4 + 2] - t0, b[(w * 16 + i) * 4 + 3] - t0);
                std::printf(" end %lld\n", b[256 + w] - t0);
            }
            std::printf("  kernel entry %lld, exit %lld, segments %lld; last segment's epilogue (wave 0): loop end %lld, barrier 1 %lld, 2 %lld, 3 %lld, laid out %lld, stores issued %lld\n",
                        b[264] - t0, b[265] - t0, b[266], b[268] - t0, b[269] - t0, b[270] - t0, b[271] - t0, b[272] - t0, b[273] - t0);
        }
    }
#endif
    if (getenv("PTAM_DEBUG_STALL"))
        std::fprintf(stderr, "[ptam] compute total %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count());
    if (accepted_out) *accepted_out = ba->accepted;
    return PTAM_OK;
}

int ptam_ba_converged(const ptam_ba* ba) { return ba && ba->converged ? 1 : 0; }

int ptam_ba_get_point(const ptam_ba* ba, int n, double pos[3]) {
    ARG_TRY(ba && pos && n >= 0 && n < (int)(ba->pts.size() / 3));   // vector::at() would throw
    std::memcpy(pos, &ba->pts[3 * (size_t)n], 24);
    return PTAM_OK;
}
int ptam_ba_get_camera(const ptam_ba* ba, int n, double pose[12]) {
    ARG_TRY(ba && pose && n >= 0 && n < (int)ba->cam_fixed.size());
    std::memcpy(pose, &ba->cam_pose[12 * (size_t)n], 96);
    return PTAM_OK;
}
int ptam_ba_get_all(const ptam_ba* ba, double* poses12, double* points3) {
    ARG_TRY(ba);
    if (poses12) std::memcpy(poses12, ba->cam_pose.data(), ba->cam_pose.size() * 8);
    if (points3) std::memcpy(points3, ba->pts.data(), ba->pts.size() * 8);
    return PTAM_OK;
}
int ptam_ba_get_outliers(const ptam_ba* ba, int32_t* pairs, int cap) {
    if (!ba) return PTAM_E_ARG;
    ba_finish_outliers(const_cast<ptam_ba*>(ba));
    const int n = (int)ba->outliers.size();
    for (int i = 0; i < n && i < cap && pairs; i++) {
        pairs[2 * i] = ba->outliers[i].first;
        pairs[2 * i + 1] = ba->outliers[i].second;
    }
    return n;
}
int ptam_ba_get_trials(const ptam_ba* ba, ptam_ba_trial* out, int cap) {
    if (!ba) return PTAM_E_ARG;
    const int n = (int)ba->trials.size();
    for (int i = 0; i < n && i < cap && out; i++) out[i] = ba->trials[i];
    return n;
}
int ptam_ba_solve_fallbacks(const ptam_ba* ba) {
    ARG_TRY(ba);
    return ba->solve_fallbacks;
}
int ptam_ba_duplicates_refused(const ptam_ba* ba) {
    ARG_TRY(ba);
    return ba->dups_refused;
}
int ptam_ba_counts(const ptam_ba* ba, int* n_cams, int* n_free, int* n_points, int* n_meas) {
    ARG_TRY(ba);
    int f = 0;
    for (uint8_t x : ba->cam_fixed) f += x ? 0 : 1;
    ba_finish_outliers(const_cast<ptam_ba*>(ba));
    const int live = (int)ba->m_dead.size() - ba->n_dead;
    if (n_cams) *n_cams = (int)ba->cam_fixed.size();
    if (n_free) *n_free = f;
    if (n_points) *n_points = (int)(ba->pts.size() / 3);
    if (n_meas) *n_meas = live;
    return PTAM_OK;
}

// K7 alone, HIP-event timed over `reps` launches (bench.py roofline leg)
int ptam_ba_bench_jacobian(ptam_ba* ba, int reps, double* avg_ms, double* algorithmic_bytes) {
    ARG_TRY(ba && reps > 0);
    ptam_ctx* ctx = ba->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ptam_ba_prepare(ba);
    if (rc) return rc;
    BaDev& d = ba->d;
    ARG_TRY(d.M > 0);
    const bool prof = ba->prof;
    ba->prof = false;
    rc = ba_pass1_sigma(ba);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        launch_k7(ba);
    HIP_TRY(ptam_stream_wait(ctx->stream));
    // one event pair around `reps` back-to-back launches: the average is the kernel's steady-state
    // duration (an event pair around a single launch adds ~6 us of record / completion latency — an empty
    // kernel measures 6.3 us that way, tools/membw — and would not agree with rocprofv3's kernel trace)
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < reps; i++) launch_k7(ba);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms_total = 0;
    HIP_TRY(hipEventElapsedTime(&ms_total, e0, e1));
    const double total = ms_total;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    ba->prof = prof;
#ifdef K7_TIMING
    {
        long long h[16];
        HIP_TRY(hipMemcpy(h, d.dbg, sizeof h, hipMemcpyDeviceToHost));
        {
            long long c[4];
            HIP_TRY(hipMemcpy(c, d.dbg + 5000, sizeof c, hipMemcpyDeviceToHost));
            if (c[3] > c[1])
                std::printf("K7 BACK TO BACK (last of %d launches, block 100): %lld shader cycles in %.2f us = %.2f GHz\n", reps, c[2] - c[0], (c[3] - c[1]) * 0.01,
                            (double)(c[2] - c[0]) / ((c[3] - c[1]) * 10.0));
        }
        std::printf("K7 stamps (10 ns ticks since kernel-body start):");
        for (int i = 1; i < 10; i++) std::printf(" [%d] %lld", i, h[i] - h[0]);
        std::printf("\n");
        const int nb = std::min(ba->d.grid_acc, 2000);
        std::vector<long long> w(2 * nb);
        HIP_TRY(hipMemcpy(w.data(), d.dbg + 16, w.size() * 8, hipMemcpyDeviceToHost));
        long long t0 = w[0];
        for (int b = 0; b < nb; b++) t0 = std::min(t0, w[2 * b]);
        std::vector<long long> st(nb), en(nb), du(nb);
        for (int b = 0; b < nb; b++) st[b] = w[2 * b] - t0, en[b] = w[2 * b + 1] - t0, du[b] = en[b] - st[b];
        std::sort(st.begin(), st.end());
        std::sort(en.begin(), en.end());
        std::sort(du.begin(), du.end());
        std::printf("K7 block 7: body starts %lld ticks after the block's first instruction\n", h[0] - w[14]);
        std::printf("K7 wall (10 ns ticks, %d blocks): start p0/p50/p90/p100 %lld %lld %lld %lld | end %lld %lld %lld %lld | dur %lld %lld %lld %lld\n",
                    nb, st[0], st[nb / 2], st[nb * 9 / 10], st[nb - 1], en[0], en[nb / 2], en[nb * 9 / 10], en[nb - 1], du[0], du[nb / 2],
                    du[nb * 9 / 10], du[nb - 1]);
    }
#endif
    if (avg_ms) *avg_ms = total / reps;
    if (algorithmic_bytes)   // 8 idx + 24 found/s + 1 state + 144 W per measurement,
                             // 96 B/camera pose read, 24 read + 72 write per point, 216 B/free camera
        *algorithmic_bytes = (double)d.M * (8 + 24 + 1 + 144) + (double)d.C * 96 + (double)d.P * (24 + 72) + (double)d.F * 216;
    return PTAM_OK;
}

// K7 with its inputs and outputs COLD in the 256 MB Infinity Cache: `n` prepared bundles of the same context (copies of one
// problem) are launched round-robin, back to back, inside ONE event bracket — the method of ptam_ba_bench_jacobian, but
// between two launches on the same bundle lie the n - 1 other working sets.  The caller picks n so that
// (n - 1) x (bytes one launch touches) exceeds the cache by a wide margin; then every launch streams from / to HBM.
int ptam_ba_schur_index_map(int variant, uint16_t* out, int cap) {   // (test hook: host only)
    ARG_TRY(out && variant >= 0 && variant < SCHUR_N_VARIANTS && cap >= SCHUR_TILE_ELEMS);
    const std::vector<unsigned short> m = schur_index_map();
    std::memcpy(out, m.data() + (size_t)variant * SCHUR_TILE_ELEMS, (size_t)SCHUR_TILE_ELEMS * 2);
    return SCHUR_TILE_ELEMS;
}

int ptam_ba_debug_lists(ptam_ba* ba, int which, void* out, size_t cap_bytes) {   // (test hook: include/ptam_hip_bench.h)
    ARG_TRY(ba);
    if (!ba->prepared) {
        ptam_set_error("ptam_ba_debug_lists: the bundle is not prepared");
        return PTAM_E_STATE;
    }
    ptam_ctx* ctx = ba->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const BaDev& d = ba->d;
    int counts[16] = {d.C, d.F, d.P, d.M, d.band, d.n_chunks, d.n_tiles, d.n_pairs, d.n_schur_wg, d.n_schur_entries, ba->n_schur_segs, d.grid_acc, 0, 0, 0, 0};
    const void* src = nullptr;
    size_t bytes = 0;
    bool host_src = false;
    int n_free_meas = 0;
    switch (which) {
        case PTAM_BL_COUNTS: src = counts, bytes = sizeof counts, host_src = true; break;
        case PTAM_BL_ROWPTR: src = d.rowptr, bytes = ((size_t)d.P + 1) * 4; break;
        case PTAM_BL_M_CAM: src = d.m_cam, bytes = (size_t)d.M * 4; break;
        case PTAM_BL_M_PT: src = d.m_pt, bytes = (size_t)d.M * 4; break;
        case PTAM_BL_M_ORIG: src = d.m_orig, bytes = (size_t)d.M * 4; break;
        case PTAM_BL_M_FIDX: src = d.m_fidx, bytes = (size_t)d.M * 4; break;
        case PTAM_BL_M_FOUND: src = d.m_found, bytes = (size_t)d.M * 16; break;
        case PTAM_BL_M_S: src = d.m_s, bytes = (size_t)d.M * 8; break;
        case PTAM_BL_PT_ORIG: src = ba->pt_orig.data(), bytes = (size_t)d.P * 4, host_src = true; break;
        case PTAM_BL_POINTS: src = d.pt[ba->cur], bytes = (size_t)d.P * 24; break;
        case PTAM_BL_CHUNKS: src = d.chunks, bytes = (size_t)d.n_chunks * sizeof(BaChunk); break;
        case PTAM_BL_S_ENTRIES: src = d.s_entries, bytes = (size_t)d.n_schur_entries * sizeof(SchurEntry); break;
        case PTAM_BL_S_SEGS: src = d.s_segs, bytes = (size_t)ba->n_schur_segs * sizeof(SchurWG); break;
        case PTAM_BL_S_WG_SEG: src = d.s_wg_seg, bytes = d.n_schur_wg > 0 ? ((size_t)d.n_schur_wg + 1) * 4 : 0; break;
        case PTAM_BL_S_PAIR_BEGIN: src = d.s_pair_wg_begin, bytes = d.F > 0 ? ((size_t)d.n_pairs + 1) * 4 : 0; break;
        case PTAM_BL_S_WG_HEAD: src = d.s_wg_head, bytes = (size_t)d.n_schur_wg * 32; break;
        case PTAM_BL_CAM_PTR: src = d.cam_ptr, bytes = ba->det ? (size_t)((d.M + DET_TILE - 1) / DET_TILE) * ((size_t)d.F + 1) * 4 : 0; break;
        case PTAM_BL_CAM_MEAS:
            if (ba->det && d.M > 0) {   // its length is the last tile's last row end
                const size_t last = (size_t)((d.M + DET_TILE - 1) / DET_TILE) * ((size_t)d.F + 1) - 1;
                HIP_TRY(hipMemcpyAsync(&n_free_meas, d.cam_ptr + last, 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ptam_stream_wait(ctx->stream));
            }
            src = d.cam_meas, bytes = (size_t)n_free_meas * 4;
            break;
        default: ARG_TRY(!"which"); 
    }
    if (out && bytes > 0) {
        ARG_TRY(cap_bytes >= bytes);
        if (host_src)
            std::memcpy(out, src, bytes);
        else {
            HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ptam_stream_wait(ctx->stream));
        }
    }
    if (bytes > (size_t)INT_MAX) return PTAM_E_LIMIT;
    return (int)bytes;
}

int ptam_ba_bench_jacobian_rotating(ptam_ba** bas, int n, int reps, double* avg_ms) {
    ARG_TRY(bas && n > 0 && reps > 0 && avg_ms);
    ptam_ctx* ctx = bas[0]->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++) {
        ARG_TRY(bas[i] && bas[i]->ctx == ctx);
        int rc = ptam_ba_prepare(bas[i]);
        if (rc) return rc;
        ARG_TRY(bas[i]->d.M > 0);
        rc = ba_pass1_sigma(bas[i]);
        if (rc) return rc;
    }
    for (int r = 0; r < 2; r++)
        for (int i = 0; i < n; i++) launch_k7(bas[i]);
    HIP_TRY(ptam_stream_wait(ctx->stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < n; i++) launch_k7(bas[i]);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms_total = 0;
    HIP_TRY(hipEventElapsedTime(&ms_total, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *avg_ms = (double)ms_total / ((double)reps * n);
    return PTAM_OK;
}

}   // extern "C"

// every kernel of this file resolved once, when a context is created: the first launch of a kernel otherwise pays for
// loading the code object / resolving the function — 10-28 ms in the middle of the first frame or the first adjustment
void ba_preload_kernels() {
    ptam_preload((const void*)project_e2_kernel);
    ptam_preload((const void*)pass1_from_trial_kernel);
    ptam_preload((const void*)pass1_keep_kernel);
    ptam_preload((const void*)purge_pass1_kernel);
    ptam_preload((const void*)purge_pass1_decide_kernel);
    ptam_preload((const void*)hist_keys_kernel);
    ptam_preload((const void*)select_compact_kernel);
    ptam_preload((const void*)select_final_kernel);
    ptam_preload((const void*)hist_to_f64_kernel);
    ptam_preload((const void*)f64_to_hist_kernel);
    ptam_preload((const void*)select_stage_kernel);
    ptam_preload((const void*)select_finish_kernel);
    ptam_preload((const void*)compact_valid_kernel);
    ptam_preload(k7_wave_fn(256, true, -1, true));
    for (int est : {(int)PTAM_EST_TUKEY, (int)PTAM_EST_CAUCHY})
        for (int lp = 0; lp < 2; lp++) ptam_preload(k7_wave_fn(lp ? 256 : 1024, lp != 0, est, false, true));
    ptam_preload(k7_wave_fn(256, true, -1, true, true));
    ptam_preload((const void*)reduce_det_kernel);
    ptam_preload((const void*)reduce_partials_kernel);
    ptam_preload((const void*)vinv_kernel);
    ptam_preload((const void*)reduce_vinv_kernel);
    ptam_preload((const void*)schur_tile_mfma_kernel);
    ptam_preload((const void*)schur_reduce_kernel);
    ptam_preload((const void*)pose_update_kernel);
    ptam_preload((const void*)point_update_kernel);
    ptam_preload((const void*)finalize_new_kernel);
    ptam_preload((const void*)purge_kernel);
    ptam_preload((const void*)readback_kernel);
    ptam_preload((const void*)stamp_kernel);
    ptam_preload((const void*)publish_scalars_kernel);
    ptam_preload((const void*)set_scalars_kernel);
    ptam_preload((const void*)set_new_kernel);
    ptam_preload((const void*)pack2_kernel);
    ptam_preload((const void*)unpack2_kernel);
    ptam_preload((const void*)place_keys_kernel);
    for (int est = 0; est < 2; est++) {
        ptam_preload(k7_wave_fn(512, false, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(1024, false, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(256, true, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(512, true, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
        ptam_preload(k7_wave_fn(1024, true, est ? PTAM_EST_CAUCHY : PTAM_EST_TUKEY));
    }
}
