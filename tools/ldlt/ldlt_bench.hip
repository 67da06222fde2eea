// tools/ldlt/ldlt_bench.hip — the camera solve (csrc/solve.hip) alone, on a synthetic banded SPD system:
//   ldlt_bench [free cameras = 49] [band (blocks) = all] [repetitions = 200]
// Builds S = G G^T + n I inside the block band (lower triangle, block-banded storage of csrc/bundle.h), a right-hand
// side, runs ba_solve() `repetitions` times (S | E restored from a master copy before each), checks da against a host
// LDL^T of the same system and prints the average wall time per solve (events around the whole loop, copies included — use
// rocprofv3 --kernel-trace --stats on this binary for per-kernel durations; -DK7_TIMING builds print the kernels' stamps).
// Not part of the product: it includes solve.hip textually so that instrumented variants need no library rebuild.
#include <cstdarg>
#include <cstdlib>
#include <random>

#include "../../ptam_cg_amd/csrc/solve.hip"

void ptam_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
void ptam_preload(const void*) {}
#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 49;
    const int n = 6 * F, npad = (n + NB - 1) / NB * NB, nblk = npad / NB;
    int band = argc > 2 ? atoi(argv[2]) : nblk - 1;
    if (band > nblk - 1) band = nblk - 1;
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    // dense symmetric matrix with the block band, strongly diagonal
    std::vector<double> A((size_t)npad * npad, 0.0), b(npad, 0.0);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (int i = 0; i < npad; i++)
        for (int j = 0; j <= i; j++) {
            double v = 0;
            if (i >= n || j >= n)
                v = i == j ? 1.0 : 0.0;   // identity padding
            else if (i / NB - j / NB <= band)
                v = i == j ? 2.0 * NB * (band + 1) + u(rng) : u(rng);
            A[(size_t)i * npad + j] = A[(size_t)j * npad + i] = v;
        }
    for (int i = 0; i < n; i++) b[i] = u(rng);
    // host reference: dense LDL^T
    std::vector<double> Lh = A, x = b;
    for (int j = 0; j < npad; j++) {
        const double dj = Lh[(size_t)j * npad + j];
        for (int i = j + 1; i < npad; i++) {   // a_iq -= a_ij a_qj / d_j, lower triangle, column j still undivided
            const double aij = Lh[(size_t)i * npad + j];
            if (aij == 0.0) continue;
            for (int q = j + 1; q <= i; q++) Lh[(size_t)i * npad + q] -= aij * Lh[(size_t)q * npad + j] / dj;
        }
        for (int i = j + 1; i < npad; i++) Lh[(size_t)i * npad + j] /= dj;
    }
    for (int i = 0; i < npad; i++)
        for (int j = 0; j < i; j++) x[i] -= Lh[(size_t)i * npad + j] * x[j];
    for (int i = 0; i < npad; i++) x[i] /= Lh[(size_t)i * npad + i];
    for (int i = npad - 1; i >= 0; i--)
        for (int j = i + 1; j < npad; j++) x[i] -= Lh[(size_t)j * npad + i] * x[j];
    // device layout
    BaDev d;
    memset(&d, 0, sizeof d);
    d.n = n, d.npad = npad, d.band = band, d.C = 0, d.F = F;
    const size_t ssz = se_size(nblk, band), tot = ssz + npad + 3;
    std::vector<double> SE(tot, 0.0);
    for (int bi = 0; bi < nblk; bi++)
        for (int bj = std::max(0, bi - band); bj <= bi; bj++)
            for (int r = 0; r < NB; r++)
                for (int c = 0; c < NB; c++)
                    SE[se_blk(bi, bj, band) + r * NB + c] = (bi == bj && c > r) ? 1e300 : A[(size_t)(bi * NB + r) * npad + bj * NB + c];
    for (int i = 0; i < npad; i++) SE[ssz + i] = b[i];
    double *master, *dbg;
    CK(hipMalloc(&master, tot * 8));
    CK(hipMalloc(&d.SE, tot * 8));
    CK(hipMalloc(&d.L, ssz * 8));
    CK(hipMalloc(&d.Dg, npad * 8));
    CK(hipMalloc(&d.y, npad * 8));
    CK(hipMalloc(&d.da, npad * 8));
    CK(hipMalloc(&d.sumsq2, 16));
    CK(hipMalloc(&d.cam_free, 16));
    CK(hipMalloc(&d.bw_scratch, (size_t)12 * npad * 8));
    CK(hipMalloc(&d.sflags, ba_solve_flag_bytes(nblk)));
    CK(hipMalloc(&d.sflags2, ba_solve_flag_bytes(nblk)));
    CK(hipMemset(d.sflags2, 0, ba_solve_flag_bytes(nblk)));
    CK(hipMalloc(&d.SE2, tot * 8));
    CK(hipMalloc(&d.L2, ssz * 8));
    CK(hipMalloc(&d.Dg2, npad * 8));
    CK(hipMalloc(&d.y2, npad * 8));
    CK(hipMemset(d.sflags, 0, ba_solve_flag_bytes(nblk)));
    d.spin_limit = CH_SPIN_DEFAULT;   // (a zero limit makes every wait of the persistent form give up at once)
    CK(hipMalloc(&d.sc, sizeof(BaScalars)));
    CK(hipMemset(d.sc, 0, sizeof(BaScalars)));
    CK(hipMalloc(&dbg, 65536 * 8));
    CK(hipMemset(dbg, 0, 65536 * 8));
    d.dbg = (long long*)dbg;
    CK(hipMemcpy(master, SE.data(), tot * 8, hipMemcpyHostToDevice));
    ptam_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    CK(hipStreamCreate(&ctx.stream));
    if (ba_solve_init() != PTAM_OK) return 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, ms = 0;
    for (int round = 0; round < 3; round++) {
        CK(hipEventRecord(e0, ctx.stream));
        for (int i = 0; i < reps; i++) {
            CK(hipMemcpyAsync(d.SE, master, tot * 8, hipMemcpyDeviceToDevice, ctx.stream));
            if (ba_solve(&ctx, d, 0) != PTAM_OK) return 1;
        }
        CK(hipEventRecord(e1, ctx.stream));
        CK(hipStreamSynchronize(ctx.stream));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<double> da(npad);
    CK(hipMemcpy(da.data(), d.da, npad * 8, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (int i = 0; i < n; i++) err = std::max(err, fabs(da[i] - x[i])), nrm = std::max(nrm, fabs(x[i]));
    {
        BaScalars sc;
        CK(hipMemcpy(&sc, d.sc, sizeof sc, hipMemcpyDeviceToHost));
        if (sc.solve_fault) printf("SOLVE FAULT: a wait of the persistent form gave up\n");
    }
    printf("F %d n %d nblk %d band %d: %.2f us per solve (copy included), max |da - ref| = %.3e (|ref| max %.3e) %s\n", F, n, nblk, band,
           1e3 * best / reps, err, nrm, err <= 1e-11 * nrm + 1e-300 ? "OK" : "MISMATCH");
#ifdef K7_TIMING
    if (!getenv("PTAM_LDLT_NO_CHAIN") && nblk > SM_USE_NB) {
        std::vector<long long> st(512);
        CK(hipMemcpy(st.data(), dbg, 512 * 8, hipMemcpyDeviceToHost));
        {
            std::vector<unsigned> fl(ch_flag_words(nblk, band));
            CK(hipMemcpy(fl.data(), d.sflags, fl.size() * 4, hipMemcpyDeviceToHost));
            printf("error word %u | F[0] %u (seq %u, plain stores %u) | XCC ids:", fl[0], fl[1], fl[1] >> 1, fl[1] & 1);
            for (int i = 0; i < ch_roles(nblk); i++) printf(" %u", fl[1 + 2 * nblk + i] & 0xf);
            printf("\n");
        }
        {
            std::vector<unsigned> fl(ch_flag_words(nblk, band));
            CK(hipMemcpy(fl.data(), d.sflags2, fl.size() * 4, hipMemcpyDeviceToHost));
            printf("second chain: F[0] %u (seq %u, plain stores %u) | XCC ids:", fl[1], fl[1] >> 1, fl[1] & 1);
            for (int i = 0; i < 24; i++) printf(" %u", fl[1 + 2 * nblk + i] & 0xf);
            printf("\n");
        }
        printf("chain stamps per step (cycles): factor | stores + wait for the row | row step, flags | (step total)\n");
        for (int k = 0; k + 1 < nblk && k < 12; k++) {
            const long long* q = &st[32 + 4 * k];
            printf("  step %2d: %6lld %6lld %6lld   (%lld)\n", k, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[0]);
        }
        printf("iteration at which the next row's flag was seen (0: never inside the loop):");
        for (int k = 0; k + 1 < nblk && k < 16; k++) printf(" %lld/%lld", st[440 + k] ? st[440 + k] - 100 : 0, st[456 + k] ? st[456 + k] - 100 : 0);
        printf("\n");
        {
            std::vector<long long> rt(4 * 16);
            CK(hipMemcpy(rt.data(), dbg + 600, rt.size() * 8, hipMemcpyDeviceToHost));
            printf("per k (10 ns units, relative to the start of the chain's loop k): F[k] raised | worker k+2 saw it | worker k+2 raised RF | (loop k+1 starts)\n");
            for (int k = 0; k + 2 < nblk && k < 15; k++)
                printf("  k %2d: %5lld %5lld %5lld (%lld)\n", k, rt[4 * k] - rt[4 * k + 3], rt[4 * k + 1] - rt[4 * k + 3], rt[4 * k + 2] - rt[4 * k + 3],
                       rt[4 * k + 7] - rt[4 * k + 3]);
        }
        {
            std::vector<long long> rt(4 * 16), r0(4);
            CK(hipMemcpy(rt.data(), dbg + 700, rt.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r0.data(), dbg + 600, 32, hipMemcpyDeviceToHost));
            printf("last row's worker, per m (10 ns units since the chain's start): saw F[m] | L stored, XF up | two tiles updated | step done\n");
            for (int m = 0; m + 2 < nblk && m < 15; m++)
                printf("  m %2d: %6lld %6lld %6lld %6lld\n", m, rt[4 * m] - r0[3], rt[4 * m + 1] - r0[3], rt[4 * m + 2] - r0[3], rt[4 * m + 3] - r0[3]);
        }
    } else {
        std::vector<long long> st(512);
        CK(hipMemcpy(st.data(), dbg, 512 * 8, hipMemcpyDeviceToHost));
        printf("stamps (cycles between consecutive ones):");
        for (int i = 1; i < 400 && st[32 + i]; i++) printf(" %lld", st[32 + i] - st[31 + i]);
        printf("\n");
    }
#endif
#ifdef K7_TIMING
    {
        std::vector<long long> st(8 * 64 + 8);
        CK(hipMemcpy(st.data(), dbg + 1392, st.size() * 8, hipMemcpyDeviceToHost));
        const long long* q0 = &st[8];
        if (st[7]) {
            printf("backward inside the launch, per block (cycles): wave 0: chain | diag request | at barrier B -> released | (wave 1: top -> barrier B released)   top at\n");
            for (int k = nblk - 1; k >= 0; k--) {
                const long long* q = q0 + 8 * k;
                printf("  k %2d: %5lld %5lld %5lld   (%lld)   top at %lld\n", k, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[7] - q[4], q[0] - st[7]);
            }
            printf("  epilogue starts at %lld (cycles after the forward pass ended)\n", st[6] - st[7]);
        }
    }
#endif
#ifdef FL_STAMPS
    {
        std::vector<long long> st(4 * 32);
        CK(hipMemcpy(st.data(), dbg + 1200, st.size() * 8, hipMemcpyDeviceToHost));
        printf("factor loop of step 2, per wave and iteration (cycles): top -> MFMAs issued | -> at the barrier | barrier -> next top   (top, relative to wave 0's first)\n");
        for (int w = 0; w < 4; w++)
            for (int t = 0; t < 8; t++) {
                const long long* q = &st[32 * w + 4 * t];
                if (!q[0]) continue;
                printf("  wave %d t %d: %5lld %5lld %5lld   (top at %lld)\n", w, t, q[1] - q[0], q[2] - q[1], t < 7 && q[4] ? q[4] - q[2] : 0, q[0] - st[0]);
            }
    }
#endif
#ifdef K7_TIMING
    {
        std::vector<long long> st(8 * 64);
        CK(hipMemcpy(st.data(), dbg + 1000, st.size() * 8, hipMemcpyDeviceToHost));
        printf("backward kernel, workgroup 0, per block (cycles): loads issued | barrier A | mat-vec, x written | barrier B | updates | (to next block)\n");
        for (int k = nblk - 1, shown = 0; k >= 0 && shown < 12; k--)
            if (st[8 * k] && ++shown) printf("  k %2d: %5lld %5lld %5lld %5lld (%lld)\n", k, st[8 * k + 1] - st[8 * k], st[8 * k + 2] - st[8 * k + 1], st[8 * k + 3] - st[8 * k + 2],
                   st[8 * k + 4] - st[8 * k + 3], k > 0 ? st[8 * (k - 1)] - st[8 * k + 4] : 0);
    }
#endif
    return err <= 1e-11 * nrm ? 0 : 2;
}
