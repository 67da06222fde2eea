// shim_demo.cc — the reference's call sequence for the hot path, written against ptam_shim.hpp:
// MakeKeyFrame_Lite -> FindPatchCoarse -> pose iterations -> Bundle::Compute.  Prints results in a
// line format that tests/test_gpu_shim.py compares with the CPU oracle.
//   g++ -O2 -std=c++17 -Iinclude examples/shim_demo.cc -Lptam_cg_amd/csrc -lptam_hip -Wl,-rpath,$PWD/ptam_cg_amd/csrc
#include <cmath>
#include <cstdio>
#include <vector>

#include "ptam_shim.hpp"

static uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }

int main() {
    using namespace ptam;
    Context ctx({1.0803, 1.43987, 0.519983, 0.548655, 0.244943}, {160, 120});
    // a blocky synthetic image (same LCG in the Python test)
    std::vector<uint8_t> im(160 * 120, 128);
    uint32_t s = 12345;
    for (int k = 0; k < 40; k++) {
        int x0 = lcg(s) % 150, y0 = lcg(s) % 110, w = 8 + lcg(s) % 40, h = 8 + lcg(s) % 40, v = 30 + lcg(s) % 190;
        for (int y = y0; y < y0 + h && y < 120; y++)
            for (int x = x0; x < x0 + w && x < 160; x++) im[y * 160 + x] = (uint8_t)v;
    }
    KeyFrame kf(ctx);
    kf.MakeKeyFrame_Lite(im.data(), 160);
    for (int l = 0; l < 4; l++) {
        const Level& L = kf.aLevels(l);
        long sx = 0, sy = 0, sl = 0;
        for (auto& c : L.vCorners) sx += c.x, sy += c.y;
        for (int v : L.vCornerRowLUT) sl += v;
        std::printf("LEVEL %d %d %d %zu %ld %ld %ld\n", l, L.w, L.h, L.vCorners.size(), sx, sy, sl);
    }
    KeyFrame copy = kf;   // the tracker -> mapmaker hand-off deep copy
    std::printf("CLONE %zu\n", copy.aLevels(0).vCorners.size());
    // search the patch around the first level-0 corner that is far enough from the border
    const Level& L0 = kf.aLevels(0);
    for (auto& c : L0.vCorners)
        if (c.x >= 8 && c.y >= 8 && c.x < 152 && c.y < 112) {
            uint8_t t[64];
            for (int r = 0; r < 8; r++)
                for (int q = 0; q < 8; q++) t[r * 8 + q] = L0.im[(c.y - 4 + r) * 160 + c.x - 4 + q];
            PatchFinder pf(ctx);
            pf.SetTemplate(t, 0);
            bool found = pf.FindPatchCoarse({c.x + 2, c.y - 1}, kf, 10);
            std::printf("PATCH %d %d %d %.1f %.1f %d\n", c.x, c.y, (int)found, pf.GetCoarsePosAsVector()[0],
                        pf.GetCoarsePosAsVector()[1], pf.ZMSSDAtPoint(kf, 0, c));
            break;
        }
    // warped template out of the same keyframe: an identity warp reproduces the 8x8 window, a second call for the
    // same point with an almost identical warp is served from the reuse test, a rotated warp regenerates
    for (auto& c : L0.vCorners)
        if (c.x >= 20 && c.y >= 20 && c.x < 140 && c.y < 100) {
            PatchFinder pf(ctx);
            const int pointId = 1;
            const double wi_id[4] = {1, 0, 0, 1}, wi_close[4] = {1.01, 0, 0, 1.01}, wi_rot[4] = {0.8, -0.6, 0.6, 0.8};
            pf.MakeTemplateCoarseCont(&pointId, kf, 0, c, wi_id, 0);
            const int z0 = pf.ZMSSDAtPoint(kf, 0, c);
            pf.MakeTemplateCoarseCont(&pointId, kf, 0, c, wi_close, 0);   // |dm2| < 0.07: template kept
            const int z1 = pf.ZMSSDAtPoint(kf, 0, c);
            pf.MakeTemplateCoarseCont(&pointId, kf, 0, c, wi_rot, 0);
            const int z2 = pf.ZMSSDAtPoint(kf, 0, c);
            std::printf("WARP %d %d %d %d %d %d\n", c.x, c.y, z0, z1, z2, (int)pf.TemplateBad());
            break;
        }
    // a toy bundle: 3 cameras on a line looking down +z, 12 points on a grid, exact measurements of a
    // pinhole-ish projection perturbed deterministically
    Context c640({1.0803, 1.43987, 0.519983, 0.548655, 0.244943}, {640, 480});
    Bundle b(c640);
    bool abort_flag = false;
    for (int j = 0; j < 3; j++) {
        SE3 T = SE3::Identity();
        T.t[0] = -0.2 * j;
        b.AddCamera(T, j == 0);
    }
    uint32_t s2 = 777;
    for (int i = 0; i < 12; i++) {
        Vec<3> X{-0.3 + 0.2 * (i % 4), -0.2 + 0.2 * (i / 4), 2.0 + 0.05 * (i % 3)};
        b.AddPoint(X);
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 12; i++) {
            double X = -0.3 + 0.2 * (i % 4) + 0.2 * j, Y = -0.2 + 0.2 * (i / 4), Z = 2.0 + 0.05 * (i % 3);
            double u = 332.3 + 691.4 * X / Z + (double)(lcg(s2) % 1000) / 1000.0 - 0.5;
            double v = 262.9 + 691.1 * Y / Z + (double)(lcg(s2) % 1000) / 1000.0 - 0.5;
            b.AddMeas(j, i, {u, v}, 1.0);
        }
    int acc = b.Compute(&abort_flag);
    std::printf("BUNDLE %d %d %zu\n", acc, (int)b.Converged(), b.GetOutlierMeasurements().size());
    for (int j = 0; j < 3; j++) {
        SE3 T = b.GetCamera(j);
        std::printf("CAM %d %.12e %.12e %.12e\n", j, T.t[0], T.t[1], T.t[2]);
    }
    Vec<3> p0 = b.GetPoint(0);
    std::printf("PT0 %.12e %.12e %.12e\n", p0[0], p0[1], p0[2]);
    return 0;
}
