"""np_oracle.py — second, independent restatement of the hot path in numpy (vectorised, dense linear
algebra), used ONLY in this container to (a) cross-check oracle/ptam_oracle.cc and (b) generate the
golden fixtures under tests/golden/ (tests/golden/make_golden.py).  TEST INFRASTRUCTURE; PARITY
UNPINNED against the reference binary for the reasons given in ptam_oracle.cc's header.

It deliberately shares no code or structure with the C++ oracle: FAST is evaluated as whole-image
boolean algebra, ZMSSD from its definition with exact integer arithmetic, the bundle adjuster builds
the normal equations from dense per-measurement Jacobian arrays and solves with numpy.linalg, the
order statistic comes from np.sort.  Reference lines each piece follows are cited per function."""
import numpy as np

LEVELS = 4
MAX_SSD = 8 * 8 * 500
FAST_THRESH = (10, 15, 15, 10)                    # src/KeyFrame.cc:35-42
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---- KeyFrame::MakeKeyFrame_Lite (src/KeyFrame.cc:18-54) ---------------------------------------
def half_sample(im, variant="R"):
    """CVD::halfSample: T = (a+b+c+d)/4 ; R = SSE2 pavgb (vertical) then pavgw (horizontal)."""
    h, w = im.shape[0] // 2, im.shape[1] // 2
    a = im[0:2 * h:2, 0:2 * w:2].astype(np.int32)
    b = im[0:2 * h:2, 1:2 * w:2].astype(np.int32)
    c = im[1:2 * h:2, 0:2 * w:2].astype(np.int32)
    d = im[1:2 * h:2, 1:2 * w:2].astype(np.int32)
    if variant == "T":
        return ((a + b + c + d) // 4).astype(np.uint8)
    v1, v2 = (a + c + 1) >> 1, (b + d + 1) >> 1
    return ((v1 + v2 + 1) >> 1).astype(np.uint8)


def fast10(im, thr):
    """fast_corner_detect_10: >= 10 contiguous ring pixels all > p+b or all < p-b; raster order."""
    h, w = im.shape
    if h < 7 or w < 7:
        return np.zeros((0, 2), np.int32)
    p = im[3:h - 3, 3:w - 3].astype(np.int32)
    ring = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING])
    out = np.zeros(p.shape, bool)
    for flags in (ring > p + thr, ring < p - thr):
        ext = np.concatenate([flags, flags[:9]])          # circular runs
        run = np.ones((16,) + p.shape, bool)
        for k in range(10):
            run &= ext[k:k + 16]
        out |= run.any(axis=0)
    ys, xs = np.nonzero(out)                               # row-major == raster order
    return np.column_stack([xs + 3, ys + 3]).astype(np.int32)


def row_lut(corners, h):
    """LUT[y] = index of the first corner with row >= y (src/KeyFrame.cc:46-52)."""
    return np.searchsorted(corners[:, 1], np.arange(h), side="left").astype(np.int32)


def make_keyframe_lite(im, variant="R"):
    levels = []
    cur = np.ascontiguousarray(im, np.uint8)
    for l in range(LEVELS):
        if l:
            cur = half_sample(cur, variant)
        c = fast10(cur, FAST_THRESH[l])
        levels.append({"im": cur, "corners": c, "rowlut": row_lut(c, cur.shape[0])})
    return levels


# ---- KeyFrame::MakeKeyFrame_Rest (src/KeyFrame.cc:61-82): fast_nonmax + Shi-Tomasi ------------------
def fast_scores(im, corners):
    """largest threshold at which each corner is still a FAST-10 corner: over the 16 arcs of 10 ring
    pixels, max of min(ring - p) (brighter) or min(p - ring) (darker), minus 1"""
    if len(corners) == 0:
        return np.zeros(0, int)
    x, y = corners[:, 0], corners[:, 1]
    p = im[y, x].astype(int)
    ring = np.stack([im[y + dy, x + dx].astype(int) for dx, dy in RING])       # (16, n)
    ext = np.concatenate([ring, ring[:9]])
    best = np.full(len(corners), -1000)
    for sign in (1, -1):
        d = sign * (ext - p)
        arcs = np.stack([d[k:k + 10].min(axis=0) for k in range(16)])
        best = np.maximum(best, arcs.max(axis=0) - 1)
    return best


def make_keyframe_rest(levels):
    out = []
    for L in levels:
        im, c = L["im"], L["corners"]
        h, w = im.shape
        sc = fast_scores(im, c)
        smap = np.zeros((h + 2, w + 2), int)                                      # padded score map, 0 = no corner
        smap[c[:, 1] + 1, c[:, 0] + 1] = sc
        nb = np.stack([smap[c[:, 1] + 1 + dy, c[:, 0] + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy])
        keep = (nb <= sc).all(axis=0) if len(c) else np.zeros(0, bool)
        mc = c[keep]
        st = np.full(len(mc), -1.0)
        I = im.astype(np.float64)
        for i, (x, y) in enumerate(mc):
            if not (10 <= x < w - 10 and 10 <= y < h - 10):
                continue
            dx = I[y - 3:y + 4, x - 2:x + 5] - I[y - 3:y + 4, x - 4:x + 3]
            dy = I[y - 2:y + 5, x - 3:x + 4] - I[y - 4:y + 3, x - 3:x + 4]
            xx, yy, xy = (dx * dx).sum() / 98.0, (dy * dy).sum() / 98.0, (dx * dy).sum() / 98.0
            st[i] = 0.5 * (xx + yy - np.sqrt((xx + yy) ** 2 - 4 * (xx * yy - xy * xy)))
        out.append({"max_corners": mc, "st_scores": st})
    return out


# ---- ZMSSD + FindPatchCoarse (src/ImageProcess.cc:130-163, src/PatchFinder.cc:160-211) -----------
def trunc_div(a, b):
    """C integer division (toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def zmssd(im, x, y, tmpl):
    h, w = im.shape
    if not (4 <= x < w - 4 and 4 <= y < h - 4):
        return MAX_SSD + 1
    I = im[y - 4:y + 4, x - 4:x + 4].astype(np.int64).ravel()
    T = np.asarray(tmpl, np.int64).ravel()
    SA, SB = int(T.sum()), int(I.sum())
    return trunc_div(2 * SA * SB - SA * SA - SB * SB, 64) + int((I * I).sum()) + int((T * T).sum()) - 2 * int((I * T).sum())


def unproject(cam, u, v):
    """ATANCamera::UnProject (src/ATANCamera.cc:125-140) with math.* scalars; cam: Camera below (focal, centre, w, k = 2 tan(w/2))"""
    import math
    dx = (u - cam.centre[0]) * (1.0 / cam.focal[0])
    dy = (v - cam.centre[1]) * (1.0 / cam.focal[1])
    dr = math.sqrt(dx * dx + dy * dy)
    rr = dr if cam.w == 0.0 else math.tan(dr * cam.w) * (1.0 / cam.k)
    f = rr / dr if dr > 0.01 else 1.0
    return f * dx, f * dy


def one_pixel_dist(cam):
    """mdOnePixelDist (src/ATANCamera.cc:69-75)"""
    import math
    a = unproject(cam, cam.size[0] / 2, cam.size[1] / 2)
    b = unproject(cam, cam.size[0] / 2 + 1, cam.size[1] / 2 + 1)
    return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2) / math.sqrt(2.0)


def implane_corners(cam, corners, level):
    """Level::vImplaneCorners (src/MapMaker.cc:605-614)"""
    scale = 1 << level
    out = np.zeros((len(corners), 2))
    for i, (x, y) in enumerate(corners):
        out[i] = unproject(cam, float(int((x + 0.5) * scale - 0.5)), float(int((y + 0.5) * scale - 0.5)))
    return out


def epipolar_search(src_level, tgt_level, ip, q):
    """corner scan of MapMaker::AddPointEpipolar (src/MapMaker.cc:598-637).  src_level / tgt_level: dict(im, corners);
    ip: in-plane corners of the target level; q: one query record.  -> dict(best, best_zmssd, n_scored, template_bad)"""
    im = src_level["im"]
    h, w = im.shape
    x0, y0 = int(q["level_x"]), int(q["level_y"])
    if not (x0 >= 5 and y0 >= 5 and x0 < w - 5 and y0 < h - 5):
        return dict(best=-1, best_zmssd=MAX_SSD + 1, n_scored=0, template_bad=1)
    tmpl = im[y0 - 4:y0 + 4, x0 - 4:x0 + 4].reshape(64)
    best, bz, ns = -1, MAX_SSD + 1, 0
    nx, ny, ax, ay = float(q["normal"][0]), float(q["normal"][1]), float(q["along"][0]), float(q["along"][1])
    nd, lo, hi, md = float(q["norm_dist"]), float(q["min_len"]), float(q["max_len"]), float(q["max_dist_sq"])
    for i, (cx, cy) in enumerate(tgt_level["corners"]):
        vx, vy = float(ip[i][0]), float(ip[i][1])
        dd = nd - (vx * nx + vy * ny)
        if dd * dd > md:
            continue
        al = vx * ax + vy * ay
        if al < lo or al > hi:
            continue
        z = zmssd(tgt_level["im"], int(cx), int(cy), tmpl)
        ns += 1
        if z < bz:
            best, bz = i, z
    return dict(best=best, best_zmssd=bz, n_scored=ns, template_bad=0)


def make_template_coarse_cont(im, cx, cy, search_level, warp_inverse):
    """PatchFinder::MakeTemplateCoarseCont (src/PatchFinder.cc:98-127) without the host-side reuse test:
    CVD::transform of level image `im` (2-D uint8) with M = M2Inverse(mm2WarpInverse) * LevelScale(search_level),
    inOrig = (cx, cy), outOrig = (4, 4), then MakeTemplateSums.  libCVD's transform / sample restated from the
    library's published vision.h (unpinned): sequential position updates, bilinear value in double, truncation.
    Returns (template uint8[64], dict(bad, n_outside, sum, sum_sq, m2))."""
    wi = [float(v) for v in np.asarray(warp_inverse, dtype=np.float64).reshape(4)]
    det = wi[0] * wi[3] - wi[2] * wi[1]
    inv = 1.0 / det
    sc = float(1 << search_level)
    m00, m11, m10, m01 = wi[3] * inv * sc, wi[0] * inv * sc, -wi[2] * inv * sc, -wi[1] * inv * sc
    ih, iw = im.shape
    w = h = 8
    ax, ay, dx, dy = m00, m10, m01, m11
    p0x = float(cx) - (m00 * 4.0 + m01 * 4.0)
    p0y = float(cy) - (m10 * 4.0 + m11 * 4.0)
    min_x = max_x = p0x
    min_y = max_y = p0y
    if ax < 0: min_x += w * ax
    else: max_x += w * ax
    if dx < 0: min_x += h * dx
    else: max_x += h * dx
    if ay < 0: min_y += w * ay
    else: max_y += w * ay
    if dy < 0: min_y += h * dy
    else: max_y += h * dy
    crx, cry = dx - w * ax, dy - w * ay
    all_inside = min_x >= 0 and min_y >= 0 and max_x < iw - 1 and max_y < ih - 1
    out = np.zeros(64, dtype=np.uint8)
    count = 0
    px, py = p0x, p0y
    for i in range(h):
        for j in range(w):
            if all_inside or (0 <= px and 0 <= py and px < iw - 1 and py < ih - 1):
                lx, ly = int(px), int(py)
                x, y = px - lx, py - ly
                a, b, c, d = float(im[ly, lx]), float(im[ly, lx + 1]), float(im[ly + 1, lx]), float(im[ly + 1, lx + 1])
                v = (1 - y) * ((1 - x) * a + x * b) + y * ((1 - x) * c + x * d)
                out[i * 8 + j] = int(v)
            else:
                count += 1
            px += ax
            py += ay
        px += crx
        py += cry
    t = out.astype(np.int64)
    return out, dict(bad=int(count != 0), n_outside=count, sum=int(t.sum()), sum_sq=int((t * t).sum()),
                     m2=np.array([m00, m01, m10, m11]))


def find_patch_coarse(levels, q, tmpl):
    x, y, level, rng = int(q["x"]), int(q["y"]), int(q["level"]), int(q["range"])
    res = dict(found=0, best_ssd=MAX_SSD + 1, best_x=-1, best_y=-1, n_scored=0, pos=(0.0, 0.0))
    if level < 0 or level >= LEVELS:
        return res
    L = levels[level]
    h, w = L["im"].shape
    s = 1 << level
    px, py = trunc_div(x, s), trunc_div(y, s)
    r = (rng + s - 1) // s
    top, bot1 = max(py - r, 0), py + r + 1
    if top >= h or bot1 <= 0:
        return res
    i0 = L["rowlut"][top]
    i1 = len(L["corners"]) if bot1 >= h else L["rowlut"][bot1]
    for cx, cy in L["corners"][i0:i1]:
        if cx < px - r or cx > px + r or (px - cx) ** 2 + (py - cy) ** 2 > r * r:
            continue
        ssd = zmssd(L["im"], int(cx), int(cy), tmpl)
        res["n_scored"] += 1
        if ssd < res["best_ssd"]:
            res.update(best_ssd=ssd, best_x=int(cx), best_y=int(cy))
    if res["best_ssd"] < MAX_SSD:
        res["found"] = 1
        res["pos"] = ((res["best_x"] + 0.5) * s - 0.5, (res["best_y"] + 0.5) * s - 0.5)
    return res


# ---- sub-pixel refinement (src/PatchFinder.cc:219-318) -----------------------------------------------
def subpix(levels, coarse_pos, level, tmpl, max_its=8):
    """-> dict(converged, iterations, pos, mean_diff).  Gradient image + 3x3 normal matrix from the
    template, then inverse-compositional iterations with a float32 bilinear mix."""
    res = dict(converged=0, iterations=0, pos=np.array(coarse_pos, float), mean_diff=0.0)
    if level < 0 or level >= LEVELS:
        return res
    im = levels[level]["im"]
    h, w = im.shape
    T = np.asarray(tmpl, np.float64).reshape(8, 8)
    gx = 0.5 * (T[1:7, 2:8] - T[1:7, 0:6])            # [y-1][x-1]
    gy = 0.5 * (T[2:8, 1:7] - T[0:6, 1:7])
    G = np.stack([gx.ravel(), gy.ravel(), np.ones(36)], axis=1)
    Hinv = np.linalg.inv(G.T @ G)
    jx, jy = gx.astype(np.float32).astype(np.float64), gy.astype(np.float32).astype(np.float64)
    pos, mean_diff, s = np.array(coarse_pos, float), 0.0, 1 << level
    for it in range(max_its):
        res["iterations"] = it + 1
        c = (pos + 0.5) / s - 0.5
        r = np.where(c > 0, c + 0.5, c - 0.5).astype(int)      # ir_rounded
        if not (5 <= r[0] < w - 5 and 5 <= r[1] < h - 5):
            break
        base = c - 4
        d = base - np.floor(base)
        f32 = np.float32
        mix = [f32((1 - d[0]) * (1 - d[1])), f32(d[0] * (1 - d[1])), f32((1 - d[0]) * d[1]), f32(d[0] * d[1])]
        ib = base.astype(int)                                     # ir(): truncation
        win = im[ib[1] + 1:ib[1] + 8, ib[0] + 1:ib[0] + 8].astype(np.float32)
        pix = ((mix[0] * win[0:6, 0:6] + mix[1] * win[0:6, 1:7]) + mix[2] * win[1:7, 0:6]) + mix[3] * win[1:7, 1:7]
        diff = pix.astype(np.float64) - T[1:7, 1:7] + mean_diff
        acc = np.array([(diff * jx).sum(), (diff * jy).sum(), diff.sum()])
        upd = Hinv @ acc
        pos = pos - upd[:2] * s
        mean_diff -= upd[2]
        if upd[0] ** 2 + upd[1] ** 2 < 0.03 ** 2:
            res["converged"] = 1
            break
    res["pos"], res["mean_diff"] = pos, mean_diff
    return res


# ---- ATANCamera (src/ATANCamera.cc:27-66, 109-121, 179-209) -------------------------------------
class Camera:
    def __init__(self, params, size):
        fx, fy, cx, cy, w = params
        self.size = np.array(size, float)
        self.focal = self.size * [fx, fy]
        self.centre = self.size * [cx, cy] - 0.5
        self.w = w
        self.k = 2 * np.tan(w / 2) if w else 0.0
        rr = np.hypot(max(cx, 1 - cx) / fx, max(cy, 1 - cy) / fy)
        self.largest_radius = np.tan(rr * w) / self.k if w else rr
        self.max_r = 1.5 * self.largest_radius

    def project(self, xy):
        """-> image (N,2), r, factor"""
        xy = np.atleast_2d(xy)
        r = np.hypot(xy[:, 0], xy[:, 1])
        small = (r < 0.001) | (self.w == 0)
        f = np.ones_like(r)
        rs = np.where(small, 1.0, r)
        f = np.where(small, 1.0, np.arctan(rs * self.k) / (self.w * rs) if self.w else 1.0)
        return self.centre + self.focal * (f[:, None] * xy), r, f

    def derivs(self, xy, r, f):
        """2x2 d(image)/d(z=1 plane), shape (N,2,2)"""
        x, y = xy[:, 0], xy[:, 1]
        k, w = self.k, self.w
        re = r * (1.0 if w else 0.0)
        ok = re >= 0.01
        rs = np.where(ok, re, 1.0)
        common = np.where(ok, ((k / w if w else 0.0) / (1 + k * k * rs * rs) - f) / (rs * rs), 0.0)
        dfx, dfy = x * common, y * common
        D = np.zeros((len(x), 2, 2))
        D[:, 0, 0] = self.focal[0] * (dfx * x + f)
        D[:, 1, 0] = self.focal[1] * (dfx * y)
        D[:, 0, 1] = self.focal[0] * (dfy * x)
        D[:, 1, 1] = self.focal[1] * (dfy * y + f)
        return D


# ---- SE3 (TooN semantics) -------------------------------------------------------------
def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(mu):
    """closed-form exponential of the twist (t, w) -> pose (12,): R row-major + translation"""
    t, w = np.asarray(mu[:3], float), np.asarray(mu[3:], float)
    th2 = w @ w
    th = np.sqrt(th2)
    K = hat(w)
    if th2 < 1e-8:
        A, B, C = 1 - th2 / 6, 0.5, 0.0
        trans = t + 0.5 * np.cross(w, t)
    else:
        if th2 < 1e-6:
            C = (1 - th2 / 20) / 6
            A = 1 - th2 * C
            B = 0.5 - th2 / 24
        else:
            A, B = np.sin(th) / th, (1 - np.cos(th)) / th2
            C = (1 - A) / th2
        trans = t + B * np.cross(w, t) + C * np.cross(w, np.cross(w, t))
    R = np.eye(3) + A * K + B * (K @ K)
    return np.concatenate([R.ravel(), trans])


def se3_mul(a, b):
    Ra, Rb = a[:9].reshape(3, 3), b[:9].reshape(3, 3)
    return np.concatenate([(Ra @ Rb).ravel(), a[9:] + Ra @ b[9:]])


def se3_apply(T, X):
    return X @ T[:9].reshape(3, 3).T + T[9:]


def generators(Xc):
    """generator_field(m, (X,Y,Z,1)) for m = 0..5 -> (N,6,3)"""
    N = len(Xc)
    G = np.zeros((N, 6, 3))
    G[:, 0, 0] = G[:, 1, 1] = G[:, 2, 2] = 1
    X, Y, Z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
    G[:, 3, 1], G[:, 3, 2] = -Z, Y
    G[:, 4, 0], G[:, 4, 2] = Z, -X
    G[:, 5, 0], G[:, 5, 1] = -Y, X
    return G


def motion_to_plane(Xc, G):
    """d(x/z, y/z) for camera-frame motions G (N,K,3) -> (N,K,2)   (include/Tracker.h:132-133)"""
    iz = 1.0 / Xc[:, 2]
    mx = (G[:, :, 0] - Xc[:, None, 0] * G[:, :, 2] * iz[:, None]) * iz[:, None]
    my = (G[:, :, 1] - Xc[:, None, 1] * G[:, :, 2] * iz[:, None]) * iz[:, None]
    return np.stack([mx, my], axis=2)


# ---- M-estimators (include/Tools.h:128-228) -------------------------------------------------------
def sigma_squared(e2, est="Tukey"):
    v = np.sort(np.asarray(e2, float))
    n = len(v)
    den = (2 * n - 6) % (1 << 64)                  # size_t arithmetic
    with np.errstate(divide="ignore"):
        s = 1.4826 * (1 + np.float64(5.0) / np.float64(den)) * np.sqrt(v[n // 2])
    s *= 1.345 if est == "Huber" else 4.6851
    return s * s


def weight(e2, s2, est="Tukey"):
    e2 = np.asarray(e2, float)
    if est == "Tukey":
        return np.where(e2 > s2, 0.0, (1 - e2 / s2) ** 2)
    if est == "Cauchy":
        return 1 / (1 + e2 / s2)
    with np.errstate(divide="ignore"):
        return np.where(e2 < s2, 1.0, np.sqrt(s2 / np.where(e2 == 0, 1, e2)))


def sqrt_weight(e2, s2, est="Tukey"):
    if est == "Tukey":
        return np.where(e2 > s2, 0.0, 1 - e2 / s2)
    return np.sqrt(weight(e2, s2, est))


def objective(e2, s2, est="Tukey"):
    e2 = np.asarray(e2, float)
    if est == "Tukey":
        return np.where(e2 > s2, 1.0, 1 - (1 - e2 / s2) ** 3)
    if est == "Cauchy":
        return np.log(1 + e2 / s2)
    return np.where(e2 < s2, 0.5 * e2, np.sqrt(s2) * (np.sqrt(e2) - 0.5 * np.sqrt(s2)))


# ---- Tracker pose Gauss-Newton (src/Tracker.cc:613-643, 928-1005; include/Tracker.h:70-142) ------
def project_points(cam, pose, world):
    """TrackerData::Project -> dict(cam, image, derivs, in_image, reached)"""
    Xc = se3_apply(pose, world)
    n = len(world)
    out = dict(cam=Xc, image=np.zeros((n, 2)), derivs=np.zeros((n, 2, 2)), in_image=np.zeros(n, bool),
               reached=np.zeros(n, bool))
    ok = ~(Xc[:, 2] < 0.001)
    xy = np.zeros((n, 2))
    xy[ok] = Xc[ok, :2] / Xc[ok, 2:3]
    ok &= ~((xy ** 2).sum(1) > cam.largest_radius ** 2)
    im, r, f = cam.project(xy)
    out["reached"] = ok
    out["image"][ok] = im[ok]
    out["derivs"][ok] = cam.derivs(xy, r, f)[ok]
    inim = ok & ~(r > cam.max_r)
    inim &= ~((im[:, 0] < 0) | (im[:, 1] < 0) | (im[:, 0] > cam.size[0]) | (im[:, 1] > cam.size[1]))
    out["in_image"] = inim
    return out


def track_pvs(cam, pose, world, pixel_right_w, pixel_down_w):
    """TrackMap's PVS loop + CalcSearchLevelAndWarpMatrix (src/Tracker.cc:453-478, src/PatchFinder.cc:52-84)
    -> dict(in_image, image, derivs, warp_inverse (N,2,2), level, counts)"""
    pr = project_points(cam, pose, world)
    R = pose[:9].reshape(3, 3)
    Xc = pr["cam"]
    with np.errstate(divide="ignore", invalid="ignore"):
        M = np.stack([pixel_right_w @ R.T, pixel_down_w @ R.T], axis=1)                  # (N,2,3)
        W = np.einsum("nab,nkb->nak", pr["derivs"], motion_to_plane(Xc, M))              # columns = right, down
        det = W[:, 0, 0] * W[:, 1, 1] - W[:, 0, 1] * W[:, 1, 0]
    level = np.zeros(len(world), int)
    for _ in range(LEVELS - 1):
        up = (det > 3) & (level < LEVELS - 1)
        level[up] += 1
        det = np.where(up, det * 0.25, det)
    bad = (det > 3) | (det < 0.25) | ~pr["in_image"] | ~np.isfinite(det)
    level[bad] = -1
    W[~pr["in_image"]] = 0
    counts = np.array([(level == l).sum() for l in range(LEVELS)], np.int32)
    return dict(in_image=pr["in_image"], image=pr["image"], derivs=pr["derivs"], cam=Xc, warp_inverse=W, level=level,
                counts=counts)


def refind(cam, levels_k, pose_k, world, pixel_right_w, pixel_down_w, src_images, centers):
    """MapMaker::ReFind_Common (src/MapMaker.cc:943-1020) for a batch of points against one keyframe, composed from the
    pieces above: projection + visibility tests, CalcSearchLevelAndWarpMatrix (whose -1 verdict the reference does not look
    at: the template is made at the level the loop stopped at), MakeTemplateCoarseCont, FindPatchCoarse with range 4,
    sub-pixel refinement for level > 0 (convergence ignored).  src_images[i]: the source level image of point i.
    -> list of dict(found, level, sub_pix, never_retry, root_pos)"""
    pr = project_points(cam, pose_k, world)
    R = pose_k[:9].reshape(3, 3)
    out = []
    for i in range(len(world)):
        r = dict(found=0, level=-1, sub_pix=0, never_retry=1, root_pos=np.zeros(2))
        out.append(r)
        if not pr["in_image"][i]:
            continue
        Xc = pr["cam"][i:i + 1]
        M = np.stack([pixel_right_w[i:i + 1] @ R.T, pixel_down_w[i:i + 1] @ R.T], axis=1)
        W = np.einsum("nab,nkb->nak", pr["derivs"][i:i + 1], motion_to_plane(Xc, M))[0]
        det = W[0, 0] * W[1, 1] - W[0, 1] * W[1, 0]
        level = 0
        while det > 3 and level < LEVELS - 1:
            level += 1
            det *= 0.25
        tmpl, tr = make_template_coarse_cont(src_images[i], int(centers[i][0]), int(centers[i][1]), level, W.reshape(4))
        r["level"] = level
        if tr["bad"]:
            continue
        q = dict(x=int(pr["image"][i][0]), y=int(pr["image"][i][1]), level=level, range=4)
        res = find_patch_coarse(levels_k, q, tmpl)
        if not res["found"]:
            continue
        r["found"], r["never_retry"] = 1, 0
        if level > 0:
            sp = subpix(levels_k, res["pos"], level, tmpl, 8)
            r["root_pos"], r["sub_pix"] = np.array(sp["pos"], float), 1
        else:
            r["root_pos"] = np.array(res["pos"], float)
    return out


def refind_pairs(cam, pairs, state):
    """MapMaker::ReFind_Common as its callers run it (src/MapMaker.cc:1046-1082): (keyframe, point) pairs in order through ONE
    PatchFinder — `static PatchFinder Finder`, :977 — whose state `state` (a dict the caller keeps) outlives the call:
    MakeTemplateCoarseCont (src/PatchFinder.cc:98-127) keeps template and mbTemplateBad when it last warped this map point and
    no column of the warp m2 moved by more than 0.07; CalcSearchLevelAndWarpMatrix sets mbTemplateBad when it rejects a warp
    (:78-81).  pairs: dicts with levels_k, pose_k, world, pixel_right_w, pixel_down_w, src_image, center, point_id, skip.
    -> list of dict(found, level, sub_pix, never_retry, root_pos, kept)"""
    out = []
    for pr_ in pairs:
        r = dict(found=0, level=-1, sub_pix=0, never_retry=0, root_pos=np.zeros(2), kept=0)
        out.append(r)
        if pr_["skip"]:                                             # :947-948
            continue
        r["never_retry"] = 1
        pose_k = pr_["pose_k"]
        pr = project_points(cam, pose_k, pr_["world"][None, :])
        if not pr["in_image"][0]:                                   # :950-975
            continue
        R = pose_k[:9].reshape(3, 3)
        M = np.stack([pr_["pixel_right_w"][None, :] @ R.T, pr_["pixel_down_w"][None, :] @ R.T], axis=1)
        W = np.einsum("nab,nkb->nak", pr["derivs"][0:1], motion_to_plane(pr["cam"][0:1], M))[0]
        det = W[0, 0] * W[1, 1] - W[0, 1] * W[1, 0]
        level = 0
        while det > 3 and level < LEVELS - 1:
            level += 1
            det *= 0.25
        if det > 3 or det < 0.25:                                   # src/PatchFinder.cc:78-81; the -1 is not looked at (:979)
            state["bad"] = True
        m2 = np.linalg.inv(W) * (1 << level)                        # :101   (rows; columns are m2.T()[i])
        need = state.get("point") != pr_["point_id"]                 # :103
        if not need:
            d = m2 - state["m2"]
            need = bool((d[:, 0] @ d[:, 0] > 0.07 ** 2) or (d[:, 1] @ d[:, 1] > 0.07 ** 2))   # :105-110
        if need:
            tmpl, tr = make_template_coarse_cont(pr_["src_image"], int(pr_["center"][0]), int(pr_["center"][1]), level, W.reshape(4))
            state.update(point=pr_["point_id"], m2=m2, tmpl=tmpl, bad=bool(tr["bad"]))   # :112-123
        else:
            r["kept"] = 1
        r["level"] = level
        if state["bad"]:                                            # :982-986
            continue
        q = dict(x=int(pr["image"][0][0]), y=int(pr["image"][0][1]), level=level, range=4)
        res = find_patch_coarse(pr_["levels_k"], q, state["tmpl"])
        if not res["found"]:
            continue
        r["found"], r["never_retry"] = 1, 0
        if level > 0:
            sp = subpix(pr_["levels_k"], res["pos"], level, state["tmpl"], 8)
            r["root_pos"], r["sub_pix"] = np.array(sp["pos"], float), 1
        else:
            r["root_pos"] = np.array(res["pos"], float)
    return out


def calc_pose_update(found, image, s, J, override=0.0, est="Tukey", prior=100.0):
    """-> (mu, weight_zero_mask)   J: (N,2,6)"""
    if len(found) == 0:
        return np.zeros(6), np.zeros(0, bool)
    e = s[:, None] * (found - image)
    e2 = (e ** 2).sum(1)
    s2 = override if override > 0 else sigma_squared(e2, est)
    wgt = weight(e2, s2, est)
    Js = s[:, None, None] * J
    C = np.eye(6) * prior + np.einsum("n,nri,nrj->ij", wgt, Js, Js)
    b = np.einsum("n,nr,nri->i", wgt, e, Js)
    return np.linalg.solve(C, b), wgt == 0


def pose_gn(cam, world, found, s, pose, iterations=10, nonlinear_mask=0x211, override_after=5, override_sigma_sq=16.0,
            mark_outliers_iter=9, est="Tukey", prior=100.0):
    pose = np.array(pose, float)
    pr = project_points(cam, pose, world)
    use = pr["in_image"].copy()
    Xc, image, D = pr["cam"].copy(), pr["image"].copy(), pr["derivs"].copy()
    J = np.zeros((len(world), 2, 6))
    flags = np.zeros(len(world), np.int32)
    updates, last = [], np.zeros(6)
    for it in range(iterations):
        nonlinear = (nonlinear_mask >> it) & 1
        if it:
            if nonlinear:
                pr = project_points(cam, pose, world)
                Xc = pr["cam"]
                upd = pr["reached"]
                image[upd], D[upd] = pr["image"][upd], pr["derivs"][upd]
            else:
                image = image + J @ last
        if nonlinear:
            with np.errstate(divide="ignore", invalid="ignore"):
                J = np.einsum("nab,nkb->nak", D, motion_to_plane(Xc, generators(Xc)))
        ov = override_sigma_sq if it > override_after else 0.0
        mu, wz = calc_pose_update(found[use], image[use], s[use], J[use], ov, est, prior)
        if it == mark_outliers_iter:
            flags[np.flatnonzero(use)[wz]] = 1
        pose = se3_mul(se3_exp(mu), pose)
        last = mu
        updates.append(mu)
    return pose, flags, np.array(updates)


# ---- Bundle (src/Bundle.cc) with dense linear algebra ---------------------------------------------
def bundle_adjust(cam, prob, max_iterations=20, conv_limit=1e-6, min_sigma=0.4, est="Tukey"):
    poses, pts = prob["poses"].copy(), prob["points"].copy()
    fixed = prob["fixed"].astype(bool)
    ci, pi = prob["cam_idx"].astype(int), prob["pt_idx"].astype(int)
    found, sn = prob["found"].copy(), np.sqrt(1.0 / prob["sigma_sq"])
    alive = np.ones(len(ci), bool)
    free_of = -np.ones(len(poses), int)
    free_of[~fixed] = np.arange((~fixed).sum())
    nF, nP = int((~fixed).sum()), len(pts)
    n = 6 * nF

    def residuals(P, X):
        Xc = np.einsum("nij,nj->ni", P[ci, :9].reshape(-1, 3, 3), X[pi]) + P[ci, 9:]
        bad = Xc[:, 2] <= 0
        z = np.where(bad, 1.0, Xc[:, 2])
        xy = Xc[:, :2] / z[:, None]
        im, r, f = cam.project(xy)
        eps = sn[:, None] * (found - im)
        return Xc, xy, r, f, eps, bad

    lam, lam_f = 1e-4, 2.0
    converged = hit_max = False
    counter = accepted = 0
    trials, outliers = [], []
    while not converged and not hit_max:
        Xc, xy, r, f, eps, bad1 = residuals(poses, pts)
        bad = alive & bad1
        ok = alive & ~bad1
        e2 = (eps ** 2).sum(1)
        s2 = max(sigma_squared(e2[ok], est), min_sigma ** 2)
        w = np.where(ok, sqrt_weight(e2, s2, est), 0.0)
        bad |= ok & (w == 0)
        use = ok & (w != 0)
        cur_err = bad.sum() + objective(e2[use], s2, est).sum()
        D = cam.derivs(xy, r, f) * (w * sn)[:, None, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            A = np.einsum("nab,nkb->nak", D, motion_to_plane(Xc, generators(Xc)))          # (N,2,6)
            Rcols = np.transpose(poses[ci, :9].reshape(-1, 3, 3), (0, 2, 1))                # k-th column of R
            B = np.einsum("nab,nkb->nak", D, motion_to_plane(Xc, Rcols))                     # (N,2,3)
        A[~use] = 0
        B[~use] = 0
        A[fixed[ci]] = 0
        epsw = eps * w[:, None]
        epsw[~use] = 0
        U = np.zeros((nF, 6, 6))
        eA = np.zeros((nF, 6))
        fi = free_of[ci]
        sel = use & (fi >= 0)
        np.add.at(U, fi[sel], np.einsum("nri,nrj->nij", A[sel], A[sel]))
        np.add.at(eA, fi[sel], np.einsum("nri,nr->ni", A[sel], epsw[sel]))
        V = np.zeros((nP, 3, 3))
        eB = np.zeros((nP, 3))
        np.add.at(V, pi[use], np.einsum("nri,nrj->nij", B[use], B[use]))
        np.add.at(eB, pi[use], np.einsum("nri,nr->ni", B[use], epsw[use]))
        W = np.einsum("nri,nrj->nij", A, B)                                                  # (N,6,3)
        new_err = cur_err + 9999
        ran = False
        while new_err > cur_err and not converged and not hit_max:
            ran = True
            Vs = V.copy()
            Vs[:, [0, 1, 2], [0, 1, 2]] *= (1 + lam)
            deg = (V[:, 0, 0] * V[:, 1, 1] * V[:, 2, 2]) == 0
            Vs[deg] = np.eye(3)
            Vinv = np.linalg.inv(Vs)
            Vinv[deg] = 0
            S = np.zeros((n, n))
            E = np.zeros(n)
            for j in range(nF):
                Uj = U[j].copy()
                Uj[range(6), range(6)] *= (1 + lam)
                S[6 * j:6 * j + 6, 6 * j:6 * j + 6] = Uj
                E[6 * j:6 * j + 6] = eA[j]
            Y = np.einsum("nij,njk->nik", W, Vinv[pi])                                        # (N,6,3)
            msel = np.flatnonzero(sel)
            order = msel[np.argsort(pi[msel], kind="stable")]
            ptr = np.searchsorted(pi[order], np.arange(nP + 1))
            for p in range(nP):
                ms = order[ptr[p]:ptr[p + 1]]
                if len(ms) == 0:
                    continue
                Yp = Y[ms].reshape(-1, 3)                                                     # (6k,3)
                Wp = W[ms].reshape(-1, 3)
                rows = (6 * fi[ms][:, None] + np.arange(6)).ravel()
                S[np.ix_(rows, rows)] -= Yp @ Wp.T
                E[rows] -= Yp @ eB[p]
            da = np.linalg.solve(S, E) if n else np.zeros(0)
            t = np.zeros((nP, 3))
            np.add.at(t, pi[sel], np.einsum("nij,ni->nj", W[sel], da.reshape(-1, 6)[fi[sel]]))
            db = np.einsum("pij,pj->pi", Vinv, eB - t)
            sumsq = float(da @ da + (db ** 2).sum())
            if sumsq < conv_limit:
                converged = True
            new_poses = poses.copy()
            for c in np.flatnonzero(~fixed):
                new_poses[c] = se3_mul(se3_exp(da[6 * free_of[c]:6 * free_of[c] + 6]), poses[c])
            new_pts = pts + db
            _, _, _, _, eps_n, bad_n = residuals(new_poses, new_pts)
            e2n = (eps_n ** 2).sum(1)
            new_err = float((alive & bad_n).sum() + objective(e2n[alive & ~bad_n], s2, est).sum())
            trials.append(dict(lam=lam, sigma_sq=s2, err_old=float(cur_err), err_new=new_err, sumsq=sumsq,
                               n_bad=int(bad.sum()), accepted=0))
            if new_err > cur_err:
                lam *= lam_f
                lam_f *= 2
            counter += 1
            if counter >= max_iterations:
                hit_max = True
        if ran and new_err < cur_err:
            lam_f = 2.0
            lam *= 0.3
            poses, pts = new_poses, new_pts
            accepted += 1
            trials[-1]["accepted"] = 1
        for m in np.flatnonzero(bad):
            outliers.append((int(pi[m]), int(ci[m])))
        alive &= ~bad
    return dict(poses=poses, points=pts, trials=trials, outliers=outliers, accepted=accepted, converged=converged)
