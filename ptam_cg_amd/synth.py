"""Seeded synthetic inputs for the hot path: frames, patch queries, pose-GN cases and
bundle problems.  Pure numpy; used by tests/ and bench.py.  Nothing here depends on the oracle."""
import numpy as np

SEED_FRAME = 0x5EED0001
SEED_QUERIES = 0x5EED0002
SEED_POSE = 0x5EED0003
SEED_BA_LOCAL = 0x5EED0004
SEED_BA_HEADLINE = 0x5EED0005
SEED_BA_GLOBAL = 0x5EED0006

DEFAULT_CAMERA = (1.0803, 1.43987, 0.519983, 0.548655, 0.244943)   # config/camera.cfg:7
LEVEL_MIX = (0.50, 0.25, 0.15, 0.10)


# ---------------------------------------------------------------------------------------------
# frames
# ---------------------------------------------------------------------------------------------
def make_frame(seed=SEED_FRAME, w=640, h=480, n_rect=400, shift=(0, 0), noise_seed=None):
    """Rectangles + integer noise frame.  `shift` translates the rectangle layout (dx, dy); the
    noise is drawn from `noise_seed` (default: seed) so a shifted frame can carry fresh noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    im = np.full((h, w), 128, dtype=np.int32)
    rw = rng.integers(32, 129, n_rect)
    rh = rng.integers(32, 129, n_rect)
    x0 = rng.integers(-64, w, n_rect)
    y0 = rng.integers(-64, h, n_rect)
    val = rng.integers(30, 226, n_rect)
    for i in range(n_rect):
        xa, ya = x0[i] + shift[0], y0[i] + shift[1]
        xb, yb = xa + rw[i], ya + rh[i]
        xa, ya, xb, yb = max(xa, 0), max(ya, 0), min(xb, w), min(yb, h)
        if xb > xa and yb > ya:
            im[ya:yb, xa:xb] = val[i]
    nrng = np.random.Generator(np.random.PCG64(seed if noise_seed is None else noise_seed))
    im += nrng.integers(-3, 4, (h, w))
    return np.clip(im, 0, 255).astype(np.uint8)


def make_frame_pair(seed=SEED_FRAME, shift=(3, -2)):
    a = make_frame(seed)
    b = make_frame(seed, shift=shift, noise_seed=seed + 1)
    return a, b


def make_patch_queries(levels_a, n=1000, seed=SEED_QUERIES, shift=(3, -2), search_range=10, jitter=4):
    """Queries for FindPatchCoarse in frame B built from corners of frame A.
    levels_a: list of 4 dicts {im, corners} (KeyFrame.level(l) of frame A).
    Returns (queries structured array fields x,y,level,range ; templates (n,64) uint8)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    q = np.zeros(n, dtype=[("x", "<i4"), ("y", "<i4"), ("level", "<i4"), ("range", "<u4")])
    tm = np.zeros((n, 64), dtype=np.uint8)
    lv = rng.choice(4, size=n, p=LEVEL_MIX)
    for i in range(n):
        l = int(lv[i])
        im, corners = levels_a[l]["im"], levels_a[l]["corners"]
        hh, ww = im.shape
        # MakeTemplateCoarseNoWarp needs in_image_with_border(pos, 5)  (src/PatchFinder.cc:141)
        ok = corners[(corners[:, 0] >= 5) & (corners[:, 1] >= 5) & (corners[:, 0] < ww - 5) & (corners[:, 1] < hh - 5)]
        if len(ok) == 0:
            q[i] = (0, 0, -1, search_range)
            continue
        cx, cy = ok[rng.integers(len(ok))]
        tm[i] = im[cy - 4:cy + 4, cx - 4:cx + 4].reshape(64)
        s = 1 << l
        px = (cx + 0.5) * s - 0.5 + shift[0] + rng.uniform(-jitter, jitter)
        py = (cy + 0.5) * s - 0.5 + shift[1] + rng.uniform(-jitter, jitter)
        q[i] = (int(px), int(py), l, search_range)   # ir(): truncation toward zero
    return q, tm


# ---------------------------------------------------------------------------------------------
# camera + SE3 helpers (numpy, for data generation only)
# ---------------------------------------------------------------------------------------------
class AtanCam:
    def __init__(self, params=DEFAULT_CAMERA, size=(640, 480)):
        fx, fy, cx, cy, w = params
        self.size = size
        self.focal = np.array([size[0] * fx, size[1] * fy])
        self.centre = np.array([size[0] * cx - 0.5, size[1] * cy - 0.5])
        self.w = w
        self.two_tan = 2.0 * np.tan(w / 2.0) if w != 0 else 0.0
        v = np.array([max(cx, 1 - cx) / fx, max(cy, 1 - cy) / fy])
        r = np.hypot(*v)
        self.largest_radius = np.tan(r * w) / self.two_tan if w != 0 else r
        self.max_r = 1.5 * self.largest_radius

    def project(self, xy):
        xy = np.asarray(xy, dtype=np.float64)
        r = np.hypot(xy[..., 0], xy[..., 1])
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.where((r < 0.001) | (self.w == 0), 1.0, np.arctan(r * self.two_tan) / (self.w * np.where(r == 0, 1, r)))
        return self.centre + self.focal * (f[..., None] * xy), r

    def unproject(self, im):
        """ATANCamera::UnProject (src/ATANCamera.cc:125-140), vectorised"""
        im = np.asarray(im, dtype=np.float64)
        dc = (im - self.centre) * (1.0 / self.focal)
        dr = np.hypot(dc[..., 0], dc[..., 1])
        rr = np.tan(dr * self.w) / self.two_tan if self.w != 0 else dr
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.where(dr > 0.01, rr / np.where(dr == 0, 1, dr), 1.0)
        return f[..., None] * dc

    def visible(self, pose, X):
        """TrackerData::Project visibility (include/Tracker.h:70-85) -> (mask, image coords)"""
        R, t = pose[:9].reshape(3, 3), pose[9:]
        Xc = X @ R.T + t
        z = Xc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            xy = Xc[:, :2] / z[:, None]
        im, r = self.project(np.nan_to_num(xy))
        ok = (z >= 0.001) & (r * r <= self.largest_radius ** 2) & (r <= self.max_r)
        ok &= (im[:, 0] >= 0) & (im[:, 1] >= 0) & (im[:, 0] <= self.size[0]) & (im[:, 1] <= self.size[1])
        return ok, im


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def se3_exp(mu):
    """4x4 matrix exponential of the twist (t, w) — closed form."""
    t, w = np.asarray(mu[:3], float), np.asarray(mu[3:], float)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = so3_exp(w)
    if th < 1e-9:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return np.concatenate([R.reshape(9), V @ t])


def se3_mul(a, b):
    Ra, ta = a[:9].reshape(3, 3), a[9:]
    Rb, tb = b[:9].reshape(3, 3), b[9:]
    return np.concatenate([(Ra @ Rb).reshape(9), ta + Ra @ tb])


def look_at(cam_pos, target, up=(0.0, 0.0, 1.0)):
    """camera-from-world pose (12,) with +z looking from cam_pos to target, image y pointing 'down'."""
    cam_pos, target = np.asarray(cam_pos, float), np.asarray(target, float)
    z = target - cam_pos
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, float))
    if np.linalg.norm(x) < 1e-9:
        x = np.array([1.0, 0, 0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    return np.concatenate([R.reshape(9), -R @ cam_pos])


# ---------------------------------------------------------------------------------------------
# pose Gauss-Newton case
# ---------------------------------------------------------------------------------------------
def make_pose_case(n=1000, seed=SEED_POSE, camera=DEFAULT_CAMERA, size=(640, 480), outlier_frac=0.05):
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = AtanCam(camera, size)
    # 1.5 m above the plane, looking down with a 0.1 rad tilt about x
    base = np.concatenate([np.diag([1.0, -1.0, -1.0]).reshape(9), [0, 0, 1.5]])
    tilt = np.concatenate([so3_exp(np.array([0.1, 0, 0])).reshape(9), [0, 0, 0]])
    true_pose = se3_mul(tilt, base)
    pts = np.zeros((0, 3))
    while len(pts) < n:
        cand = np.column_stack([rng.uniform(-0.45, 0.45, 2 * n), rng.uniform(-0.45, 0.45, 2 * n),
                                rng.uniform(-0.1, 0.1, 2 * n)])
        ok, _ = cam.visible(true_pose, cand)
        pts = np.vstack([pts, cand[ok]])
    pts = pts[:n]
    _, im = cam.visible(true_pose, pts)
    lv = rng.choice(4, size=n, p=LEVEL_MIX)
    found = im + rng.normal(0, 1, (n, 2)) * (0.5 * 2.0 ** lv)[:, None]
    out = rng.random(n) < outlier_frac
    sign = rng.choice([-1.0, 1.0], (n, 2))
    found[out] += sign[out] * rng.uniform(10, 30, (n, 2))[out]
    xi = np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.01, 3)])
    init_pose = se3_mul(se3_exp(xi), true_pose)
    # keep only points visible from the initial pose too (TrackMap's PVS test)
    ok, _ = cam.visible(init_pose, pts)
    keep = np.flatnonzero(ok)
    return {"world": pts[keep], "found": found[keep], "sqrt_inv_noise": 1.0 / 2.0 ** lv[keep],
            "init_pose": init_pose, "true_pose": true_pose, "is_outlier": out[keep]}


def make_pvs_case(n=4000, seed=0x5EED0007, camera=DEFAULT_CAMERA, size=(640, 480)):
    """Map points for TrackMap's PVS loop: positions (many outside the view), and the world-frame
    one-pixel-right / one-pixel-down vectors MapPoint::RefreshPixelVectors (src/Map.cc:40-65) would give
    for a fronto-parallel patch seen from a source keyframe at pyramid level l."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = AtanCam(camera, size)
    pc = make_pose_case(n=16, seed=seed + 1, camera=camera, size=size)
    src_pose, cur_pose = pc["true_pose"], pc["init_pose"]
    world = np.column_stack([rng.uniform(-0.9, 0.9, n), rng.uniform(-0.9, 0.9, n), rng.uniform(-0.3, 1.6, n)])
    R = src_pose[:9].reshape(3, 3)
    depth = (world @ R.T + src_pose[9:])[:, 2]
    lv = rng.choice(4, size=n, p=LEVEL_MIX)
    scale = rng.uniform(0.3, 3.0, n) * (2.0 ** lv) * np.abs(depth) / cam.focal[0]
    right = scale[:, None] * R[0] + rng.normal(0, 0.02, (n, 3)) * scale[:, None]
    down = scale[:, None] * (cam.focal[0] / cam.focal[1]) * R[1] + rng.normal(0, 0.02, (n, 3)) * scale[:, None]
    return {"world": world, "pixel_right_w": right, "pixel_down_w": down, "pose": cur_pose}


def make_trackmap_case(levels_a, counts=(800, 300, 80, 40), seed=0x5EED000A, camera=DEFAULT_CAMERA, size=(640, 480),
                       shift=(3, -2), height=1.5, pose_noise=(0.004, 0.004), n_junk=24):
    """A map for Tracker::TrackMap against the synthetic frame pair: the source keyframe (frame A) looks straight down at the
    plane z = 0 from `height`; every map point is the back-projection of a FAST corner of frame A at its pyramid level
    (`counts` corners of levels 0..3), with the world-frame one-pixel-right / -down vectors of that level
    (MapPoint::RefreshPixelVectors, src/Map.cc:40-65).  The current frame (frame B = A shifted by `shift` pixels) corresponds
    to a camera translation parallel to the plane; `pose_in` is that pose perturbed by N(0, pose_noise) (m, rad) — the
    motion model's prediction.  A few junk points (behind the camera, outside the view, absurd pixel vectors) ride along.
    levels_a: KeyFrame.level(l) of frame A for l = 0..3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = AtanCam(camera, size)
    src_pose = np.concatenate([np.diag([1.0, -1.0, -1.0]).reshape(9), [0, 0, height]])
    R, t = src_pose[:9].reshape(3, 3), src_pose[9:]

    def back_project(px0):                      # level-0 pixel -> world point on the plane
        xy = cam.unproject(px0)
        camc = np.column_stack([xy * height, np.full(len(xy), height)])
        return (camc - t) @ R                   # R^T (c - t)

    world, right, down, lev, cen = [], [], [], [], []
    for l, n_l in enumerate(counts):
        c = levels_a[l]["corners"]
        hh, ww = levels_a[l]["im"].shape
        m = 14
        c = c[(c[:, 0] >= m) & (c[:, 1] >= m) & (c[:, 0] < ww - m) & (c[:, 1] < hh - m)]
        c = c[rng.permutation(len(c))[:n_l]]
        s_ = float(1 << l)
        p0 = (c + 0.5) * s_ - 0.5               # LevelZeroPos
        w0 = back_project(p0)
        world.append(w0)
        right.append(back_project(p0 + [s_, 0.0]) - w0)
        down.append(back_project(p0 + [0.0, s_]) - w0)
        lev.append(np.full(len(c), l, np.int32))
        cen.append(c.astype(np.int32))
    world, right, down = np.vstack(world), np.vstack(right), np.vstack(down)
    lev, cen = np.concatenate(lev), np.vstack(cen)
    # junk: behind the camera / far outside / degenerate warps
    j = n_junk
    jw = np.column_stack([rng.uniform(-6, 6, j), rng.uniform(-6, 6, j), rng.uniform(-1, 4, j)])
    jr = rng.normal(0, 1, (j, 3)) * rng.choice([1e-6, 1e-3, 1.0], j)[:, None]
    jd = rng.normal(0, 1, (j, 3)) * rng.choice([1e-6, 1e-3, 1.0], j)[:, None]
    world, right, down = np.vstack([world, jw]), np.vstack([right, jr]), np.vstack([down, jd])
    lev = np.concatenate([lev, rng.integers(0, 4, j).astype(np.int32)])
    cen = np.vstack([cen, np.column_stack([rng.integers(14, 60, j), rng.integers(14, 40, j)]).astype(np.int32)])
    perm = rng.permutation(len(world))          # map order is not level order
    world, right, down, lev, cen = world[perm], right[perm], down[perm], lev[perm], cen[perm]
    # current pose: the scene moves by `shift` pixels in the image
    tcam = np.array([shift[0] * height / cam.focal[0], shift[1] * height / cam.focal[1], 0.0])
    cur_pose = se3_mul(np.concatenate([np.eye(3).reshape(9), tcam]), src_pose)
    xi = np.concatenate([rng.normal(0, pose_noise[0], 3), rng.normal(0, pose_noise[1], 3)])
    pose_in = se3_mul(se3_exp(xi), cur_pose)
    n = len(world)
    return {"world": world, "pixel_right_w": right, "pixel_down_w": down, "src_level": lev, "center": cen,
            "src_pose": src_pose, "cur_pose": cur_pose, "pose_in": pose_in,
            "shuffle_levels": rng.permutation(n).astype(np.int32), "shuffle_fine": rng.permutation(n).astype(np.int32)}


# ---------------------------------------------------------------------------------------------
# a tracked SEQUENCE: a camera that moves over a textured plane (Tracker::TrackFrame, src/Tracker.cc:86-200)
# ---------------------------------------------------------------------------------------------
SEED_SEQUENCE = 0x5EED000B
_TEXEL = 0.00125          # metres per texel of the plane's texture
_TEX_HALF = 2.0           # the texture covers [-2, 2]^2 m of the plane z = 0


def make_plane_texture(seed=SEED_SEQUENCE):
    """rectangles + integer noise like make_frame, on a 3200 x 3200 texel plane (same rectangle density per pixel seen from
    1.5 m as the 640 x 480 frames)"""
    n = int(round(2 * _TEX_HALF / _TEXEL))
    return make_frame(seed, w=n, h=n, n_rect=int(400 * n * n / (640 * 480)))


def sequence_pose(k, n, height=1.5):
    """camera-from-world pose (12,) of frame k of a closed n-frame trajectory: the camera looks down at the plane from about
    `height`, swings sideways and fore-and-aft, rises and sinks by 8 %, rolls about its optical axis by +-0.25 rad and tilts
    by a few degrees — per frame (n = 64) up to ~5 px of image motion and ~1.4 degrees of roll, so that a share of the warps
    crosses MakeTemplateCoarseCont's 0.07 refresh limit (src/PatchFinder.cc:103-111) every frame."""
    ph = 2.0 * np.pi * k / n
    pos = np.array([0.12 * np.sin(ph), 0.08 * np.sin(2 * ph), height * (1.0 + 0.08 * np.sin(ph + 0.7))])
    base = np.diag([1.0, -1.0, -1.0])                                    # looking straight down, image y = -world y
    w = np.array([0.05 * np.sin(2 * ph + 1.0), 0.04 * (np.cos(ph) - 1.0), 0.25 * np.sin(ph)])
    R = so3_exp(w) @ base
    return np.concatenate([R.reshape(9), -R @ pos])


def render_plane_view(cam, pose, texture, noise_rng=None):
    """the image the ATAN camera at `pose` sees of the textured plane z = 0: every pixel's ray (ATANCamera::UnProject) is
    intersected with the plane and the texture sampled bilinearly; +-2 grey levels of sensor noise"""
    w, h = cam.size
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xy = cam.unproject(np.stack([u, v], axis=-1))
    R, t = pose[:9].reshape(3, 3), pose[9:]
    ray = np.concatenate([xy, np.ones(xy.shape[:2] + (1,))], axis=-1) @ R      # R^T d  (world direction), row vectors
    org = -R.T @ t                                                              # camera centre in the world
    with np.errstate(divide="ignore", invalid="ignore"):
        s = -org[2] / ray[..., 2]
    X = org[0] + s * ray[..., 0]
    Y = org[1] + s * ray[..., 1]
    tx = (X + _TEX_HALF) / _TEXEL - 0.5
    ty = (Y + _TEX_HALF) / _TEXEL - 0.5
    n = texture.shape[0]
    ok = (s > 0) & (tx >= 0) & (ty >= 0) & (tx < n - 1) & (ty < n - 1)
    tx, ty = np.where(ok, tx, 0.0), np.where(ok, ty, 0.0)
    x0, y0 = tx.astype(np.int64), ty.astype(np.int64)
    fx, fy = tx - x0, ty - y0
    tex = texture.astype(np.float64)
    val = ((1 - fy) * ((1 - fx) * tex[y0, x0] + fx * tex[y0, x0 + 1]) + fy * ((1 - fx) * tex[y0 + 1, x0] + fx * tex[y0 + 1, x0 + 1]))
    val = np.where(ok, val, 128.0)
    if noise_rng is not None:
        val = val + noise_rng.integers(-2, 3, val.shape)
    return np.clip(np.floor(val + 0.5), 0, 255).astype(np.uint8)


def sequence_keyframe_pose(height=1.5):
    """the pose of the map's source keyframe: near the trajectory's start but on none of its frames (a tracked frame that IS
    the patch source matches to the last bit — half of the reprojection errors are then exactly zero and Tukey's sigma with
    them, include/Tools.h:156-165)"""
    pos = np.array([0.017, -0.023, 0.97 * height])
    R = so3_exp(np.array([0.012, -0.02, 0.03])) @ np.diag([1.0, -1.0, -1.0])
    return np.concatenate([R.reshape(9), -R @ pos])


def make_tracking_frames(n_frames=64, period=64, seed=SEED_SEQUENCE, camera=DEFAULT_CAMERA, size=(640, 480)):
    """-> (frames (n, h, w) u8, true poses (n, 12), keyframe image, keyframe pose): the first n_frames of the closed trajectory
    `sequence_pose` of `period` frames (n_frames == period: frame 0 follows the last one) and the view from
    `sequence_keyframe_pose` the map is made of"""
    cam = AtanCam(camera, size)
    tex = make_plane_texture(seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    poses = np.stack([sequence_pose(k, period) for k in range(n_frames)])
    frames = np.stack([render_plane_view(cam, poses[k], tex, rng) for k in range(n_frames)])
    kf_pose = sequence_keyframe_pose()
    return frames, poses, render_plane_view(cam, kf_pose, tex, rng), kf_pose


def make_sequence_map(levels_0, pose_0, counts=(800, 300, 80, 40), seed=SEED_SEQUENCE + 2, camera=DEFAULT_CAMERA, size=(640, 480),
                      n_junk=24):
    """The map of the sequence: every point is the back-projection of a FAST corner of the source keyframe (levels_0 =
    KeyFrame.level(l) of its image, `counts` corners per level) onto the plane z = 0 through the camera at pose_0, with the
    one-pixel-right / -down world vectors of its level (MapPoint::RefreshPixelVectors, src/Map.cc:40-65); junk points as in
    make_trackmap_case.  That keyframe is every point's patch source."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = AtanCam(camera, size)
    R, t = pose_0[:9].reshape(3, 3), pose_0[9:]
    org = -R.T @ t

    def back_project(px0):
        xy = cam.unproject(px0)
        ray = np.column_stack([xy, np.ones(len(xy))]) @ R
        s = -org[2] / ray[:, 2]
        return org + s[:, None] * ray

    world, right, down, lev, cen = [], [], [], [], []
    for l, n_l in enumerate(counts):
        c = levels_0[l]["corners"]
        hh, ww = levels_0[l]["im"].shape
        m = 14
        c = c[(c[:, 0] >= m) & (c[:, 1] >= m) & (c[:, 0] < ww - m) & (c[:, 1] < hh - m)]
        c = c[rng.permutation(len(c))[:n_l]]
        s_ = float(1 << l)
        p0 = (c + 0.5) * s_ - 0.5
        w0 = back_project(p0)
        world.append(w0)
        right.append(back_project(p0 + [s_, 0.0]) - w0)
        down.append(back_project(p0 + [0.0, s_]) - w0)
        lev.append(np.full(len(c), l, np.int32))
        cen.append(c.astype(np.int32))
    world, right, down = np.vstack(world), np.vstack(right), np.vstack(down)
    lev, cen = np.concatenate(lev), np.vstack(cen)
    j = n_junk
    jw = np.column_stack([rng.uniform(-6, 6, j), rng.uniform(-6, 6, j), rng.uniform(-1, 4, j)])
    jr = rng.normal(0, 1, (j, 3)) * rng.choice([1e-6, 1e-3, 1.0], j)[:, None]
    jd = rng.normal(0, 1, (j, 3)) * rng.choice([1e-6, 1e-3, 1.0], j)[:, None]
    world, right, down = np.vstack([world, jw]), np.vstack([right, jr]), np.vstack([down, jd])
    lev = np.concatenate([lev, rng.integers(0, 4, j).astype(np.int32)])
    cen = np.vstack([cen, np.column_stack([rng.integers(14, 60, j), rng.integers(14, 40, j)]).astype(np.int32)])
    perm = rng.permutation(len(world))
    n = len(world)
    return {"world": world[perm], "pixel_right_w": right[perm], "pixel_down_w": down[perm], "src_level": lev[perm], "center": cen[perm],
            "shuffle_levels": rng.permutation(n).astype(np.int32), "shuffle_fine": rng.permutation(n).astype(np.int32)}


def make_template_cases(size, n=600, seed=0x5EED0008):
    """Inputs of PatchFinder::MakeTemplateCoarseCont against a keyframe of level-0 size `size` (w, h): source level
    (50/25/15/10 mix), patch centre at that level (mostly interior, some hugging the border so that the walk
    leaves the image), search level, and warp-inverse matrices like CalcSearchLevelAndWarpMatrix produces them
    (rotation x anisotropic scale with determinant in the accepted band 0.25..3 after the level shift).
    Entry 0 is a skipped query (search level -1), entry 1 sits in the corner."""
    rng = np.random.Generator(np.random.PCG64(seed))
    src_level = rng.choice(4, size=n, p=LEVEL_MIX).astype(np.int32)
    search_level = rng.integers(0, 4, n).astype(np.int32)
    w = (size[0] >> src_level).astype(np.int64)
    h = (size[1] >> src_level).astype(np.int64)
    cx = rng.integers(0, w)
    cy = rng.integers(0, h)
    inner = rng.random(n) < 0.8
    cx = np.where(inner, np.clip(cx, np.minimum(14, w // 2), np.maximum(w - 15, w // 2)), cx)
    cy = np.where(inner, np.clip(cy, np.minimum(14, h // 2), np.maximum(h - 15, h // 2)), cy)
    ang = rng.uniform(-np.pi, np.pi, n)
    det_at_level = rng.uniform(0.3, 2.8, n)                # dDet after the *0.25 steps
    det = det_at_level * 4.0 ** search_level
    aniso = rng.uniform(0.7, 1.4, n)
    sx, sy = np.sqrt(det) * aniso, np.sqrt(det) / aniso
    c, s_ = np.cos(ang), np.sin(ang)
    shear = rng.normal(0, 0.1, n)
    wi = np.stack([c * sx, -s_ * sy + shear * sx, s_ * sx, c * sy], axis=1)
    search_level[0] = -1
    cx[1], cy[1], src_level[1] = 1, 1, 0
    return {"src_level": src_level, "search_level": search_level,
            "center": np.stack([cx, cy], axis=1).astype(np.int32), "warp_inverse": wi}


def make_epipolar_queries(cam, corners_a, level, one_pixel_dist, n=400, seed=0x5EED0009, shift=(3, -2)):
    """Queries for the corner scan of MapMaker::AddPointEpipolar between the synthetic frame pair: candidates are
    corners of frame A at `level`; the epipolar segment of each is a random line through (or, for a quarter of
    them, a few pixels beside) the in-plane position of its true match in frame B, with a random extent.
    `cam`: AtanCam-like with unproject(); corners_a: (m, 2) int32.  Entry 0 is a candidate in the border (bad
    template), entry 1 has an empty segment."""
    rng = np.random.Generator(np.random.PCG64(seed))
    scale = 1 << level
    idx = rng.integers(0, len(corners_a), n)
    q = np.zeros(n, dtype=[("level_x", "<i4"), ("level_y", "<i4"), ("normal", "<f8", (2,)), ("norm_dist", "<f8"),
                           ("along", "<f8", (2,)), ("min_len", "<f8"), ("max_len", "<f8"), ("max_dist_sq", "<f8")])
    c = corners_a[idx].astype(np.float64)
    lvl0 = (c + 0.5) * scale - 0.5 + np.asarray(shift, dtype=np.float64)
    v = cam.unproject(lvl0)
    ang = rng.uniform(0, np.pi, n)
    along = np.stack([np.cos(ang), np.sin(ang)], axis=1)
    normal = np.stack([along[:, 1], -along[:, 0]], axis=1)
    off = np.where(rng.random(n) < 0.25, rng.normal(0, 6.0, n), rng.normal(0, 1.0, n)) * one_pixel_dist
    a = (v * along).sum(1)
    q["level_x"], q["level_y"] = corners_a[idx][:, 0], corners_a[idx][:, 1]
    q["normal"], q["along"] = normal, along
    q["norm_dist"] = (v * normal).sum(1) + off
    q["min_len"] = a - rng.uniform(0.01, 0.3, n)
    q["max_len"] = a + rng.uniform(0.01, 0.3, n)
    q["max_dist_sq"] = (one_pixel_dist * (4.0 + scale)) ** 2
    q["level_x"][0], q["level_y"][0] = 2, 3
    q["min_len"][1], q["max_len"][1] = 5.0, 5.5
    return q


# ---------------------------------------------------------------------------------------------
# bundle problems
# ---------------------------------------------------------------------------------------------
def make_ba_problem(n_cams, n_pts, seed, window=None, camera=DEFAULT_CAMERA, size=(640, 480),
                    outlier_frac=0.02, n_fixed=1, pt_noise=0.01, pose_noise=0.005, dup=1):
    """Cameras on a 120 degree arc (radius 2 m, height 1 m) looking at the origin; points in
    [-0.5,0.5]^2 x [-0.1,0.1].  window=k limits each point to k consecutive cameras (banded
    covisibility).  Measurements are emitted in the reference's marshalling order: keyframe
    order, then point order (src/MapMaker.cc:871-882).  dup=k replicates every point (position,
    start value and measurements) k times: thousands of bit-identical errors, the degenerate input of
    the order-statistic select."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = AtanCam(camera, size)
    ang = np.linspace(-np.pi / 3, np.pi / 3, n_cams)
    poses_true = np.stack([look_at([2 * np.sin(a), -2 * np.cos(a), 1.0], [0, 0, 0]) for a in ang])
    pts_true = np.column_stack([rng.uniform(-0.5, 0.5, n_pts), rng.uniform(-0.5, 0.5, n_pts),
                                rng.uniform(-0.1, 0.1, n_pts)])
    if window is not None and window < n_cams:
        start = rng.integers(0, n_cams - window + 1, n_pts)
    cam_idx, pt_idx, found, sig = [], [], [], []
    for c in range(n_cams):
        ok, im = cam.visible(poses_true[c], pts_true)
        if window is not None and window < n_cams:
            ok &= (start <= c) & (c < start + window)
        ids = np.flatnonzero(ok)
        lv = rng.choice(4, size=len(ids), p=LEVEL_MIX)
        f = im[ids] + rng.normal(0, 1, (len(ids), 2)) * (0.5 * 2.0 ** lv)[:, None]
        out = rng.random(len(ids)) < outlier_frac
        sign = rng.choice([-1.0, 1.0], (len(ids), 2))
        f[out] += (sign * rng.uniform(10, 30, (len(ids), 2)))[out]
        cam_idx.append(np.full(len(ids), c, np.int32))
        pt_idx.append(ids.astype(np.int32))
        found.append(f)
        sig.append(4.0 ** lv)
    poses = poses_true.copy()
    for c in range(n_fixed, n_cams):
        poses[c] = se3_mul(se3_exp(rng.normal(0, pose_noise, 6)), poses_true[c])
    fixed = np.zeros(n_cams, np.uint8)
    fixed[:n_fixed] = 1
    prob = {"poses": poses, "fixed": fixed, "points": pts_true + rng.normal(0, pt_noise, (n_pts, 3)),
            "cam_idx": np.concatenate(cam_idx), "pt_idx": np.concatenate(pt_idx),
            "found": np.concatenate(found), "sigma_sq": np.concatenate(sig),
            "poses_true": poses_true, "points_true": pts_true}
    if dup > 1:
        j = np.arange(dup, dtype=np.int32)
        prob["points"] = np.repeat(prob["points"], dup, axis=0)
        prob["points_true"] = np.repeat(pts_true, dup, axis=0)
        prob["pt_idx"] = (prob["pt_idx"][:, None] * dup + j[None, :]).reshape(-1).astype(np.int32)
        for k in ("cam_idx", "found", "sigma_sq"):
            prob[k] = np.repeat(prob[k], dup, axis=0)
    return prob


def load_into(bundle, prob):
    """Marshal a problem dict into a host.Bundle exactly as MapMaker::BundleAdjust does
    (non-fixed cameras first is NOT required by Bundle; ids follow array order)."""
    bundle.add_problem(prob["poses"], prob["fixed"], prob["points"], prob["cam_idx"], prob["pt_idx"],
                       prob["found"], prob["sigma_sq"])
    return bundle
