"""The bundle's index structures, built on the device by ptam_ba_prepare (csrc/ba_prepare.inc), against an independent numpy
restatement of what Bundle::Compute's GenerateMeasLUTs / GenerateOffDiagScripts stand for here (src/Bundle.cc:558-599): the
point-major sort, the CSR, the chunk table, the Schur entry lists and — with csrc/ba_split.h compiled for the host
(tests/split_ref.cc) — the work split, segment for segment."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ptam_cg_amd import host, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
(BL_COUNTS, BL_ROWPTR, BL_M_CAM, BL_M_PT, BL_M_ORIG, BL_M_FIDX, BL_M_FOUND, BL_M_S, BL_PT_ORIG, BL_POINTS, BL_CHUNKS, BL_S_ENTRIES,
 BL_S_SEGS, BL_S_WG_SEG, BL_S_PAIR_BEGIN, BL_S_WG_HEAD, BL_CAM_PTR, BL_CAM_MEAS) = range(18)
SOLVE_NB, BA_CHUNK, TC, DET_TILE = 32, 256, 8, 1024
CFG = dict(seg_cost=4900, second_lag=7000, min_room=4900, min_seg=16, n_first=32, cost_model=36, slots=64, greedy=0)


@pytest.fixture(scope="module")
def split_ref(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("split") / "libsplit_ref.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "split_ref.cc")])
    lib = C.CDLL(so)
    lib.split_ref_list.restype = C.c_int
    lib.split_ref_entry_cost.restype = C.c_int
    lib.split_ref_full_products.restype = C.c_int
    return lib


def _ip(a):
    return a.ctypes.data_as(C.c_void_p)


def expected_lists(prob, dead, split_ref):
    """-> dict of the structures, from the problem as it was added (insertion order) and the erased measurements"""
    fixed = np.asarray(prob["fixed"], np.uint8)
    cam, pt = np.asarray(prob["cam_idx"], np.int64), np.asarray(prob["pt_idx"], np.int64)
    n_c = len(fixed)
    cam_free = np.full(n_c, -1, np.int64)
    cam_free[fixed == 0] = np.arange(int((fixed == 0).sum()))
    F = int((fixed == 0).sum())
    live = np.flatnonzero(~dead)
    f = cam_free[cam[live]]
    key = np.where(f >= 0, f, F + cam[live])
    order = live[np.lexsort((key, pt[live]))]   # point-major; inside a point the free cameras by free index, then the fixed ones
    E = {"F": F, "M": len(order), "m_orig": order.astype(np.int32), "m_cam": cam[order].astype(np.int32)}
    upt, dense = np.unique(pt[order], return_inverse=True)
    P = len(upt)
    E["P"], E["pt_orig"], E["m_pt"] = P, upt.astype(np.int32), dense.astype(np.int32)
    E["m_fidx"] = cam_free[cam[order]].astype(np.int32)
    E["m_found"] = np.asarray(prob["found"], np.float64)[order]
    E["m_s"] = np.sqrt(1.0 / np.asarray(prob["sigma_sq"], np.float64)[order])
    rowptr = np.zeros(P + 1, np.int64)
    np.add.at(rowptr, dense + 1, 1)
    rowptr = np.cumsum(rowptr)
    E["rowptr"] = rowptr.astype(np.int32)
    E["points"] = np.asarray(prob["points"], np.float64)[upt]
    # chunks: consecutive whole points, at most BA_CHUNK measurements (and points); a longer point alone
    chunks, p = [], 0
    while p < P:
        b, cnt, npts = p, 0, 0
        while p < P and cnt + (rowptr[p + 1] - rowptr[p]) <= BA_CHUNK and npts < BA_CHUNK:
            cnt += rowptr[p + 1] - rowptr[p]
            p += 1
            npts += 1
        if npts == 0:
            p += 1
        chunks.append((b, p, rowptr[b], rowptr[p]))
    E["chunks"] = np.array(chunks, np.int32).reshape(-1, 4)
    # band, tiles, entries
    n_tiles = (F + TC - 1) // TC
    n_pairs = n_tiles * (n_tiles + 1) // 2
    E["n_tiles"], E["n_pairs"] = n_tiles, n_pairs
    band = 0
    fidx = E["m_fidx"]
    per_point = []   # per point: list of (tile, first, off0, off1, code)
    for q in range(P):
        fs = fidx[rowptr[q]:rowptr[q + 1]]
        fr = fs[fs >= 0]
        if len(fr):
            band = max(band, (6 * int(fr.max()) + 5) // SOLVE_NB - (6 * int(fr.min())) // SOLVE_NB)
        recs = {}
        for i, fv in enumerate(fs):
            if fv < 0:
                continue
            t, slot = int(fv) // TC, int(fv) % TC
            if t not in recs:
                recs[t] = [rowptr[q] + i, [0xff] * 8]
            recs[t][1][slot] = rowptr[q] + i - recs[t][0]
        lst = []
        for t in sorted(recs):
            first, offs = recs[t]
            o0 = sum(int(offs[s]) << (8 * s) for s in range(4))
            o1 = sum(int(offs[4 + s]) << (8 * s) for s in range(4))
            s01 = (o0 != 0xffffffff) or ((o1 & 0xffff) != 0xffff)
            s2 = (o1 >> 8) != 0xffffff
            lst.append((t, int(first), o0, o1, int(s01) | (int(s2) << 1)))
        per_point.append(lst)
    E["band"] = band
    if F == 0 or E["M"] == 0:
        E["n_entries"] = 0
        return E
    cost_of = lambda a, b, pat: split_ref.split_ref_entry_cost(F, a, b, pat, CFG["cost_model"])
    ent = []   # (pt, pair, pattern, cost, ma, mb, offa0, offa1, offb0, offb1)
    pt_cost = np.zeros(P + 1, np.int64)
    for q, lst in enumerate(per_point):
        for ia in range(len(lst)):
            for ib in range(ia + 1):
                ta, fa, a0, a1, ca = lst[ia]
                tb, fb, b0, b1, cb = lst[ib]
                pat = ca | (cb << 2)
                cst = cost_of(ta, tb, pat)
                ent.append((q, ta * (ta + 1) // 2 + tb, pat, cst, fa, fb, a0, a1, b0, b1))
                pt_cost[q + 1] += cst
    ent = np.array(ent, np.int64).reshape(-1, 10)
    E["n_entries"] = len(ent)
    pre = np.cumsum(pt_cost)
    bound = [0] + [min(int(np.searchsorted(pre.astype(np.float64), float(pre[P]) * x / 8, side="left")), P) for x in range(1, 8)] + [P]
    E["bound"] = bound
    pa, pb = np.zeros(n_pairs, np.int64), np.zeros(n_pairs, np.int64)
    for a in range(n_tiles):
        for b in range(a + 1):
            pa[a * (a + 1) // 2 + b], pb[a * (a + 1) // 2 + b] = a, b
    by_products = (len(ent) // 8 // (4 * CFG["min_seg"])) >= CFG["slots"]
    pair_order = list(range(n_pairs))
    if by_products:
        pair_order = sorted(pair_order, key=lambda pr: split_ref.split_ref_full_products(F, int(pa[pr]), int(pb[pr])))   # (stable)
    rank_of = np.zeros(n_pairs, np.int64)
    rank_of[pair_order] = np.arange(n_pairs)
    x_of = np.searchsorted(np.array(bound[1:8]), ent[:, 0], side="right")
    # entries in (XCD range, list position of the pair, pattern, point) order
    srt = np.lexsort((ent[:, 0], ent[:, 2], rank_of[ent[:, 1]], x_of))
    ent, x_of = ent[srt], x_of[srt]
    E["s_entries"] = np.column_stack([ent[:, 0], ent[:, 4], ent[:, 5], ent[:, 3] | (ent[:, 2] << 16), ent[:, 6], ent[:, 7], ent[:, 8],
                                      ent[:, 9]]).astype(np.uint32).view(np.int32)
    # the split, per XCD list
    cfg8 = np.array([CFG[k] for k in ("seg_cost", "second_lag", "min_room", "min_seg", "n_first", "cost_model", "slots", "greedy")], np.int32)
    cuts_x, n_wgs_x = [], []
    for x in range(8):
        sel = np.flatnonzero(x_of == x)
        if len(sel) == 0:
            cuts_x.append([])
            n_wgs_x.append(0)
            continue
        e0 = int(sel[0])
        ex = ent[sel]
        run_cnt, run_cost, pl_run0, pl_n, pl_pair, pl_e0 = [], [], [], [], [], []
        i = 0
        while i < len(ex):
            pr = ex[i, 1]
            j = i
            pl_run0.append(len(run_cnt))
            while j < len(ex) and ex[j, 1] == pr:
                k = j
                while k < len(ex) and ex[k, 1] == pr and ex[k, 2] == ex[j, 2]:
                    k += 1
                run_cnt.append(k - j)
                run_cost.append(int(ex[j, 3]))
                j = k
            pl_pair.append(int(pr))
            pl_n.append(j - i)
            pl_e0.append(e0 + i)
            i = j
        pl_run0.append(len(run_cnt))
        run_cnt, run_cost, pl_run0, pl_n = (np.array(v, np.int32) for v in (run_cnt, run_cost, pl_run0, pl_n))
        pcost = np.array([int((run_cnt[pl_run0[k]:pl_run0[k + 1]].astype(np.int64) * run_cost[pl_run0[k]:pl_run0[k + 1]]).sum()) for k in range(len(pl_n))], np.int64)
        pl_e = np.concatenate([[0], np.cumsum(pl_n)]).astype(np.int32)
        pl_h = np.concatenate([[0], np.cumsum(pcost + CFG["seg_cost"])]).astype(np.int64)
        cap = 4 * (CFG["slots"] + len(pl_n) + 8)
        cuts = np.zeros(cap * 4, np.int32)
        n_wgs, t_cut = C.c_int(), C.c_longlong()
        n = split_ref.split_ref_list(_ip(run_cnt), _ip(run_cost), _ip(pl_run0), _ip(pl_n), len(pl_n), C.c_longlong(int(ex[:, 3].sum())),
                                     C.c_longlong(len(ex)), _ip(cfg8), _ip(cuts), cap, C.byref(n_wgs), C.byref(t_cut), _ip(pl_e), _ip(pl_h))
        assert n >= 0
        cuts_x.append([(pl_pair[k], pl_e0[k] + b, pl_e0[k] + e, wg) for k, b, e, wg in cuts[:4 * n].reshape(-1, 4)])
        n_wgs_x.append(n_wgs.value)
    # slots: contiguous per pair, in creation order (XCD range, then position)
    per_pair = [[] for _ in range(n_pairs)]
    for x in range(8):
        for ci, (pr, _, _, _) in enumerate(cuts_x[x]):
            per_pair[pr].append((x, ci))
    slot_of, pwb, slot = {}, np.zeros(n_pairs + 1, np.int32), 0
    for pr in range(n_pairs):
        pwb[pr] = slot
        for key_ in per_pair[pr]:
            slot_of[key_] = slot
            slot += 1
    pwb[n_pairs] = slot
    E["s_pair_begin"] = pwb
    longest = max(n_wgs_x)
    segs, wseg = [], []
    for i in range(longest):
        for x in range(8):
            wseg.append(len(segs))
            for ci, (pr, b, e, wg) in enumerate(cuts_x[x]):
                if wg == i:
                    segs.append((pr, b, e, slot_of[(x, ci)]))
    wseg.append(len(segs))
    E["s_segs"] = np.array(segs, np.int32).reshape(-1, 4)
    E["s_wg_seg"] = np.array(wseg, np.int32)
    head = np.zeros((8 * longest, 8), np.int32)
    for bq in range(8 * longest):
        head[bq, 0], head[bq, 1] = wseg[bq], wseg[bq + 1]
        if wseg[bq + 1] > wseg[bq]:
            head[bq, 2:6] = segs[wseg[bq]]
    E["s_wg_head"] = head
    return E


def check_lists(ba, prob, dead, split_ref, det=False):
    E = expected_lists(prob, dead, split_ref)
    g = lambda which, dt: ba.debug_lists(which, dt)
    cn = g(BL_COUNTS, np.int32)
    assert (cn[1], cn[2], cn[3]) == (E["F"], E["P"], E["M"])
    assert cn[4] == E["band"]
    for which, name, dt in ((BL_ROWPTR, "rowptr", np.int32), (BL_M_CAM, "m_cam", np.int32), (BL_M_PT, "m_pt", np.int32),
                            (BL_M_ORIG, "m_orig", np.int32), (BL_M_FIDX, "m_fidx", np.int32), (BL_PT_ORIG, "pt_orig", np.int32)):
        assert np.array_equal(g(which, dt), E[name]), name
    assert np.array_equal(g(BL_M_FOUND, np.float64).reshape(-1, 2), E["m_found"])
    assert np.array_equal(g(BL_M_S, np.float64), E["m_s"])            # sqrt(1 / sigma^2) on the device: the same bits
    assert np.array_equal(g(BL_POINTS, np.float64).reshape(-1, 3), E["points"])
    assert np.array_equal(g(BL_CHUNKS, np.int32).reshape(-1, 4), E["chunks"])
    assert cn[9] == E["n_entries"]
    if E["n_entries"]:
        assert np.array_equal(g(BL_S_ENTRIES, np.int32).reshape(-1, 8), E["s_entries"])
        assert np.array_equal(g(BL_S_PAIR_BEGIN, np.int32), E["s_pair_begin"])
        assert np.array_equal(g(BL_S_WG_SEG, np.int32), E["s_wg_seg"])
        assert np.array_equal(g(BL_S_SEGS, np.int32).reshape(-1, 4), E["s_segs"])
        assert np.array_equal(g(BL_S_WG_HEAD, np.int32).reshape(-1, 8), E["s_wg_head"])
        # what the kernels rely on, whatever the split decided: every pair's entries covered exactly once, in whole segments
        segs = E["s_segs"]
        cover = np.zeros(E["n_entries"], np.int32)
        for pr, b, e, _ in segs:
            cover[b:e] += 1
        assert (cover == 1).all()
        assert len(set(segs[:, 3])) == len(segs) and cn[8] <= 512
    if det:
        fidx, M, F = E["m_fidx"], E["M"], E["F"]
        ptr, meas = [], []
        for t in range((M + DET_TILE - 1) // DET_TILE):
            m0, m1 = t * DET_TILE, min(M, (t + 1) * DET_TILE)
            row = [len(meas)]
            for f in range(F):
                meas.extend((m0 + np.flatnonzero(fidx[m0:m1] == f)).tolist())
                row.append(len(meas))
            ptr.append(row)
        assert np.array_equal(g(BL_CAM_PTR, np.int32).reshape(-1, F + 1), np.array(ptr, np.int32).reshape(-1, F + 1))
        assert np.array_equal(g(BL_CAM_MEAS, np.int32), np.array(meas, np.int32))
    return E


def _shuffled(prob, seed):
    """the same problem with its measurements added in a random order (the reference's marshalling order is not a precondition)"""
    perm = np.random.default_rng(seed).permutation(len(prob["cam_idx"]))
    out = dict(prob)
    for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
        out[k] = prob[k][perm]
    return out


CASES = {
    "dense_12x700": (dict(n_cams=12, n_pts=700, seed=2), False),
    "local_20x3000": (dict(n_cams=20, n_pts=3000, seed=3), False),
    "headline_50x5000": (dict(n_cams=50, n_pts=5000, seed=11), False),
    "banded_80x6000_w10_shuffled": (dict(n_cams=80, n_pts=6000, seed=5, window=10), True),
    "three_fixed_33x900_shuffled": (dict(n_cams=33, n_pts=900, seed=6, n_fixed=3), True),
    "one_free_camera": (dict(n_cams=4, n_pts=200, seed=7, n_fixed=3), False),
    "long_points_300x40": (dict(n_cams=300, n_pts=40, seed=31), False),
    "many_fixed_309x60": (dict(n_cams=309, n_pts=60, seed=32, n_fixed=300), True),
}


@pytest.mark.parametrize("case", list(CASES))
def test_device_built_lists_equal_the_numpy_restatement(hip, split_ref, case):
    kw, shuffle = CASES[case]
    prob = synth.make_ba_problem(**kw)
    if shuffle:
        prob = _shuffled(prob, 99)
    ctx = host.Context(lib=hip)
    ba = synth.load_into(host.Bundle(ctx), prob)
    ba.prepare()
    check_lists(ba, prob, np.zeros(len(prob["cam_idx"]), bool), split_ref)
    assert ba.duplicates_refused() == 0
    ba.close()
    ctx.close()


def test_lists_with_unobserved_points_fixed_cameras_in_between_and_deterministic_tiles(hip, split_ref):
    prob = synth.make_ba_problem(n_cams=24, n_pts=1500, seed=8, window=7)
    # cameras 3, 10, 11 fixed (fixed cameras anywhere: their measurements go behind the free ones of a point); points 100..399 unobserved
    prob["fixed"] = np.zeros(24, np.uint8)
    prob["fixed"][[3, 10, 11]] = 1
    keep = ~((prob["pt_idx"] >= 100) & (prob["pt_idx"] < 400))
    for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
        prob[k] = prob[k][keep]
    prob = _shuffled(prob, 5)
    ctx = host.Context(lib=hip)
    ba = synth.load_into(host.Bundle(ctx, deterministic=1), prob)
    ba.prepare()
    E = check_lists(ba, prob, np.zeros(len(prob["cam_idx"]), bool), split_ref, det=True)
    assert E["P"] == 1200
    ba.close()
    ctx.close()


def test_lists_after_outliers_were_erased(hip, split_ref):
    """the second Compute() of a bundle rebuilds its lists without the measurements the first one purged (src/Bundle.cc:536-547)"""
    prob = synth.make_ba_problem(n_cams=14, n_pts=900, seed=9, outlier_frac=0.05)
    ctx = host.Context(lib=hip)
    ba = synth.load_into(host.Bundle(ctx, max_iterations=6), prob)
    ba.Compute()
    out = ba.GetOutlierMeasurements()
    assert len(out) > 10
    dead = np.zeros(len(prob["cam_idx"]), bool)
    idx = {(int(p), int(c)): i for i, (p, c) in enumerate(zip(prob["pt_idx"], prob["cam_idx"]))}
    for p, c in out:
        dead[idx[(int(p), int(c))]] = True
    ba.prepare()
    prob2 = dict(prob)
    poses, pts = ba.get_all()
    prob2["points"] = pts   # (the adjusted positions are what the next prepare uploads)
    E = check_lists(ba, prob2, dead, split_ref)
    assert E["M"] == len(dead) - int(dead.sum())
    ba.close()
    ctx.close()


def test_duplicate_measurement_is_refused_with_its_point_and_camera(hip):
    prob = synth.make_ba_problem(n_cams=9, n_pts=300, seed=10)
    for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
        prob[k] = np.concatenate([prob[k], prob[k][1234:1235]])
    ctx = host.Context(lib=hip)
    ba = synth.load_into(host.Bundle(ctx), prob)
    with pytest.raises(host.PtamError) as ei:
        ba.prepare()
    assert f"duplicate measurement of point {prob['pt_idx'][1234]} by camera {prob['cam_idx'][1234]}" in str(ei.value)
    assert ba.duplicates_refused() == 2   # (both twins count; 0 for an accepted bundle)
    ba.close()
    ctx.close()


def test_measurements_added_between_two_computes_and_exact_chunk_sizes(hip, oracle, split_ref):
    """The measurements go to the device in chunks of 32 768 while they are added (MeasStore): a bundle of exactly one chunk, one
    of a chunk + 1, and a bundle that is adjusted, given MORE measurements (the partial chunk goes up again) and adjusted again —
    lists against the numpy restatement, results against the checker driven the same way."""
    from tests import util
    for n_meas in (32768, 32769):
        prob = synth.make_ba_problem(n_cams=30, n_pts=1400, seed=21)
        assert len(prob["cam_idx"]) > n_meas
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = prob[k][:n_meas]
        ctx = host.Context(lib=hip)
        ba = synth.load_into(host.Bundle(ctx), prob)
        ba.prepare()
        check_lists(ba, prob, np.zeros(n_meas, bool), split_ref)
        ba.close()
        ctx.close()
    prob = synth.make_ba_problem(n_cams=16, n_pts=900, seed=22)
    half = len(prob["cam_idx"]) // 2
    res = []
    for lib in (hip, oracle):
        ctx = host.Context(lib=lib)
        ba = host.Bundle(ctx, max_iterations=4)
        ba.add_problem(prob["poses"], prob["fixed"], prob["points"], prob["cam_idx"][:half], prob["pt_idx"][:half], prob["found"][:half],
                       prob["sigma_sq"][:half])
        ba.Compute()
        first = ba.trials().copy()
        c = ctx._check
        n2 = len(prob["cam_idx"]) - half
        c(lib.ba_add_measurements(ba.h, n2, host._ptr(np.ascontiguousarray(prob["cam_idx"][half:])), host._ptr(np.ascontiguousarray(prob["pt_idx"][half:])),
                                  host._ptr(np.ascontiguousarray(prob["found"][half:])), host._ptr(np.ascontiguousarray(prob["sigma_sq"][half:]))), "add")
        ba.Compute()
        poses, pts = ba.get_all()
        res.append({"accepted": 0, "converged": ba.Converged(), "trials": np.concatenate([first, ba.trials()]), "poses": poses, "points": pts,
                    "outliers": ba.GetOutlierMeasurements()})
        ba.close()
        ctx.close()
    util.assert_ba_equal(res[0], res[1], rel=1e-6)


@pytest.mark.parametrize("what", ["no_measurements", "all_cameras_fixed", "one_measurement", "every_measurement_erased_point"])
def test_degenerate_bundles_prepare_and_compute(hip, oracle, what):
    from tests import util
    prob = synth.make_ba_problem(n_cams=5, n_pts=40, seed=23)
    if what == "no_measurements":
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = prob[k][:0]
    elif what == "all_cameras_fixed":
        prob["fixed"] = np.ones(5, np.uint8)
    elif what == "one_measurement":
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = prob[k][7:8]
    else:   # most points unobserved, the observed ones in the middle of the id range
        keep = (prob["pt_idx"] >= 17) & (prob["pt_idx"] < 21)
        for k in ("cam_idx", "pt_idx", "found", "sigma_sq"):
            prob[k] = prob[k][keep]
    rh = util.run_ba(hip, prob, max_iterations=5)
    if what == "no_measurements":
        # (the reference asserts in FindSigmaSquared on an empty error vector, include/MEstimator.h, and so does the checker: here the
        #  call returns at once with nothing adjusted)
        assert len(rh["trials"]) == 0 and rh["accepted"] == 0 and len(rh["outliers"]) == 0
        assert np.array_equal(rh["poses"], np.asarray(prob["poses"]).reshape(-1, 12)) and np.array_equal(rh["points"], prob["points"])
        return
    ro = util.run_ba(oracle, prob, max_iterations=5)
    util.assert_ba_equal(rh, ro, rel=1e-6)
