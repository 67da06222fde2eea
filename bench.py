#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json on MI355X.

  metric  : bundle-adjustment LM iterations/s (one iteration = one lambda trial, the unit mnCounter
            counts, src/Bundle.cc:518) on the synthetic 50-keyframe x 5000-point problem
            (seed 0x5EED0005, M = 250 000 measurements), plus — as extra keys of the
            same JSON line — tracked frames/s at 640x480 (pyramid + FAST-10 + 1000-patch ZMSSD search
            + 10-iteration pose Gauss-Newton per frame).
  step    : one lambda trial.  W warm-up trials run in a separate Compute(); the timed region is ONE
            Compute() with max_iterations = K and the convergence limit disabled, so exactly K trials
            execute.  Inputs are resident in HBM (ptam_ba_prepare) before the clock starts.
  N > 1   : STRONG scaling of the global adjustment north_star shards (BASELINE.json configs[4]): ONE
            problem of 200 keyframes x 50 000 points, 16-camera covisibility window (M ~ 0.8 M), its
            points (and all their measurements) dealt out by point id modulo N; per trial one RCCL
            all-reduce of the in-band lower-triangle blocks of S + E (+ three small exchanges for the
            exact global median, + two scalar pairs).  value = trials of that ONE problem per second
            (never multiplied by N).  The same line carries `single_gpu_same_workload` (rank 0 alone,
            same run) so that the speed-up can be read off one record; the N = 1 default line carries
            the same single-GPU measurement as `global_ba_single_gpu`.
  roofline: K7 (fused Jacobian + normal-equation kernel), algorithmic bytes / HIP-event time.
  cpu_baseline: the CPU oracle (oracle/ptam_oracle.cc, a single-thread restatement of the
            reference's loops — kind "port") on the same problem, rank 0, N = 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help="lambda trials in the timed Compute() (default 20 = Bundle.MaxIterations, src/Bundle.cc:40)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cams", type=int, default=None, help="keyframes (default: 50 at N = 1, 200 at N > 1)")
    ap.add_argument("--points", type=int, default=None, help="map points (default: 5000 at N = 1, 50000 at N > 1)")
    ap.add_argument("--window", type=int, default=None, help="covisibility window, 0 = dense (default: 0 at N = 1, 16 at N > 1)")
    ap.add_argument("--no-global", action="store_true", help="skip the single-GPU run of the 200 x 50000 global adjustment")
    ap.add_argument("--no-local", action="store_true", help="skip the local bundle adjustment leg (BASELINE.json configs[3]: 20 keyframes x 3000 points)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tracking", action="store_true")
    ap.add_argument("--no-replicas", action="store_true", help="tracking: one context only (no k-thread replica runs; used under rocprofv3)")
    ap.add_argument("--deterministic", action="store_true", help="ptam_ba_opts.deterministic = 1 (camera sums in a fixed order): what the mode costs")
    ap.add_argument("--jac-reps", type=int, default=200)
    return ap.parse_args()


def tracking_bench(hip, host, synth, seq, frames=250, replicas=True):
    """tracked frames/s: K1+K2 (keyframe) + K3 (1000 patches) + K4 (pose GN, 10 iterations)."""
    C = ctypes
    ctx = host.Context(lib=hip)
    a, b = synth.make_frame_pair()
    kfa = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    q, t = synth.make_patch_queries([kfa.level(l) for l in range(4)], n=1000)
    kfb = host.KeyFrame(ctx)
    pc = synth.make_pose_case()
    # resident inputs
    d_im, d_q, d_t, d_r = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    chk = ctx._check
    chk(hip.dev_alloc(ctx.h, b.size, C.byref(d_im)), "alloc")
    chk(hip.dev_alloc(ctx.h, q.nbytes, C.byref(d_q)), "alloc")
    chk(hip.dev_alloc(ctx.h, t.nbytes, C.byref(d_t)), "alloc")
    chk(hip.dev_alloc(ctx.h, len(q) * 32, C.byref(d_r)), "alloc")
    chk(hip.dev_upload(ctx.h, d_im, b.ctypes.data, b.size), "up")
    chk(hip.dev_upload(ctx.h, d_q, q.ctypes.data, q.nbytes), "up")
    chk(hip.dev_upload(ctx.h, d_t, t.ctypes.data, t.nbytes), "up")
    n = len(pc["world"])
    meas = np.zeros(n, dtype=host.POSE_MEAS_DT)
    meas["world"], meas["found"], meas["sqrt_inv_noise"] = pc["world"], pc["found"], pc["sqrt_inv_noise"]
    opts = ctx.gn_opts()
    pose = pc["init_pose"].copy()
    stage, spread = {}, {}
    # device-resident frame: SearchForPoints' bookkeeping (gather) and the pose solve consume device data only; the
    # pose prediction travels as a kernel argument and the refined pose comes back through host-mapped memory.  (As in
    # the staged variant the synthetic patch queries and the pose case are separate workloads: the gather runs on the
    # patch results into scratch, the solve on the resident 1000-measurement pose case.)
    pts = np.zeros(len(q), dtype=[("world", "<f8", (3,))])
    pts["world"] = pc["world"][np.arange(len(q)) % n]
    d_w, d_gm, d_gi = host.DevBuf(ctx, pts), host.DevBuf(ctx, len(q) * 48), host.DevBuf(ctx, len(q) * 4)
    d_gc, d_pm, d_pn, d_pp = host.DevBuf(ctx, 32), host.DevBuf(ctx, meas), host.DevBuf(ctx, np.array([n], dtype=np.int32)), host.DevBuf(ctx, 96)
    pose_out = np.zeros(12)

    def one(parts):
        if "kf" in parts:
            chk(hip.make_keyframe_lite_dev(ctx.h, kfb.h, d_im), "kf")
        if "patch" in parts:
            chk(hip.find_patch_coarse_batch_dev(ctx.h, kfb.h, len(q), d_q, d_t, d_r), "patch")
        if "pose" in parts:
            p = pose.copy()
            chk(hip.pose_gn(ctx.h, n, meas.ctypes.data, None, p.ctypes.data_as(C.POINTER(C.c_double)),
                            C.byref(opts), None, None), "pose")
        if "gather" in parts:
            chk(hip.gather_pose_meas_dev(ctx.h, len(q), d_q, d_r, None, d_w.p, 24, d_gm.p, d_gi.p, d_gc.p, None), "gather")
        if "pose_dev" in parts:
            chk(hip.pose_gn_dev_counted(ctx.h, n, d_pn.p, d_pm.p, None, d_pp.p, C.byref(opts), None, None,
                                        pose.ctypes.data_as(C.POINTER(C.c_double)),
                                        pose_out.ctypes.data_as(C.POINTER(C.c_double))), "pose_dev")

    # (the first stage measured used to catch the chip waking up: one block in five ten times slower)
    for _ in range(400):
        one(("kf", "patch", "gather", "pose_dev"))
    ctx.sync()
    for name, parts in (("frame", ("kf", "patch", "gather", "pose_dev")), ("frame_staged", ("kf", "patch", "pose")),
                        ("keyframe", ("kf",)), ("patch", ("kf", "patch")), ("gather", ("gather",)), ("pose_dev", ("pose_dev",)),
                        ("pose", ("pose",))):
        for _ in range(50):
            one(parts)
        ctx.sync()
        # five blocks, the median block counts: these are tiny launches on an otherwise idle chip, and every few runs one
        # block lands in a low power state (the single-workgroup pose kernel then takes 2.7x as long; min / max are reported)
        blocks = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(frames // 5):
                one(parts)
            ctx.sync()
            blocks.append((time.perf_counter() - t0) / (frames // 5))
        blocks.sort()
        stage[name] = blocks[2]
        spread[name] = (blocks[0], blocks[-1])
    # the resident solve is the host-entry solve: same kernel, same list
    ref = pose.copy()
    chk(hip.pose_gn(ctx.h, n, meas.ctypes.data, None, ref.ctypes.data_as(C.POINTER(C.c_double)), C.byref(opts), None, None), "pose")
    assert np.array_equal(ref, pose_out), "device-resident pose solve differs from the host entry"
    # ---- the resident TrackMap chain (ptam_track_map, src/Tracker.cc:442-696): one dependent pipeline per frame —
    # pyramid + FAST of the new frame, PVS over the map, set choice, warped templates, coarse search (range 30) + sub-pixel,
    # ten coarse pose iterations, re-projection, fine search (+ sub-pixel on the top level), gather, ten fine pose iterations.
    # The pose solves consume what the searches of the same frame found.
    chain = trackmap_bench(hip, host, synth, ctx, kfa, b, d_im, frames, replicas)
    for p in (d_im, d_q, d_t, d_r):
        hip.dev_free(ctx.h, p)
    for bfr in (d_w, d_gm, d_gi, d_gc, d_pm, d_pn, d_pp):
        bfr.free()
    # the headline tracking figure: one dependent chain per frame through the C ABI, driven by a native host thread
    # (ptam_bench_track_frames, one context); frame_chain["fps"] is the same loop driven from Python (ctypes + interpreter per call)
    native1 = chain["aggregate_fps_by_concurrent_contexts"]["1"]
    moving = tracking_moving_bench(hip, host, synth, ctx, seq)
    return {"tracked_fps": moving["fps"], "frame_us": moving["frame_us"], "moving_camera": moving,
            "tracked_fps_moving": moving["fps"],   # (the same figure under a key that says what it is: rounds 1-3 reported the stationary frame as tracked_fps)
            "tracked_fps_stationary": native1, "frame_us_stationary": 1e6 / native1,
            "frame_chain": chain, "kernels": tracking_kernel_record(),
            "fine_stage_only_fps": 1.0 / stage["frame"], "fine_stage_only_frame_us": stage["frame"] * 1e6,
            "keyframe_us": stage["keyframe"] * 1e6, "keyframe_plus_patch_us": stage["patch"] * 1e6,
            "pose_gn_us": stage["pose_dev"] * 1e6, "pose_gn_host_buffers_us": stage["pose"] * 1e6,
            "frame_host_staged_pose_us": stage["frame_staged"] * 1e6,
            "gather_us": stage["gather"] * 1e6,
            "frame_us_min_max_of_5_blocks": [spread["frame"][0] * 1e6, spread["frame"][1] * 1e6], "patches_per_frame": int(len(q)), "pose_meas": int(n),
            "note": "tracked_fps / frame_us = the moving-camera sequence (moving_camera: closed loop with the motion model, templates re-warped as the warps move); "
                    "*_stationary = one image tracked from one prediction over and over, every template kept (the figure of rounds 2-3); frame_chain: "
                    "the stationary frame driven from Python, and k contexts; fine_stage_only_* = round 1's frame: pyramid + FAST + "
                    "1000-patch search + gather + one 10-iteration pose solve fed from a separate pose case"}


SEQ_FRAMES = 64   # frames of the closed trajectory (synth.sequence_pose), visited round and round


def moving_sequence(synth):
    """the moving-camera workload of the tracked-frames half of the metric: 64 rendered 640x480 frames of a camera that
    translates, rises, rolls and tilts over a textured plane, and the view the map is made of"""
    frames, poses, kim, kpose = synth.make_tracking_frames(SEQ_FRAMES)
    return {"frames": frames, "poses": poses, "kf_image": kim, "kf_pose": kpose}


def tracking_moving_bench(hip, host, synth, ctx, seq, passes=16):
    """Tracker::TrackFrame's tracking branch (src/Tracker.cc:94,134-137) frame after frame on the moving sequence, closed loop
    from one native host thread (ptam_bench_track_sequence): keyframe of the new image, motion-model prediction, bTryCoarse from
    the velocity, TrackMap, motion-model update.  The per-point PatchFinders re-warp their templates whenever the warp has moved
    by more than 0.07 (src/PatchFinder.cc:103-111)."""
    kf0 = host.KeyFrame(ctx).MakeKeyFrame_Lite(seq["kf_image"])
    m = synth.make_sequence_map([kf0.level(l) for l in range(4)], seq["kf_pose"])
    tr = host.Tracker(ctx, len(m["world"]))
    tr.set_map(m["world"], m["pixel_right_w"], m["pixel_down_w"], kf0, m["src_level"], m["center"])
    kf = host.KeyFrame(ctx)
    d_frames = [host.DevBuf(ctx, f) for f in seq["frames"]]
    opts = tr.opts()
    mm = tr.motion_model(seq["poses"][0])
    tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=2)             # warm-up: two rounds
    blocks, stats = [], None
    for _ in range(5):
        secs, st = tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=max(1, passes // 5),
                                            poses_true=seq["poses"])
        blocks.append(secs / st["frames"])
        stats = st if stats is None else {k: (max(stats[k], v) if k == "max_position_error_m" else stats[k] + v) for k, v in st.items()}
    blocks.sort()
    # per-stage breakdown measured in this run (HIP events after every launch of the frame; profiled frames are not the timed ones)
    tr.set_profiling(True)
    tr.track_sequence_native(kf, d_frames, mm, opts, m["shuffle_levels"], m["shuffle_fine"], passes=2)
    stages = tr.stage_times()
    tr.set_profiling(False)
    out = {"fps": 1.0 / blocks[2], "frame_us": blocks[2] * 1e6, "frame_us_min_max_of_5_blocks": [blocks[0] * 1e6, blocks[-1] * 1e6],
           "frames_timed": int(stats["frames"]), "sequence_frames": len(d_frames), "map_points": int(len(m["world"])),
           "patches_searched_per_frame": stats["searched"] / stats["frames"], "patches_found_per_frame": stats["measurements"] / stats["frames"],
           "templates_reused_frac": stats["templates_reused"] / max(1.0, stats["searched"]),
           "templates_rewarped_per_frame": (stats["searched"] - stats["templates_reused"]) / stats["frames"],
           "frames_with_coarse_stage_frac": stats["frames_did_coarse"] / stats["frames"],
           "max_position_error_m": stats["max_position_error_m"], "frames_below_50_measurements": int(stats["frames_below_50_measurements"]),
           "stage_us_profiled": stages, "stage_us_profiled_sum": sum(stages.values()),
           "workload": "64 rendered frames of a camera translating (up to 5 px / frame), rising by 8 %, rolling +-0.25 rad and tilting over a "
                       "textured plane; closed trajectory visited round and round; constant-velocity motion model; one context, native host thread"}
    for d in d_frames:
        d.free()
    tr.close()
    return out


def tracking_kernel_record():
    """the tracked frame kernel by kernel: the committed rocprofv3 summary of the same chain (profiles/tracking_kernels.json,
    written by tools/collect_profiles.sh), with the algorithmic bytes of the two image kernels.  Every kernel of the frame is a
    few microseconds of a dependent chain on an otherwise idle chip: they are latency-bound, none is near a roofline, and the
    frame is their sum plus one host round trip."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "tracking_kernels.json")
    try:
        from ptam_cg_amd._srchash import source_sha16
        with open(path) as f:
            rec = json.load(f)
        if rec.get("source_sha16") != source_sha16():
            return {"file": "profiles/tracking_kernels.json", "stale": True,
                    "note": "taken from another build of the library (source_sha16 differs): not quoted; moving_camera.stage_us_profiled is measured in this run"}
    except (OSError, ValueError):
        return None
    px = 640 * 480
    alg = {"tm_pyr_pvs_kernel": px + px // 4 + px // 16 + px // 64,         # level 0 read; levels 1-3 written
           "fast_detect_kernel": px + px // 4 + px // 16 + px // 64}        # the four levels read once
    for name, k in rec.get("kernels", {}).items():
        for key, nbytes in alg.items():
            if key in name and k.get("avg_us"):
                k["algorithmic_bytes"] = nbytes
                k["achieved_GBps"] = round(nbytes / k["avg_us"] / 1e3, 1)
    rec["file"] = "profiles/tracking_kernels.json"
    return rec


def trackmap_bench(hip, host, synth, ctx, kfa, frame_b, d_im, frames, replicas=True):
    import threading
    C = ctypes
    case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])

    def make(ctx_, kfa_):
        tr = host.Tracker(ctx_, len(case["world"]))
        tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa_, case["src_level"], case["center"])
        return tr

    tr = make(ctx, kfa)
    kfb = host.KeyFrame(ctx)
    opts = tr.opts()
    pose = np.ascontiguousarray(case["pose_in"])
    res = None

    def frame(ctx_, kf_, tr_, d_im_):
        tr_.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])     # the frame's random orders travel with it
        return tr_.TrackFrame(kf_, d_im_, pose, opts)                      # ptam_track_map_frame: keyframe of the new image + TrackMap

    for _ in range(30):
        res = frame(ctx, kfb, tr, d_im)
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(frames // 5):
            res = frame(ctx, kfb, tr, d_im)
        blocks.append((time.perf_counter() - t0) / (frames // 5))
    blocks.sort()
    out = {"fps": 1.0 / blocks[2], "frame_us": blocks[2] * 1e6, "frame_us_min_max_of_5_blocks": [blocks[0] * 1e6, blocks[-1] * 1e6],
           "map_points": int(len(case["world"])), "did_coarse": int(res["did_coarse"]), "n_coarse": int(res["n_coarse"]),
           "n_top": int(res["n_top"]), "n_fine": int(res["n_fine"]), "patches_searched": int(sum(res["attempted"])),
           "patches_found": int(res["n_meas"]),
           "pose_err_in_out": [float(np.abs(case["pose_in"] - case["cur_pose"]).max()), float(np.abs(res["pose"] - case["cur_pose"]).max())]}
    tr.close()
    # replicas (frames scale as independent units): k contexts, each with its own stream, map and keyframes,
    # driven by k host threads INSIDE the library (ptam_bench_track_frames: per frame set_shuffle + ptam_track_map_frame, the
    # calls the reference's tracker thread would make); aggregate frames/s.  `python_threads` keeps the earlier figure whose
    # host side was k Python threads (interpreter-bound).
    conc, conc_py, batched = {}, {}, {}
    sl = np.ascontiguousarray(case["shuffle_levels"], dtype=np.int32)
    sf = np.ascontiguousarray(case["shuffle_fine"], dtype=np.int32)
    for k in ((1, 4, 8, 16, 32, 64) if replicas else (1,)):
        workers = []
        for _ in range(k):
            cx = host.Context(lib=hip)
            ka = host.KeyFrame(cx).MakeKeyFrame_Lite(synth.make_frame_pair()[0])
            di = host.DevBuf(cx, frame_b)
            workers.append((cx, ka, host.KeyFrame(cx), make(cx, ka), di))
        for w in workers:
            frame(w[0], w[2], w[3], w[4].p)
        nf = max(40, 1600 // k)
        raw = lambda h: h.value if hasattr(h, "value") else int(h)
        trs = (C.c_void_p * k)(*[raw(w[3].h) for w in workers])
        kfs = (C.c_void_p * k)(*[raw(w[2].h) for w in workers])
        dis = (C.c_void_p * k)(*[raw(w[4].p) for w in workers])
        secs = C.c_double()
        for rep in range(2):   # (first pass: warm-up)
            ctx._check(hip.bench_track_frames(k, trs, kfs, dis, pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p),
                                              sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p), nf, C.byref(secs)), "bench_track_frames")
        conc[str(k)] = k * nf / secs.value
        if k in (4, 16, 64):
            # the same k trackers as ONE chain of launches per round of frames (ptam_track_map_frames_batch): not bound by the
            # process' four hardware queues
            rounds = max(20, 1600 // k)
            for rep in range(2):
                ctx._check(hip.bench_track_batch(k, trs, kfs, dis, pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p),
                                                 sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p), rounds, 1, C.byref(secs)), "bench_track_batch")
            batched[str(k)] = {"fps": k * rounds / secs.value, "batch_us": 1e6 * secs.value / rounds}
        if k in (1, 8, 32):
            nfp = max(20, 400 // k)

            def run(w):
                cx, _, kb, t_, di = w
                for _ in range(nfp):
                    frame(cx, kb, t_, di.p)

            th = [threading.Thread(target=run, args=(w,)) for w in workers]
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            conc_py[str(k)] = k * nfp / (time.perf_counter() - t0)
        for cx, ka, kb, t_, di in workers:
            t_.close()
            di.free()
            kb.close()
            ka.close()
            cx.close()
    out["aggregate_fps_python_threads"] = conc_py
    out["batched_fps_by_frames_per_batch"] = batched
    out["aggregate_fps_by_concurrent_contexts"] = conc
    return out


def tracking_replica_rank(hip, host, synth, ctx, k=64):
    """this rank's share of the N-device tracking figure: frames/s of k cameras tracked as batches, and of one camera alone"""
    C = ctypes
    a, b = synth.make_frame_pair()
    kfa0 = host.KeyFrame(ctx).MakeKeyFrame_Lite(a)
    case = synth.make_trackmap_case([kfa0.level(l) for l in range(4)])
    sl = np.ascontiguousarray(case["shuffle_levels"], dtype=np.int32)
    sf = np.ascontiguousarray(case["shuffle_fine"], dtype=np.int32)
    pose = np.ascontiguousarray(case["pose_in"], dtype=np.float64)
    raw = lambda h: h.value if hasattr(h, "value") else int(h)
    ws = []
    for _ in range(k):
        cx = host.Context(lib=hip, device=ctx.device)
        ka = host.KeyFrame(cx).MakeKeyFrame_Lite(a)
        tr = host.Tracker(cx, len(case["world"]))
        tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], ka, case["src_level"], case["center"])
        ws.append((cx, ka, host.KeyFrame(cx), tr, host.DevBuf(cx, b)))
    opts = ws[0][3].opts()
    trs = (C.c_void_p * k)(*[raw(w[3].h) for w in ws])
    kfs = (C.c_void_p * k)(*[raw(w[2].h) for w in ws])
    dis = (C.c_void_p * k)(*[raw(w[4].p) for w in ws])
    secs = C.c_double()
    args_ = (pose.ctypes.data_as(C.POINTER(C.c_double)), opts.ctypes.data_as(C.c_void_p), sl.ctypes.data_as(C.c_void_p), sf.ctypes.data_as(C.c_void_p))
    for rep in range(2):
        ctx._check(hip.bench_track_batch(k, trs, kfs, dis, *args_, 40, 1, C.byref(secs)), "bench_track_batch")
    fps_b = k * 40 / secs.value
    for rep in range(2):
        ctx._check(hip.bench_track_frames(1, trs, kfs, dis, *args_, 400, C.byref(secs)), "bench_track_frames")
    fps_1 = 400 / secs.value
    for cx, ka, kb, tr, di in ws:
        tr.close()
        di.free()
        kb.close()
        ka.close()
        cx.close()
    kfa0.close()
    return fps_b, fps_1


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def pmc_traffic(workload):
    """HBM bytes per K7 launch from the committed rocprofv3 PMC passes (profiles/k7_pmc_traffic.json, written by
    tools/profile_k7.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected from inside this process, so the
    number is only reported for the workload it was measured on; anything else gets null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "k7_pmc_traffic.json")
    try:
        from ptam_cg_amd._srchash import source_sha16
        with open(path) as f:
            rec = json.load(f)
        if rec.get("source_sha16") != source_sha16():   # (counters of another build of the library: not this run's)
            return None
        return float(rec[workload]["traffic_bytes_per_launch"]) if workload in rec else None
    except (OSError, ValueError, KeyError):
        return None


GLOBAL_BA = dict(cams=200, points=50000, window=16)   # BASELINE.json configs[4]
LOCAL_BA = dict(cams=20, points=3000)                  # BASELINE.json configs[3]


def workload_name(cams, points, window):
    return f"bundle_{cams}kf_x_{points}pts" + (f"_window{window}" if window else "_dense")


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON record: whatever libraries print there on the way (gloo's connection notes,
    # RCCL's version banner) goes to stderr — file descriptor 1 is pointed at stderr until the record is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch   # before libptam_hip.so: one shared HIP runtime (see ptam_cg_amd/_lib.py)
    import torch.distributed as dist
    # PTAM_BENCH_ONE_GPU=1 (development only): all ranks share device 0 and exchange through gloo with host staging, so that
    # the N > 1 code path of this file can be exercised on a one-GPU box; its numbers mean nothing.
    one_gpu = world > 1 and os.environ.get("PTAM_BENCH_ONE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from ptam_cg_amd import _abi, host, synth
    from ptam_cg_amd._srchash import source_sha16
    from ptam_cg_amd.sharding import shard_problem
    from ptam_cg_amd._lib import load
    hip = load()
    device = local_rank if world > 1 else 0
    ctx = host.Context(lib=hip, device=device)

    # ---- the workload -------------------------------------------------------------------------------------------
    # N = 1: the configuration BASELINE.json's metric is quoted on, 50 keyframes x 5000 points (dense).
    # N > 1: STRONG scaling of BASELINE.json configs[4], the global adjustment that north_star shards: 200 keyframes x
    #        50 000 points, 16-camera covisibility window, ONE problem whose points are dealt out over the N ranks.
    #        value = lambda trials of that one global problem per second (a trial over N shards is one trial).
    # --cams / --points / --window override either.
    dflt = GLOBAL_BA if world > 1 else dict(cams=50, points=5000, window=0)
    if args.cams is None and args.points is None and args.window is None:
        args.cams, args.points, args.window = dflt["cams"], dflt["points"], dflt["window"]
    else:   # an explicit shape: what is not given is dense / the single-GPU default
        args.cams = 50 if args.cams is None else args.cams
        args.points = 5000 if args.points is None else args.points
        args.window = 0 if args.window is None else args.window
    is_global = (args.cams, args.points, args.window) == (GLOBAL_BA["cams"], GLOBAL_BA["points"], GLOBAL_BA["window"])
    seed = synth.SEED_BA_GLOBAL if is_global else synth.SEED_BA_HEADLINE
    prob_full = synth.make_ba_problem(args.cams, args.points, seed, window=args.window if args.window > 0 else None)
    prob = shard_problem(prob_full, rank, world)

    comm = hook = None
    if one_gpu:
        from ptam_cg_amd.sharding import torch_allreduce_hook
        hook, comm = torch_allreduce_hook(ctx, device_ptr=True), ctypes.c_void_p()
    elif world > 1:
        ident = (ctypes.c_uint8 * 128)()
        if rank == 0:
            ctx._check(hip.rccl_unique_id(ident), "rccl_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        comm = ctypes.c_void_p()
        ctx._check(hip.rccl_create(ctx.h, ident, rank, world, ctypes.byref(comm)), "rccl_create")
        hook = ctypes.cast(hip.lib.ptam_rccl_allreduce_f64, _abi.ALLREDUCE_FN)

    prepare_ms = []   # host + upload time of every ptam_ba_prepare of this run (never part of `value`; docs/LOG_r01_r04.md section 5)

    def new_bundle(max_it, problem=None, sharded=True):
        ba = synth.load_into(host.Bundle(ctx, max_iterations=max_it, update_sq_conv_limit=0.0, deterministic=1 if args.deterministic else 0),
                             prob if problem is None else problem)
        if comm is not None and sharded:
            ba.set_comm(rank, world, hook, comm)
        t0 = time.perf_counter()
        ba.prepare()     # sort + upload: inputs resident in HBM before any timed region
        ctx.sync()
        prepare_ms.append(1e3 * (time.perf_counter() - t0))
        return ba

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        ctx.sync()

    def spin_up(problem):
        # blocks of 1000 back-to-back K7 launches on a LOCAL copy (no communicator: ranks may need different block
        # counts) until the launch time has settled (at least 6, at most 150 blocks) — the launch time is the clock probe
        sb = synth.load_into(host.Bundle(ctx), problem)
        best, calm, spin = None, 0, []
        for blk in range(150):
            ms1, _ = sb.bench_jacobian(1000)
            spin.append(ms1)
            calm = calm + 1 if best is not None and ms1 <= best * 1.03 else 0
            best = ms1 if best is None else min(best, ms1)
            if blk >= 5 and calm >= 3:
                break
        sb.close()
        return spin

    def timed_compute(problem, steps, warmup, sharded=True, sync_ranks=True):
        """W warm-up trials in their own Compute(), then ONE timed Compute() of exactly `steps` trials, bracketed by
        barrier + synchronise on both sides.  All bundles are built and made resident first, so that the warm-up runs
        IMMEDIATELY before the timed region (building a bundle is ~100 ms of host work with nothing queued).  DESIGN.md
        section 5 lists what once made this bench bimodal and what was changed."""
        wb = new_bundle(warmup, problem, sharded) if warmup > 0 else None
        ba = new_bundle(steps, problem, sharded)
        spin = spin_up(problem)
        if wb is not None:
            wb.Compute()
        if sync_ranks:
            barrier()
        else:
            ctx.sync()
        t0 = time.perf_counter()
        ba.Compute()
        ctx.sync()
        if sync_ranks:
            barrier()
        dt = time.perf_counter() - t0
        trials = ba.trials()
        assert len(trials) == steps, (len(trials), steps)
        counts = ba.counts()
        ba.close()
        if wb is not None:
            wb.close()
        return dt, trials, counts, spin

    def trial_mix(tr):
        acc = int(tr["accepted"].sum())
        stay = int(((tr["err_new"] == tr["err_old"]) & (tr["accepted"] == 0)).sum())
        return {"accepted": acc, "rejected": int(len(tr) - acc - stay), "stay": stay}

    def schur_roofline(problem, kms):
        """the Schur complement's useful fp64 work — per point with n free cameras n(n+1)/2 blocks of (6x3)(3x3)... = 216 flop
        each (src/Bundle.cc:374-446) — over the measured K8 time (tile kernel + reduce, HIP events), against the chip's fp64
        peak (MI355X_MICROARCH.md gives no fp64 row: 78.6 TFLOP/s vector = matrix, AMD's datasheet figure)"""
        free = problem["fixed"][problem["cam_idx"]] == 0
        _, npc = np.unique(problem["pt_idx"][free], return_counts=True)
        flops = float((npc.astype(np.float64) * (npc + 1) / 2 * 216).sum())
        t = kms.get("schur", 0.0) * 1e-3
        return {"kernel": "schur_tile_mfma_kernel + schur_reduce_kernel", "bound": "mfma (fp64)", "useful_flop_per_trial": flops,
                "us_per_trial": t * 1e6, "achieved": flops / t / 1e12 if t > 0 else None, "peak": 78.6, "unit": "TFLOP/s",
                "frac": flops / t / 78.6e12 if t > 0 else None,
                # what the matrix pipe sustains under THIS kernel's load: the shader clock reads 2.1 GHz inside the tile kernel
                # (s_memtime against the 100 MHz clock, docs/LOG_r05.md), and a v_mfma_f64_16x16x4 holds the pipe 64 cycles
                "sustained_peak": 78.6 * 2.1 / 2.4,
                "frac_of_sustained": flops / t / (78.6e12 * 2.1 / 2.4) if t > 0 else None}

    def kernel_breakdown(problem, steps, sharded=True):
        # two profiled runs, per kernel the smaller average: now and then one event bracket of a run catches a stall of
        # several milliseconds (seen in the solve bracket: 0.4 - 0.8 ms "per trial" instead of 0.1)
        out = {}
        for _ in range(2):
            kb = new_bundle(steps, problem, sharded)
            kb.set_profiling(True)
            kb.Compute()
            for k, (ms, n) in kb.kernel_times().items():
                if n > 0:
                    out[k] = min(out.get(k, float("inf")), ms / n)
            kb.close()
        return out

    dt, trials, (n_cams, n_free, n_points, n_meas), spin = timed_compute(prob, args.steps, args.warmup)
    prepare_headline_ms = min(prepare_ms) if prepare_ms else None   # (the bundles of the timed workload; later legs append more)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = None
    if rank == 0:
        value = args.steps / dt   # trials of the ONE (global) problem per second — never multiplied by the rank count
        wl = workload_name(args.cams, args.points, args.window)
        out = {
            "metric": "BA LM iterations/s" + (" (global BA, 200 KF x 50k pts, points sharded)" if is_global else
                                              " (50 KF x 5k pts)" if (args.cams, args.points) == (50, 5000) else "")
                      + " [+ tracked frames/s @640x480 in 'tracking']",
            "value": value, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl + (f"_strong_x{world}" if world > 1 else ""),
                       "keyframes": args.cams, "points_total": args.points,
                       "measurements_total": int(len(prob_full["cam_idx"])),
                       "measurements_rank0": int(len(prob["cam_idx"])), "estimator": "Tukey",
                       "parallelism": (f"points sharded x{world} (p % {world}), per trial one RCCL all-reduce of the in-band "
                                       f"lower-triangle blocks of S + E, camera solve replicated") if world > 1 else "1 GPU",
                       "halfsample": "R"},
            "trial_mix": trial_mix(trials),
            "accepted_trials": int(trials["accepted"].sum()),
            "spinup_k7_us_first_last_blocks": [spin[0] * 1e3, spin[-1] * 1e3, len(spin)],
            "err_first_last": [float(trials["err_old"][0]), float(trials["err_new"][-1])],
            "deterministic": bool(args.deterministic),
            "source_sha16": source_sha16(),   # fingerprint of the library sources this run was built from (profiles/ are keyed by it)
            "prepare_ms": prepare_headline_ms,   # upload + device-built index structures of one Bundle of this workload: outside the timed region
        }
    if world > 1:
        # per-kernel breakdown of the sharded run (HIP events, separate Compute; every rank takes part in its collectives)
        kms = kernel_breakdown(prob, args.steps)
        if rank == 0:
            out["kernel_ms_per_trial"] = kms
        # the SAME global problem on ONE device in the same run (rank 0 alone; the others wait at the barrier): the
        # denominator of the strong-scaling speed-up, so that one JSON line carries both
        barrier()
        if rank == 0:
            dt1, tr1, _, _ = timed_compute(prob_full, args.steps, args.warmup, sharded=False, sync_ranks=False)
            out["single_gpu_same_workload"] = {"value": args.steps / dt1, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt1 / args.steps,
                                               "workload": wl, "trial_mix": trial_mix(tr1),
                                               "kernel_ms_per_trial": kernel_breakdown(prob_full, args.steps, sharded=False)}
            out["speedup_vs_single_gpu"] = out["value"] / out["single_gpu_same_workload"]["value"]
        barrier()
        # tracking on N devices: replicas only — every rank tracks its own 64 cameras as batches on its own
        # device at the same time, no exchange; the total is the sum.  Kept out of the way of the BA record: a failure here
        # only drops these keys.
        if not args.no_tracking:
            try:
                fps64, fps1 = tracking_replica_rank(hip, host, synth, ctx)
                tt = torch.tensor([fps64, fps1], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                if rank == 0:
                    out["tracking_replicas"] = {"devices": world, "frames_per_batch": 64, "batched_fps_total": float(tt[0].item()),
                                                "single_camera_fps_sum": float(tt[1].item()),
                                                "note": "one process per device, each tracking 64 independent cameras per "
                                                        "ptam_track_map_frames_batch call (and one camera alone) concurrently; no data-path collective"}
            except Exception as e:   # noqa: BLE001
                if rank == 0:
                    out["tracking_replicas"] = {"error": repr(e)}
        barrier()
    if world == 1:
        # ---- steady accepted-trial rate: a Compute() of exactly the leading accepted trials of this problem ----------
        n_lead = 0
        for a in trials["accepted"]:
            if not a:
                break
            n_lead += 1
        if n_lead >= 2:
            dta, tra, _, _ = timed_compute(prob, n_lead, args.warmup)
            out["accepted_trial_us"] = 1e6 * dta / n_lead
            out["accepted_trial_note"] = (f"one Compute() of {n_lead} trials, {int(tra['accepted'].sum())} of them accepted (includes the "
                                          f"first step's full projection pass and the read-back at the end of Compute(); near the "
                                          f"noise floor the last trials of the run may come out rejected in one run and accepted in the next)")
        # ---- roofline leg: K7 alone, HIP events on the library's stream -------------------------
        pb = new_bundle(args.steps)
        pb.bench_jacobian(1000)   # (untimed: the bundle was just built, the chip idled meanwhile — see the spin-up above)
        avg_ms, alg_bytes = pb.bench_jacobian(args.jac_reps)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "jac_accum_wave_kernel", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": pmc_traffic(out["config"]["workload"]), "algorithmic_bytes_per_launch": alg_bytes,
                           "avg_launch_us": avg_ms * 1e3, "launches_timed": args.jac_reps}
        pb.close()
        # the same bracket with the working set COLD in the 256 MB Infinity Cache: copies of the problem launched
        # round-robin, (n - 1) x bytes-per-launch >= 768 MB between two launches on the same copy
        n_rot = max(4, int(768e6 // alg_bytes) + 2)
        rot = [new_bundle(args.steps) for _ in range(n_rot)]
        host.Bundle.bench_jacobian_rotating(rot, 20)
        cold_ms = host.Bundle.bench_jacobian_rotating(rot, max(3, args.jac_reps // n_rot))
        for b_ in rot:
            b_.close()
        out["roofline"].update({"achieved_cold": alg_bytes / (cold_ms * 1e-3) / 1e9,
                                "frac_cold": alg_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "avg_launch_us_cold": cold_ms * 1e3, "cold_copies": n_rot,
                                "cold_note": f"{n_rot} copies of the problem launched round-robin in one event bracket: "
                                             f"{(n_rot - 1) * alg_bytes / 1e6:.0f} MB of other working sets pass between two launches "
                                             f"on the same copy (Infinity Cache 256 MB + L2 32 MB)"})
        out["kernel_ms_per_trial"] = kernel_breakdown(prob, args.steps)
        out["schur_roofline"] = schur_roofline(prob, out["kernel_ms_per_trial"])
        # the same kernel as it runs INSIDE Compute() (between the select and the Schur complement, its inputs and W in whatever cache
        # state the trial leaves them): HIP events around the launch of a profiled Compute(), event overhead included
        jac_ms = out["kernel_ms_per_trial"].get("jacobian")
        if jac_ms:
            out["roofline"]["avg_launch_us_in_compute"] = 1e3 * jac_ms
            out["roofline"]["frac_in_compute"] = alg_bytes / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["note"] = ("frac / achieved / avg_launch_us: back-to-back launches in one HIP-event bracket after the spin-up (the figure "
                                       "the rocprofv3 kernel stats' average agrees with); *_in_compute: HIP events around the ONE launch inside a profiled "
                                       "Compute(), ~2.7 us of event bracket included — the kernel trace has 14.1 us there = 0.40 "
                                       "(profiles/r06_k7_in_situ.txt: 2.07 GHz and 15 % more cycles than spun up, docs/LOG_r06.md section 3); *_cold: working "
                                       "set out of the Infinity Cache")
        # ---- a cold call: the mapmaker thread calls Compute() from idle — no spin-up, no warm-up trials, 50 ms of nothing queued
        cb = new_bundle(args.steps)
        ctx.sync()
        time.sleep(0.05)
        t0 = time.perf_counter()
        cb.Compute()
        ctx.sync()
        dtc = time.perf_counter() - t0
        out["cold_call"] = {"value": args.steps / dtc, "unit": "LM iterations/s", "ms_per_step": 1e3 * dtc / args.steps, "idle_ms_before": 50,
                            "trial_mix": trial_mix(cb.trials()),
                            "note": "one Compute() of the same workload entered 50 ms after the device went idle, without the K7 spin-up and "
                                    "the warm-up trials that precede `value`"}
        cb.close()
        # ---- one adjustment as the mapmaker thread runs it (src/MapMaker.cc:838-900): a NEW Bundle, Add*, Compute(), Get*, destroyed.
        # The reference's Compute() builds its index structures inside itself (src/Bundle.cc:116-123, :558-599); here that is
        # ptam_ba_prepare (device-built lists, csrc/ba_prepare.inc), timed as its own phase.  Host buffers -> PCIe upload included.
        e2e = []
        for _ in range(7):
            t0 = time.perf_counter()
            eb = synth.load_into(host.Bundle(ctx, max_iterations=args.steps, update_sq_conv_limit=0.0), prob)
            t1 = time.perf_counter()
            eb.prepare()
            t2 = time.perf_counter()
            eb.Compute()
            t3 = time.perf_counter()
            eb.get_all()
            eb.GetOutlierMeasurements()
            t4 = time.perf_counter()
            n_tr = len(eb.trials())
            eb.close()
            t5 = time.perf_counter()
            e2e.append((t5 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, n_tr))
        e2e = sorted(e2e[1:])   # (the first call finds the context's caches empty)
        med = e2e[len(e2e) // 2]
        out["compute_call_end_to_end"] = {
            "ms": 1e3 * med[0], "add_ms": 1e3 * med[1], "prepare_ms": 1e3 * med[2], "compute_ms": 1e3 * med[3], "get_ms": 1e3 * med[4],
            "destroy_ms": 1e3 * med[5], "trials": med[6], "calls_ms": [round(1e3 * e[0], 3) for e in e2e],
            "note": "median of 6 calls: new Bundle + Add* (bulk marshalling) + ptam_ba_prepare + Compute() + Get* + destroy, host buffers in, "
                    "host buffers out — one MapMaker::BundleAdjust (src/MapMaker.cc:838-900)"}
        # ---- the deterministic mode's figure beside the default one (fixed-order camera sums: bit-identical runs, one trajectory)
        if not args.deterministic:
            args.deterministic = True
            try:
                dtd, trd, _, _ = timed_compute(prob, args.steps, args.warmup)
                out["deterministic_mode"] = {"value": args.steps / dtd, "unit": "LM iterations/s", "ms_per_step": 1e3 * dtd / args.steps,
                                             "trial_mix": trial_mix(trd),
                                             "note": "ptam_ba_opts.deterministic = 1: the accept / reject / stay mix of this problem is the same in every run"}
            finally:
                args.deterministic = False
        # ---- BASELINE configs[4] on ONE device: the N = 1 point of the strong-scaling curve that `--gpus N` measures ----
        if not is_global and not args.no_global:
            big = synth.make_ba_problem(GLOBAL_BA["cams"], GLOBAL_BA["points"], synth.SEED_BA_GLOBAL, window=GLOBAL_BA["window"])
            dtg, trg, cg, _ = timed_compute(big, args.steps, args.warmup)
            gb = {"workload": workload_name(**GLOBAL_BA), "value": args.steps / dtg, "unit": "LM iterations/s",
                  "ms_per_step": 1e3 * dtg / args.steps, "steps": args.steps, "measurements": int(cg[3]),
                  "trial_mix": trial_mix(trg), "kernel_ms_per_trial": kernel_breakdown(big, args.steps)}
            gb["schur_roofline"] = schur_roofline(big, gb["kernel_ms_per_trial"])
            bb = synth.load_into(host.Bundle(ctx), big)
            bb.bench_jacobian(300)
            bms, bby = bb.bench_jacobian(50)
            gb["roofline_k7"] = {"kernel": "jac_accum_wave_kernel", "achieved": bby / (bms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": bby / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": bms * 1e3,
                                 "algorithmic_bytes_per_launch": bby}
            bb.close()
            out["global_ba_single_gpu"] = gb
        # ---- BASELINE configs[3]: PTAM's local adjustment (src/MapMaker.cc:788-829), 20 keyframes x 3000 points ----------
        if not args.no_local and (args.cams, args.points) != (LOCAL_BA["cams"], LOCAL_BA["points"]):
            try:
                loc = synth.make_ba_problem(LOCAL_BA["cams"], LOCAL_BA["points"], synth.SEED_BA_LOCAL)
                lsteps = 10   # (configs[3] is quoted at 10 LM iterations)
                # FIVE timed Compute() calls of the same problem, the median reported: a call is ten trials = 1 ms, and one host
                # hiccup of 80 - 100 us in it (the mailbox spin pre-empted once) was the "bimodal" 100 / 113 - 122 us of round 4 —
                # the per-kernel times and the trial mix of slow and fast runs are identical (tools/dev/r05_local_spread.sh)
                reps_l = []
                for _ in range(5):
                    dtl, trl, cl, _ = timed_compute(loc, lsteps, args.warmup)
                    reps_l.append(dtl)
                dtl = sorted(reps_l)[len(reps_l) // 2]
                lb = {"workload": workload_name(LOCAL_BA["cams"], LOCAL_BA["points"], 0), "value": lsteps / dtl, "unit": "LM iterations/s",
                      "ms_per_step": 1e3 * dtl / lsteps, "steps": lsteps, "keyframes_free": int(cl[1]), "measurements": int(cl[3]),
                      "ms_per_step_of_5_calls": [round(1e3 * x / lsteps, 4) for x in reps_l],
                      "trial_mix": trial_mix(trl), "kernel_ms_per_trial": kernel_breakdown(loc, lsteps)}
                n_lead_l = 0
                for a in trl["accepted"]:
                    if not a:
                        break
                    n_lead_l += 1
                if n_lead_l >= 2:
                    dta, _, _, _ = timed_compute(loc, n_lead_l, args.warmup)
                    lb["accepted_trial_us"] = 1e6 * dta / n_lead_l
                if not args.no_cpu_baseline:
                    from tests.oracle_lib import load_oracle
                    ob = synth.load_into(host.Bundle(host.Context(lib=load_oracle()), max_iterations=lsteps, update_sq_conv_limit=0.0), loc)
                    t0 = time.perf_counter()
                    ob.Compute()
                    cdt = time.perf_counter() - t0
                    otl = ob.trials()
                    lb["cpu_baseline"] = {"value": len(otl) / cdt, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                          "sample": f"the same problem, {len(otl)} lambda trials, oracle/ptam_oracle.cc, one thread"}
                    lb["parity_rel_err_final_trial"] = float(abs(otl["err_new"][-1] - trl["err_new"][-1]) / abs(otl["err_new"][-1]))
                    ob.close()
                out["local_ba_config4"] = lb
            except Exception as e:   # noqa: BLE001
                out["local_ba_config4"] = {"error": repr(e)}
        # the legs below report beside the headline record; a failure in one of them must not cost the record itself
        seq = None
        if not args.no_tracking:
            try:
                seq = moving_sequence(synth)
                out["tracking"] = tracking_bench(hip, host, synth, seq, replicas=not args.no_replicas)
            except Exception as e:   # noqa: BLE001
                out["tracking"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], otr = cpu_baseline(args, host, synth, prob, seq)
                # parity of the timed workload itself (oracle as checker, cheap: it already ran)
                rel = abs(otr["err_new"][-1] - trials["err_new"][-1]) / abs(otr["err_new"][-1])
                out["parity_rel_err_final_trial"] = float(rel)
            except Exception as e:   # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        # the LAST key of the line (the driver keeps the line's tail): both halves of BASELINE's metric and what a mapmaker thread sees
        def _g(d_, *ks, scale=1.0, nd=1):
            for k_ in ks:
                d_ = d_.get(k_) if isinstance(d_, dict) else None
            return round(scale * d_, nd) if isinstance(d_, (int, float)) else None
        out["record_version"] = 6   # 4: tracking.tracked_fps became the moving-camera sequence (stationary: tracked_fps_stationary); 5: + summary, tracked_fps_moving; 6: + compute_call_end_to_end, summary.prepare_ms / trial_mix
        out["value_note"] = ("`value` is the WARM figure (bundles built, K7 spun up until its launch time settles, warm-up trials, then K timed "
                             "trials); `cold_call` is one Compute() 50 ms after the device went idle — what PTAM's mapmaker thread sees")
        out["summary"] = {"ba_it_s": round(out["value"], 1), "cold_call_it_s": _g(out, "cold_call", "value"),
                          "accepted_trial_us": _g(out, "accepted_trial_us"), "k7_roofline_frac": _g(out, "roofline", "frac", nd=3),
                          "solve_us": _g(out, "kernel_ms_per_trial", "solve", scale=1e3), "schur_us": _g(out, "kernel_ms_per_trial", "schur", scale=1e3),
                          "local_ba_it_s": _g(out, "local_ba_config4", "value"), "global_ba_it_s": _g(out, "global_ba_single_gpu", "value"),
                          "tracked_fps": _g(out, "tracking", "tracked_fps"), "frame_us": _g(out, "tracking", "frame_us"),
                          "prepare_ms": _g(out, "prepare_ms", nd=3), "compute_call_ms": _g(out, "compute_call_end_to_end", "ms", nd=3),
                          "k7_frac_in_compute": _g(out, "roofline", "frac_in_compute", nd=3), "trial_mix": out.get("trial_mix")}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        if comm is not None and not one_gpu:
            hip.rccl_destroy(comm)
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, host, synth, prob, seq=None):
    """the oracle (kind "port") on the host cores: one thread on the timed problem (a bounded sample: at most 20 trials),
    a tracked-frame loop, and one independent replica per PHYSICAL core for the per-node figure"""
    from tests.oracle_lib import load_oracle
    oracle = load_oracle()
    octx = host.Context(lib=oracle)
    ob = synth.load_into(host.Bundle(octx, max_iterations=args.steps, update_sq_conv_limit=0.0), prob)
    t0 = time.perf_counter()
    ob.Compute()
    cdt = time.perf_counter() - t0
    otr = ob.trials()
    cpu = {"value": len(otr) / cdt, "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"same {args.cams}x{args.points} problem, {len(otr)} lambda trials, oracle/ptam_oracle.cc "
                     f"(single-thread restatement of src/Bundle.cc; the upstream binary cannot be built here)",
           "host_threads_available": os.cpu_count(), "cpu_model": cpu_model()}
    if not args.no_tracking:
        a, b = synth.make_frame_pair()
        kfa = host.KeyFrame(octx).MakeKeyFrame_Lite(a)
        # the SAME chain as tracking.tracked_fps — MakeKeyFrame_Lite of the new frame + the whole Tracker::TrackMap (PVS loop,
        # set choice, coarse search + ten coarse pose iterations, fine search, ten fine pose iterations) on the same map,
        # the same frame and the same prediction — through the oracle's twin of the entry point (ptamo_track_map_frame)
        case = synth.make_trackmap_case([kfa.level(l) for l in range(4)])
        tr = host.Tracker(octx, len(case["world"]))
        tr.set_map(case["world"], case["pixel_right_w"], case["pixel_down_w"], kfa, case["src_level"], case["center"])
        kfc = host.KeyFrame(octx)
        oo = tr.opts()
        pose_in = np.ascontiguousarray(case["pose_in"])
        for _ in range(3):
            tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
            rr = tr.TrackFrame(kfc, b.ctypes.data, pose_in, oo)
        nf = 200
        t0 = time.perf_counter()
        for _ in range(nf):
            tr.set_shuffle(case["shuffle_levels"], case["shuffle_fine"])
            rr = tr.TrackFrame(kfc, b.ctypes.data, pose_in, oo)
        cpu["tracked_fps_stationary"] = nf / (time.perf_counter() - t0)
        cpu["tracked_stationary_sample"] = (f"{nf} frames of the chain tracking.tracked_fps_stationary runs: keyframe of the 640x480 frame + TrackMap over "
                                 f"{len(case['world'])} map points, {int(sum(rr['attempted']))} patches searched, {int(rr['n_meas'])} found, "
                                 f"oracle/ptam_oracle.cc ptamo_track_map_frame, one thread")
        tr.close()
        if seq is not None:
            # the moving-camera sequence tracking.tracked_fps runs, closed loop through the oracle's twin of ptam_track_frame
            kf0 = host.KeyFrame(octx).MakeKeyFrame_Lite(seq["kf_image"])
            m = synth.make_sequence_map([kf0.level(l) for l in range(4)], seq["kf_pose"])
            tr = host.Tracker(octx, len(m["world"]))
            tr.set_map(m["world"], m["pixel_right_w"], m["pixel_down_w"], kf0, m["src_level"], m["center"])
            mm = tr.motion_model(seq["poses"][0])
            oo = tr.opts()
            nseq = len(seq["frames"])
            tot = {"n": 0, "searched": 0, "reused": 0, "found": 0, "err": 0.0}
            t0 = None
            for k in range(3 * nseq):
                if k == nseq:                       # the first round warms the finders up, two rounds are timed
                    t0 = time.perf_counter()
                f = k % nseq
                tr.set_shuffle(m["shuffle_levels"], m["shuffle_fine"])
                rr = tr.TrackFrameMoving(kfc, seq["frames"][f].ctypes.data, mm, oo)
                if k >= nseq:
                    tot["n"] += 1
                    tot["searched"] += int(rr["n_coarse"] + rr["n_top"] + rr["n_fine"])
                    tot["reused"] += int(rr["templates_reused"])
                    tot["found"] += int(rr["n_meas"])
                    tot["err"] = max(tot["err"], float(np.abs(rr["pose"] - seq["poses"][f]).max()))
            cpu["tracked_fps"] = tot["n"] / (time.perf_counter() - t0)
            cpu["tracked_sample"] = (f"{tot['n']} frames of the moving-camera sequence tracking.tracked_fps runs (keyframe + motion model + TrackMap, "
                                     f"{tot['searched'] / tot['n']:.0f} patches searched and {tot['found'] / tot['n']:.0f} found per frame, "
                                     f"{100.0 * tot['reused'] / max(1, tot['searched']):.0f} % of the templates kept, max pose error {tot['err']:.1e}), "
                                     f"oracle/ptam_oracle.cc ptamo_track_frame, one thread")
            tr.close()
        # round 1-2's figure, comparable to tracking.fine_stage_only_fps: keyframe + 1000-patch search + ONE pose loop
        q, t = synth.make_patch_queries([kfa.level(l) for l in range(4)], n=1000)
        kfb = host.KeyFrame(octx)
        pc = synth.make_pose_case()
        pf = host.PatchFinder(octx)
        t0 = time.perf_counter()
        nf = 100
        for _ in range(nf):
            kfb.MakeKeyFrame_Lite(b)
            pf.FindPatchCoarse(kfb, q, t)
            octx.pose_gn(pc["world"], pc["found"], pc["sqrt_inv_noise"], pc["init_pose"])
        cpu["fine_stage_only_fps"] = nf / (time.perf_counter() - t0)
    # per-node figure: one independent problem per physical core, the same restatement (each replica holds ~0.6 KB per
    # measurement: the replica count is also bounded by a quarter of the free memory)
    import threading
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
        mem_cap = int(psutil.virtual_memory().available * 0.25 // (len(prob["cam_idx"]) * 1024 + (64 << 20)))
    except Exception:
        phys, mem_cap = max(1, (os.cpu_count() or 2) // 2), 64
    n_rep = max(1, min(phys, mem_cap))
    reps = []
    for _ in range(n_rep):
        rc = host.Context(lib=oracle)
        reps.append((rc, synth.load_into(host.Bundle(rc, max_iterations=args.steps, update_sq_conv_limit=0.0), prob)))
    th = [threading.Thread(target=b.Compute) for _, b in reps]   # ctypes drops the GIL inside the call
    t0 = time.perf_counter()
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    rdt = time.perf_counter() - t0
    cpu["node_replicas"] = {"value": sum(len(b.trials()) for _, b in reps) / rdt, "unit": "LM iterations/s",
                            "cores": n_rep, "physical_cores": phys,
                            "note": "one independent copy of the problem per physical host core (fewer if memory bounds it)"}
    for _, b in reps:
        b.close()
    return cpu, otr


if __name__ == "__main__":
    main()
