"""CPU: the C-ABI library loads and exports every symbol include/ptam_hip.h declares (no compute
calls — there is no GPU here), struct layouts match the header, host-side logic is sound."""
import ctypes
import os
import re

import numpy as np
import pytest

from ptam_cg_amd import _abi, host, synth
from ptam_cg_amd.sharding import shard_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ptam_cg_amd", "csrc", "libptam_hip.so")
HEADERS = [os.path.join(ROOT, "include", "ptam_hip.h"), os.path.join(ROOT, "include", "ptam_hip_bench.h")]


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def header_functions():
    src = "\n".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptam_[a-z0-9_]+)\s*\(", src)) - {"ptam_allreduce_f64_fn"})


def test_library_exports_every_declared_symbol(built):
    names = header_functions()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(built, n)]
    assert not missing, missing


def test_python_prototypes_cover_the_header():
    assert sorted("ptam_" + n for n in _abi.DECLARED) == header_functions()


def test_no_gpu_means_loud_failure(built):
    """the product has no CPU fallback: creating a context without a device must fail with PTAM_E_HIP"""
    n = ctypes.c_int(-1)
    rc = built.ptam_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    cam = _abi.CamParams(*host.DEFAULT_CAMERA, 640, 480)
    h = ctypes.c_void_p()
    assert built.ptam_ctx_create(ctypes.byref(cam), 0, ctypes.byref(h)) == -2
    built.ptam_last_error.restype = ctypes.c_char_p
    assert built.ptam_last_error()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_abi.PatchQuery) == 16 and ctypes.sizeof(_abi.PatchResult) == 40
    assert ctypes.sizeof(_abi.Projection) == 80 and ctypes.sizeof(_abi.PoseMeas) == 48
    assert ctypes.sizeof(_abi.PoseUpdateMeas) == 136 and ctypes.sizeof(_abi.BaTrial) == 48
    assert ctypes.sizeof(_abi.CamParams) == 48 and ctypes.sizeof(_abi.GnOpts) == 40 and ctypes.sizeof(_abi.BaOpts) == 40
    assert ctypes.sizeof(_abi.MotionModel) == 12 * 8 * 2 + 6 * 8 + 4 * 8 + 16


def test_product_never_touches_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may reference oracle/"""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "ptam_cg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"oracle/|ptamo_|np_oracle|oracle_lib", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    assert "oracle" not in open(os.path.join(ROOT, "include", "ptam_hip.h")).read().lower()


def test_synthetic_inputs_are_deterministic_and_sane():
    a1, b1 = synth.make_frame_pair()
    a2, _ = synth.make_frame_pair()
    assert np.array_equal(a1, a2) and a1.shape == (480, 640) and a1.dtype == np.uint8
    assert not np.array_equal(a1, b1)
    pc = synth.make_pose_case()
    assert 900 <= len(pc["world"]) <= 1000 and 0.02 < pc["is_outlier"].mean() < 0.09
    p = synth.make_ba_problem(20, 3000, synth.SEED_BA_LOCAL)
    assert len(p["cam_idx"]) == 60000 and p["fixed"].sum() == 1              # every point visible
    g = synth.make_ba_problem(40, 500, 3, window=16)
    per_pt = np.bincount(g["pt_idx"], minlength=500)
    assert per_pt.max() <= 16


def test_shard_problem_is_a_partition():
    prob = synth.make_ba_problem(6, 101, 8)
    shards = [shard_problem(prob, r, 3) for r in range(3)]
    ids = np.concatenate([s["global_point_ids"] for s in shards])
    assert sorted(ids) == list(range(101))
    assert sum(len(s["cam_idx"]) for s in shards) == len(prob["cam_idx"])
    mids = np.concatenate([s["global_meas_ids"] for s in shards])
    assert sorted(mids) == list(range(len(prob["cam_idx"])))
    for s in shards:
        assert np.array_equal(s["points"], prob["points"][s["global_point_ids"]])
        assert np.array_equal(s["global_point_ids"][s["pt_idx"]], prob["pt_idx"][s["global_meas_ids"]])
        assert s["pt_idx"].max() < len(s["points"]) and len(s["poses"]) == 6


def _schur_row(mode, t, rid):
    """csrc/ba_schur.inc: which row (6 * camera slot + parameter) of a 48-row camera tile MFMA row `rid` of fragment t stands for"""
    if mode == 1:
        return rid
    if t < 2:
        return 6 * (rid // 3) + 2 * (rid % 3) + t
    q = 16 + (rid >> 1)
    return 6 * (q // 3) + 2 * (q % 3) + (rid & 1)


@pytest.mark.parametrize("variant", range(6))
def test_schur_index_map_is_the_inverse_of_the_row_mapping(built, variant):
    """round 5: the Schur tile kernel's epilogue gathers a partial tile through a host-built index map.  Re-derived here from the
    row mapping and the D layout of v_mfma_f64_16x16x4_f64 (column = lane & 15, row = (lane >> 4) + 4 v): every element of the
    [8][8][6][6] + E[48] layout that a fragment covers points at exactly one accumulator slot, no slot is used twice, and the
    elements no fragment covers are marked as zeros."""
    out = (ctypes.c_uint16 * 2352)()
    assert built.ptam_ba_schur_index_map(variant, out, 2352) == 2352
    got = np.frombuffer(out, dtype=np.uint16).copy()
    diag = variant < 3
    ma = 3 - variant if diag else 6 - variant
    mb = ma if diag else 3
    want = np.full(2352, 0xFFFF, dtype=np.uint16)
    for ti in range(ma):
        for tj in range(mb):
            for lane in range(64):
                for v in range(4):
                    row, col = _schur_row(ma, ti, (lane >> 4) + 4 * v), _schur_row(mb, tj, lane & 15)
                    if row < 48 and col < 48:
                        e = ((row // 6) * 8 + col // 6) * 36 + (row % 6) * 6 + col % 6
                        assert want[e] == 0xFFFF
                        want[e] = ((ti * 3 + tj) * 4 + v) * 64 + lane
    if diag:
        for t in range(ma):
            for rid in range(16):
                row = _schur_row(ma, t, rid)
                if row < 48:
                    want[2304 + row] = (36 + t) * 64 + rid
    assert np.array_equal(got, want)
    used = got[got != 0xFFFF]
    assert len(set(used.tolist())) == len(used) and used.max() < 39 * 64
    assert built.ptam_ba_schur_index_map(6, out, 2352) < 0 and built.ptam_ba_schur_index_map(0, out, 100) < 0


def test_persistent_solve_keeps_its_in_flight_registers_out_of_scratch():
    """ldlt_chain.inc requests tiles with inline-asm loads whose destination registers the compiler believes valid at
    once (ch_ld2_issue / ch_ld2_wait).  That holds as long as nothing moves those registers between request and wait — which a
    spill would.  The persistent solve's kernels must therefore use no scratch at all: checked on the compiler's own summary."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this host")
    src = os.path.join(ROOT, "ptam_cg_amd", "csrc", "solve.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "solve.s")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-o", out, src],
                              stderr=subprocess.DEVNULL)
        txt = open(out).read()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*ldlt_chain\S*)(.*?)\.end_amdhsa_kernel", txt, re.S):
        seen += 1
        body = m.group(2)
        size = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert size and int(size.group(1)) == 0, (m.group(1), size and size.group(1))
    assert seen >= 1
