"""CPU: the C++ oracle against the committed golden vectors (independent numpy restatement)."""
import pytest

from ptam_cg_amd import _abi
from tests import golden_util as G


@pytest.mark.parametrize("variant", [_abi.HALFSAMPLE_R, _abi.HALFSAMPLE_T])
def test_keyframe(oracle, variant):
    G.check_keyframe(oracle, variant)


def test_patch(oracle):
    G.check_patch(oracle)


def test_pose(oracle):
    G.check_pose(oracle)


@pytest.mark.parametrize("name", ["ba_8x50", "ba_20x300", "ba_banded_30x200"])
def test_bundle(oracle, name):
    G.check_ba(oracle, name)


def test_subpix(oracle):
    G.check_subpix(oracle)


def test_template_cont(oracle):
    G.check_template_cont(oracle)


def test_epipolar(oracle):
    G.check_epipolar(oracle)


def test_pvs(oracle):
    G.check_pvs(oracle)


def test_keyframe_rest(oracle):
    G.check_keyframe_rest(oracle)


def test_refind(oracle):
    G.check_refind(oracle)


def test_refind_pairs(oracle):
    G.check_refind_pairs(oracle)
