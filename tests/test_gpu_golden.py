"""-m gpu: the HIP path against the committed golden vectors."""
import pytest

from ptam_cg_amd import _abi
from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [_abi.HALFSAMPLE_R, _abi.HALFSAMPLE_T])
def test_keyframe(hip, variant):
    G.check_keyframe(hip, variant)


def test_patch(hip):
    G.check_patch(hip)


def test_pose(hip):
    G.check_pose(hip)


@pytest.mark.parametrize("name", ["ba_8x50", "ba_20x300", "ba_banded_30x200"])
def test_bundle(hip, name):
    G.check_ba(hip, name)


def test_subpix(hip):
    G.check_subpix(hip)


def test_template_cont(hip):
    G.check_template_cont(hip)


def test_epipolar(hip):
    G.check_epipolar(hip)


def test_pvs(hip):
    G.check_pvs(hip)


def test_keyframe_rest(hip):
    G.check_keyframe_rest(hip)


def test_refind(hip):
    G.check_refind(hip)


def test_refind_pairs(hip):
    G.check_refind_pairs(hip)
