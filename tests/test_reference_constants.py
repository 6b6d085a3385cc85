"""Tables and constants of the hot path, checked against the TEXT of the reference's sources where the reference is present (this
container: /root/reference) and against committed digests everywhere else.  This is the only place where the oracle and the
product are held against something the reference itself ships: it has no golden vectors and cannot be built here (no OpenCV /
Eigen / boost), but its rBRIEF sampling pattern and its thresholds are literals in its sources."""
import hashlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/orb_slam3_ros/orb_slam3"
# sha256 of the 1024 integers of bit_pattern_31_ (S/ORBextractor.cc:148-406) joined by commas, computed from the reference by
# this file's _reference_pattern() on 2026-10-03
PATTERN_SHA256 = "88df8ca875cc8db56799edd57bb914edad8acb2d48c202b7a464a575b55dbdb8"


def _ints_of_inc(path):
    text = "\n".join(ln for ln in open(path).read().splitlines() if not ln.strip().startswith("//"))
    return [int(x) for x in re.findall(r"-?\d+", text)]


def _digest(nums):
    return hashlib.sha256(",".join(map(str, nums)).encode()).hexdigest()


def _reference_pattern():
    src = open(os.path.join(REF, "src", "ORBextractor.cc")).read()
    m = re.search(r"static int bit_pattern_31_\[256\*4\]\s*=\s*\{(.*?)\};", src, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return [int(x) for x in re.findall(r"-?\d+", body)]


def test_rbrief_pattern_is_the_references_table():
    for p in (os.path.join(ROOT, "oracle", "orb_pattern_data.inc"), os.path.join(ROOT, "multi_orbslam3_amd", "csrc", "orb_pattern_data.inc")):
        nums = _ints_of_inc(p)
        assert len(nums) == 1024 and max(abs(v) for v in nums) <= 13, p
        assert _digest(nums) == PATTERN_SHA256, p
    if os.path.isdir(REF):
        ref = _reference_pattern()
        assert len(ref) == 1024 and _digest(ref) == PATTERN_SHA256


def _src(*parts):
    return open(os.path.join(ROOT, *parts)).read()


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_thresholds_are_the_references_literals():
    ex = open(os.path.join(REF, "src", "ORBextractor.cc")).read()
    ma = open(os.path.join(REF, "src", "ORBmatcher.cc")).read()
    op = open(os.path.join(REF, "src", "Optimizer.cc")).read()
    lm = open(os.path.join(REF, "Thirdparty", "g2o", "g2o", "core", "optimization_algorithm_levenberg.cpp")).read()

    def ref_int(text, name):
        return int(re.search(r"\b%s\s*=\s*(-?\d+)\s*;" % re.escape(name), text).group(1))

    mine_common = _src("multi_orbslam3_amd", "csrc", "common.hpp")
    mine_oex = _src("oracle", "extractor.cc")
    mine_omat = _src("oracle", "matching.cc")
    for ref_name, my_name in (("PATCH_SIZE", "kPatch"), ("HALF_PATCH_SIZE", "kHalfPatch"), ("EDGE_THRESHOLD", "kEdge")):
        v = ref_int(ex, ref_name)
        assert ref_int(mine_common, my_name) == v and ref_int(mine_oex, my_name) == v, ref_name
    for name in ("TH_HIGH", "TH_LOW", "HISTO_LENGTH"):
        v = ref_int(ma, "ORBmatcher::" + name)
        assert ref_int(mine_omat, name) == v, name
        assert re.search(r"\b%s\s*=\s*%d\b" % (name if name != "HISTO_LENGTH" else "kHistoLength|HISTO_LENGTH", v), _src("multi_orbslam3_amd", "csrc", "matcher.hip")) or \
            str(v) in _src("multi_orbslam3_amd", "csrc", "matcher.hip"), name
    # chi-square thresholds / Huber deltas of the local BA and the pose optimisation (95 %: 2 and 3 degrees of freedom)
    for lit in ("5.991", "7.815"):
        assert lit in op
        assert lit in _src("oracle", "lba.cc") and lit in _src("multi_orbslam3_amd", "csrc", "lba.hip"), lit
    # g2o's Levenberg-Marquardt constants (G/core/optimization_algorithm_levenberg.cpp:47-51)
    assert re.search(r"_tau\s*=\s*1e-5", lm) and re.search(r"_goodStepUpperScale\s*=\s*2\./3\.", lm) and re.search(r"_goodStepLowerScale\s*=\s*1\./3\.", lm)
    assert re.search(r'"maxTrialsAfterFailure",\s*10\)', lm)
    for mine in (_src("oracle", "lba.cc"), _src("multi_orbslam3_amd", "csrc", "lba.hip")):
        assert "1e-5" in mine and re.search(r"2\.\s*/\s*3\.", mine) and re.search(r"1\.\s*/\s*3\.", mine)
        assert re.search(r"qmax\s*(<|==)\s*10", mine)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_keyframe_wire_block_is_the_references_message_layout():
    """The KF wire block of the server tick (47 bytes per feature) is the reference's ROS message fields, packed: CvKeyPoint.msg =
    f32 x, f32 y, u8 size, f32 angle, u8 response, i8 octave (15 bytes), Descriptor.msg = u8[32]."""
    size = {"float32": 4, "uint8": 1, "int8": 1}

    def fields(name):
        out = []
        for ln in open(os.path.join("/root/reference/src/orb_slam3_ros/msg", name)).read().splitlines():
            ln = ln.split("#")[0].strip()
            if ln:
                t, f = ln.split()
                out.append((t, f))
        return out

    kp = fields("CvKeyPoint.msg")
    assert [t for t, _ in kp] == ["float32", "float32", "uint8", "float32", "uint8", "int8"]
    assert sum(size[t] for t, _ in kp) == 15
    assert fields("Descriptor.msg") == [("uint8[32]", "mDescriptor")]
    mine = _src("multi_orbslam3_amd", "csrc", "matcher.hip")
    assert re.search(r"kWireKp\s*=\s*15\s*,\s*kWireDesc\s*=\s*32", mine)
    assert "f32 x, f32 y, u8 size, f32 angle, u8 response, i8 octave" in mine
