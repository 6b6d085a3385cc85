"""The vocabulary in the reference's own on-disk format (ORBvoc.txt): bool TemplatedVocabulary::loadFromTextFile,
Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1427.  Host-only tests: the product's parser (csrc/vocab_text.cpp: mmap + strtol),
the oracle's restatement (oracle/vocab_text.cc: getline + stringstream + operator>>, as the reference reads) and the oracle's writer
(saveToTextFile's bytes, :1431-1450) against each other and against the tree that was written -- up to the reference's size
(k = 10, L = 6: 1 111 111 nodes, 10^6 words; ORBvoc.txt itself is not in the repository).  The GPU tests
(tests/test_gpu_parity.py::test_vocabulary_from_text_at_the_references_size) put a tree loaded this way on the device."""
import os

import numpy as np
import pytest

from multi_orbslam3_amd import _capi as capi, api, synth, views
from oracle import binding as ob

KEYS = ("child_start", "child_ids", "desc", "weight", "word_id")


def _same(a, b):
    return all(np.array_equal(a[k], b[k]) for k in KEYS)


def _write(tmp_path, v, name="voc.txt", scoring=0):
    vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], v["L"], weighting=v.get("weighting", capi.ORBV_TF_IDF))
    path = str(tmp_path / name)
    ob.vocab_save_text(vv, v["k"], path, scoring=scoring)
    return path


def _ragged(seed=5):
    v = synth.make_vocabulary(k=10, L=3, seed=seed)
    v["weight"] = np.array([float("%.6g" % w) for w in v["weight"]])      # what survives saveToTextFile's operator<<(double)
    v["word_id"] = np.where(v["word_id"] < 0, 0, v["word_id"]).astype(np.int32)      # Node(): word_id(0) for inner nodes (:316)
    v["desc"] = v["desc"].copy()
    v["desc"][0] = 0                                                      # the root is not in the file (saveToTextFile starts at node 1)
    return v


def test_round_trip_of_a_ragged_tree(tmp_path):
    v = _ragged()
    path = _write(tmp_path, v)
    head = open(path).readline()
    assert head == "10 3  0 0\n"                                          # "k L  scoring weighting" (two blanks: :1436)
    a, b = api.load_text_vocabulary(path), ob.vocab_load_text(path)
    assert _same(a, b) and _same(a, v)
    assert a["k"] == b["k"] == 10 and a["L"] == 3 and a["scoring"] == 0 and a["weighting"] == capi.ORBV_TF_IDF and a["scoring_norm"] == capi.ORBV_NORM_L1
    leaves = np.diff(v["child_start"]) == 0
    assert a["n_words"] == b["n_words"] == int(leaves.sum())
    assert np.array_equal(a["word_id"][leaves], np.arange(leaves.sum()))  # words in file order (:1413-1419)


def test_trailing_newline_makes_the_references_stray_node_only_on_request(tmp_path):
    """`while(!f.eof()) getline` reads the empty line after the last newline as one more node: parent 0, no children, weight 0, word
    id 0, descriptor never written.  Default: the tree the file describes; ORBV_TEXT_KEEP_TRAILING_NODE: that node, descriptor zero."""
    v = _ragged(7)
    path = _write(tmp_path, v)
    n = len(v["weight"])
    a, b = api.load_text_vocabulary(path, True), ob.vocab_load_text(path, True)
    assert _same(a, b) and len(a["weight"]) == n + 1
    root = a["child_ids"][a["child_start"][0]: a["child_start"][1]]
    assert root[-1] == n and a["child_start"][n + 1] == a["child_start"][n]          # last child of the root, childless
    assert a["weight"][n] == 0 and a["word_id"][n] == 0 and not a["desc"][n].any()
    assert a["n_words"] == api.load_text_vocabulary(path)["n_words"]
    # a file that does NOT end with a newline has no such line: the flag changes nothing
    raw = open(path).read()
    assert raw.endswith("\n")
    p2 = str(tmp_path / "nonl.txt")
    open(p2, "w").write(raw[:-1])
    for keep in (False, True):
        a2, b2 = api.load_text_vocabulary(p2, keep), ob.vocab_load_text(p2, keep)
        assert _same(a2, b2) and _same(a2, v)


@pytest.mark.parametrize("header", ["25 6 0 0", "10 0 0 0", "10 11 0 0", "10 6 6 0", "10 6 0 4", "-1 6 0 0", "10 6", "hello"])
def test_headers_the_reference_refuses(tmp_path, header):
    """:1361-1365: k in 0..20, L in 1..10, scoring in 0..5, weighting in 0..3."""
    path = str(tmp_path / "bad.txt")
    open(path, "w").write(header + "\n0 1 " + " ".join(["1"] * 32) + " 0.5\n")
    with pytest.raises(capi.OrbGpuError):
        api.load_text_vocabulary(path)
    with pytest.raises(Exception):
        ob.vocab_load_text(path)


def test_malformed_node_lines_are_errors_not_garbage(tmp_path):
    good = "0 1 " + " ".join(str(i) for i in range(32)) + " 0.25"
    for body in ("0 1 " + " ".join(["7"] * 20),            # a truncated line: the reference would read stale stream state
                 "5 1 " + " ".join(["7"] * 32) + " 1.0",   # a parent that does not exist yet
                 good + "\n\n" + good):                     # a node line after a blank one (the reference shifts every later id by one)
        path = str(tmp_path / "m.txt")
        open(path, "w").write("10 2 0 0\n" + body + "\n")
        with pytest.raises(capi.OrbGpuError):
            api.load_text_vocabulary(path)
    with pytest.raises(capi.OrbGpuError):
        api.load_text_vocabulary(str(tmp_path / "does_not_exist.txt"))


def test_number_forms_and_scoring_types(tmp_path):
    """Weights as operator<<(double) writes them (1e-05, 0, 12.3457), descriptor integers cast to unsigned char, the four weighting and
    six scoring types (mustNormalize: L1 for L1_NORM / CHI_SQUARE / KL / BHATTACHARYYA, L2 for L2_NORM, none for DOT_PRODUCT)."""
    lines = ["0 0 " + " ".join(["255"] * 32) + "  0", "0 1 " + " ".join(["0"] * 31) + " 300  1e-05",
             "1 1 " + " ".join(str((7 * i) % 256) for i in range(32)) + "  12.3457", "1 1 " + " ".join(["1"] * 32) + " 2"]
    for scoring, norm in ((0, capi.ORBV_NORM_L1), (1, capi.ORBV_NORM_L2), (2, capi.ORBV_NORM_L1), (3, capi.ORBV_NORM_L1), (4, capi.ORBV_NORM_L1), (5, capi.ORBV_NORM_NONE)):
        for weighting in range(4):
            path = str(tmp_path / "n.txt")
            open(path, "w").write("9 2  %d %d\n" % (scoring, weighting) + "\n".join(lines) + "\n")
            a, b = api.load_text_vocabulary(path), ob.vocab_load_text(path)
            assert _same(a, b)
            assert a["scoring_norm"] == b["scoring_norm"] == norm and a["weighting"] == b["weighting"] == weighting and a["k"] == 9
    assert list(a["weight"]) == [0.0, 0.0, 1e-05, 12.3457, 2.0]
    assert a["desc"][2][31] == 300 % 256 and list(a["word_id"]) == [0, 0, 0, 1, 2]
    assert list(a["child_start"]) == [0, 2, 4, 4, 4, 4] and list(a["child_ids"]) == [1, 2, 3, 4]


def test_the_references_size_round_trips_and_parses_fast(tmp_path):
    """k = 10, L = 6: 1 111 111 nodes / 10^6 words / 140 MB of text, the size of ORBvoc.txt.  Product parser, oracle parser and the
    written tree agree array for array."""
    import time
    v = synth.make_full_vocabulary(10, 6)
    assert len(v["weight"]) == 1111111
    path = _write(tmp_path, v, "voc6.txt")
    try:
        assert os.path.getsize(path) > 100e6
        t0 = time.time()
        a = api.load_text_vocabulary(path)
        t_product = time.time() - t0
        b = ob.vocab_load_text(path)
        assert _same(a, v) and _same(a, b) and a["n_words"] == 10 ** 6 and a["k"] == 10 and a["L"] == 6
        assert t_product < 10.0
    finally:
        os.remove(path)
