"""The drop-in glue (include/orbgpu_dropin.hpp) against the ACCESS RULES of the reference's own classes.

The glue is a set of templates over the reference's Frame / KeyFrame / MapPoint / Map; in this repository it is compiled against the
header-only mocks of tests/cpp/mock_orbslam3.hpp.  Two mechanical checks keep the mocks from hiding what the reference would refuse:

1. tests/cpp/glue_access_check.cpp instantiates every glue template with -DMOCK_STRICT_ACCESS, the build in which the mocks'
   MOCK_PROTECTED sections really are protected (runs everywhere: g++ only).
2. Where the reference is present (the build container: /root/reference never travels), the mocks' partition is held against
   I/MapPoint.h, I/KeyFrame.h, I/Frame.h and I/Map.h: every member the mocks call "public in the reference" is declared in a public
   section of that class there, every MOCK_PROTECTED member in a non-public one, and every member of a "reference-side edits"
   section is absent from the reference AND listed in INTEGRATION.md's table of edits -- so a member the glue needs that is neither
   public nor listed fails here (round 5: MapPoint::mfMinDistance / mfMaxDistance were read directly; they are protected,
   I/MapPoint.h:244,281-282)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/src/orb_slam3_ros/orb_slam3/include"
CLASSES = ("MapPoint", "KeyFrame", "Frame", "Map")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#"))


def _class_body(text, name):
    """Text between the braces of `class name ... {` (the definition, not a forward declaration)."""
    for m in re.finditer(r"\bclass\s+%s\b[^;{]*\{" % re.escape(name), text):
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        return text[m.end():i - 1]
    raise AssertionError("class %s not found" % name)


def _declared_names(stmt):
    """Names a member declaration introduces: the function name, or every declarator of a data member."""
    stmt = re.sub(r"\btemplate\s*<[^>]*>", " ", stmt)
    stmt = stmt.strip()
    if not stmt or re.match(r"(friend|using|typedef|enum|static_assert)\b", stmt):
        return []
    m = re.match(r"(class|struct)\s+(\w+)\s*$", stmt)
    if m:
        return [m.group(2)]
    # drop template argument lists (std::map<KeyFrame*, std::tuple<int, int>> x) so that their commas do not split declarators
    prev = None
    while prev != stmt:
        prev = stmt
        stmt = re.sub(r"<[^<>()]*>", " ", stmt)
    if "(" in stmt:
        head = stmt[:stmt.index("(")]
        if "operator" in head:
            return []
        ids = re.findall(r"[A-Za-z_]\w*", head)
        return ids[-1:] if ids else []
    names = []
    for decl in stmt.split(","):
        decl = re.sub(r"=.*", "", decl)
        decl = re.sub(r"\[[^\]]*\]", "", decl)
        ids = re.findall(r"[A-Za-z_]\w*", decl)
        if ids:
            names.append(ids[-1])
    return names


def _members_by_section(body, labels):
    """[(label text, [names])] for the depth-0 declarations of a class body; `labels` = regex of access labels."""
    out, cur, names = [], "private", []
    i, n, stmt = 0, len(body), ""
    lab = re.compile(r"\s*(%s)\s*:(?!:)" % labels)
    while i < n:
        if not stmt.strip():
            m = lab.match(body, i)
            if m:
                out.append((cur, names))
                cur, names = m.group(1), []
                i = m.end()
                continue
        c = body[i]
        if c == "(":                      # parameter lists / constructor calls: skipped, recorded as "()"
            depth = 1
            i += 1
            while depth and i < n:
                depth += {"(": 1, ")": -1}.get(body[i], 0)
                i += 1
            stmt += "()"
            continue
        if c == "{":                      # a function body, an in-class initialiser or a nested type: skipped
            depth = 1
            i += 1
            while depth and i < n:
                depth += {"{": 1, "}": -1}.get(body[i], 0)
                i += 1
            if "(" in stmt or re.match(r"\s*(class|struct|enum)\b", stmt):      # a function body / nested type ends the declaration
                if not re.match(r"\s*(class|struct|enum)\b", stmt):
                    names += _declared_names(stmt)
                    stmt = ""
                else:
                    stmt += " "
            continue
        if c == ";":
            names += _declared_names(stmt)
            stmt = ""
            i += 1
            continue
        stmt += c
        i += 1
    out.append((cur, names))
    return out


def _mock_partition():
    raw = open(os.path.join(ROOT, "tests", "cpp", "mock_orbslam3.hpp")).read()
    # keep the section comments: they say which kind of section a label opens
    kinds = {}
    for cls in CLASSES:
        text = raw[raw.index("class %s {" % cls):]
        depth, i = 0, text.index("{")
        start = i + 1
        while True:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
            if depth == 0:
                break
        body = text[start:i - 1]
        # tag the labels with their kind before the comments go
        tagged = re.sub(r"(public|MOCK_PROTECTED)\s*:\s*//\s*----\s*(public in the reference|reference-side edits|protected in the reference|test instrumentation)[^\n]*",
                        lambda m: {"public in the reference": "REFPUBLIC", "reference-side edits": "EDIT", "protected in the reference": "REFPROTECTED",
                                   "test instrumentation": "INSTR"}[m.group(2)] + ":", body)
        assert "public:" not in _strip_comments(tagged) and "MOCK_PROTECTED:" not in _strip_comments(tagged), "%s: a section without a kind comment" % cls
        part = {"REFPUBLIC": [], "EDIT": [], "REFPROTECTED": [], "INSTR": []}
        for label, names in _members_by_section(_strip_comments(tagged), "REFPUBLIC|EDIT|REFPROTECTED|INSTR"):
            if label in part:
                part[label] += names
            else:
                assert not names, (cls, label, names)
        kinds[cls] = part
    return kinds


def _reference_access(cls):
    text = _strip_comments(open(os.path.join(REF_INC, cls + ".h")).read())
    acc = {}
    for label, names in _members_by_section(_class_body(text, cls), "public|protected|private"):
        for nm in names:
            acc.setdefault(nm, set()).add(label)
    return acc


def _integration_edits():
    """Rows `| E<n> | file | `Class::member` ... |` of INTEGRATION.md's table of reference-side edits -> {(class, member)}."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rows = re.findall(r"^\|\s*E\d+\s*\|([^\n]*)$", text, flags=re.M)
    listed = set()
    for row in rows:
        for cls, member in re.findall(r"`(\w+)::(\w+)", row):
            if cls in CLASSES:
                listed.add((cls, member))
    return listed


def test_glue_compiles_under_the_references_access_rules():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DMOCK_STRICT_ACCESS", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "glue_access_check.cpp")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]


def test_strict_access_build_really_is_strict(tmp_path):
    """The same compile with edit E1 taken away must fail on the protected members -- otherwise the check above checks nothing."""
    src = open(os.path.join(ROOT, "tests", "cpp", "mock_orbslam3.hpp")).read()
    assert "float GetMinDistance() const" in src
    (tmp_path / "mock_orbslam3.hpp").write_text(src.replace("float GetMinDistance() const", "float GetMinDistanceRenamed() const"))
    (tmp_path / "glue_access_check.cpp").write_text(open(os.path.join(ROOT, "tests", "cpp", "glue_access_check.cpp")).read())
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DMOCK_STRICT_ACCESS", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
                        str(tmp_path / "glue_access_check.cpp")], capture_output=True, text=True, timeout=600, env=dict(os.environ, LC_ALL="C"))
    assert r.returncode != 0 and re.search(r"mfMinDistance\W+ is protected", r.stderr), r.stderr[-2000:]


def test_mock_sections_are_labelled_and_the_glue_never_names_instrumentation():
    part = _mock_partition()
    glue = _strip_comments(open(os.path.join(ROOT, "include", "orbgpu_dropin.hpp")).read())
    for cls, p in part.items():
        assert p["REFPUBLIC"], cls
        for nm in p["INSTR"]:
            assert not re.search(r"\b%s\b" % re.escape(nm), glue), "the glue names the mock's test instrumentation %s::%s" % (cls, nm)
    # every edit the mocks carry is a row of INTEGRATION.md's table, and the other way round
    listed = _integration_edits()
    carried = {(cls, nm) for cls, p in part.items() for nm in p["EDIT"]}
    assert carried == listed, "mock edit sections %s vs INTEGRATION.md's table %s" % (sorted(carried), sorted(listed))


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference is only present in the build container")
def test_mock_partition_matches_the_references_headers():
    part = _mock_partition()
    problems = []
    for cls, p in part.items():
        acc = _reference_access(cls)
        assert len(acc) > 30, (cls, len(acc))           # the parser saw the class
        for nm in p["REFPUBLIC"]:
            if "public" not in acc.get(nm, set()):
                problems.append("%s::%s is used as public; the reference declares it %s" % (cls, nm, sorted(acc.get(nm, {"nowhere"}))))
        for nm in p["REFPROTECTED"]:
            if not acc.get(nm) or "public" in acc[nm]:
                problems.append("%s::%s is marked protected in the mock; the reference declares it %s" % (cls, nm, sorted(acc.get(nm, {"nowhere"}))))
        for nm in p["EDIT"]:
            if nm in acc:
                problems.append("%s::%s is listed as a reference-side edit but the reference already declares it (%s)" % (cls, nm, sorted(acc[nm])))
    assert not problems, "\n".join(problems)


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference is only present in the build container")
def test_the_parser_sees_what_round_5_missed():
    acc = _reference_access("MapPoint")
    assert acc["mfMinDistance"] == {"protected"} and acc["mfMaxDistance"] == {"protected"}        # I/MapPoint.h:244,281-282
    assert acc["GetMinDistanceInvariance"] == {"public"} and acc["mnBALocalForKF"] == {"public"} and acc["mWorldPos"] == {"protected"}
    assert "GetMinDistance" not in acc and "mnChangeStamp" not in acc
    assert _reference_access("Map")["mMutexMapUpdate"] == {"public"} and _reference_access("Map")["mnMapChange"] == {"protected"}
