"""The generated constant tables: digests (always) and, in the build container,
value-for-value equality with the reference header read as text
(corintho_ai/cpp/include/util.h:85-702)."""
import json
import os
import re

import numpy as np
import pytest

from tools import gen_tables as gt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tables():
    return gt.build_all()


def test_digests_match_golden(tables):
    with open(os.path.join(ROOT, "tests/golden/tables.json")) as f:
        golden = json.load(f)
    assert gt.digests(*tables) == golden


def test_committed_inc_files_are_current(tables):
    text = gt.emit(*tables)
    for rel in ("corintho_ai_amd/csrc/tables.inc", "oracle/tables.inc"):
        with open(os.path.join(ROOT, rel)) as f:
            assert f.read() == text, rel + " is stale: run tools/gen_tables.py"


def test_symmetries_are_permutations_and_group(tables):
    _, _, sp, mv = tables
    for k in range(8):
        assert sorted(sp[k]) == list(range(16))
        assert sorted(mv[k]) == list(range(96))
    assert list(sp[0]) == list(range(16)) and list(mv[0]) == list(range(96))
    # closed under composition
    rows = {tuple(r) for r in sp}
    for a in sp:
        for b in sp:
            assert tuple(a[b]) in rows


def test_gamma_table_shape(tables):
    gm = tables[1]
    assert gm.dtype == np.float32 and gm.shape == (1024,)
    assert np.all(np.diff(gm) > 0)
    assert abs(float(gm.astype(np.float64).mean()) - 0.3) < 1e-6  # E[Gamma(0.3,1)] = 0.3


def _parse_ints(block):
    return [int(x) for x in re.findall(r"-?\d+", block)]


def test_line_breakers_equal_reference(tables, reference_util_h):
    src = reference_util_h
    s = src[src.index("line_breakers = {"):src.index("// Number of buckets")]
    strs = re.findall(r'bitset<kNumMoves>\(\s*((?:"[01]+"\s*)+)\)', s)
    assert len(strs) == 102
    for idx, t in enumerate(strs):
        bits = "".join(re.findall(r'"([01]+)"', t))
        assert len(bits) == 96
        m = 0
        for k, ch in enumerate(bits):  # bitset string: char k is bit 95-k
            if ch == "1":
                m |= 1 << (95 - k)
        assert m == tables[0][idx], "line %d" % idx


def test_gamma_samples_equal_reference(tables, reference_util_h):
    src = reference_util_h
    s = src[src.index("gamma_samples[kNumGammaBuckets] = {"):]
    s = s[s.index("{") + 1:s.index("};")]
    ref = np.array([float(x) for x in re.findall(r"[-+0-9.e]+", s)]).astype(np.float32)
    assert ref.shape == (1024,)
    assert np.array_equal(ref.view(np.uint32), tables[1].view(np.uint32))


def test_symmetries_equal_reference(tables, reference_util_h):
    src = reference_util_h
    s = src[src.index("space_symmetries[kNumSymmetries][kBoardSize] = {"):]
    a = s[s.index("{"):s.index("};")]
    sp = np.array(_parse_ints(a)).reshape(8, 16)
    s2 = src[src.index("move_symmetries[kNumSymmetries][kNumMoves] = {"):]
    b = s2[s2.index("{"):s2.index("};")]
    mv = np.array(_parse_ints(b)).reshape(8, 96)
    assert np.array_equal(sp, tables[2])
    assert np.array_equal(mv, tables[3])
