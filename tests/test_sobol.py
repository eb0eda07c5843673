"""Sobol generator: the only known-answer data the reference holds."""
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "sobol_kat.json")))
P5_FSH = "/root/reference/part 5 -- Importance Sampling & Low Discrepancy Sequence/source code/shaders/fshader.fsh"


def _table():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_sobol_table import direction_numbers
    return direction_numbers()


def _py_sobol(table, d, i):
    g = i ^ (i >> 1)
    r, j = 0, 0
    while g:
        if g & 1:
            r ^= table[d][j]
        g >>= 1
        j += 1
    return np.float32(r) * (np.float32(1.0) / np.float32(0xFFFFFFFF))


def test_oracle_matches_tutorial_known_answers(oracle):
    got = oracle.sobol(0, 30, 3)
    want = np.array(KAT["points"], np.float32)
    assert got.shape == want.shape
    assert np.array_equal(got, want)  # exact dyadics


def test_oracle_all_eight_dims_match_table_formula(oracle):
    table = _table()
    got = oracle.sobol(0, 300, 8)
    want = np.array([[_py_sobol(table, d, i) for d in range(8)] for i in range(300)], np.float32)
    assert np.array_equal(got, want)
    assert (got >= 0).all() and (got <= 1).all()


def test_generated_table_is_the_committed_include():
    inc = open(os.path.join(ROOT, "include", "ezrt_sobol_v.inc")).read()
    nums = [int(x) for x in re.findall(r"(\d+)u,", inc)]
    assert nums == [x for row in _table() for x in row]


@pytest.mark.skipif(not os.path.exists(P5_FSH), reason="reference not mounted")
def test_table_equals_the_shader_literal():
    src = open(P5_FSH).read()
    m = re.search(r"const uint V\[8\*32\] = \{\s*([0-9u,\s]+)\};", src)
    ref = [int(x.strip().rstrip("u")) for x in m.group(1).split(",") if x.strip()]
    assert ref == [x for row in _table() for x in row]


def test_joe_kuo_dims_are_stratified_quirk_dims_documented():
    """Dims 0-4 and 6 are true (0,m,1)-nets in base 2: the first 2^k points hit every 2^-k
    interval once.  Dims 5 and 7 carry the reference's corrupted rows and do not have to."""
    table = _table()
    for d in (0, 1, 2, 3, 4, 6):
        pts = np.array([_py_sobol(table, d, i) for i in range(64)])
        assert sorted((pts * 64).astype(int).tolist()) == list(range(64)), d


@pytest.mark.gpu
def test_gpu_sobol_kernel_matches_known_answers_and_oracle(hip, oracle):
    got = hip.sobol(0, 30, 3)
    assert np.array_equal(got, np.array(KAT["points"], np.float32))
    assert np.array_equal(hip.sobol(1, 4096, 8), oracle.sobol(1, 4096, 8))
