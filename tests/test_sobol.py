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


# ----------------------------------------------------------------------------- dimensions 8-15
def _table16():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_sobol_table import direction_numbers, direction_numbers_ext
    return direction_numbers() + direction_numbers_ext()


def test_extension_table_is_the_committed_include():
    inc = open(os.path.join(ROOT, "include", "ezrt_sobol_v16.inc")).read()
    nums = [int(x) for x in re.findall(r"(\d+)u,", inc)]
    assert nums == [x for row in _table16()[8:] for x in row] and len(nums) == 8 * 32


def test_extension_dims_are_nets_with_unit_triangular_generator_matrices():
    """Word k of a Joe-Kuo row is m_k << (31 - k) with m_k odd and < 2^(k+1): bit (31 - k) is set and no higher
    bit (the generator matrix is unit upper triangular), which is what makes every 2^m prefix a (0, m, 1)-net;
    checked on the bits and on the points."""
    table = _table16()
    for d in range(8, 16):
        for k, w in enumerate(table[d]):
            assert (w >> (31 - k)) & 1 == 1 and w % (1 << (31 - k)) == 0, (d, k)
        for m in (4, 6, 9):
            pts = np.array([_py_sobol(table, d, i) for i in range(1 << m)], np.float64)
            assert sorted((pts * (1 << m)).astype(int).tolist()) == list(range(1 << m)), (d, m)


def test_no_two_of_the_sixteen_rows_share_a_tail():
    """No pair of rows may agree in more than a few (early, small-m) words -- two dimensions with a common tail
    would be the same sequence from some sample index on."""
    table = _table16()
    for i in range(16):
        for j in range(i + 1, 16):
            same = sum(1 for k in range(2, 32) if table[i][k] == table[j][k])
            assert same <= 3, (i, j, same)


def test_oracle_sixteen_dims_match_table_formula(oracle):
    table = _table16()
    got = oracle.sobol(0, 300, 16)
    want = np.array([[_py_sobol(table, d, i) for d in range(16)] for i in range(300)], np.float32)
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, :8], oracle.sobol(0, 300, 8))


@pytest.mark.gpu
def test_gpu_sobol_sixteen_dims_match_oracle(hip, oracle):
    assert np.array_equal(hip.sobol(1, 4096, 16), oracle.sobol(1, 4096, 16))
