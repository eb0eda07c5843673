"""bench.py's host-side helpers that need no GPU: the VALU issue ceiling is read from the round's microbenchmark log
(profiles/r5/valu_issue_microbench.txt, tools/exp_valu_issue.hip) -- VERDICT r4 #8 -- and the algorithmic-bytes formula is SURVEY 8(d)'s."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_issue_ceiling_comes_from_the_rounds_microbenchmark_log():
    b = _bench()
    rows = open(os.path.join(ROOT, "profiles", "r5", "valu_issue_microbench.txt")).read()
    assert "v_mov_b32" in rows and "traceq4 opcode mix (r5)" in rows
    # the best v_mov_b32 row is the ceiling, the kernel's own opcode mix sits below it, both below the nominal 2-cycle rate
    assert b.VALU_ISSUE_PEAK_SOURCE.startswith("profiles/r5/valu_issue_microbench.txt")
    assert 0.9 < b.VALU_ISSUE_PEAK_T < b.VALU_ISSUE_PEAK_NOMINAL_T
    assert 0.5 < b.VALU_ISSUE_PEAK_KERNEL_MIX_T < b.VALU_ISSUE_PEAK_T
    best = max(float(r.split(" chip ")[1].split()[0]) for r in rows.splitlines() if r.startswith("v_mov_b32"))
    assert abs(best - b.VALU_ISSUE_PEAK_T) < 1e-9


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    c = {"node_pops": 10, "inner_pops": 7, "tri_tests": 5, "mat_fetch": 2, "samples": 3, "env_map": 4, "env_cache": 1}
    base = 48 * 10 + 96 * 7 + 72 * 5 + 72 * 2 + 32 * 3
    assert b.alg_bytes(c, bilinear=True) == base + 48 * 5      # four texels of 12 B per bilinear lookup
    assert b.alg_bytes(c, bilinear=False) == base + 12 * 5
