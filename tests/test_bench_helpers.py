"""bench.py's host-side helpers that need no GPU: the measured VALU issue ceiling is the best row over every recorded microbenchmark log
(profiles/r*/valu_issue_microbench.txt, tools/exp_valu_issue.hip), and the algorithmic-bytes formula is SURVEY 8(d)'s."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_the_measured_issue_ceiling_is_the_best_row_over_all_recorded_logs(tmp_path):
    """the ceiling used to be the latest log's best v_mov_b32 row, so a noisy, lower re-measurement (1.031 T in r5 after
    1.086 T in r2) raised the reported fraction with no kernel change.  Now: the maximum over every recorded log -- and the headline
    `roofline.frac` divides by the NOMINAL constant anyway."""
    import glob
    b = _bench()
    best, mix = 0.0, 0.0
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*", "valu_issue_microbench.txt")):
        for r in open(path).read().splitlines():
            if r.startswith("v_mov_b32"):
                best = max(best, float(r.split(" chip ")[1].split()[0]))
            if r.startswith("traceq4 opcode mix"):
                mix = max(mix, float(r.split(" chip ")[1].split()[0]))
    assert abs(best - b.VALU_ISSUE_PEAK_T) < 1e-9 and best >= 1.086
    assert abs(mix - b.VALU_ISSUE_PEAK_KERNEL_MIX_T) < 1e-9
    assert 0.9 < b.VALU_ISSUE_PEAK_T < b.VALU_ISSUE_PEAK_NOMINAL_T and abs(b.VALU_ISSUE_PEAK_NOMINAL_T - 1.2288) < 1e-9
    assert 0.5 < b.VALU_ISSUE_PEAK_KERNEL_MIX_T < b.VALU_ISSUE_PEAK_T
    # a later, LOWER log does not lower the ceiling; a later, higher one raises it
    for rnd, val in (("r7", 0.9), ("r8", 1.15)):
        os.makedirs(tmp_path / rnd)
        (tmp_path / rnd / "valu_issue_microbench.txt").write_text(
            "v_mov_b32              W=8  wall 0.8 ms  chip %.3f T wave-instr/s  per SIMD 1.0 G wave-instr/s\n" % val)
    os.makedirs(tmp_path / "r2")
    (tmp_path / "r2" / "valu_issue_microbench.txt").write_text(
        "v_mov_b32              W=8  wall 0.8 ms  chip 1.086 T wave-instr/s  per SIMD 1.0 G wave-instr/s\n")
    b._read_valu_peak(str(tmp_path))
    assert abs(b.VALU_ISSUE_PEAK_T - 1.15) < 1e-9 and "r8" in b.VALU_ISSUE_PEAK_SOURCE


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    c = {"node_pops": 10, "inner_pops": 7, "tri_tests": 5, "mat_fetch": 2, "samples": 3, "env_map": 4, "env_cache": 1}
    base = 48 * 10 + 96 * 7 + 72 * 5 + 72 * 2 + 32 * 3
    assert b.alg_bytes(c, bilinear=True) == base + 48 * 5      # four texels of 12 B per bilinear lookup
    assert b.alg_bytes(c, bilinear=False) == base + 12 * 5


def test_the_rccl_floor_file_the_scaling_model_reads_is_parseable():
    """profiles/r6/rccl_floor.json (tools/exp_rccl_floor.py; RCCL prints a version banner to stdout in front of the JSON line, which the
    first committed copy still carried -- and bench.py's reader then fell back to a floor of 0 silently): one JSON line with the three
    payloads, every floor a few microseconds."""
    import json
    lines = open(os.path.join(ROOT, "profiles", "r6", "rccl_floor.json")).read().splitlines()
    d = json.loads([ln for ln in lines if ln.startswith("{")][-1])
    assert set(d["payloads"]) == {"0.5MiB", "2MiB", "8MiB"}
    for v in d["payloads"].values():
        assert 0.001 < v["rccl_floor_ms"] < 0.1 and v["mgpu_rccl_loopback_gather_ms"]["median"] > v["rccl_floor_ms"]
