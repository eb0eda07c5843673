#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: Mrays/s at fixed spp.

N = 1 (the headline line): workload C2 = BASELINE.json configs[1] made concrete as SURVEY.md 8(d): the P3 scene with
the Stanford Bunny subdivided to ~70k triangles (79 820 total), SAH BVH leaf 8, 512x512, reference camera (r = 4),
integrator 50 (P5 pathTracing: Sobol + Cranley-Patterson hemisphere sampling, Disney BRDF), 4 bounces, 64 spp, env
map = the reference's only shipped HDR (P4/HDR/peppermint_powerplant_4k.hdr, 1024x512, bilinear).  One "step" = one
full render of that frame (64 spp) through libezrt_hip.so; scene, env map and the frame buffer are resident in HBM
before the timed region.  ray := one hitBVH call, counted by the kernels.

The N = 1 line also carries (round 4):
  * `parity`: the frame the TIMED loop left (cloned right after the windows) compared on the bits with the frame of the
    instrumented route (binary unpruned in-order kernel) and, in `cpu_baseline`, with the CPU oracle's complete frame;
  * `configs`: BASELINE.json configs[2..4] (C3 / C4 / C5) timed at THEIR stated spp (128 / 256 / 512) -- one warm-up +
    3 calls each -- with a crop of the timed frame checked on the bits against the oracle at the full spp;
  * `scaling_model`: C2 and C4 rendered as shard r of N in {2, 4, 8} ALONE on this GPU (max over r = the critical
    path of an N-GPU split) + pack + payload / 153 GB/s + un-permute = a predicted strong-scaling curve;
  * `cpu_baseline.reference_shader_one_thread_Mrays_s`: the reference's own fshader.fsh (compiled by g++ into
    oracle/_ref) timed on a crop of the same frames, next to the port's one-thread figure.

N > 1 (one process per GPU, torchrun): the SAME workload, so that the per-N values of a scaling run are comparable:
the C2 frame is cut into 16x16 tiles dealt round-robin to the ranks, the scene is replicated, every rank traces all
spp of its tiles and ONE gather of the packed tiles to rank 0 over RCCL closes the frame.  Default "scaling": "strong"
(round 4: the metric reads "at fixed spp" -- the fixed 64-spp frame is split N ways; a weak run, spp x N, is
near-linear by construction and is reported as the extra field `weak_variant`); `--scaling weak` makes the weak run
the headline instead.  Extra fields report per-rank render times, the gather time, the imbalance, the same frame
rendered by rank 0 alone with a bitwise comparison (so the line carries its own 1-GPU reference), and BASELINE.json
configs[3] (C4: P5 scene, integrator 51 = env importance sampling + MIS, 2 bounces, 1024x1024, 256 spp) as a FIXED
frame split N ways (`c4_strong_variant`).  `--workload c2|c4` picks the main workload for any N.

Timing (VERDICT r2 / SURVEY 8(d) "median of >= 5 runs"): after W warm-up steps the script times `--windows` (default 7)
windows of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and reduced with MAX over ranks;
`value` and `ms_per_step` come from the MEDIAN window, every window is in the line (`timing.window_ms`), and so are the
per-step GPU times of the slowest window (stream events, no host synchronisation inside a window), the host-side
enqueue time per step and `wall_over_gpu`, so that a host-side stall or a clock ramp is visible instead of averaged in.

Prints one JSON line on rank 0 (contract in the task statement) with `roofline` and, at N = 1, `cpu_baseline`
(the CPU oracle timed on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's device-memory sharing across the ranks of a
# node fails with `hipIpcGetMemHandle: invalid argument` (already exported on the boxes; kept for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
LDS_PEAK_GBS = 150000.0      # MI355X_MICROARCH.md "LDS": ~150 TB/s for ds_read_b64/b128 with every CU streaming
# VALU issue ceiling: tools/exp_valu_issue.hip on this chip: wave64 fp32 add/mul/mov and the slab test's own opcode mix issue at
# 1.0-1.09 T wave-instructions/s chip-wide (~2 cycles per instruction per SIMD at the ~2.1-2.4 GHz the chip sustains),
# fma/min3/cndmask alone at 0.58 T.  Read from the round's microbenchmark log (no constant two rounds old): the best
# row of the fastest opcode (v_mov_b32) = the ceiling `frac` divides by, as in every round; the rate of the timed kernel's own
# static opcode mix is printed beside it (`peak_kernel_opcode_mix`).  The constants are the fall-back when the log is missing.
VALU_ISSUE_PEAK_T = 1.086
VALU_ISSUE_PEAK_KERNEL_MIX_T = None
VALU_ISSUE_PEAK_SOURCE = "constant (profiles/r2/valu_issue_microbench.txt)"


def _read_valu_peak(profiles_dir=None):
    """The MEASURED issue ceiling = the best v_mov_b32 row over ALL recorded microbenchmark logs (profiles/r*/valu_issue_microbench.txt).
    ADVICE r5: reading only the latest log let a noisy, lower re-measurement (1.031 T in r5 vs 1.086 T in r2, same chip model) inflate
    the fraction with no kernel change.  Since round 6 `roofline.frac` does not use it at all -- it divides by the NOMINAL figure below,
    a constant -- and this number is printed beside it as `peak_measured`."""
    global VALU_ISSUE_PEAK_T, VALU_ISSUE_PEAK_KERNEL_MIX_T, VALU_ISSUE_PEAK_SOURCE
    import glob
    import re
    base = profiles_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    best, src = {}, {}
    for path in sorted(glob.glob(os.path.join(base, "r*", "valu_issue_microbench.txt"))):
        try:
            rows = open(path).read().splitlines()
        except OSError:
            continue
        for r in rows:
            m = re.match(r"(.+?)\s+W=(\d+)\s+wall .*? chip ([0-9.]+) T wave-instr/s", r)
            if m and float(m.group(3)) > best.get(m.group(1).strip(), 0.0):
                best[m.group(1).strip()] = float(m.group(3))
                src[m.group(1).strip()] = os.path.relpath(path, os.path.dirname(base))
    if "v_mov_b32" in best:
        VALU_ISSUE_PEAK_T = best["v_mov_b32"]
        VALU_ISSUE_PEAK_SOURCE = "%s (tools/exp_valu_issue.hip: best v_mov_b32 row over all recorded logs)" % src["v_mov_b32"]
    mix = [v for k, v in best.items() if k.startswith("traceq4 opcode mix")]
    VALU_ISSUE_PEAK_KERNEL_MIX_T = max(mix) if mix else None


_read_valu_peak()
# ... and the guide's NOMINAL figure for the same ceiling: one wave64 fp32 VALU instruction per 2 cycles per SIMD x 4 SIMDs x
# 256 CUs x 2.4 GHz (MI355X_MICROARCH.md "Chip-level parameters") = 1.229 T wave-instructions/s.  Both are printed.
VALU_ISSUE_PEAK_NOMINAL_T = 256 * 4 * 2.4e9 / 2 / 1e12
XGMI_LINK_GBS = 153.0        # MI355X_MICROARCH.md: one xGMI link, per direction (each peer's payload crosses its own link to rank 0)


def alg_bytes(c, bilinear=True):
    """bytes(ray) = 48 P + 96 I + 72 T + 72 M in the reference's record sizes,
    + 32 B per pixel-sample (lastFrame read + write) + 48/12 B per env-map / cache lookup."""
    tex = 48 if bilinear else 12
    return (48 * c["node_pops"] + 96 * c["inner_pops"] + 72 * c["tri_tests"] + 72 * c["mat_fetch"]
            + 32 * c["samples"] + tex * (c["env_map"] + c["env_cache"]))


def host_cores():
    cores = os.cpu_count() or 1
    try:  # a container's CPU quota (cgroup v2) is what the oracle really gets
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def physical_roofline(trace_ms_per_step, launches_per_step):
    """Counters of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r2/pmc_summary.json, made by
    tools/profile.sh + tools/summarize_profile.py), combined with THIS run's launch durations.  The profile is stamped
    with a hash of the GPU sources; a stale stamp means the counters describe other code and nothing is quoted."""
    from ezrt_amd.srchash import gpu_source_hash
    sha = gpu_source_hash()
    pm, path, seen = None, None, []
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        cand = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        if os.path.exists(cand):
            seen.append("profiles/%s/pmc_summary.json" % rnd)
            got = json.load(open(cand))
            if got.get("source_sha") == sha:
                pm, path = got, "profiles/%s/pmc_summary.json" % rnd
                break
    if pm is None:
        return None, ("no pmc_summary.json under profiles/" if not seen else
                      "%s: stale (GPU sources changed since they were collected: %s)" % (", ".join(seen), sha))
    k = pm["dominant"]                      # per STEP sums over the dominant kernel's launches
    secs = trace_ms_per_step * 1e-3
    insts = k["SQ_INSTS_VALU"]
    out = {
        "kernel_instances": k["names"],
        "valu_wave_instr_per_step": int(insts),
        "issue_rate_T": round(insts / secs / 1e12, 4),
        "issue_frac": round(insts / secs / 1e12 / VALU_ISSUE_PEAK_T, 4),
        "issue_frac_of_nominal_peak": round(insts / secs / 1e12 / VALU_ISSUE_PEAK_NOMINAL_T, 4),
        "lane_fill": round(k["SQ_THREAD_CYCLES_VALU"] / (64.0 * insts), 4),
        "salu_per_valu": round(k["SQ_INSTS_SALU"] / insts, 3),
        "wave_cycles_waiting": round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 3),
        "lds_conflict_frac": round(k["SQ_LDS_BANK_CONFLICT"] / max(1.0, k["SQ_ACTIVE_INST_LDS"]), 3),
        "hbm_bytes_per_step": int(k["hbm_bytes"]),
        "hbm_frac": round(k["hbm_bytes"] / secs / 1e9 / HBM_PEAK_GBS, 4),
        "l2_bytes_per_step": int(k["l2_bytes"]),
        "l2_frac": round(k["l2_bytes"] / secs / 1e9 / L2_PEAK_GBS, 4),
        "lds_bytes_per_step": int(k["lds_bytes"]),
        "lds_frac": round(k["lds_bytes"] / secs / 1e9 / LDS_PEAK_GBS, 4),
        "source": "%s @ %s" % (path, pm["source_sha"]),
    }
    out["traffic_per_launch"] = int(k["hbm_bytes"] / max(1, launches_per_step))
    # where the wave-cycles go (MI355X_MICROARCH.md "SQ": WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint): parked on an
    # s_waitcnt (the trace loop has no s_barrier), stalled at issue (of which: the LDS pipe), issuing
    if k.get("SQ_WAVE_CYCLES"):
        wc = k["SQ_WAVE_CYCLES"]
        out["stall_split"] = {"parked_on_s_waitcnt": round(k.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                              "stalled_at_issue": round(k.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                              "stalled_at_issue_lds_pipe": round(k["SQ_WAIT_INST_LDS"] / wc, 4) if "SQ_WAIT_INST_LDS" in k else None,
                              "issuing": round(k.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                              "issuing_valu": round(k.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                              "issuing_lds": round(k.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4),
                              "vmem_read_instr_per_valu": round(k.get("SQ_INSTS_VMEM_RD", 0.0) / insts, 4),
                              "lds_instr_per_valu": round(k.get("SQ_INSTS_LDS", 0.0) / insts, 4),
                              "note": "fractions of SQ_WAVE_CYCLES, summed over the dominant kernel's launches of a step; the loop has no s_barrier, "
                                      "so 'parked' is s_waitcnt (vmcnt / lgkmcnt drains of the record, triangle and LDS-stack loads)"}
    if pm.get("all_kernels"):
        out["all_kernels_valu_wave_instr_per_step"] = int(pm["all_kernels"]["valu_wave_instr_per_step"])
    return out, None


# crops of the timed full-spp frames that the CPU oracle re-renders (x0, y0, x1, y1).  Round 5 ("the driver-line crop
# checks are small"): sixteen times the pixels of round 4's crops, around the same centres -- C3 256x192 over the sphere grid (round 4: the
# 64x48 of tests/test_gpu_configs.py), C4 256x192 over the Bunny, its mirror floor and sky, C5 128x96 across spheres, floor tiles and the
# horizon -- still a few seconds of oracle time each at the full 128 / 256 / 512 spp on the box's 16 cores.
CONFIG_CROPS = {"C3": (384, 308, 640, 500), "C4": (352, 216, 608, 408), "C5": (976, 1092, 1104, 1188)}


def load_oracle():
    """The CPU oracle (test infrastructure): only the checker / cpu_baseline legs of this script call it."""
    import ctypes
    from ezrt_amd import _abi, trace
    return trace.TraceLib(_abi.declare_trace_abi(ctypes.CDLL(os.path.join(ROOT, "oracle", "libezrt_oracle.so"))))


def same_bits_nan_ok(a, b):
    """bit equality, except that a NaN equals a NaN (include/ezrt.h: sign and payload of a NaN are not part of the contract)"""
    import numpy as np
    ua, ub = np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32)
    return bool(((ua == ub) | (np.isnan(a) & np.isnan(b))).all())


def build_config_scene(name, env):
    from ezrt_amd import scenes
    if name == "C3":
        return scenes.disney_grid_scene(subdiv=3, hdr=env)
    if name == "C4":
        return scenes.p5_scene(subdiv=2, hdr=env)
    if name == "C5":
        return scenes.mega_scene(hdr=env)
    raise ValueError(name)


def time_config(name, hip, torch, dev, stream, env, spp_override=0, check=True, calls=3):
    """One BASELINE config at its stated resolution / bounces / spp on this GPU: scene build + create timed separately, one
    warm-up call + `calls` timed calls (each bracketed by synchronize; median), rays counted by the kernels, the trace
    launches' share from one more call with launch events, and a crop of the frame the TIMED calls left compared on the
    bits with the CPU oracle rendering that crop at the same spp."""
    import numpy as np
    from ezrt_amd import scene as S, scenes, trace
    cfg = dict(scenes.CONFIGS[name])
    if spp_override:
        cfg["spp"] = spp_override
    t0 = time.perf_counter()
    bs = build_config_scene(name, env)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    sc = bs.upload(hip)
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    eye, cam = S.camera(*cfg["camera"])
    W, H = cfg["width"], cfg["height"]
    p = trace.make_params(W, H, eye, cam, cfg["integrator"], cfg["max_bounce"], spp=cfg["spp"], tile=(16, 16))
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    sc.render_device(p, acc.data_ptr(), stream)          # warm-up (allocates the chunk scratch)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    sc.counters_reset()
    ms = []
    for _ in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.render_device(p, acc.data_ptr(), stream)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    rays = sc.counters()["rays"] // calls
    frame = acc.clone()                                   # what the timed calls left
    # the trace launches' share: one more call with a timing-event pair around every trace launch, its chunks NOT overlapped
    # (pipeline_calls 0 for this call: under overlap an event interval includes the launch's wait for wave slots)
    sc.set_option("launch_events", 1)
    sc.set_option("pipeline_calls", 0)
    sc.render_device(p, acc.data_ptr(), stream)
    torch.cuda.synchronize()
    total_ms, trace_ms, n_launch = sc.last_render_ms()
    sc.set_option("launch_events", 0)
    sc.set_option("pipeline_calls", 1)
    med = statistics.median(ms)
    out = {"workload": "%s: %d tris, %d BVH nodes, %dx%d, integrator %d, %d bounces, %d spp" %
                       (name, bs.tri.shape[0], bs.nodes.shape[0], W, H, cfg["integrator"], cfg["max_bounce"], cfg["spp"]),
           "Mrays_s": round(rays / (med * 1e-3) / 1e6, 2), "ms_per_frame": round(med, 3), "ms_per_frame_calls": [round(x, 3) for x in ms],
           "rays": int(rays), "trace_ms": round(trace_ms, 3), "trace_launches": int(n_launch), "gpu_ms_unpipelined_with_launch_events": round(total_ms, 3),
           "scene_build_s": round(t_build, 3), "scene_build": bs.build_stats if isinstance(bs.build_stats, dict) else None,
           "scene_create_s": round(t_create, 3), "first_call_s": round(t_first, 3),
           "non_finite_pixels": int((~torch.isfinite(frame[..., :3]).all(dim=2)).sum()),
           "non_finite_pixels_ezrt_frame_nonfinite": hip.frame_nonfinite(frame.data_ptr(), W, H, stream),
           "roofline": config_roofline(name)}
    if check:
        x0, y0, x1, y1 = CONFIG_CROPS[name]
        ora = load_oracle()
        so = bs.upload(ora)
        t0 = time.perf_counter()
        want = so.render(trace.make_params(W, H, eye, cam, cfg["integrator"], cfg["max_bounce"], spp=cfg["spp"], rect=(x0, y0, x1, y1)))
        got = frame[y0:y1, x0:x1].cpu().numpy()
        out["crop_vs_oracle"] = {"rect": [x0, y0, x1, y1], "spp": cfg["spp"], "bit_identical": same_bits_nan_ok(got, want[y0:y1, x0:x1]),
                                 "what": "the frame the timed calls left vs the CPU oracle on this crop at the same spp (NaN == NaN)",
                                 "crop_max": float(np.nanmax(want[y0:y1, x0:x1, :3])), "oracle_s": round(time.perf_counter() - t0, 2)}
        so.close()
    sc.close()
    return out


def config_roofline(name):
    """The dominant kernel of a BASELINE config and what bounds it, from the committed rocprofv3 summary of that config
    (profiles/rN/<cfg>_pmc_summary.json: tools/profile_configs.sh + tools/summarize_config_profile.py; kernels profiled ALONE at the
    BASELINE spp, counters in separate --pmc passes).  Every fraction has a fixed denominator: VALU issue / 1.2288 T wave-instr/s
    (nominal), HBM (2 FETCH + WRITE) / 8 TB/s.  A stale stamp (other GPU sources) means nothing is quoted."""
    from ezrt_amd.srchash import gpu_source_hash
    sha = gpu_source_hash()
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        cand = os.path.join(ROOT, "profiles", rnd, "%s_pmc_summary.json" % name.lower())
        if not os.path.exists(cand):
            continue
        pm = json.load(open(cand))
        if pm.get("source_sha") != sha:
            return {"note_profile": "profiles/%s/%s_pmc_summary.json: stale (GPU sources changed since it was collected: %s)" % (rnd, name.lower(), sha)}
        ks = {k: v for k, v in pm["kernels"].items() if v.get("share_of_gpu_time") and "issue_rate_T" in v}
        if not ks:
            return None
        k = max(ks, key=lambda n: ks[n]["share_of_gpu_time"])
        v = ks[k]
        fr = {"valu_issue": round(v["issue_rate_T"] / VALU_ISSUE_PEAK_NOMINAL_T, 4), "hbm": v.get("hbm_frac_of_8TBs")}
        bound = max((b for b in fr if fr[b] is not None), key=lambda b: fr[b])
        out = {"kernel": "ezd::" + k, "share_of_gpu_time": v["share_of_gpu_time"], "launch_us": v["avg_us"], "bound": bound, "frac": fr[bound],
               "achieved": v["issue_rate_T"] if bound == "valu_issue" else v.get("hbm_GBs"),
               "peak": round(VALU_ISSUE_PEAK_NOMINAL_T, 4) if bound == "valu_issue" else HBM_PEAK_GBS,
               "unit": "T wave-instr/s" if bound == "valu_issue" else "GB/s", "ceilings": fr,
               "lane_fill": v.get("lane_fill"), "wave_cycles_waiting": v.get("wave_cycles_waiting"), "stall_split": v.get("stall_split"),
               "traffic": v.get("hbm_bytes_per_dispatch_in_counter_pass"),
               "hbm_bytes_per_pixel_sample": (pm.get("hbm_bytes_per_pixel_sample") or {}).get("by_kernel", {}).get(k),
               "hbm_bytes_per_pixel_sample_all_kernels": (pm.get("hbm_bytes_per_pixel_sample") or {}).get("all_kernels"),
               "source": "profiles/%s/%s_pmc_summary.json @ %s (kernels profiled alone: EZRT_PIPELINE_CALLS=0)" % (rnd, name.lower(), pm["source_sha"])}
        return out
    return None


def scaling_model(tag, sc, hip, torch, dev, stream, make_p, W, H, tile, shards=(2, 4, 8), reps=3):
    """What an N-GPU strong split of this frame would cost, measured on ONE GPU: shard r of N rendered alone (the tiles rank r
    would own; median of `reps` bursts of 4 back-to-back calls), max over r = the critical path of the render phase; + the pack kernel, the payload
    over one xGMI link (each peer has its own link to rank 0), and the N - 1 un-permute kernels on rank 0.  + RCCL's
    launch / protocol floor as far as ONE GPU can measure it (profiles/r6/rccl_floor.json, tools/exp_rccl_floor.py: a grouped
    ncclSend / ncclRecv of the payload to the device itself, pack and un-permute kernels subtracted: 10-15 us; a floor, not the cost of an
    exchange between distinct devices)."""
    from ezrt_amd import tiles
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    floor_ms, floor_src = 0.0, None
    try:
        fl = json.loads([ln for ln in open(os.path.join(ROOT, "profiles", "r6", "rccl_floor.json")).read().splitlines() if ln.startswith("{")][-1])["payloads"]
        floor_ms = max(v["rccl_floor_ms"] for v in fl.values())
        floor_src = "profiles/r6/rccl_floor.json (max over 0.5 / 2 / 8 MiB payloads of loop-back gather - pack - un-permute)"
    except (OSError, KeyError, ValueError):
        pass

    def timed(fn, n, burst=1):
        """median over n measurements of `burst` back-to-back calls (ms per call).  Render calls are measured in bursts of 4: the
        steps of a timed window are queued back to back too, and consecutive calls overlap (pipeline_calls) -- a lone call
        between two synchronisations would overstate what a shard costs in steady state."""
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            for _ in range(burst):
                fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3 / burst)
        return statistics.median(ts)

    t1 = timed(lambda: sc.render_device(make_p((0, 1)), acc.data_ptr(), stream), reps, 4)
    out = {"workload": tag, "one_gpu_ms": round(t1, 4), "shards": {}}
    for n in shards:
        per = [timed(lambda r=r: sc.render_device(make_p((r, n)), acc.data_ptr(), stream), reps, 4) for r in range(n)]
        plan = tiles.TilePlan(W, H, tile, tile, n)
        nfl = plan.per_rank * tile * tile * 4
        packed = torch.zeros(nfl, dtype=torch.float32, device=dev)
        a = (W, H, tile, tile)
        pack = timed(lambda: hip.lib.ezrt_tiles_pack_device(acc.data_ptr(), *a, 1, n, packed.data_ptr(), stream), 5)
        unpack = timed(lambda: hip.lib.ezrt_tiles_unpack_device(packed.data_ptr(), *a, 1, n, acc.data_ptr(), stream), 5)
        wire = nfl * 4 / (XGMI_LINK_GBS * 1e9) * 1e3
        crit = max(per)
        total = crit + pack + wire + floor_ms + (n - 1) * unpack
        out["shards"][str(n)] = {"render_ms_per_shard": [round(x, 4) for x in per], "critical_path_ms": round(crit, 4),
                                 "imbalance_max_over_mean": round(crit / (sum(per) / n), 4),
                                 "pack_ms": round(pack, 4), "wire_ms_at_153GBs": round(wire, 4), "unpack_ms_each": round(unpack, 4),
                                 "payload_bytes_per_peer": nfl * 4, "rccl_floor_ms": round(floor_ms, 4), "predicted_ms": round(total, 4),
                                 "predicted_speedup": round(t1 / total, 3), "render_only_speedup": round(t1 / crit, 3)}
    out["rccl_floor_source"] = floor_src
    out["note"] = ("pack / un-permute figures are host-synchronised single launches (they include ~10-20 us of launch + sync overhead each, "
                   "i.e. pessimistic); rccl_floor_ms = what one GPU can measure of RCCL's launch + protocol latency (a loop-back grouped "
                   "send/receive); an exchange between distinct devices over xGMI has never run here and will cost more")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("auto", "c2", "c4"), default="auto", help="auto = c2 for every N (the fixed C2 frame, split N ways at N > 1); c4: BASELINE's C4 instead")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=-1)
    ap.add_argument("--integrator", type=int, default=0)
    ap.add_argument("--subdiv", type=int, default=2)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--env", choices=("shipped", "synthetic"), default="shipped")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1 -- strong (default: the metric reads 'at fixed spp'): the fixed frame is split N ways; "
                         "weak: spp x N, so every GPU keeps the 1-GPU number of pixel-samples")
    ap.add_argument("--configs", default="C3,C4,C5",
                    help="N = 1: BASELINE configs timed after the headline at their stated spp (comma list of C3,C4,C5; 'none' skips)")
    ap.add_argument("--config-spp", type=int, default=0, help="override the spp of --configs (tests; 0 = BASELINE's 128 / 256 / 512)")
    ap.add_argument("--model", type=int, default=1, help="N = 1: 0 skips the single-GPU scaling model")
    ap.add_argument("--windows", type=int, default=7, help="timed windows of --steps steps each; value = the median window")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample time (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0: skip the extra fields (second camera, 1-GPU reference, weak variant)")
    ap.add_argument("--save-png", default="")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace has no CPU fallback")
    # EZRT_BENCH_BACKEND=gloo is a debugging aid for boxes with fewer GPUs than ranks: the ranks share
    # GPU 0 and the gather is staged through host memory.  The driver's runs use nccl (= RCCL).
    backend = os.environ.get("EZRT_BENCH_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ezrt_amd import scene as S, scenes, tiles, trace
    hip = trace.hip()  # after torch: shares torch's HIP runtime (same soname)

    wl = args.workload if args.workload != "auto" else "c2"
    cfg = dict(scenes.CONFIGS["C2" if wl == "c2" else "C4"])
    for k_arg, k_cfg in (("width", "width"), ("height", "height"), ("spp", "spp"), ("integrator", "integrator")):
        if getattr(args, k_arg):
            cfg[k_cfg] = getattr(args, k_arg)
    if args.bounces >= 0:
        cfg["max_bounce"] = args.bounces
    t_build = time.perf_counter()
    if wl == "c2":
        bs = scenes.bunny_scene(subdiv=args.subdiv, want_cache=(cfg["integrator"] == 51), hdr=args.env)
        desc = "C2: P3 scene, Stanford Bunny subdivided x%d" % args.subdiv
    else:
        bs = scenes.p5_scene(subdiv=args.subdiv, hdr=args.env)
        desc = "C4: P5 scene (Bunny subdivided x%d with the teapot's material, mirror floor)" % args.subdiv
    t_build = time.perf_counter() - t_build
    sc = bs.upload(hip)
    eye, cam = S.camera(*cfg["camera"])
    W, H, integ, mb = cfg["width"], cfg["height"], cfg["integrator"], cfg["max_bounce"]
    weak = world > 1 and args.scaling == "weak"
    spp = cfg["spp"] * world if weak else cfg["spp"]
    env_name = ("the reference's shipped P4/HDR/peppermint_powerplant_4k.hdr 1024x512 (asset copy), bilinear"
                if args.env == "shipped" else "procedural 1024x512 env, bilinear")

    def params(shard, n_spp=None, camera=None):
        e, c = (eye, cam) if camera is None else S.camera(*camera)
        return trace.make_params(W, H, e, c, integ, mb, spp=spp if n_spp is None else n_spp, tile=(args.tile, args.tile),
                                 shard=shard)

    p = params((rank, world))
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    plan = tiles.TilePlan(W, H, args.tile, args.tile, world) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream
    tdev = dev if backend == "nccl" else "cpu"

    gather_ev = []

    def make_step(sc_, p_, accum_, plan_, log_gather=None):
        def step():
            sc_.render_device(p_, accum_.data_ptr(), stream)
            if world > 1:
                if rank == 0 and log_gather is not None:   # (events on the launch stream, read after the window: no host sync inside it)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                out_img = tiles.gather_frame(accum_, plan_, rank, dist, via_cpu=(backend != "nccl"), lib=hip.lib)
                if rank == 0 and log_gather is not None:
                    e1.record()
                    log_gather.append((e0, e1))   # rank 0's render done -> frame assembled: includes waiting for the slowest rank
                return out_img
            return accum_
        return step

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_windows(step, n_windows, n_steps):
        """n_windows windows of exactly n_steps steps: barrier + synchronize on both sides of each, wall clock per window
        (MAX over ranks), one stream event per step (GPU time between step starts; read after the window) and the host's
        enqueue time per step.  Returns (window seconds after MAX over ranks, per-window per-step GPU ms, per-window
        per-step host enqueue ms, the last frame)."""
        wall, gpu_ms, host_ms, last = [], [], [], None
        for _ in range(n_windows):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
            enq = []
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n_steps):
                evs[k].record()
                h0 = time.perf_counter()
                last = step()
                enq.append((time.perf_counter() - h0) * 1e3)
            evs[n_steps].record()
            torch.cuda.synchronize()
            barrier()
            wall.append(time.perf_counter() - t0)
            gpu_ms.append([evs[k].elapsed_time(evs[k + 1]) for k in range(n_steps)])
            host_ms.append(enq)
        wt = torch.tensor(wall, dtype=torch.float64, device=tdev)
        if world > 1:
            dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        return [float(x) for x in wt], gpu_ms, host_ms, last

    step = make_step(sc, p, accum, plan, gather_ev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sc.counters_reset()
    gather_ev.clear()
    n_windows = max(1, args.windows)
    win_s, win_gpu_ms, win_host_ms, final = timed_windows(step, n_windows, args.steps)
    # the frame the TIMED loop left, before anything else renders: every later pass (launch events, the instrumented route,
    # the extras) writes into `scratch`, so `final` is what the parity fields below compare
    final = final.clone()
    scratch = torch.zeros_like(accum)
    gather_s = [a.elapsed_time(b) * 1e-3 for a, b in gather_ev]
    rays_timed = sc.counters()["rays"]          # over all windows (the workload is deterministic: the same rays every step)
    # per-step GPU times with a timing-event pair around every trace launch (hipEvents on the launch stream; reading them
    # synchronises and each record costs the stream ~5 us, so this is a separate, untimed pass of the same step):
    # median of 7 -- (all kernels, the trace launches, number of trace launches)
    step_ms = []
    sc.set_option("launch_events", 1)
    for _ in range(7):
        sc.render_device(p, scratch.data_ptr(), stream)
        step_ms.append(sc.last_render_ms())
    sc.set_option("launch_events", 0)
    # ... and ONE call at a time without any per-launch event: a call between two synchronisations, nothing to overlap with
    # (the headline windows queue their steps back to back, so a chunk's late stages run under the next chunk's primary stage:
    # `value` is that steady-state rate, `value_lone_call` the rate of a host that waits for every frame; VERDICT r4 #5c)
    lone_ms = []
    e_l0, e_l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(9):
        torch.cuda.synchronize()
        e_l0.record()
        sc.render_device(p, scratch.data_ptr(), stream)
        e_l1.record()
        torch.cuda.synchronize()
        lone_ms.append(e_l0.elapsed_time(e_l1))
    lone_ms_med = statistics.median(lone_ms)

    tt = torch.tensor([float(rays_timed)], dtype=torch.float64, device=tdev)
    rank_ms = torch.tensor([statistics.median(m[0] for m in step_ms)], dtype=torch.float64, device=tdev)
    all_rank_ms = [float(rank_ms[0])]
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        bufs = [torch.zeros_like(rank_ms) for _ in range(world)]
        dist.all_gather(bufs, rank_ms)
        all_rank_ms = [float(b[0]) for b in bufs]
    rays_per_step = float(tt[0]) / max(1, args.steps * n_windows)
    order = sorted(range(n_windows), key=lambda i: win_s[i])
    i_med, i_slow = order[(n_windows - 1) // 2], order[-1]      # (lower median for an even count)
    elapsed = win_s[i_med]                                      # the median window: K steps, MAX over ranks
    value = rays_per_step * args.steps / elapsed / 1e6
    ms_per_step = elapsed / max(1, args.steps) * 1e3
    gpu_step_med = statistics.median(x for w in win_gpu_ms for x in w)   # this rank's stream events, all windows
    timing = {
        "protocol": "%d warm-up steps, then %d windows of exactly %d steps, barrier + synchronize around each, MAX over ranks; "
                    "value = median window" % (args.warmup, n_windows, args.steps),
        "window_ms": [round(x * 1e3, 3) for x in win_s],
        "window_ms_min_median_max": [round(win_s[order[0]] * 1e3, 3), round(elapsed * 1e3, 3), round(win_s[i_slow] * 1e3, 3)],
        "first_window_Mrays_s": round(rays_per_step * args.steps / win_s[0] / 1e6, 2),
        "gpu_ms_per_step_median_in_windows": round(gpu_step_med, 4),
        "wall_over_gpu": round(ms_per_step / gpu_step_med, 4) if gpu_step_med > 0 else None,
        "slowest_window": {"index": i_slow, "gpu_ms_per_step": [round(x, 3) for x in win_gpu_ms[i_slow]],
                           "host_enqueue_ms_per_step": [round(x, 3) for x in win_host_ms[i_slow]]},
        "first_window_gpu_ms_per_step": [round(x, 3) for x in win_gpu_ms[0]],
    }
    if timing["wall_over_gpu"] and timing["wall_over_gpu"] > 1.1:
        timing["flag"] = "wall time of the median window is more than 1.1x the GPU time between step starts: host-side stall"

    bounce_txt = "%d bounces" % mb
    out = {
        "metric": "Mrays/s at fixed spp (Bunny ~70k tris, %s)" % bounce_txt,
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "value_lone_call": round(rays_per_step / world / (lone_ms_med * 1e-3) / 1e6, 3) if world == 1 else None,
        "ms_per_step_lone_call": round(lone_ms_med, 4),
        "value_note": "value = steps queued back to back (chunks of consecutive calls overlap: pipeline_calls); value_lone_call = one call "
                      "between two synchronisations, no per-launch events (this rank's shard at N > 1: ms only)",
        "scaling": "n/a" if world == 1 else ("weak" if weak else "strong"),
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s (%d tris, %d BVH nodes, SAH leaf 8), %dx%d, camera rot %g up %g r %g, integrator %d, %d bounces, "
                               "%d spp%s, env = %s"
                               % (desc, bs.tri.shape[0], bs.nodes.shape[0], W, H, cfg["camera"][0], cfg["camera"][1], cfg["camera"][2],
                                  integ, mb, spp, (" (= %d spp x %d GPUs: per-GPU pixel-samples fixed)" % (cfg["spp"], world)) if weak else "",
                                  env_name),
                   "rays_per_step": int(rays_per_step), "ray_definition": "one hitBVH call",
                   "parallelism": "tiles%dx%d round-robin over %d GPU(s), 1 RCCL gather/frame" % (args.tile, args.tile, world),
                   "scene_build_s": round(t_build, 3),
                   "median_gpu_ms_per_step": round(statistics.median(m[0] for m in step_ms), 4)},
        "timing": timing,
    }

    if world > 1:
        # ---- what the N-GPU frame cost where (rank 0 prints; every rank takes part in the collectives above)
        mg = {"tiles_total": plan.n_tiles, "tiles_per_rank": [len(range(r, plan.n_tiles, world)) for r in range(world)],
              "render_ms_per_rank_median": [round(x, 4) for x in all_rank_ms],
              "imbalance_max_over_mean": round(max(all_rank_ms) / (sum(all_rank_ms) / world), 4),
              "payload_bytes_per_peer": plan.per_rank * args.tile * args.tile * 16,
              "transport": "torch.distributed gather on the %s backend (nccl = RCCL: one grouped send/recv, peers -> rank 0), "
                           "pack / un-permute by the library's kernels" % backend}
        if rank == 0 and gather_s:
            mg["gather_ms_median_incl_wait_for_slowest_rank"] = round(statistics.median(gather_s) * 1e3, 4)
        if args.extras:
            # the same frame by rank 0 alone (the others wait at the barrier): the line's own 1-GPU reference
            barrier()
            if rank == 0:
                one = torch.zeros_like(accum)
                p1 = params((0, 1))
                sc.counters_reset()
                sc.render_device(p1, one.data_ptr(), stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sc.render_device(p1, one.data_ptr(), stream)
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                r1 = sc.counters()["rays"] / 2
                mg["one_gpu_same_frame"] = {"ms": round(d1 * 1e3, 3), "Mrays_s": round(r1 / d1 / 1e6, 2),
                                            "speedup_of_this_line": round(d1 / elapsed * args.steps, 3) if elapsed > 0 else None,
                                            # (a sample of chapter 5's estimator can be non-finite -- 0/0 in the MIS weights, as in the
                                            # reference -- so the comparison is on the bits, and the L-inf over the finite pixels)
                                            "bit_identical_to_n_gpu_frame": bool(torch.equal(one.view(torch.int32), final.view(torch.int32))),
                                            "linf_vs_n_gpu_frame": float(torch.nan_to_num(one - final, nan=0.0, posinf=0.0, neginf=0.0).abs().max()),
                                            "non_finite_pixels": int((~torch.isfinite(one[..., :3]).all(dim=2)).sum())}
            barrier()
            if not weak:
                # the weak variant (spp x N: every GPU keeps the 1-GPU number of pixel-samples) as an extra field: near-linear by
                # construction -- each GPU repeats the 1-GPU work and only the closing gather is new -- so it is never `value` by default
                pw = params((rank, world), n_spp=cfg["spp"] * world)
                accw = torch.zeros_like(accum)
                stepw = make_step(sc, pw, accw, plan)
                stepw()
                torch.cuda.synchronize()
                sc.counters_reset()
                ww, _, _, _ = timed_windows(stepw, 3, 2)
                rw = torch.tensor([float(sc.counters()["rays"])], dtype=torch.float64, device=tdev)
                dist.all_reduce(rw, op=dist.ReduceOp.SUM)
                medw = sorted(ww)[1]
                mg["weak_variant"] = {"workload": "the same scene and frame at %d spp (= %d spp x %d GPUs)" % (cfg["spp"] * world, cfg["spp"], world),
                                      "scaling": "weak", "ms_per_step": round(medw / 2 * 1e3, 3), "window_ms": [round(x * 1e3, 3) for x in ww],
                                      "Mrays_s": round(float(rw[0]) / 6 / (medw / 2) / 1e6, 2)}
                barrier()
            # BASELINE.json configs[3] as a fixed frame split N ways (strong scaling): an extra field, never `value`
            c4 = dict(scenes.CONFIGS["C4"])
            if args.spp:
                c4["spp"] = args.spp
            if args.width and args.height:
                c4["width"], c4["height"] = args.width, args.height
            bs4 = scenes.p5_scene(subdiv=args.subdiv, hdr=args.env)
            sc4 = bs4.upload(hip)
            e4, cam4 = S.camera(*c4["camera"])
            W4, H4 = c4["width"], c4["height"]
            p4 = trace.make_params(W4, H4, e4, cam4, c4["integrator"], c4["max_bounce"], spp=c4["spp"], tile=(args.tile, args.tile),
                                   shard=(rank, world))
            acc4 = torch.zeros((H4, W4, 4), dtype=torch.float32, device=dev)
            plan4 = tiles.TilePlan(W4, H4, args.tile, args.tile, world)
            step4 = make_step(sc4, p4, acc4, plan4)
            step4()
            torch.cuda.synchronize()
            sc4.counters_reset()
            w4, _, _, fin4 = timed_windows(step4, 3, 2)
            r4 = torch.tensor([float(sc4.counters()["rays"])], dtype=torch.float64, device=tdev)
            dist.all_reduce(r4, op=dist.ReduceOp.SUM)
            med4 = sorted(w4)[1]
            c4v = {"workload": "C4: P5 scene, integrator %d, %d bounces, %dx%d, %d spp, fixed frame split over %d GPUs"
                               % (c4["integrator"], c4["max_bounce"], W4, H4, c4["spp"], world),
                   "scaling": "strong", "ms_per_step": round(med4 / 2 * 1e3, 3), "window_ms": [round(x * 1e3, 3) for x in w4],
                   "Mrays_s": round(float(r4[0]) / 6 / (med4 / 2) / 1e6, 2)}
            barrier()
            if rank == 0:
                one4 = torch.zeros_like(acc4)
                p41 = trace.make_params(W4, H4, e4, cam4, c4["integrator"], c4["max_bounce"], spp=c4["spp"], tile=(args.tile, args.tile))
                sc4.render_device(p41, one4.data_ptr(), stream)
                torch.cuda.synchronize()
                sc4.counters_reset()
                t1 = time.perf_counter()
                sc4.render_device(p41, one4.data_ptr(), stream)
                torch.cuda.synchronize()
                d41 = time.perf_counter() - t1
                c4v["one_gpu_same_frame"] = {"ms": round(d41 * 1e3, 3), "Mrays_s": round(sc4.counters()["rays"] / d41 / 1e6, 2),
                                             "speedup": round(d41 / (med4 / 2), 3),
                                             "bit_identical_to_n_gpu_frame": bool(torch.equal(one4.view(torch.int32), fin4.view(torch.int32)))}
            barrier()
            mg["c4_strong_variant"] = c4v
        out["multi_gpu"] = mg

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel
        sc.set_instrumentation(1)
        sc.counters_reset()
        scratch.zero_()
        sc.render_device(p, scratch.data_ptr(), stream)
        torch.cuda.synchronize()
        c = sc.counters()
        sc.set_instrumentation(0)
        # two routes, one frame: the timed pipeline (4-wide pruning nearest-first traceq4_kernel, split shading, in-launch ray
        # generation) and the instrumented one (binary UNPRUNED in-order traceq_kernel<true,6> = the reference's visit order)
        out["parity"] = {"timed_frame_equals_instrumented_frame": bool(torch.equal(final.view(torch.int32), scratch.view(torch.int32))),
                         "timed_frame": "cloned right after the timed windows (spp %d, frames 0..%d)" % (spp, spp - 1),
                         "timed_frame_finite": bool(torch.isfinite(final).all()),
                         "rays_timed_route_equals_instrumented_route": int(rays_per_step) == int(c["rays"])}
        # Dominant kernel = the persistent hitBVH over a ray queue (traceq4_kernel; 1 + max_bounce launches per step
        # + as many normally-empty redo launches of the in-order kernel).
        bytes_trace = 48 * c["node_pops"] + 96 * c["inner_pops"] + 72 * c["tri_tests"] + 72 * c["mat_fetch"]
        bytes_step = alg_bytes(c, bilinear=(bs.env_filter == 1))
        ms_total = statistics.median(m[0] for m in step_ms)       # all kernels of a step (hipEvents)
        ms_trace = statistics.median(m[1] for m in step_ms)       # trace launches of a step (hipEvents around each)
        launches = max(1, step_ms[0][2])
        ach = bytes_trace / (ms_trace * 1e-3) / 1e9
        phys, why = physical_roofline(ms_trace, launches)
        rf = {"kernel": "ezd::traceq4_kernel<6,*> (persistent hitBVH over a ray queue; + redo launches of ezd::traceq_kernel<false,6>)",
              "launch_ms": round(ms_trace / launches, 4), "launches_per_step": launches, "trace_ms_per_step": round(ms_trace, 4)}
        if phys:
            # every fraction on a FIXED denominator: VALU issue / the guide's nominal 1.2288 T wave-instr/s, HBM / 8 TB/s,
            # L2 / 34.5 TB/s, LDS / 150 TB/s -- recomputable from profiles/rN/pmc_summary.json + trace_ms_per_step of this line
            fr = {"valu_issue": phys["issue_frac_of_nominal_peak"], "hbm": phys["hbm_frac"], "l2": phys["l2_frac"], "lds": phys["lds_frac"]}
            bound = max(fr, key=fr.get)
            rf.update({"bound": bound, "achieved": phys["issue_rate_T"] if bound == "valu_issue" else None,
                       "peak": round(VALU_ISSUE_PEAK_NOMINAL_T, 4), "unit": "T wave-instr/s", "frac": fr[bound],
                       "peak_measured": VALU_ISSUE_PEAK_T, "peak_nominal": round(VALU_ISSUE_PEAK_NOMINAL_T, 4),
                       "frac_of_measured_peak": phys["issue_frac"],
                       "useful_lane_frac": round(phys["issue_frac_of_nominal_peak"] * phys["lane_fill"], 4),
                       "stall_split": phys.get("stall_split"),
                       "peak_kernel_opcode_mix": VALU_ISSUE_PEAK_KERNEL_MIX_T,
                       "frac_of_kernel_opcode_mix_peak": round(phys["issue_rate_T"] / VALU_ISSUE_PEAK_KERNEL_MIX_T, 4) if VALU_ISSUE_PEAK_KERNEL_MIX_T else None,
                       "regime": "the dominant kernel ALONE: per-launch events, calls one at a time, chunks not overlapped (pairs with "
                                 "value_lone_call, not with value)",
                       "peak_note": "peak = peak_nominal = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md): a "
                                    "constant, so the fraction only moves when the kernel does; peak_measured = this chip's measured wave64 issue ceiling: "
                                    + VALU_ISSUE_PEAK_SOURCE + "; peak_kernel_opcode_mix = the same microbenchmark on the timed bounce-stage kernel's "
                                    "static opcode histogram; useful_lane_frac = frac x lane_fill = issued lane-instructions / the chip's lane-issue peak",
                       "traffic": phys["traffic_per_launch"], "ceilings": fr, "physical": phys})
            if phys.get("all_kernels_valu_wave_instr_per_step"):
                # the CHIP over a whole step, every kernel: what overlapping consecutive calls changes (the dominant kernel's own fraction
                # above is measured with the calls one at a time and does not move)
                av = phys["all_kernels_valu_wave_instr_per_step"]
                rf["whole_step"] = {"valu_wave_instr_all_kernels": av,
                                    "issue_frac_in_the_timed_windows": round(av / (ms_per_step * 1e-3) / 1e12 / VALU_ISSUE_PEAK_NOMINAL_T, 4),
                                    "issue_frac_one_call_at_a_time": round(av / (ms_total * 1e-3) / 1e12 / VALU_ISSUE_PEAK_NOMINAL_T, 4),
                                    "note": "all kernels' VALU wave-instructions per step / step time / %.4f T (nominal): ms_per_step of the median window "
                                            "(calls queued back to back, chunks overlapped) vs the median GPU time of a call between two synchronisations" % VALU_ISSUE_PEAK_NOMINAL_T}
        else:
            rf.update({"bound": "valu_issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK_NOMINAL_T, 4), "unit": "T wave-instr/s", "frac": None,
                       "peak_measured": VALU_ISSUE_PEAK_T, "peak_nominal": round(VALU_ISSUE_PEAK_NOMINAL_T, 4),
                       "traffic": None, "note_profile": why})
        rf["work_rate_vs_hbm"] = {
            "definition": "SURVEY.md 8(d): algorithmic bytes of the reference's unpruned traversal in the reference's record sizes "
                          "(48 P + 96 I + 72 T + 72 M) / trace time / 8 TB/s.  A WORK-RATE figure, not a roofline: the timed kernel walks a "
                          "4-wide re-tree with results-neutral pruning and compact records and the scene is cache-resident, so the ratio exceeds 1 "
                          "and north_star's '>= 40 % of HBM roofline' can neither be met nor missed on this definition (VERDICT r5 weak #2); the "
                          "physical HBM fraction of the same kernel is roofline.ceilings.hbm.",
            "achieved_GBs": round(ach, 2), "peak_GBs": HBM_PEAK_GBS, "ratio": round(ach / HBM_PEAK_GBS, 4),
            "ratio_vs_measured_copy_peak_6290": round(ach / 6290.0, 4),
            "alg_bytes_per_launch": int(bytes_trace // launches), "alg_bytes_per_ray": round(bytes_trace / c["rays"], 1),
            "counters_per_step": {k: c[k] for k in ("rays", "node_pops", "inner_pops", "tri_tests", "mat_fetch", "samples", "env_map", "env_cache")},
            "counters_from": "one extra, untimed step with ezrt_set_instrumentation(1): the binary in-order kernel traceq_kernel<true,6> counts "
                             "the REFERENCE's unpruned traversal (P, I, T, M of SURVEY 8(d)) -- not the timed 4-wide, pruning kernel, which "
                             "visits fewer boxes by construction",
            "whole_step": {"alg_bytes": int(bytes_step), "gpu_ms": round(ms_total, 4),
                           "achieved_GBs": round(bytes_step / (ms_total * 1e-3) / 1e9, 2)}}
        out["roofline"] = rf

        if args.extras and wl == "c2":
            # second camera (SURVEY.md 8d "worth adding"): the Bunny-filling P5 preset, same scene and settings
            p2 = params((0, 1), camera=scenes.CONFIGS["C4"]["camera"])
            acc2 = torch.zeros_like(accum)
            sc.render_device(p2, acc2.data_ptr(), stream)
            torch.cuda.synchronize()
            sc.counters_reset()
            t1 = time.perf_counter()
            for _ in range(5):
                sc.render_device(p2, acc2.data_ptr(), stream)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            out["config"]["second_camera_p5_preset"] = {"camera": "rot 90 up 10 r 2 (P5/main.cpp:796-798): the Bunny fills the frame",
                                                        "Mrays_s": round(sc.counters()["rays"] / d2 / 1e6, 2),
                                                        "ms_per_step": round(d2 / 5 * 1e3, 4),
                                                        "rays_per_step": int(sc.counters()["rays"] / 5)}

            # the host-buffer entry point (ezrt_render: lastFrame crosses PCIe in and out on every call) -- never `value`
            host_acc = np.zeros((H, W, 4), np.float32)
            sc.render(p, host_acc)
            sc.counters_reset()
            t1 = time.perf_counter()
            for _ in range(5):
                sc.render(p, host_acc)
            dh = time.perf_counter() - t1
            out["config"]["host_buffer_entry_ezrt_render"] = {"Mrays_s_pcie_inclusive": round(sc.counters()["rays"] / dh / 1e6, 2),
                                                              "ms_per_step": round(dh / 5 * 1e3, 4),
                                                              "frame_bytes_each_way": int(H * W * 16)}

        # ---- single-GPU scaling model: what an N-way strong split of C2 and of C4 would cost
        c4_for_model = None
        if args.model and args.extras and wl == "c2":
            mdl = {"C2": scaling_model(out["config"]["workload"].split(",")[0] + ", %d spp" % spp, sc, hip, torch, dev, stream,
                                       lambda sh: params(sh), W, H, args.tile)}
            c4 = dict(scenes.CONFIGS["C4"])
            if args.config_spp:
                c4["spp"] = args.config_spp
            bs4 = scenes.p5_scene(subdiv=args.subdiv, hdr=args.env)
            sc4 = bs4.upload(hip)
            e4, cam4 = S.camera(*c4["camera"])
            mdl["C4"] = scaling_model("C4: P5 scene, integrator %d, %d bounces, %dx%d, %d spp" % (c4["integrator"], c4["max_bounce"], c4["width"],
                                                                                            c4["height"], c4["spp"]), sc4, hip, torch, dev, stream,
                                      lambda sh: trace.make_params(c4["width"], c4["height"], e4, cam4, c4["integrator"], c4["max_bounce"], spp=c4["spp"],
                                                                   tile=(args.tile, args.tile), shard=sh), c4["width"], c4["height"], args.tile)
            sc4.close()
            mdl["predicted_speedup"] = {k: {n: v["shards"][n]["predicted_speedup"] for n in v["shards"]} for k, v in mdl.items() if "shards" in v}
            out["scaling_model"] = mdl
            # (at the top level too, VERDICT r5 #6: C2 is a 1.7-ms frame whose 1/8 shard is launch floors; C4 -- BASELINE configs[3], the
            # config BASELINE actually puts on 8 GPUs -- is the one the >= 6x target can be judged on)
            out["predicted_strong_scaling"] = {"C2_512x512_64spp": mdl["predicted_speedup"].get("C2"), "C4_1024x1024_%dspp" % c4["spp"]: mdl["predicted_speedup"].get("C4"),
                                               "from": "scaling_model: shard r of N rendered alone on this GPU, max over r + pack + wire at 153 GB/s + "
                                                       "rccl_floor_ms + (N - 1) un-permutes; no N-GPU node has been available in any round"}

        # ---- BASELINE configs[2..4] at their stated spp (VERDICT r3 #1b), each with a crop of the timed frame checked against the oracle
        names = [] if args.configs.strip().lower() in ("", "none") else [x.strip().upper() for x in args.configs.split(",")]
        if names and wl == "c2":
            out["configs"] = {}
            for nm in names:
                try:
                    out["configs"][nm] = time_config(nm, hip, torch, dev, stream, args.env, args.config_spp, check=(args.cpu_seconds > 0))
                except Exception as e:  # (one config failing must not cost the headline line; it is named in the line)
                    out["configs"][nm] = {"error": "%s: %s" % (type(e).__name__, e)}

        # ---- CPU baseline: the oracle (a port, not the reference binary) on a bounded sample
        if args.cpu_seconds > 0:
            ora = load_oracle()
            so = bs.upload(ora)
            cores = host_cores()
            try:
                ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
            except OSError:
                pass
            n_spp = cfg["spp"]
            img = np.zeros((H, W, 4), np.float32)
            t1 = time.perf_counter()
            so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=1), img)
            d1 = time.perf_counter() - t1
            n = int(max(1, min(n_spp - 1, args.cpu_seconds / max(d1, 1e-3))))
            so.counters_reset()
            t1 = time.perf_counter()
            so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=n, frame0=1), img)
            dn = time.perf_counter() - t1
            gpu_img = final.detach().cpu().numpy()
            linf = float(np.abs(gpu_img - img).max()) if n + 1 == n_spp else None
            # many-core hosts finish the 64-spp frame in under a second: keep sampling further frames of the
            # same workload (into a scratch image) until the sample is ~cpu_seconds of CPU work
            n_total, d_total = n, dn
            if dn < args.cpu_seconds and n + 1 == n_spp:
                extra = int(min(8192, (args.cpu_seconds - dn) / max(dn / n, 1e-4)))
                if extra > 0:
                    t1 = time.perf_counter()
                    so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=extra, frame0=n_spp), img.copy())
                    d_total += time.perf_counter() - t1
                    n_total += extra
            cr = so.counters()["rays"]
            out["cpu_baseline"] = {"value": round(cr / d_total / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                                   "sample": "frames 1..%d of the same %dx%d workload (%d rays, %.1f s), OpenMP over 32-pixel row pieces"
                                             % (n_total, W, H, cr, d_total)}
            if linf is not None:
                out["cpu_baseline"]["linf_vs_gpu"] = linf
            # per-core figure: one more frame on ONE thread
            try:
                gomp = ctypes.CDLL("libgomp.so.1")
                gomp.omp_set_num_threads(1)
                so.counters_reset()
                t1 = time.perf_counter()
                so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=1, frame0=n_spp), img.copy())
                d1t = time.perf_counter() - t1
                out["cpu_baseline"]["one_thread_Mrays_s"] = round(so.counters()["rays"] / d1t / 1e6, 4)
                gomp.omp_set_num_threads(cores)
            except OSError:
                pass
            # The reference's OWN trace on one host thread: part 5's shaders/fshader.fsh compiled by g++ into
            # oracle/_ref/libezrt_ref_fsh_p5.so (oracle/ref_recipe/: a syntax-only source pass + a GLSL language shim; the file
            # is built where /root/reference exists and travels with the snapshot).  Same scene, camera, integrator and frames on
            # a centre crop; rays = the oracle's count for exactly that crop and those frames (same traversal definition).
            try:
                sys.path.insert(0, ROOT)
                from oracle import ref as R
                if R.fsh_available(5) and integ == 50:
                    # (64x64 at 13/32, 11/32 of the frame: for C2's camera the Bunny's flank, floor and sky -- the crop of
                    # tests/golden/make_fsh_golden.py's "c2_cam_r4"; the centre of that frame is sky only)
                    cw = min(64, W, H)
                    rx, ry = min((W * 13) // 32, W - cw), min((H * 11) // 32, H - cw)
                    rect = (rx, ry, rx + cw, ry + cw)
                    f = R.Fsh(5)
                    f.set_scene(bs.tri, bs.nodes)
                    f.set_env(bs.hdr, bs.cache, 1 if bs.env_filter == 1 else 0)
                    f.set_camera(eye, cam, W, H)
                    f.set_integrator(mb, 0)
                    nfr, dref = 0, 0.0
                    ref_img = np.zeros((H, W, 4), np.float32)
                    while dref < 3.0 and nfr < max(n_spp, 1024):   # ~3 s of the compiled shader, whole frames (any frame index is a valid sample)
                        t1 = time.perf_counter()
                        f.render(nfr, 1, ref_img, rect)
                        dref += time.perf_counter() - t1
                        nfr += 1
                    gomp = ctypes.CDLL("libgomp.so.1")
                    gomp.omp_set_num_threads(1)
                    so.counters_reset()
                    port_img = np.zeros((H, W, 4), np.float32)
                    t1 = time.perf_counter()
                    so.render(trace.make_params(W, H, eye, cam, integ, mb, spp=nfr, rect=rect), port_img)
                    dport = time.perf_counter() - t1
                    gomp.omp_set_num_threads(cores)
                    rr = so.counters()["rays"]
                    sl = (slice(rect[1], rect[3]), slice(rect[0], rect[2]))
                    out["cpu_baseline"]["reference_shader_one_thread_Mrays_s"] = round(rr / dref / 1e6, 4)
                    out["cpu_baseline"]["reference_shader"] = {
                        "what": "the reference's part-5 fshader.fsh compiled for the host (oracle/_ref/libezrt_ref_fsh_p5.so), ONE thread",
                        "sample": "crop %s, frames 0..%d (%d rays, %.2f s)" % (list(rect), nfr - 1, rr, dref),
                        "port_one_thread_same_sample_Mrays_s": round(rr / dport / 1e6, 4),
                        "port_frame_equals_reference_shader_frame": same_bits_nan_ok(port_img[sl], ref_img[sl])}
            except (OSError, ImportError) as e:
                out["cpu_baseline"]["reference_shader"] = {"skipped": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        if args.save_png:
            from ezrt_amd import imageio
            rgb = hip.tonemap(final.detach().cpu().numpy().reshape(-1, 4)).reshape(H, W, 3)
            imageio.write_png(args.save_png, rgb)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
