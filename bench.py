#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: Mrays/s at fixed spp.

N = 1 (the headline line): workload C2 = BASELINE.json configs[1] made concrete as SURVEY.md 8(d): the P3 scene with
the Stanford Bunny subdivided to ~70k triangles (79 820 total), SAH BVH leaf 8, 512x512, reference camera (r = 4),
integrator 50 (P5 pathTracing: Sobol + Cranley-Patterson hemisphere sampling, Disney BRDF), 4 bounces, 64 spp, env
map = the reference's only shipped HDR (P4/HDR/peppermint_powerplant_4k.hdr, 1024x512, bilinear).  One "step" = one
full render of that frame (64 spp) through libezrt_hip.so; scene, env map and the frame buffer are resident in HBM
before the timed region.  ray := one hitBVH call, counted by the kernels.

N > 1 (one process per GPU, torchrun): the SAME workload, so that the per-N values of a scaling run are comparable:
the C2 frame is cut into 16x16 tiles dealt round-robin to the ranks, the scene is replicated, every rank traces all
spp of its tiles and ONE gather of the packed tiles to rank 0 over RCCL closes the frame.  Default "scaling": "weak"
(spp x N: every GPU keeps the 1-GPU number of pixel-samples, the units are pixel-samples and they shard with no
data-path collective before the closing gather); `--scaling strong` splits the fixed 64-spp frame N ways instead.
Extra fields report per-rank render times, the gather time, the imbalance, the same frame rendered by rank 0 alone
with a bitwise comparison (so the line carries its own 1-GPU reference), and BASELINE.json configs[3] (C4: P5 scene,
integrator 51 = env importance sampling + MIS, 2 bounces, 1024x1024, 256 spp) as a FIXED frame split N ways
(`c4_strong_variant`).  `--workload c2|c4` picks the main workload for any N.

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
# VALU issue ceiling: tools/exp_valu_issue.hip on this chip (profiles/r2/valu_issue_microbench.txt): wave64 fp32
# add/mul/mov and the slab test's own opcode mix issue at 1.0-1.09 T wave-instructions/s chip-wide (~2 cycles per
# instruction per SIMD at the ~2.1-2.4 GHz the chip sustains), fma/min3/cndmask alone at 0.58 T.
VALU_ISSUE_PEAK_T = 1.086


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
    return out, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("auto", "c2", "c4"), default="auto", help="auto: C2 at 1 GPU, C4 at N > 1")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=-1)
    ap.add_argument("--integrator", type=int, default=0)
    ap.add_argument("--subdiv", type=int, default=2)
    ap.add_argument("--tile", type=int, default=16)
    ap.add_argument("--env", choices=("shipped", "synthetic"), default="shipped")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="weak",
                    help="N > 1 -- weak (default): spp x N, so every GPU keeps the 1-GPU number of pixel-samples; "
                         "strong: the fixed frame is split N ways")
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
    gather_s = [a.elapsed_time(b) * 1e-3 for a, b in gather_ev]
    rays_timed = sc.counters()["rays"]          # over all windows (the workload is deterministic: the same rays every step)
    # per-step GPU times with a timing-event pair around every trace launch (hipEvents on the launch stream; reading them
    # synchronises and each record costs the stream ~5 us, so this is a separate, untimed pass of the same step):
    # median of 7 -- (all kernels, the trace launches, number of trace launches)
    step_ms = []
    sc.set_option("launch_events", 1)
    for _ in range(7):
        sc.render_device(p, accum.data_ptr(), stream)
        step_ms.append(sc.last_render_ms())
    sc.set_option("launch_events", 0)

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
        sc.render_device(p, accum.data_ptr(), stream)
        torch.cuda.synchronize()
        c = sc.counters()
        sc.set_instrumentation(0)
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
            fr = {"valu_issue": phys["issue_frac"], "hbm": phys["hbm_frac"], "l2": phys["l2_frac"], "lds": phys["lds_frac"]}
            bound = max(fr, key=fr.get)
            rf.update({"bound": bound, "achieved": phys["issue_rate_T"] if bound == "valu_issue" else None,
                       "peak": VALU_ISSUE_PEAK_T, "unit": "T wave-instr/s", "frac": fr[bound],
                       "traffic": phys["traffic_per_launch"], "ceilings": fr, "physical": phys})
        else:
            rf.update({"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK_T, "unit": "T wave-instr/s", "frac": None,
                       "traffic": None, "note_profile": why})
        rf["work_rate_vs_hbm"] = {
            "definition": "SURVEY.md 8(d): algorithmic bytes of the reference's unpruned traversal in the reference's record sizes "
                          "(48 P + 96 I + 72 T + 72 M) / trace time / 8 TB/s.  A WORK-RATE figure, not a roofline: the device layout "
                          "moves fewer bytes and the scene is cache-resident, so it can exceed 1.",
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

        # ---- CPU baseline: the oracle (a port, not the reference binary) on a bounded sample
        if args.cpu_seconds > 0:
            from ezrt_amd import _abi
            opath = os.path.join(ROOT, "oracle", "libezrt_oracle.so")
            ora = trace.TraceLib(_abi.declare_trace_abi(ctypes.CDLL(opath)))
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
