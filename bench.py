#!/usr/bin/env python3
"""bench.py — scored RANSAC hypotheses/s of the triplet-match hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, all inputs already resident
in HBM: hypothesis generation from a batch of scene sample pairs (descriptor -> hashed match ->
base transform, <= 200 poses per pair, capped at 65,536) -> inlier scoring of every hypothesis
against the whole scene -> best-(count, score, pose) record (+ one RCCL all-gather of the 96-byte
records when N > 1). Workload (--workload, default c2 = BASELINE.json configs[1], the configuration the metric is quoted on: identity
traits, 100k-pt model, 1M-pt scene, 64k hypotheses per batch). The other single-GPU-sized configurations run
through the same code path and print the same line (roofline + cpu_baseline):
  c3 = configs[2] cylinder_traits, 500k-pt pipe plant (scene and model projected into the cylinder's uvw space
       on the device, scoring under the wrap-around metric);
  c4 = configs[3] plane2_traits, 1M-pt indoor scan (2-D metric);
  c5 = configs[4] 32 models x 4M-pt scene: a step = one 64k batch per model (32 scoring launches) and ONE reduce
       of the 32 per-model best records.
Data: seeded synthetic clouds (triplet_match_amd/synth.py).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); scene / model / table are
replicated, sample pairs are sharded round-robin, every rank scores its own 64k batch (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
INFINITY_CACHE_RANDOM_GBS = 8600.0  # same guide, "Indexed rows": 38 MB table, uniformly random rows, chip-wide


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="BASELINE.json configs[1] (default, the headline), [2], [3] or [4]")
    ap.add_argument("--model-points", type=int, default=100_000)
    ap.add_argument("--scene-points", type=int, default=1_000_000)
    ap.add_argument("--hyp-per-batch", type=int, default=65_536)
    ap.add_argument("--copies", type=int, default=4)
    ap.add_argument("--mode", choices=["auto", "brute", "culled"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU baseline time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-figures measured after the headline region (profiling passes)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a 1-GPU box: every rank uses cuda:0 and the 96-byte exchange goes "
                         "over gloo (RCCL refuses two ranks on one device); not a measurement")
    ap.add_argument("--ed-profile", action="store_true",
                    help="profiling aid: after the setup run only the device-resident early-drop scoring of the batch "
                         "(three calls) and print its figure; tools/profile_round5.sh counts its fabric bytes")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="plumbing check without a GPU: the ranks come up over gloo, exchange one synthetic best "
                         "record each and rank 0 prints a line with \"selftest\": true and no value (tests/)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher around it: start the N ranks here. Nothing in this process has
    # touched HIP yet (torch is not even imported), the ranks are fresh child processes, rank 0's JSON line is
    # relayed and the children's status becomes ours.
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly one JSON line: send everything libraries print there (gloo / RCCL banners,
    # compiler output) to stderr until the line is written
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # a 1-GPU figure must never be recorded under "--gpus 8" (or the reverse)
        log("FATAL: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); no line is printed" % (args.gpus, world))
        sys.exit(2)
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world, real_stdout)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from triplet_match_amd import build as tm_build
    if rank == 0:
        tm_build.build()
    if world > 1:
        dist.barrier()
    from triplet_match_amd import capi, synth
    from triplet_match_amd import dist as tmd

    mode = {"auto": capi.SCORE_AUTO, "brute": capi.SCORE_BRUTEFORCE, "culled": capi.SCORE_CULLED}[args.mode]
    DIST_THRES, ACCEPT_PROB = 3.0, 0.5

    # ---- the workload: clouds (identical on every rank), model(s) and scene on the device, sample pairs ----
    H = args.hyp_per_batch
    W = prepare_workload(args, ctx_factory=lambda: capi.Context(local_rank), rank=rank, world=world, log=log)
    ctx, units, gs = W["ctx"], W["units"], W["gs"]
    u0 = units[0]
    gm, pairs, model, res, subset, scene = u0["gm"], u0["pairs"], u0["M"], u0["res"], u0["subset"], W["scene"]
    d_poses = ctx.alloc(H * 64)
    d_prov = ctx.alloc(H * 16)
    d_counts = ctx.alloc(H * 4)
    d_scores = ctx.alloc(H * 8)
    for u in units:
        u["d_ij"] = ctx.alloc(max(u["pairs"].nbytes, 8))
        ctx.h2d(u["d_ij"], u["pairs"])
        u["n_hyp"], total = u["gs"].generate_hypotheses_dev(u["gm"], u["d_ij"], len(u["pairs"]), d_poses, d_prov, H)
        log("[rank %d] %s: %d sample pairs -> %d hypotheses (%d before the batch cap)"
            % (rank, u["name"], len(u["pairs"]), u["n_hyp"], total))
        if u["n_hyp"] < H:
            log("warning: batch not full (%d < %d); throughput is reported for the hypotheses scored" % (u["n_hyp"], H))
    # the buffers hold the LAST unit's batch after the loop; the single-unit helpers below (early-drop figure, CPU
    # baseline) work on unit 0's batch, which step() leaves there when there is one unit and which is regenerated
    # for them otherwise
    n_hyp = u0["n_hyp"]

    if args.ed_profile:
        out = early_drop_dev_figure(ctx, gm, gs, pairs, d_poses, d_prov, n_hyp, DIST_THRES, ACCEPT_PROB, reps=3, parity=False)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"ed_profile": True, "early_drop_ball_dev": out}), flush=True)
        return

    cuda_dev = torch.device("cpu") if args.rehearse_on_one_gpu else torch.device("cuda", local_rank)

    # the library's own RCCL communicator (C ABI: tm_comm_create / tm_best_reduce_begin / _end). torch.distributed
    # only carries the 128-byte unique id to the other ranks, the barriers and the timing reduction.
    # Order matters: (1) every rank probes whether RCCL is usable and the ranks agree on it, (2) rank 0 makes the id,
    # (3) only then do all ranks enter the collective tm_comm_create — a rank that cannot load librccl never leaves
    # the others waiting inside ncclCommInitRank.
    comm = None
    comm_route = "single_process"
    if world > 1:
        comm_route = "torch_fallback"
    if not args.rehearse_on_one_gpu and world > 1:
        ok = 1 if capi.comm_probe() else 0
        t = torch.zeros(capi.COMM_ID_BYTES + 1, dtype=torch.uint8, device=cuda_dev)
        if rank == 0 and ok:
            try:
                t[:capi.COMM_ID_BYTES] = torch.tensor(list(capi.comm_unique_id()), dtype=torch.uint8)
                t[capi.COMM_ID_BYTES] = 1
            except Exception as e:
                log("[rank 0] tm_comm_get_unique_id failed (%s)" % e)
        dist.broadcast(t, src=0)
        have_id = int(t[capi.COMM_ID_BYTES].item())
        flag = torch.tensor([ok & have_id], dtype=torch.int32, device=cuda_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            uid = bytes(t[:capi.COMM_ID_BYTES].cpu().numpy().tobytes())
            try:
                comm = capi.Comm(ctx, rank, world, uid)
            except Exception as e:  # past the probe this is a real RCCL failure: say so, then agree on the fallback
                log("[rank %d] tm_comm_create failed (%s)" % (rank, e))
                comm = None
            flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=cuda_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                comm = None
        else:
            log("[rank %d] RCCL not usable on every rank (probe here: %d): the records travel through torch.distributed"
                % (rank, ok))
        if comm is not None:
            comm_route = "tm_best_reduce"
    comm_world = int(capi.lib().tm_comm_world(comm.h)) if comm is not None else 1
    n_units = len(units)
    d_best = ctx.alloc(96 * n_units)
    import ctypes

    class _First:
        def __init__(self, pending):
            self._p = pending

        def result(self):
            return self._p.result()[0]

    def step():
        # one batch per unit (one unit = one model; c5 has 32), everything on the context's one stream: generation,
        # scoring under the unit's metric, local argmax on the device with ids that do not depend on the number of
        # ranks (global sample pair index * query_limit + entry) into the unit's 96-byte record; then ONE
        # ncclAllGather of the n_units records on the library's stream, the deterministic selection per model and an
        # asynchronous copy of the winners into pinned memory (tm_best_reduce_begin): nothing here waits for the GPU.
        # The wait (tm_best_reduce_end) happens in .result(), which the timed loop calls only after the NEXT step has
        # been queued.
        total = 0
        for k, u in enumerate(units):
            n, _ = u["gs"].generate_hypotheses_dev(u["gm"], u["d_ij"], len(u["pairs"]), d_poses, d_prov, H)
            u["gs"].score_hypotheses_dev(u["gm"], d_poses, n, DIST_THRES, d_counts, d_scores, mode=mode, traits=u["K"])
            ctx.best_of_batch_prov(d_counts, d_scores, d_poses, d_prov, n, ctypes.c_void_p(d_best.value + 96 * k),
                                   pair_offset=rank, pair_stride=world, query_limit=200)
            total += n
        if comm is not None or world == 1:
            return total, _First(capi.best_reduce_begin(ctx, comm, d_best, n_units))
        # rehearsal of N ranks on one GPU (RCCL refuses two ranks per device) or no usable RCCL: the same
        # records through torch.distributed (first unit's record; the rehearsal is plumbing, not a measurement)
        rec = capi.BestRecord()
        ctx.d2h(np.frombuffer(rec, dtype=np.uint8), d_best)
        return total, tmd.allgather_best_async(rec, device=cuda_dev)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # Stage counters (evaluations, in-grid, near, passing, staged items) are a property of the batch, and every step
    # scores the same batch: they are read from the WARM-UP launches (measurement window with counters). The timed
    # launches then run what a caller outside any measurement window runs — the kernel instantiation without the
    # counters' bookkeeping — with the HIP events only (tm_ctx_enable_timing(ctx, 2)).
    cnt = None
    if args.warmup > 0:
        ctx.enable_timing(1)
    for _ in range(args.warmup):
        step()[1].result()
    if args.warmup > 0:
        cnt = ctx.score_timing_totals()
        ctx.enable_timing(False)
        if cnt["launches"] != args.warmup * n_units:
            cnt = None
    # kernel timing = HIP events recorded on the library's own stream around each scoring launch; read AFTER the
    # timed region. Inside it the host waits once per step, for the PREVIOUS step's 96-byte winner (an event behind
    # its pinned copy).
    ctx.enable_timing(2 if cnt else 1)
    hyps = 0
    sync_all()
    t_start = time.perf_counter()
    pending = None
    for _ in range(args.steps):
        n, nxt = step()
        if pending is not None:
            best = pending.result()
        pending = nxt
        hyps += n
    if pending is None:
        raise SystemExit("--steps must be at least 1")
    best = pending.result()  # the last exchange completes inside the timed region
    sync_all()
    elapsed = time.perf_counter() - t_start
    tot = ctx.score_timing_totals()
    ctx.enable_timing(False)
    if tot["launches"] != args.steps * n_units:
        log("warning: %d timed scoring launches for %d steps of %d unit(s)" % (tot["launches"], args.steps, n_units))
    if cnt:  # per-launch counters of the warm-up window, scaled to the timed launches (same batch every step)
        f = tot["launches"] / float(cnt["launches"])
        for k in ("evaluations", "passing", "in_grid", "near", "staged_chunks"):
            tot[k] = cnt[k] * f
    evals, passing = tot["evaluations"], tot["passing"]

    per_rank_ms = [round(elapsed / args.steps * 1e3, 3)]
    if world > 1:
        mine = torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=cuda_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(x.item()), 3) for x in every]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cuda_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        h = torch.tensor([hyps, evals], dtype=torch.float64, device=cuda_dev)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        hyps_all, evals_all = float(h[0].item()), float(h[1].item())
    else:
        hyps_all, evals_all = float(hyps), float(evals)

    value = hyps_all / elapsed
    # roofline of the dominant kernel (inlier scoring), rank 0's launches, per launch:
    #   formula  = BASELINE.md section 3: 40*E + 32*P + 64*H (E = evaluations the kernel performed)
    #   physical = the bytes those evaluations really request: scene xyz (12 B per evaluation; the grouped
    #              kernel stages 512-point chunks of 6 KiB + 1 KiB of poses and spheres once per item instead),
    #              1 B brick code per in-grid point, 4 B voxel + 12 B model xyz per near point, 16 + 16 B
    #              orientation records per distance-passing point, 64 B pose per hypothesis
    L = max(1, tot["launches"])
    k_ms = tot["ms"] / L
    E, P, Hh = evals / L, passing / L, hyps / float(L)
    N_in, N_near = tot["in_grid"] / L, tot["near"] / L
    alg_bytes = 40.0 * E + 32.0 * P + 64.0 * Hh
    N_items = tot["staged_chunks"] / L
    scene_bytes = (6144.0 + 1408.0) * N_items if N_items else 12.0 * E
    phys_bytes = scene_bytes + 1.0 * N_in + 16.0 * N_near + 32.0 * P + 64.0 * Hh
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    achieved_phys = phys_bytes / (k_ms * 1e-3) / 1e9
    pmc = measured_traffic(args)
    traffic = pmc["hbm_bytes_per_launch"] if pmc else None
    copy_gbs = measured_copy_bandwidth(torch, local_rank) if rank == 0 else None
    # ONE basis for roofline.frac, on every line of every round from here on: the bytes the L2s request from the
    # fabric for this kernel (TCC_EA0_RDREQ by size + write requests, separate rocprofv3 --pmc passes of this same
    # command: profiles/rNN/traffic.json, quoted only while the kernel sources still hash to what was profiled)
    # / this run's kernel time / the 8 TB/s HBM peak. Infinity-Cache hits are inside those counters, so this is
    # "Infinity Cache + HBM": an UPPER bound of the HBM traffic. The formula and the requested-bytes
    # model stay as side fields; neither is ever promoted to `frac`.
    k_s = k_ms * 1e-3
    meas_gbs = traffic / k_s / 1e9 if traffic else None
    roofline = {"bound": "hbm", "achieved": round(meas_gbs, 1) if meas_gbs else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(meas_gbs / HBM_PEAK_GBS, 4) if meas_gbs else None,
                "frac_basis": ("measured_l2_fabric_bytes: Infinity Cache + HBM, upper bound of HBM" if meas_gbs else
                               "unavailable: the scoring kernel's sources changed since the committed PMC passes"),
                "traffic": traffic,
                "kernel": "k_pose_aux + k_group_cull + select + k_score_items", "kernel_ms": round(k_ms, 4),
                "stage_counters": ("warm-up launches of the same batch (the timed launches run without the counters' "
                                   "bookkeeping, as every launch outside a measurement window does)" if cnt else
                                   "timed launches"),
                # the guide's rate for uniformly random lines out of a table that fits the Infinity Cache (8.6 TB/s):
                # the roof the measured line rate is really up against (the hot set, ~60 MB, is Infinity-Cache resident)
                "infinity_cache_peak": INFINITY_CACHE_RANDOM_GBS,
                "infinity_cache_frac": round(meas_gbs / INFINITY_CACHE_RANDOM_GBS, 4) if meas_gbs else None,
                "evaluations_per_launch": E, "in_grid_per_launch": N_in, "near_per_launch": N_near,
                "staged_chunks_per_launch": N_items, "passing_per_launch": P,
                # side figures, never `frac`: SURVEY 8d's formula 40E + 32P + 64H prices scene bytes that come out
                # of LDS (above the peak since round 2: not evidence); the requested-bytes model = 7,552 B per staged
                # item + 1 B per in-grid point + 16 B per near point + 32 B per passing point + 64 B per hypothesis
                "algorithmic_bytes_per_launch": alg_bytes, "achieved_formula": round(achieved, 1),
                "frac_formula": round(achieved / HBM_PEAK_GBS, 4),
                "physical_bytes_per_launch": phys_bytes, "achieved_physical": round(achieved_phys, 1),
                "frac_physical": round(achieved_phys / HBM_PEAK_GBS, 4),
                "measured_copy_GBs": copy_gbs}
    # Two figures that say more about this kernel than bytes do:
    #   valu_issue_frac = wave-level VALU instructions of a launch (SQ_INSTS_VALU) x 4 shader cycles each (a wave64
    #                     instruction occupies its SIMD for four cycles; tools/ubench_valu.hip reads 3.75-4.2 per
    #                     instruction against s_memtime with two or more waves per SIMD) over 1024 SIMDs x the shader
    #                     cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs): counters of the PMC passes only, i.e. under the
    #                     profiler (the launch is slower there and its clock is not this run's: the figure is not
    #                     rescaled to this run). `wave_wait_frac` = SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of its resident
    #                     cycles a wave spends in s_waitcnt (four waves per SIMD).
    #   line_rate_frac  = fabric lines per second of this run over the rate at which this chip delivers uniformly
    #                     random 128-B lines to all CUs (tools/ubench_l2.hip, profiles/r04/ubench_l2.txt): 3.12 ns per
    #                     line and CU out of an Infinity-Cache-resident table, 4.59 ns out of HBM.
    if pmc and pmc.get("sq_insts_valu") and pmc.get("shader_cycles_per_launch"):
        roofline["valu_issue_frac"] = round(4.0 * pmc["sq_insts_valu"] / (1024.0 * pmc["shader_cycles_per_launch"]), 4)
        roofline["valu_issue_basis"] = ("4 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), both from the PMC passes "
                                        "of the same command (under the profiler)")
        if pmc.get("sq_wait_any") and pmc.get("sq_wave_cycles"):
            roofline["wave_wait_frac"] = round(pmc["sq_wait_any"] / pmc["sq_wave_cycles"], 4)
    if pmc and pmc.get("l2_fabric_read_requests"):
        lines_per_s = pmc["l2_fabric_read_requests"] / k_s
        roofline["line_rate_frac"] = round(lines_per_s / (256.0 / 3.12e-9), 4)
        roofline["line_rate_frac_hbm_resident"] = round(lines_per_s / (256.0 / 4.59e-9), 4)
        roofline["line_rate_basis"] = ("L2->fabric lines / kernel time over the measured random-line rate of "
                                       "tools/ubench_l2.hip (profiles/r04/ubench_l2.txt): 8.2e10 lines/s out of the "
                                       "Infinity Cache, 5.6e10 out of HBM")
    if pmc:
        # how much of a fetched voxel line is used: 4 B per near point over the 128-B lines the voxel gathers
        # bring in (fabric lines of the kernel minus those of the same kernel with its voxel index folded into
        # 1 MB, measured in the same profiling session with the timing-experiments build)
        vox_lines = pmc.get("voxel_lines_per_launch")
        if vox_lines:
            roofline["voxel_lines_per_launch"] = vox_lines
            roofline["voxel_line_utilisation"] = round(4.0 * N_near / (128.0 * vox_lines), 4)
        roofline["pmc"] = {k: pmc[k] for k in ("source", "l2_fabric_read_requests", "l2_fabric_read_128B",
                                               "tcp_cache_accesses", "l1_frac_of_peak_at_profiled_ms",
                                               "kernel_ms_under_profiler")
                           if k in pmc}

    # sub-figures measured after the headline region (rank 0, N = 1): the reference's default inner loop
    # (project_ on the neighbourhood of the first sample point, with the early-drop rule) on the same batch,
    # and the whole drop-in search scene<Point>::find_all_parallel on the same clouds
    if n_units > 1 and rank == 0 and world == 1:  # the buffers hold the last unit's batch: put unit 0's back
        n_hyp, _ = u0["gs"].generate_hypotheses_dev(gm, u0["d_ij"], len(pairs), d_poses, d_prov, H)
        u0["gs"].score_hypotheses_dev(gm, d_poses, n_hyp, DIST_THRES, d_counts, d_scores, mode=mode, traits=u0["K"])
    extras = None
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2":
        extras = run_extras(args, model, res, scene, gm, gs, ctx, pairs, d_poses, d_prov, n_hyp, DIST_THRES,
                            ACCEPT_PROB)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # the CPU leg runs at N = 1 only
        cpu_baseline = run_cpu_baseline(args, model, res, subset, scene, gm, ctx, d_poses, d_counts, n_hyp,
                                        DIST_THRES, ACCEPT_PROB, pairs=pairs, d_prov=d_prov, K=u0["K"], radius=u0["radius"],
                                        unit_note=("" if n_units == 1 else "model 0 of the %d; " % n_units))

    if rank == 0 and cpu_baseline is not None and not cpu_baseline["gpu_counts_equal_oracle_on_sample"]:
        log("FATAL: the GPU's inlier counts differ from the oracle's on the sampled hypotheses; no value is reported")
        value = None
    if rank == 0:
        out = {
            "metric": "scored RANSAC hypotheses/sec (%s-pt scene); inlier-score Gpoints/sec in gpoints_per_s"
                      % {1_000_000: "1M", 500_000: "500k", 4_000_000: "4M"}.get(int(W["scene"]["surfels"].shape[0]),
                                                                                 str(int(W["scene"]["surfels"].shape[0]))),
            "value": round(value, 1) if value is not None else None, "unit": "hypotheses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": W["description"], "baseline_config": args.workload,
                       "models_per_step": n_units,
                       "hypotheses_per_step_scored": int(hyps / args.steps), "dist_thres": DIST_THRES,
                       "query_limit": 200, "score_mode": args.mode, "parallelism": "hyp-shard x%d" % world},
            "comm_route": comm_route, "comm_world": comm_world, "per_rank_ms_per_step": per_rank_ms,
            "gpoints_per_s": round(evals_all / elapsed / 1e9, 2),
            "best": {"inlier_count": int(best.inlier_count), "hyp_id": int(best.hyp_id)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "extras": extras,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
        if value is None:
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _pick_cylinder_samples(s, sel, rng):
    """two points of the selected surface with clearly different normals (cylinder_traits::init_from_samples takes two)"""
    idx = np.nonzero(sel)[0]
    a = int(idx[rng.integers(len(idx))])
    for _ in range(1000):
        b = int(idx[rng.integers(len(idx))])
        if abs(float(s[a, 4:7] @ s[b, 4:7])) < 0.3:
            return [a, b]
    raise SystemExit("no cylinder sample pair")


def _pick_plane_samples(s, sel, rng):
    idx = np.nonzero(sel)[0]
    for _ in range(1000):
        a, b, c = (int(idx[k]) for k in rng.integers(len(idx), size=3))
        d1, d2 = s[b, 0:3] - s[a, 0:3], s[c, 0:3] - s[a, 0:3]
        if np.linalg.norm(np.cross(d1, d2)) > 0.5 * np.linalg.norm(d1) * np.linalg.norm(d2) > 0.05:
            return [a, b, c]
    raise SystemExit("no plane sample triple")


def prepare_workload(args, ctx_factory, rank, world, log):
    """Clouds, model(s) and scene on the device, and this rank's share of the scene sample pairs, for one of the
    BASELINE.json configurations. Returns dict(ctx, gs, scene (what is scored: surfels + tangent_mask, in the
    projector's uvw space for c3 / c4), units, description); a unit = one model: dict(name, gm, gs, K (traits kind),
    radius, M (model cloud as scored), res, subset, pairs)."""
    from triplet_match_amd import capi, synth
    from triplet_match_amd import dist as tmd
    H = args.hyp_per_batch
    t0 = time.time()
    ctx = ctx_factory()
    wl = args.workload

    def init_model(M, res, subset, K=0, radius=0.0):
        t1 = time.time()
        gm = capi.Model(ctx, M, res)
        if K:
            gm.set_metric(K, radius)
        gm.build_voxels()
        t_vox = time.time() - t1
        t1 = time.time()
        gm.build_pairs(subset)
        return gm, t_vox, time.time() - t1

    if wl == "c2":
        model, subset = synth.synth_model(args.model_points, seed=1)
        res = synth.resolution(model)
        scene = synth.synth_scene(args.scene_points, args.model_points, seed=2, copies=args.copies, model_resolution=res)
        log("[rank %d] synth %.1fs: model %d (tangent %d), scene %d (tangent %d), resolution %.5f"
            % (rank, time.time() - t0, model.shape[0], len(subset), scene["surfels"].shape[0],
               int(scene["tangent_mask"].sum()), res))
        gm, t_vox, t_pairs = init_model(model, res, subset)
        gs = capi.Scene(ctx, scene["surfels"], scene["tangent_mask"])
        pc = gm.pair_counts()
        log("[rank %d] model init: voxels %s in %.2fs, pair table %d entries / %d keys in %.2fs"
            % (rank, list(gm.grid()[0]), t_vox, pc["entries"], pc["distinct_keys"], t_pairs))
        n_first = max(64, H // 400)
        pairs_all = synth.sample_scene_pairs(scene, gm.diameter, n_first * world, 24, seed=13)
        unit = dict(name="model", gm=gm, gs=gs, K=capi.IDENTITY, radius=0.0, M=model, res=res, subset=subset,
                    pairs=tmd.shard_pairs(pairs_all, rank, world))
        desc = ("identity_traits: %d-pt model / %d-pt scene, %d hypotheses/batch, full-scene scoring"
                % (args.model_points, args.scene_points, H))
        return dict(ctx=ctx, gs=gs, scene=scene, units=[unit], description=desc, planted=scene)
    if wl in ("c3", "c4"):
        if wl == "c3":
            K, rng = capi.CYLINDER, np.random.default_rng(2)
            model, _ = synth.synth_pipe_section(50_000, seed=3)
            sc = synth.synth_pipe_plant(500_000, 50_000, seed=4)
            msamples = _pick_cylinder_samples(model, np.ones(len(model), bool), rng)
            ssamples = _pick_cylinder_samples(sc["surfels"], sc["pipe_id"] == 0, rng)
            what = "cylinder_traits (cylinder.cl path): 500k-pt pipe-plant scene, 50k-pt pipe-section model"
        else:
            K, rng = capi.PLANE2, np.random.default_rng(3)
            model, _ = synth.synth_wall_panel(15_000, seed=5)
            sc = synth.synth_indoor(1_000_000, 15_000, seed=6)
            msamples = _pick_plane_samples(model, np.ones(len(model), bool), rng)
            ssamples = _pick_plane_samples(sc["surfels"], sc["wall_id"] == 0, rng)
            what = "plane2_traits: 1M-pt indoor scan, 15k-pt wall-panel model"
        S, tm = sc["surfels"], sc["tangent_mask"]
        threshold = 0.05
        ms = np.zeros(capi.STATE_FLOATS, np.float32)
        ms[1] = threshold
        mstate = capi.traits_init_from_samples(K, ms, np.concatenate([model[i] for i in msamples]))
        sstate = capi.traits_init_from_samples(K, ms, np.concatenate([S[i] for i in ssamples]))
        radius = float(mstate[0]) if K == capi.CYLINDER else 0.0
        raw = capi.Scene(ctx, S, tm)
        gp = raw.project(K, sstate)  # the per-projector step on the device (tm_scene_create_projected)
        su, tmask = gp.download()
        mu, mv = ctx.traits_project_cloud(K, mstate, model)
        mu = mu[mv == 1]
        res = synth.resolution(mu)
        subset = np.nonzero(np.linalg.norm(mu[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
        gm, t_vox, t_pairs = init_model(mu, res, subset, K, radius)
        log("[rank %d] %s: projected scene %d live of %d, model %d (tangent %d), voxels %s in %.2fs, pair table in %.2fs"
            % (rank, wl, int(np.isfinite(su[:, 0]).sum()), len(su), len(mu), len(subset), list(gm.grid()[0]), t_vox, t_pairs))
        # first points among the tangent points on the projector's surface (|w| small)
        near = tmask * (np.abs(np.nan_to_num(su[:, 2], nan=1e9)) < (threshold / radius if radius else threshold))
        pairs_all = synth.sample_scene_pairs(dict(surfels=su, tangent_mask=near), gm.diameter, 400 * world, 3, seed=13)
        unit = dict(name="model", gm=gm, gs=gp, K=K, radius=radius, M=mu, res=res, subset=subset,
                    pairs=tmd.shard_pairs(pairs_all, rank, world))
        desc = "%s, %d hypotheses/batch, scoring of the whole projected scene under the traits' intrinsic_distance" % (what, H)
        return dict(ctx=ctx, gs=gp, scene=dict(surfels=su, tangent_mask=tmask), units=[unit], description=desc,
                    planted=None, keep=raw)
    # c5: 32 models against one 4M-point scene
    models, scene = synth.synth_multi_model_scene(4_000_000, 32, seed=7)
    S, tm = scene["surfels"], scene["tangent_mask"]
    log("[rank %d] synth %.1fs: 32 models (%d..%d points), scene %d" % (rank, time.time() - t0, min(len(m[0]) for m in models),
                                                                       max(len(m[0]) for m in models), len(S)))
    gs = capi.Scene(ctx, S, tm)
    units = []
    for k, (M, sub, res) in enumerate(models):
        gm, _, _ = init_model(M, res, sub)
        own = dict(surfels=S, tangent_mask=tm * (scene["instance_id"] == k))
        pairs_all = np.concatenate([synth.sample_scene_pairs(own, gm.diameter, 100 * world, 4, seed=13 + k),
                                    synth.sample_scene_pairs(dict(surfels=S, tangent_mask=tm), gm.diameter, 300 * world, 4,
                                                             seed=113 + k)])
        pairs_all = np.ascontiguousarray(pairs_all[np.random.default_rng(k).permutation(len(pairs_all))])
        units.append(dict(name="model %d" % k, gm=gm, gs=gs, K=capi.IDENTITY, radius=0.0, M=M, res=res, subset=sub,
                          pairs=tmd.shard_pairs(pairs_all, rank, world)))
    log("[rank %d] 32 models on the device in %.1fs" % (rank, time.time() - t0))
    desc = ("multi-model search: 32 models (20k..100k pts) x 4M-pt scene, one %d-hypothesis batch per model and step, "
            "per-model best-pose reduce" % H)
    return dict(ctx=ctx, gs=gs, scene=scene, units=units, description=desc, planted=None)


def launch_ranks(n, argv):
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child process (this
    process has not touched the GPU), passes rank 0's stdout through and returns the child's exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("bench.py: no launcher in the environment, starting %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def launcher_selftest(args, rank, world, real_stdout):
    """No GPU, no kernels, no figure: checks that N ranks come up, agree on a best record over gloo and that rank
    0 alone prints the line. Used by tests/test_distributed.py to cover the `--gpus N` launcher path on the CPU."""
    import torch.distributed as dist
    from triplet_match_amd import dist as tmd
    if world > 1:
        dist.init_process_group("gloo")
    rec = tmd.make_record(100 + rank, 7 * rank + 1, 0.5, [float(rank)] * 16)
    best = tmd.allgather_best(rec) if world > 1 else rec
    if rank == 0:
        out = {"selftest": True, "metric": "launcher self-test (no measurement)", "value": None, "n_gpus": world,
               "best": {"inlier_count": int(best.inlier_count), "hyp_id": int(best.hyp_id)},
               "comm_route": "gloo_selftest", "comm_world": world}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_copy_bandwidth(torch, local_rank):
    """Device-to-device copy of 1 GiB on this GPU (read + write bytes / time, HIP events through torch):
    the achievable streaming bandwidth next to the 8 TB/s spec figure (SURVEY.md section 8d asks for both)."""
    try:
        n = 1 << 28  # 1 GiB of fp32
        a = torch.empty(n, dtype=torch.float32, device="cuda:%d" % local_rank).normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del a, b
        return round(2.0 * n * 4 / (ms * 1e-3) / 1e9, 1)
    except Exception as exc:  # the figure is informative; never fail the bench for it
        log("copy bandwidth not measured: %s" % exc)
        return None


def _profile_rounds():
    """profiles/rNN directories, newest first"""
    try:
        names = [d for d in os.listdir(os.path.join(ROOT, "profiles")) if d[:1] == "r" and d[1:].isdigit()]
    except OSError:
        return []
    return sorted(names, key=lambda d: -int(d[1:]))


def source_hash():
    """SHA-256 over the sources of the scoring kernel: a PMC file is only quoted for the build it profiled."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "triplet_match_amd", "csrc")
    text = open(os.path.join(csrc, "tm_scene.hip"), "rb").read()
    # the kernels' part of tm_scene.hip (markers in the file); the entry points and host helpers around them can
    # change without new PMC passes
    a, b = text.find(b"/* >>> scoring kernels"), text.find(b"/* <<< scoring kernels")
    h.update(text[a:b] if 0 <= a < b else text)
    for f in ("tm_score.h", "tm_math.h", "tm_table.h"):
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(args):
    """HBM bytes per scoring launch from the committed PMC passes (profiles/rNN/traffic.json, written by
    tools/make_traffic_json.py from separate rocprofv3 --pmc runs of this same command). Null when the
    workload differs from the profiled one or the kernel sources changed since."""
    t = None
    fname = "traffic.json" if args.workload == "c2" else "traffic_%s.json" % args.workload
    for rnd in _profile_rounds():  # the newest round whose PMC passes were taken on these kernel sources
        try:
            cand = json.load(open(os.path.join(ROOT, "profiles", rnd, fname)))
        except (OSError, ValueError):
            continue
        if cand.get("source_hash") == source_hash():
            t = cand
            t["file"] = "profiles/%s/%s" % (rnd, fname)
            break
    if t is None:
        log("no profiles/r*/%s was taken on these kernel sources: roofline.traffic = null" % fname)
        return None
    w = t.get("workload", {})
    if w.get("workload", "c2") != args.workload or w.get("hyp_per_batch") != args.hyp_per_batch or w.get("mode") != args.mode:
        return None
    if args.workload == "c2" and (w.get("model_points") != args.model_points or w.get("scene_points") != args.scene_points):
        return None
    return t


def preprocessing_extra(ctx, model, res, scene, gs):
    """SURVEY 8(f) row 1 at the bench's sizes, whole C-ABI calls (host buffers in, host results out), best of three:
    kNN-30 curvature of the scene's tangent candidates on the scene's device copy (what tangent_mask_ needs), of all
    points of the model; cloud->resolution() of the model; the voxel grid and the pair table of a new model."""
    from triplet_match_amd import capi

    def best(fn, reps=3):
        b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            b = min(b, time.perf_counter() - t0)
        return round(b * 1e3, 3)

    S = scene["surfels"]
    cand = np.nonzero(np.linalg.norm(S[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    subset = np.nonzero(np.linalg.norm(model[:, 9:12], axis=1) > 0.7)[0].astype(np.uint32)
    out = {"scene_points": int(len(S)), "scene_tangent_candidates": int(len(cand)), "model_points": int(len(model)),
           "curvature_k": 30, "unit": "ms"}
    try:
        out["scene_candidates_curvature_ms"] = best(lambda: gs.curvature_subset(cand, 30))
        out["model_curvature_all_points_ms"] = best(lambda: ctx.cloud_curvature(model, 30))
        out["model_resolution_ms"] = best(lambda: ctx.cloud_resolution(model))
        gm2 = capi.Model(ctx, model, res)
        out["model_voxel_grid_ms"] = best(gm2.build_voxels, reps=2)
        out["model_pair_table_ms"] = best(lambda: gm2.build_pairs(subset), reps=2)
        gm2.close()
    except Exception as e:  # an extra must not take the line down
        out["error"] = str(e)
    return out


def run_extras(args, model, res, scene, gm, gs, ctx, pairs, d_poses, d_prov, n_hyp, dist_thres, accept_prob):
    """(1) early_drop_ball: the batch scored the way find_in_subset scores (include/impl/scene.hpp:273,326,
    492-506): every hypothesis on nn = radius_search_inclusive(model diameter, its first sample point) with
    the early-drop rule, through tm_score_hypotheses_early_drop_ball (host poses in, host results out: the
    PCIe copies are inside this figure). (2) e2e: tools/e2e_demo.cpp = model<Point>::init +
    scene<Point>::find_all_parallel (early-drop on, 10 ICP iterations, reference-default masks) on the same
    clouds, compiled here with g++ against libtm_hip.so."""
    import subprocess
    out = {}
    try:
        poses = np.zeros((n_hyp, 16), np.float32)
        prov = np.zeros((n_hyp, 4), np.uint32)
        ctx.d2h(poses, d_poses)
        ctx.d2h(prov, d_prov)
        centers = np.ascontiguousarray(pairs[prov[:, 0], 0], dtype=np.uint32)
        gs.score_hypotheses_early_drop_ball(gm, poses[:4096], centers[:4096], gm.diameter, dist_thres, accept_prob)
        t0 = time.perf_counter()
        cnt, _, dropped, saved = gs.score_hypotheses_early_drop_ball(gm, poses, centers, gm.diameter, dist_thres,
                                                                    accept_prob)
        dt = time.perf_counter() - t0
        out["preprocessing"] = preprocessing_extra(ctx, model, res, scene, gs)
        out["early_drop_ball"] = {"value": round(n_hyp / dt, 1), "unit": "hypotheses/s", "ms": round(dt * 1e3, 3),
                                  "hypotheses": int(n_hyp), "dropped_fraction": round(float((dropped >= 0).mean()), 4),
                                  "max_inliers": int(cnt.max()),
                                  "note": "project_ on nn(first point, model diameter) with early drop; PCIe-inclusive"}
    except Exception as exc:  # sub-figures never fail the bench
        log("extras: early_drop_ball not measured: %s" % exc)
    try:
        out["early_drop_ball_dev"] = early_drop_dev_figure(ctx, gm, gs, pairs, d_poses, d_prov, n_hyp, dist_thres, accept_prob)
    except Exception as exc:
        log("extras: early_drop_ball_dev not measured: %s" % exc)
    try:
        fx = "/tmp/tm_bench_c2_%d.bin" % os.getpid()
        with open(fx, "wb") as f:
            f.write(np.uint32(model.shape[0]).tobytes())
            f.write(model.tobytes())
            f.write(np.uint32(scene["surfels"].shape[0]).tobytes())
            f.write(scene["surfels"].tobytes())
            f.write(np.float32(res).tobytes())
            f.write(np.uint32(len(scene["poses_scene_to_model"])).tobytes())
            for T in scene["poses_scene_to_model"]:
                from triplet_match_amd import synth
                f.write(synth.pose_colmajor(T).tobytes())
        exe = "/tmp/tm_e2e_demo_%d" % os.getpid()
        libdir = os.path.join(ROOT, "triplet_match_amd")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "e2e_demo.cpp"), "-o", exe, "-L" + libdir, "-ltm_hip",
                               "-Wl,-rpath," + libdir, "-pthread"])
        # early drop on, 10 ICP iterations, the search repeated twelve times on the kept scene, two frames with a new cloud, then
        # a stream of new clouds on 1..4 host threads with a context each
        r = subprocess.run([exe, fx, "1", "10", "12", "0", "2", "6"], capture_output=True, text=True, timeout=300)
        os.unlink(fx)
        os.unlink(exe)
        import re
        m1 = re.search(r"model init ([0-9.]+) s; find_all_parallel\(early_drop=1, icp=10\) ([0-9.]+) s -> (\d+) matches", r.stdout)
        m2 = re.search(r"second call ([0-9.]+) s -> (\d+) matches", r.stdout)
        if m1:
            out["e2e_find_all_parallel"] = {
                "model_init_s": float(m1.group(1)), "find_all_parallel_s": float(m1.group(2)),
                "matches": int(m1.group(3)), "planted": int(len(scene["poses_scene_to_model"])),
                "second_call_s": float(m2.group(1)) if m2 else None,
                "steady_call_s": (lambda v: sorted(v)[len(v) // 2] if v else None)(
                    [float(x) for x in re.findall(r"further call ([0-9.]+) s", r.stdout)]),
                "new_cloud_frame_s": (lambda v: min(v) if v else None)(
                    [float(x) for x in re.findall(r"new cloud \d+: ([0-9.]+) s", r.stdout)]),
                "stream_frames_per_s": {int(t): float(f) for t, f in
                                        re.findall(r"stream: \d+ frames on (\d+) threads?, a context each: ([0-9.]+) frames/s",
                                                   r.stdout)} or None,
                "note": "drop-in scene<Point>::find_all_parallel, early drop on, 10 ICP iterations, curvature masks on; "
                        "model_init_s is the process's first use of the library: it includes the HIP start-up behind the first "
                        "context (~0.1 s; a further model<Point>::init is extras.preprocessing's 14 ms); "
                        "second call = the same search again with the prepared scene kept on the device (the first call that takes "
                        "the kept-scene route: its one-time allocations are in it); steady_call_s = median of eleven further calls; "
                        "new_cloud_frame_s = "
                        "new cloud + new scene object + search (upload, curve order, tangent_mask_ on the device); "
                        "stream_frames_per_s = such frames worked off by 1..6 host threads with a context (stream) each"}
        else:
            log("extras: e2e demo output not understood:\n" + r.stdout[-800:] + r.stderr[-800:])
    except Exception as exc:
        log("extras: e2e not measured: %s" % exc)
    return out or None


def ed_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("tm_earlydrop.hip", "tm_score.h", "tm_math.h"):
        h.update(open(os.path.join(ROOT, "triplet_match_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def early_drop_dev_figure(ctx, gm, gs, pairs, d_poses, d_prov, n_hyp, dist_thres, accept_prob, reps=3, parity=True):
    """The reference's default inner loop on the batch with everything resident (include/impl/scene.hpp:273,326,
    492-506): every hypothesis scans nn(its first sample point, model diameter) under the early-drop rule,
    tm_score_hypotheses_early_drop_ball_dev: poses in, results out on the device; only the centre indices (4 B per
    hypothesis) go up. Time = HIP events on the library's stream around the call (its two internal read-backs of a
    few words included). Bytes: L2->fabric requests of all its kernels from the committed PMC pass
    (profiles/rNN/early_drop_traffic.json, guarded by a hash of the kernel sources)."""
    prov = np.zeros((n_hyp, 4), np.uint32)
    ctx.d2h(prov, d_prov)
    centers = np.ascontiguousarray(pairs[prov[:, 0], 0], dtype=np.uint32)
    d_c, d_s, d_d, d_v = ctx.alloc(n_hyp * 4), ctx.alloc(n_hyp * 8), ctx.alloc(n_hyp * 4), ctx.alloc(n_hyp * 4)
    ms = []
    for _ in range(reps):
        ctx.synchronize()
        ctx.time_begin()
        gs.score_hypotheses_early_drop_ball_dev(gm, d_poses, n_hyp, centers, gm.diameter, dist_thres, accept_prob,
                                                d_c, d_s, d_d, d_v)
        ms.append(ctx.time_end())
    best = min(ms[1:]) if len(ms) > 1 else ms[0]
    dropped = np.zeros(n_hyp, np.int32)
    cnt = np.zeros(n_hyp, np.uint32)
    ctx.d2h(dropped, d_d)
    ctx.d2h(cnt, d_c)
    out = {"value": round(n_hyp / (best * 1e-3), 1), "unit": "hypotheses/s", "ms": round(best, 3), "ms_all": [round(x, 3) for x in ms],
           "hypotheses": int(n_hyp), "dropped_fraction": round(float((dropped >= 0).mean()), 4), "max_inliers": int(cnt.max()),
           "timing": "HIP events on the library's stream around tm_score_hypotheses_early_drop_ball_dev",
           "note": "project_ on nn(first point, model diameter) with early drop; poses and results device-resident"}
    if parity:  # the device-resident call == the host-buffer call
        poses = np.zeros((n_hyp, 16), np.float32)
        ctx.d2h(poses, d_poses)
        c2, _, d2, _ = gs.score_hypotheses_early_drop_ball(gm, poses, centers, gm.diameter, dist_thres, accept_prob)
        out["equals_host_buffer_call"] = bool(np.array_equal(c2, cnt) and np.array_equal(d2, dropped))
    try:
        t = {}
        for rnd in _profile_rounds():
            try:
                cand = json.load(open(os.path.join(ROOT, "profiles", rnd, "early_drop_traffic.json")))
            except (OSError, ValueError):
                continue
            if cand.get("source_hash") == ed_source_hash():
                t = cand
                break
        if t.get("source_hash") == ed_source_hash() and t.get("hypotheses") == int(n_hyp):
            gbs = t["fabric_bytes_per_call"] / (best * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "traffic": t["fabric_bytes_per_call"], "achieved": round(gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "frac_basis": "measured_l2_fabric_bytes: Infinity Cache + HBM, upper bound of HBM",
                               "infinity_cache_frac": round(gbs / INFINITY_CACHE_RANDOM_GBS, 4),
                               "kernel_busy_ms_under_profiler": t.get("kernel_busy_ms_per_call"),
                               "note": "bytes by kernel in profiles/rNN/early_drop_traffic.json: the continuation "
                                       "(k_ed_hist_list) streams the neighbourhood of every pending hypothesis once (scene "
                                       "entries, voxel lines, model points and refs of ~100k positions per hypothesis), the "
                                       "first step (k_ed_prefix_group) the voxel side of the first 5 % of every hypothesis"}
    except (OSError, ValueError, KeyError):
        pass
    return out


def run_cpu_baseline(args, model, res, subset, scene, gm, ctx, d_poses, d_counts, n_hyp, dist_thres,
                     accept_prob, pairs=None, d_prov=None, K=0, radius=0.0, unit_note=""):
    """Times the oracle's project_ (the CPU restatement of the reference path, kind "port") on a
    bounded sample of the same hypothesis batch, threads = hardware_concurrency() - 1 as the
    reference does (include/impl/scene.hpp:146). The voxel table is imported from the GPU build
    (their equality is what tests/test_gpu_parity.py::test_voxel_grid_bit_exact shows)."""
    from oracle import tmo
    threads = max(1, (os.cpu_count() or 2) - 1)
    om = tmo.Model(model, res)
    if K:
        om.set_metric(K, radius)  # the traits' intrinsic_distance (plane2: 2-D, cylinder: wrap-around)
    om.set_voxels(gm.voxels())
    os_ = tmo.Scene(scene["surfels"], scene["tangent_mask"])
    poses = np.zeros((n_hyp, 16), np.float32)
    ctx.d2h(poses, d_poses)
    # calibrate on a few hypotheses, then size the sample for the target time
    k0 = min(n_hyp, 2 * threads)
    t0 = time.perf_counter()
    os_.score_hypotheses(om, poses[:k0], accept_prob, dist_thres, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-4)
    k = int(min(n_hyp, max(k0, args.cpu_seconds * k0 / dt)))
    sel = np.linspace(0, n_hyp - 1, k).astype(np.int64)
    t0 = time.perf_counter()
    c_full, _, ev_full = os_.score_hypotheses(om, poses[sel], accept_prob, dist_thres, threads=threads)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    c_ed, _, ev_ed = os_.score_hypotheses(om, poses[sel], accept_prob, dist_thres, early_drop=True,
                                          threads=threads)
    t_ed = time.perf_counter() - t0
    # checker role of the oracle: the GPU's inlier counts for the sampled hypotheses must be identical
    counts = np.zeros(n_hyp, np.uint32)
    ctx.d2h(counts, d_counts)
    parity = bool(np.array_equal(counts[sel], c_full))
    if not parity:
        log("ERROR: GPU inlier counts differ from the oracle on the CPU-baseline sample")
    # like for like with what the GPU's culling leaves: the reference's own inner loop scores on
    # nn = radius_search_inclusive(model diameter, first sample point), with the early-drop rule
    ref_shaped = None
    if pairs is not None and d_prov is not None and not K:
        prov = np.zeros((n_hyp, 4), np.uint32)
        ctx.d2h(prov, d_prov)
        centers = pairs[prov[:, 0], 0]
        uniq = np.unique(centers)[:8]
        n_ref, t_ref = 0, 0.0
        for c in uniq:
            hyps = np.nonzero(centers == c)[0][:threads * 4]
            nn = os_.radius_subset(int(c), om.diameter)
            t0 = time.perf_counter()
            os_.score_hypotheses(om, poses[hyps], accept_prob, dist_thres, subset=nn, early_drop=True, threads=threads)
            t_ref += time.perf_counter() - t0
            n_ref += len(hyps)
        if t_ref > 0:
            ref_shaped = {"value": round(n_ref / t_ref, 2), "unit": "hypotheses/s", "hypotheses": int(n_ref),
                          "note": "project_ on nn(first point, model diameter) with early drop, %d first points" % len(uniq)}
    return {"value": round(k / t_full, 2), "unit": "hypotheses/s", "cores": threads, "kind": "port",
            "reference_shaped_ball_early_drop": ref_shaped,
            "sample": "%s%d of the %d hypotheses of one batch, full pass over the %d-pt scene (early drop off), %.1fs"
                      % (unit_note, k, n_hyp, scene["surfels"].shape[0], t_full),
            "gpoints_per_s": round(ev_full / t_full / 1e9, 4),
            "early_drop_on": {"value": round(k / t_ed, 2), "unit": "hypotheses/s",
                              "gpoints_per_s": round(ev_ed / t_ed / 1e9, 4), "seconds": round(t_ed, 2)},
            "max_inliers_in_sample": int(c_full.max()), "gpu_counts_equal_oracle_on_sample": parity}


if __name__ == "__main__":
    main()
