#!/usr/bin/env python3
"""Benchmark driver: simulation steps/s (+ PCG iterations/s, HBM roofline) of the blub fluid step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (N=1): scenes/corner_dams_256.json -- 968 688 particles on a 256^3 grid, dt = 1/120 s, solver defaults
(tolerance 0.1, 32 iterations, check every 4), rebinning every 60 steps: the "1M particles @ 256^3" configuration of
BASELINE.json.  A "step" is one HybridFluid::step.  Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_FILES = {256: os.path.join("profiles", "r06_pmc_dense_pcg_256.json"),   # FETCH_SIZE / WRITE_SIZE captures of the dense PCG benchmark (tools/dense_pmc.sh),
             512: os.path.join("profiles", "r06_pmc_dense_pcg_512.json")}   # stamped with the hash of the sources they were taken from


def algorithmic_bytes(kernel, F, P, A, Fb):
    """Per-launch algorithmic HBM bytes (derived from SURVEY.md 8(d)).
    F = FLUID cells, P = particles, A = cells of the active bricks the kernel sweeps (N for a dense-row kernel),
    Fb = cells of the bricks that hold fluid (N for a dense-row kernel).  f32 fields 4 B, marker / descriptor 1 B."""
    table = {
        "reset_bricks": 13 * A,                  # marker + 3 list-head volumes (the advect variant writes 5 B/cell)
        "build_lists": 16 * P + 12 * P + 12 * P, # pos read, 3 atomic exchanges, 3 next pointers
        "gather_velocity": 3 * (5 * A + 32 * P + 4 * F),   # the three components are one launch
        "divergence": Fb + 28 * F,
        "pcg_init": 6 * A + 12 * F,              # marker + descriptor + p everywhere, r rw + s on FLUID
        "pcg_dir": Fb + 12 * F,                  # descriptor, r, s read, s write
        "pcg_update": Fb + 20 * F,               # descriptor, s, p rw, r rw
        "pcg_iter": Fb + 40 * F,                 # single-reduction iteration: descriptor; r, w, q, d, p read and written
        "divergence_remove": 13 * A + 16 * F,
        "extrapolate": 13 * A,                   # marker + the three velocity volumes over the active bricks (a brick-sparse D3 reads them to decide what to write; round-5 review: A + 8 F was too kind)
        "advect": 176 * P,
        "density_gather": 5 * Fb + 16 * P + 4 * F,
        "position_change": 13 * A + 4 * F,
        "correct": 40 * P,
    }
    return float(table.get(kernel, 0))


def dense_pcg_benchmark(n=256, iterations=32, tuning=None, repeats=1):
    """M3 of BASELINE.md: SOLID shell + all-FLUID interior, b = sin*sin*sin, fixed iteration count, per-kernel HIP-event timing.
    tuning: {knob: value} for blub_fluid_set_tuning (tile-geometry sweeps, tools/dense_sweep.sh)."""
    import blub_amd
    tuning = dict(tuning or {})
    h = blub_amd.HybridFluid((n, n, n), 16, binning="off", volume_shift_kib=tuning.pop("volume_shift_kib", 0))
    for k_, v_ in tuning.items():
        h.set_tuning(k_, v_)
    marker = np.zeros((n, n, n), np.int8)
    marker[1:-1, 1:-1, 1:-1] = 1
    ax = np.sin(2 * np.pi * (np.arange(n) + 0.5) / n).astype(np.float32)
    b = (ax[:, None, None] * ax[None, :, None] * ax[None, None, :]).astype(np.float32)
    b[marker != 1] = 0
    h.write_volume("marker", marker)
    h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=iterations, error_check_frequency=4)
    dt = blub_amd.default_simulation_delta()
    N, F = n ** 3, int((marker == 1).sum())
    per_repeat = []
    for rep in range(1 + max(1, repeats)):   # first repetition warms up
        h.write_volume("residual", b)
        h.mark_pressure_initialised(0, False)
        h.profile_enable(rep >= 1)
        h.profile_reset()
        h.run_stage("solve_velocity", dt)
        h.synchronize()
        if rep >= 1:
            prof = h.profile_read()
            per_repeat.append({k: round(prof[k]["total_ms"] / prof[k]["launches"] * 1e3, 2) for k in ("pcg_dir", "pcg_update")})
    h.profile_enable(False)
    out = {}
    for k in ("pcg_dir", "pcg_update"):
        avg_ms = prof[k]["total_ms"] / prof[k]["launches"]
        gbs = algorithmic_bytes(k, F, 0, N, N) / (avg_ms * 1e-3) / 1e9
        out[k] = {"avg_us": round(avg_ms * 1e3, 2), "launches": prof[k]["launches"], "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    iter_us = sum(out[k]["avg_us"] for k in out)
    err, iters = h.solver_stats(0)
    h.close()
    pmc, pmc_file = {}, PMC_FILES.get(n)
    try:   # HBM bytes per launch from the committed PMC capture of this same benchmark (rocprofv3 cannot run inside bench.py)
        pmc = json.load(open(os.path.join(ROOT, pmc_file)))
    except (OSError, ValueError, TypeError):
        pass
    from blub_amd.build import source_hash
    stamp, now = pmc.get("kernel_source_sha16"), source_hash()
    for k in out:
        out[k]["traffic_bytes_pmc"] = pmc.get(k, {}).get("traffic")
        out[k]["traffic_source"] = ("committed rocprofv3 --pmc capture of this benchmark: %s (counters cannot be read from inside bench.py)" % pmc_file) if out[k]["traffic_bytes_pmc"] else None
        # the capture is a file: say whether it was taken from the sources this run was built from
        out[k]["traffic_capture_matches_sources"] = (stamp == now) if out[k]["traffic_bytes_pmc"] else None
        if out[k]["traffic_bytes_pmc"] and stamp != now:
            out[k]["traffic_warning"] = "PMC capture stamped %s, sources are %s: traffic is from an older code state" % (stamp, now)
        out[k]["algorithmic_bytes"] = algorithmic_bytes(k, F, 0, N, N)
    # one iteration = pcg_dir + pcg_update.  "iter_bytes" is SURVEY 8(d)'s figure for the UNFUSED three-phase iteration
    # (3N + 36F); the fused pair itself only has to move 2N + 32F ("iter_bytes_fused"), both fractions are reported.
    fused = 2 * N + 32 * F
    return {"grid": "%d^3" % n, "fluid_cells": F, "iterations": iters, "us_per_iteration_kernels": round(iter_us, 1), "kernel_source_sha16": now, "per_repeat_avg_us": per_repeat if repeats > 1 else None,
            "iter_bytes": 3 * N + 36 * F, "iter_GBs": round((3 * N + 36 * F) / (iter_us * 1e-6) / 1e9, 1),
            "iter_frac": round((3 * N + 36 * F) / (iter_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "iter_bytes_fused": fused, "iter_frac_fused": round(fused / (iter_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "kernels": out}


def transfer_microbenchmark(n=256, seed=1234, tune=()):
    """M4 of BASELINE.md / SURVEY 8(d): n^3 grid, 8 jittered particles in every interior cell of the lower half, smooth
    velocity field; the transfer kernels (list building + P2G gathers), advection and the density gather are timed per
    kernel class with HIP events, with the particles in random order and again after the engine's own binning pass."""
    import blub_amd
    rng = np.random.default_rng(seed)
    xs, ys, zs = np.arange(1, n - 1), np.arange(1, n // 2), np.arange(1, n - 1)
    cells = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pos = np.repeat(cells, 8, axis=0)
    pos += rng.random(pos.shape, dtype=np.float32)
    P = pos.shape[0]
    perm = rng.permutation(P)
    pos = pos[perm]
    vel = []
    for c, fn in enumerate((np.sin, np.cos, None)):
        rows = np.zeros((P, 4), np.float32)
        if fn is not None:
            rows[:, 3] = fn(pos[:, c] * 0.1) * 3.0
        vel.append(rows)
    dt = blub_amd.default_simulation_delta()
    h = blub_amd.HybridFluid((n, n, n), P)
    for kv in tune:
        k_, v_ = kv.split("=")
        h.set_tuning(k_, int(v_))
    h.set_gravity_grid((0.0, -9.81 * n / 1.28, 0.0))
    h.set_particles(pos, *vel)
    del pos, vel, cells
    N = n ** 3
    out = {"grid": "%d^3" % n, "particles": P}

    def timed(label):
        res = {}
        for rep in range(2):   # first repetition warms up
            h.profile_enable(rep == 1)
            h.profile_reset()
            h.run_stage("transfer", dt)
            h.run_stage("divergence", dt)
            h.run_stage("advect", dt)
            h.run_stage("density_gather", dt)
            h.synchronize()
        prof = h.profile_read()
        h.profile_enable(False)
        F = int((h.read_volume("marker") == 1).sum())
        bc = h.brick_counts()
        A, Fb = bc["active"] * bc["cells_per_brick"], bc["fluid"] * bc["cells_per_brick"]
        for k in ("build_lists", "gather_velocity", "advect", "density_gather", "reset_bricks"):
            if k not in prof:
                continue
            per = 2 if k == "reset_bricks" else 1      # (the transfer's and the advection's reset: two different launches, each moves its own bytes)
            avg_ms = prof[k]["total_ms"] / per         # the CLASS per pass -- the list-centric gather is two launches (walk + finishing kernel) for ONE set of algorithmic bytes
            gbs = algorithmic_bytes(k, F, P, A, Fb) / (avg_ms * 1e-3) / 1e9
            res[k] = {"avg_us": round(avg_ms * 1e3, 1), "launches": prof[k]["launches"], "algorithmic_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        res["fluid_cells"] = F
        out[label] = res
    timed("random_order")
    h.step_counter = 0
    h.run_stage("binning", dt)
    timed("after_binning")
    h.close()
    return out


def cpu_baseline(scene_path, dt, steps=12, budget_s=25.0):
    """The CPU oracle (a restatement of the reference, kind "port": pinned bit for bit against the reference's own shaders,
    tests/test_oracle_vs_ref.py) on the same scene, timed as SURVEY 8(d) prescribes: two warm-up steps (the first holds the step-0
    rebinning), then every step timed by itself with a steady clock, value = 1 / median over >= 10 steps (fewer only if the wall-clock
    budget runs out first); all usable cores (OpenMP), nproc and the thread count recorded."""
    import blub_amd
    from oracle.oracle import Oracle, num_threads
    sc = blub_amd.Scene.parse(path=scene_path).config
    dim = list(sc.grid_dimension)
    o = Oracle(dim[0], dim[1], dim[2], sc.max_num_particles)
    scale = np.float32(sc.grid_to_world_scale)
    for i in range(sc.num_fluid_cubes):
        o.add_fluid_cube(np.float32(list(sc.cube_min[i])) / scale, np.float32(list(sc.cube_max[i])) / scale)
    o.set_gravity_grid(np.float32(list(sc.gravity)) / scale)
    for _ in range(2):
        o.step(dt)
    it0, s0 = o.solver_totals()
    times = []
    t_start = time.perf_counter()
    while len(times) < steps and (len(times) < 3 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        o.step(dt)
        times.append(time.perf_counter() - t0)
    it1, s1 = o.solver_totals()
    threads = num_threads()
    med = float(np.median(times))
    return {"value": round(1.0 / med, 4), "unit": "steps/s", "cores": threads, "kind": "port", "nproc": os.cpu_count(),
            "sample": "median of %d individually timed steps of the same scene after 2 warm-up steps (oracle/libbluboracle.so, OpenMP, %d threads; min %.3f / max %.3f s per step)"
                      % (len(times), threads, min(times), max(times)),
            "mean_steps_per_s": round(len(times) / sum(times), 4), "pcg_iters_per_sec": round((it1 - it0) / max(s1 - s0, 1e-9), 2)}


def fallback_to_replicas(reason):
    """N > 1 only: re-execute this rank in replicas mode (same PID, so the launcher keeps tracking it).  Used when the z-slab
    group fails or stalls on one rank: a rank that raised would otherwise leave its peers blocked inside a transport operation.
    Every rank ends up here (the failing one at once, its peers through the watchdog) and they meet again on MASTER_PORT + 17.
    The line printed in that mode is NOT a scaling result: "value" is null, "scaling" is "fallback-replicas", the reason is kept."""
    sys.stderr.write("rank %s: leaving the z-slab path (%s); re-running as independent replicas\n" % (os.environ.get("RANK", "0"), reason))
    sys.stderr.flush()
    env = dict(os.environ)
    env["BLUB_BENCH_REPLICAS"] = "1"
    env["BLUB_BENCH_FALLBACK_REASON"] = str(reason)[:400]
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"   # rank 0 hosts a fresh store there (the launcher's own store keeps the keys of the first rendezvous)
    os.execve(sys.executable, [sys.executable] + sys.argv, env)


def fallback_to_rccl(reason):
    """N > 1, direct transport only: a failure or stall of the peer-store transport gets ONE second attempt over RCCL before the replicas
    fallback (same mechanism: every rank re-executes itself and they meet again on MASTER_PORT + 17)."""
    sys.stderr.write("rank %s: the direct transport failed (%s); re-running over RCCL\n" % (os.environ.get("RANK", "0"), reason))
    sys.stderr.flush()
    env = dict(os.environ)
    env["BLUB_BENCH_TRANSPORT"] = "rccl"
    env["BLUB_BENCH_DIRECT_FAILED"] = str(reason)[:400]
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
    os.execve(sys.executable, [sys.executable] + sys.argv, env)


METRIC = "simulation steps/sec, 1M particles @ 256^3 grid"


def roofline_object(dense, where):
    """The dominant kernel of the dense PCG iteration (the update kernel: N + 20F of the 2N + 32F the fused pair moves), timed live with HIP
    events on the engine's own stream; the direction kernel rides along as `second_kernel`."""
    ku, kd = dense["kernels"]["pcg_update"], dense["kernels"]["pcg_dir"]
    obj = {"bound": "hbm", "kernel": "k_pcg_update_z (PCG stencil update, dense %s micro-benchmark M3, %s)" % (dense["grid"], where), "achieved": ku["achieved_GBs"],
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ku["frac"], "traffic": ku["traffic_bytes_pmc"],
           "traffic_source": ku.get("traffic_source"), "traffic_capture_matches_sources": ku.get("traffic_capture_matches_sources"),
           "algorithmic_bytes": ku["algorithmic_bytes"], "avg_us": ku["avg_us"], "launches": ku["launches"],
           "second_kernel": {"kernel": "k_pcg_dir_z", "achieved": kd["achieved_GBs"], "frac": kd["frac"], "avg_us": kd["avg_us"], "algorithmic_bytes": kd["algorithmic_bytes"],
                             "traffic": kd["traffic_bytes_pmc"]},
           "iteration_frac_fused_pair": dense["iter_frac_fused"]}
    if ku.get("traffic_warning"):
        obj["traffic_warning"] = ku["traffic_warning"]
    return obj


def shared_device_queue_limit(world):
    """Development boxes only: when the ranks of a job outnumber the GPUs, several processes share a device and every process asks for its own four
    hardware queues; beyond the device's queue slots the scheduler rotates them on a timer, and kernels that spin on words another process' kernels
    write (the direct transport) only make progress when the timer fires.  GPU_MAX_HW_QUEUES must be in the environment before the HIP runtime
    starts, i.e. before `import torch` (profiles/r05_multiproc_direct.jsonl: what it changes)."""
    if world <= 1:
        return None
    # (the GPUs THIS job can see -- not the host's: a container's kfd topology lists every GPU of the machine.  Asked in a child process: the answer must be
    #  known before this process makes its first HIP call, which is when the runtime reads GPU_MAX_HW_QUEUES)
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import ctypes; h = ctypes.CDLL('libamdhip64.so'); n = ctypes.c_int(0); h.hipGetDeviceCount(ctypes.byref(n)); print(n.value)"],
                             capture_output=True, text=True, timeout=60)
        gpus = int(out.stdout.strip() or "0")
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return None
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if gpus and local_world > gpus and os.environ.get("GPU_MAX_HW_QUEUES"):
        os.environ["BLUB_BENCH_SHARED_DEVICE"] = "1"      # (the caller chose the queue limit itself)
        return None
    if gpus and local_world > gpus:
        # measured with 4 processes on one MI355X (profiles/r05_multiproc_direct.jsonl): the runtime's default of 4 hardware queues per process and 2 both
        # run the direct transport at the same speed (984 / 987 steps/s); with ONE queue per process a flag wait times out (the run falls back to RCCL)
        os.environ["GPU_MAX_HW_QUEUES"] = "2"
        os.environ["BLUB_BENCH_SHARED_DEVICE"] = "1"
        return "2"
    return None


def slab_run(args, torch, dist, rank, world, dev, ctl, transport, scene_name, steps, warmup, scaling, memory="coarse"):
    """One z-slab group over the ranks of the job stepping `scene_name`: creation (fluid-weighted cut planes unless BLUB_BENCH_CUTS=uniform), warm-up,
    the timed window bracketed by barriers, one more step to count transport operations.  Raises on any failure (the caller decides what a failure
    costs); every rank returns the same timing, rank-local details under "local"."""
    import blub_amd
    from blub_amd import slab_scene
    dt = blub_amd.default_simulation_delta()
    scene_path = os.path.join(ROOT, "scenes", scene_name + ".json")
    ok = torch.ones(1, device=ctl)
    group, err = None, ""
    res = {}
    try:
        cfg = blub_amd.Scene.parse(path=scene_path).config
        if scaling == "weak":
            dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, world)
            res["workload"] = "%s stacked x%d along z (weak scaling: every slab is one copy)" % (scene_name, world)
        else:
            dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
            res["workload"] = "%s, ONE domain cut into %d z-slabs (strong scaling)" % (scene_name, world)
        pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
        P = len(pos)
        # Cut planes: every rank holds the same particles here, so every rank derives the same cuts.  Weighted (default): every slab starts with
        # about 1/N of the FLUID bricks (round-4 review: uniform cuts of the metric's scene leave six of eight ranks without fluid).
        uniform = [blub_amd.SlabGroup.slab_range(dim[2], world, i)[0] for i in range(world)] + [dim[2]]
        # dynamic (default): weighted at t = 0, then blub_slab_group_rebalance every BLUB_BENCH_REBALANCE_EVERY (16) steps INSIDE the timed loop -- cuts
        # balanced for t = 0 go stale within ~30 steps of a dam break (profiles/r05_slab_cuts_uniform_vs_weighted.jsonl); the slabs then hold the whole grid
        cuts_mode = os.environ.get("BLUB_BENCH_CUTS", "dynamic" if scaling == "strong" else "uniform")
        dynamic = cuts_mode == "dynamic"
        rebalance_every = int(os.environ.get("BLUB_BENCH_REBALANCE_EVERY", "16"))
        cuts = blub_amd.SlabGroup.balanced_cuts(dim, pos, world)[0] if cuts_mode in ("weighted", "dynamic") else uniform
        res.update(grid=list(dim), particles=P, dt=dt, cuts=list(cuts), cuts_mode=cuts_mode,
                   fluid_bricks_per_rank=blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, cuts),
                   fluid_bricks_per_rank_uniform_cuts=blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, uniform))
        # capacity per slab: the whole particle set (strong scaling: a slab may come to own all of it)
        if transport == "loopback":
            if rank == 0:
                group = blub_amd.SlabGroup(dim, P + 64, local=world, device=dev, cuts=cuts, memory=memory, movable_cuts=dynamic)
        else:
            group = blub_amd.SlabGroup.from_torch_distributed(dim, P + 64, device=dev, cuts=cuts, memory=memory, movable_cuts=dynamic)
        if group is not None and transport == "direct":
            # peer-mapped slabs over hipIpc, kernels store into the neighbours' memory themselves (after the probe, or forced)
            if not group.connect_direct_over_torch_distributed():
                sys.stderr.write("rank %d: hipIpc mapping unavailable on some rank; staying on the RCCL transport\n" % rank)
        if group is not None and os.environ.get("BLUB_BENCH_SHARED_DEVICE") and transport != "direct":
            # (development box: several ranks on one GPU -- the one-launch brick-list build waits for co-resident workgroups of ITS process and sits out its
            #  bound while another process holds the CUs; the two-kernel build has no such wait.  Over the direct transport the library notices the shared
            #  device by itself when it maps a peer: blub_slab_group_connect)
            for i in range(group.num_local()):
                group.local_fluid(i).set_tuning("spin_free", 1)
        if group is not None:
            group.set_gravity_grid(gravity)
            if args.pcg_schedule != "default":
                group.set_pcg_schedule(args.pcg_schedule)   # (every rank passes the same flag; "default" = the library's own choice, no call)
            group.set_particles(pos)
        del pos
    except Exception as e:   # all ranks must take the same path
        err = "%s: %s" % (type(e).__name__, e)
        sys.stderr.write("rank %d: z-slab group unavailable (%s)\n" % (rank, err))
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if float(ok.item()) == 0.0:
        if group is not None:
            group.close()
        raise RuntimeError("z-slab group could not be created on every rank" + (" (%s)" % err if err else ""))
    active = group is not None

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
        if active:
            group.synchronize()
    def all_agree_ok(local_ok):
        """collective: did every rank get through without a (recoverable) transport error?"""
        flag = torch.tensor([0.0 if local_ok else 1.0], device=ctl)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return float(flag.item()) == 0.0

    def guarded(fn):
        """A timed-out wait of the direct transport (BLUB_ERR_COMM: a peer seconds late) is recoverable in place; anything else is not."""
        try:
            fn()
            return True
        except blub_amd.BlubError as e:
            if e.status != -8 or transport != "direct":
                raise
            sys.stderr.write("rank %d: %s\n" % (rank, e))
            return False
    res["recovered_in_place"] = 0
    res["recuts"] = 0
    def one_step():
        # (a re-balance is part of stepping this domain on N GPUs: it sits inside the timed loop, one host synchronisation every `rebalance_every` steps.
        #  The cadence follows the GROUP's step counter, which an in-place recovery puts back to the same restored step on every rank -- a per-process
        #  counter of loop iterations ran apart when the ranks left a failed window at different iterations, and the collective re-balance with it:
        #  round-5 ADVICE.)
        if active and dynamic:
            at = group.local_fluid(0).step_counter
            if at > 0 and at % rebalance_every == 0:
                res["recuts"] += int(group.rebalance(min_layers=2))
        if active:
            group.step(dt)
    try:
        if active and transport == "direct":
            group.set_checkpoint_interval(int(os.environ.get("BLUB_BENCH_CHECKPOINT_INTERVAL", "16")))      # (one pass over the particles + pressure every 16th step)
        attempts = 0
        stall = os.environ.get("BLUB_BENCH_STALL", "").split(":") if os.environ.get("BLUB_BENCH_STALL") else None      # "rank:step:seconds"
        while True:
            def window():
                for _ in range(warmup):
                    one_step()
                if transport == "direct" and os.environ.get("BLUB_BENCH_FAIL_DIRECT"):      # (test hook: the second attempt over RCCL)
                    raise RuntimeError("injected failure of the direct transport")
                if active:
                    group.synchronize()
            ok_w = guarded(window)
            dist.barrier()
            torch.cuda.synchronize()
            fluid0 = group.local_fluid(0) if active else None
            it0 = fluid0.total_solver_iterations() if active else 0
            t0 = time.perf_counter()

            def timed():
                for k in range(steps):
                    if stall and attempts == 0 and rank == int(stall[0]) and k == int(stall[1]):      # (test hook: this rank falls seconds behind once)
                        time.sleep(float(stall[2]))
                    one_step()
                if active:
                    group.synchronize()
            ok_t = guarded(timed) if ok_w else False
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            if all_agree_ok(ok_w and ok_t):
                break
            attempts += 1
            if attempts > 2 or transport != "direct":
                raise RuntimeError("the direct transport timed out again after %d in-place recoveries" % (attempts - 1))
            # round-4 review, item 5: one late peer must not end in `value: null` -- every rank goes back to the newest checkpoint all of them hold and
            # the window is run again from there
            if active:
                back = group.recover_over_torch_distributed()
                if rank == 0:
                    sys.stderr.write("bench.py: a wait of the direct transport timed out; recovered in place to step %d, running the window again\n" % back)
            res["recovered_in_place"] = attempts
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res["elapsed"] = float(t.item())
        dist.barrier()
        if active:
            it1 = fluid0.total_solver_iterations()
            ops0 = group.transport_ops()
            group.step(dt)
            group.synchronize()
            res.update(pcg_iters_per_step=round((it1 - it0) / steps, 2), transport_ops_per_step=round(group.transport_ops() - ops0, 1),
                       transport=group.transport_description(), transport_kind=group.transport(),
                       particles_per_local_slab=[group.local_fluid(i).num_particles() for i in range(group.num_local())], cuts_at_end=group.cuts())
        dist.barrier()
    finally:
        if active:
            group.close()
    res["active"] = active
    return res


def multi_gpu(args, torch, dist, rank, world, dev, ctl):
    """N > 1: ONE domain cut into z-slabs.  --scaling strong (default): the scene itself (the 256^3 / 1 M particle
    domain of the metric, BASELINE configs 4-5) split over N slabs, value = steps/s of that one domain; the line also carries `secondary`:
    corner_dams_512 (512^3, 8 M particles: BASELINE configs[4], the size the decomposition is for) cut the same way.  --scaling weak: N copies
    stacked along z, value = steps/s of the N-times larger domain.  Transport: RCCL (one slab per rank / GPU) or, after a probe, the direct one;
    BLUB_BENCH_TRANSPORT=loopback keeps all N slabs on rank 0's GPU (development on a 1-GPU box: the other ranks only take part in the control
    plane).  Cut planes: fluid-weighted unless BLUB_BENCH_CUTS=uniform."""
    import threading
    # auto (default): the direct transport when a probe on THIS node's GPUs says it works (blub_amd/direct_probe.py: child processes, so a bad
    # peer mapping costs the probe and not the job), RCCL otherwise; rccl / direct force one (direct without the probe); loopback: see above
    transport = os.environ.get("BLUB_BENCH_TRANSPORT", "auto")
    probe, memory = None, os.environ.get("BLUB_BENCH_SLAB_MEMORY", "coarse")      # (slab memory mode: what the probe found to work, or forced)
    if transport == "auto":
        from blub_amd import direct_probe
        try:
            ok_probe, why, memory = direct_probe.run(rank, world, dev)
        except Exception as e:   # (a rendezvous problem inside the probe itself: all ranks see it)
            ok_probe, why, memory = False, "%s: %s" % (type(e).__name__, e), None
        probe = {"passed": bool(ok_probe), "detail": why, "slab_memory": memory,
                 "checks": "121 + 3 x 121 PCG iterations, RCCL vs direct, per-iteration scalars and final pressure bit for bit on every rank; 3 free-running steps vs the single domain"}
        if rank == 0:
            sys.stderr.write("direct-transport probe: %s\n" % ("passed" if ok_probe else "failed (%s); using RCCL" % why))
        transport = "direct" if ok_probe else "rccl"
        memory = memory or "coarse"
    if os.environ.get("BLUB_BENCH_DIRECT_FAILED"):
        probe = {"passed": True, "detail": "the probe passed but the run over the direct transport failed (%s); this line is the second attempt, over RCCL" % os.environ["BLUB_BENCH_DIRECT_FAILED"]}
    bail = fallback_to_rccl if transport == "direct" else fallback_to_replicas
    watchdog = None
    if transport != "loopback":
        watchdog = threading.Timer(float(os.environ.get("BLUB_BENCH_SLAB_DEADLINE", "180")), bail, args=("no progress within the deadline",))
        watchdog.daemon = True
        watchdog.start()
    try:
        res = slab_run(args, torch, dist, rank, world, dev, ctl, transport, args.scene, args.steps, args.warmup, args.scaling, memory)
    except Exception as e:
        if watchdog is not None:
            watchdog.cancel()
        bail("z-slab run failed: %s" % e)
    if watchdog is not None:
        watchdog.cancel()
    elapsed = res["elapsed"]
    line = None
    if res["active"]:
        if transport == "loopback":
            parallelism = "%d z-slabs EMULATED on one GPU (loopback transport, rank 0 only): protocol cost without a wire, not a scaling result" % world
        else:
            over = "peer-mapped memory (hipIpc)" if res["transport_kind"] == "direct" else "RCCL"
            if os.environ.get("BLUB_BENCH_SHARED_DEVICE"):      # (more ranks than GPUs: a development box)
                parallelism = "z-slab decomposition over %s: %d slabs in %d processes SHARING fewer GPUs than ranks -- the protocol between processes, not a scaling result" % (over, world, world)
            else:
                parallelism = "z-slab decomposition over %s: %d slabs, 1 rank per GPU" % (over, world)
        line = {
            "metric": METRIC, "value": round(args.steps / elapsed, 3), "unit": "steps/s (global steps of the whole domain)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling if transport != "loopback" else args.scaling + "-emulated-on-one-gpu", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": res["workload"], "grid": res["grid"], "particles": res["particles"], "dt": res["dt"], "solver": "tol 0.1 / 32 it / check 4", "rebinning": 60,
                       "parallelism": parallelism, "pcg_schedule": args.pcg_schedule if args.pcg_schedule != "default" else "single_reduction (library default)",
                       "slab_cuts": res["cuts"], "slab_cuts_mode": res["cuts_mode"], "slab_cuts_at_end": res.get("cuts_at_end"), "recuts_in_run": res["recuts"]},
            "fluid_bricks_per_rank": res["fluid_bricks_per_rank"], "fluid_bricks_per_rank_uniform_cuts": res["fluid_bricks_per_rank_uniform_cuts"],
            "pcg_iters_per_step": res["pcg_iters_per_step"], "transport_ops_per_step": res["transport_ops_per_step"],
            "transport": res["transport"], "recovered_in_place": res["recovered_in_place"], "direct_transport_probe": probe, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "roofline": None, "cpu_baseline": None}
    # ---- secondary: the size the decomposition is for (BASELINE configs[4]): corner_dams_512, strong scaling over the same ranks.  A failure or stall
    # here costs only this object: the primary result above is printed either way.
    secondary = None
    want_secondary = args.scaling == "strong" and args.scene == "corner_dams_256" and not args.no_secondary and os.environ.get("BLUB_BENCH_NO_SECONDARY", "0") in ("", "0")
    if want_secondary:
        def give_up():
            if rank == 0 and line is not None:
                line["secondary"] = {"workload": "corner_dams_512", "error": "no progress within the deadline"}
                print(json.dumps(line))
                sys.stdout.flush()
            os._exit(0)
        wd2 = None
        if transport != "loopback":
            wd2 = threading.Timer(float(os.environ.get("BLUB_BENCH_SLAB_DEADLINE", "180")), give_up)
            wd2.daemon = True
            wd2.start()
        s_steps, s_warm = max(4, min(args.steps, 30)), max(2, min(args.warmup, 5))
        try:
            r2 = slab_run(args, torch, dist, rank, world, dev, ctl, transport, "corner_dams_512", s_steps, s_warm, "strong", memory)
            if r2["active"]:
                secondary = {"workload": r2["workload"], "grid": r2["grid"], "particles": r2["particles"], "value": round(s_steps / r2["elapsed"], 3), "unit": "steps/s",
                             "ms_per_step": round(r2["elapsed"] / s_steps * 1e3, 4), "steps": s_steps, "warmup": s_warm, "single_gpu_reference": "bench.py --scene corner_dams_512 (profiles/r06_other_scenes.txt)",
                             "slab_cuts": r2["cuts"], "slab_cuts_at_end": r2.get("cuts_at_end"), "recuts_in_run": r2["recuts"], "fluid_bricks_per_rank": r2["fluid_bricks_per_rank"], "fluid_bricks_per_rank_uniform_cuts": r2["fluid_bricks_per_rank_uniform_cuts"],
                             "pcg_iters_per_step": r2["pcg_iters_per_step"], "transport_ops_per_step": r2["transport_ops_per_step"], "transport": r2["transport"]}
        except Exception as e:
            secondary = {"workload": "corner_dams_512", "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if wd2 is not None:
                wd2.cancel()
            if rank == 0 and line is not None:      # (the other ranks may be blocked in a transport operation of the failed run: do not wait for them)
                line["secondary"] = secondary
                print(json.dumps(line))
                sys.stdout.flush()
            os._exit(0)
        if wd2 is not None:
            wd2.cancel()
    if rank == 0 and line is not None:
        line["secondary"] = secondary
        if not args.no_dense_pcg:   # the roofline kernel is a single-GPU micro-benchmark: rank 0 runs it while the others wait
            line["roofline"] = roofline_object(dense_pcg_benchmark(256, 32), "rank 0")
        print(json.dumps(line))
        sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="corner_dams_256")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1: split ONE domain (default) or stack N copies along z")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-pcg", action="store_true")
    ap.add_argument("--no-dense-512", action="store_true", help="skip the 512^3 repetition of the dense PCG micro-benchmark (roofline_512)")
    ap.add_argument("--no-other-schedule", action="store_true", help="skip the second window with the other PCG schedule (kernel traces of ONE schedule: tools/kstats.sh)")
    ap.add_argument("--tune", action="append", default=[], help="name=value for blub_fluid_set_tuning on every scene of the run (A/B measurements)")
    ap.add_argument("--pcg-schedule", default="default", choices=["default", "single_reduction", "reference"],
                    help="schedule of the headline window: the LIBRARY'S default (no call at all) unless named; the other one is timed beside it")
    ap.add_argument("--no-fast-forward", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: skip the corner_dams_512 strong-scaling run that rides along as `secondary`")
    ap.add_argument("--profile-steps", type=int, default=1000000, help="steps of the instrumented pass (a fresh scene, same warm-up; default: the whole timed window; 0: skip)")
    ap.add_argument("--dense-only", action="store_true", help="only run the dense PCG micro-benchmark (tuning)")
    ap.add_argument("--dense-size", type=int, default=256)
    ap.add_argument("--dense-tile-quads", type=int, default=0, help="--dense-only: tile width of the dense PCG kernels (256 | 512 | 1024 quads; tuning)")
    ap.add_argument("--dense-tile-planes", type=int, default=0, help="--dense-only: planes marched per tile (tuning)")
    ap.add_argument("--volume-shift-kib", type=int, default=0, help="--dense-only: blub_fluid_desc::volume_shift_kib (0 = default 64, -1 = one allocation per volume; placement study)")
    ap.add_argument("--dense-repeats", type=int, default=1, help="--dense-only: timed repetitions of the solve on the same allocation (variance study)")
    ap.add_argument("--dense-grid", type=int, default=0, help="--dense-only: launch grid of the dense PCG kernels (tuning)")
    ap.add_argument("--pcg-mapping", default="auto", choices=["auto", "rows", "bricks", "bricks_staged"], help="work mapping of the PCG kernels (tuning)")
    ap.add_argument("--transfer-only", action="store_true", help="only run the 256^3 transfer micro-benchmark M4 (65 M particles)")
    args = ap.parse_args()

    hwq = shared_device_queue_limit(int(os.environ.get("WORLD_SIZE", "1")))      # (before the HIP runtime starts)
    import torch
    import blub_amd
    if hwq and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("bench.py: more ranks than GPUs on this node: GPU_MAX_HW_QUEUES=%s\n" % hwq)

    if args.dense_only:
        tuning = {k_: v_ for k_, v_ in (("dense_tile_quads", args.dense_tile_quads), ("dense_tile_planes", args.dense_tile_planes), ("dense_grid", args.dense_grid)) if v_}
        if args.volume_shift_kib:
            tuning["volume_shift_kib"] = args.volume_shift_kib
        for kv in args.tune:
            tuning[kv.split("=")[0]] = int(kv.split("=")[1])
        res = dense_pcg_benchmark(args.dense_size, 32, tuning, repeats=args.dense_repeats)
        res["tuning"] = tuning
        print(json.dumps(res))
        return
    if args.transfer_only:
        print(json.dumps(transfer_microbenchmark(256, tune=args.tune)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    ctl = "cpu"
    force_slab = bool(os.environ.get("BLUB_BENCH_FORCE_SLAB"))   # development: run the z-slab path (RCCL transport) with a single rank
    if world > 1 or force_slab:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank %= max(1, torch.cuda.device_count())   # (development: several ranks on the one GPU of the test box)
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("BLUB_BENCH_BACKEND", "nccl")   # control-plane collectives of this script only
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        ctl = "cuda" if backend == "nccl" else "cpu"
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    replicas = world > 1 and bool(os.environ.get("BLUB_BENCH_REPLICAS"))
    if (world > 1 or force_slab) and not replicas:
        return multi_gpu(args, torch, dist, rank, world, dev, ctl)

    scene_path = os.path.join(ROOT, "scenes", args.scene + ".json")
    dt = blub_amd.default_simulation_delta()

    def new_scene(schedule):
        sc_ = blub_amd.Scene(path=scene_path, device=dev)
        fl_ = sc_.fluid()
        fl_.set_pcg_work_mapping(args.pcg_mapping)
        if schedule != "default":      # the headline window makes NO such call: value = what blub_fluid_create + blub_fluid_step do by themselves
            fl_.set_pcg_schedule(schedule)
        for kv in args.tune:
            k_, v_ = kv.split("=")
            fl_.set_tuning(k_, int(v_))
        return sc_, fl_

    # The headline window runs the library exactly as blub_fluid_create leaves it (round-3 review: `value` must be what a drop-in caller gets).
    # Since round 4 that default is the single-reduction form of the PCG recurrence (one kernel per iteration; include/blubhip.h:
    # blub_fluid_set_pcg_schedule); the same window is timed again with the reference's literal two-reduction order: both numbers are printed.
    scene, fluid = new_scene(args.pcg_schedule)
    headline_schedule = fluid.pcg_schedule()
    step, sync = (lambda: scene.step(dt)), fluid.synchronize
    nx, ny, nz = fluid.grid_dimension()
    N = nx * ny * nz
    P = fluid.num_particles()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        sync()

    for _ in range(args.warmup):
        step()
    barrier()
    it0 = fluid.total_solver_iterations()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    it1 = fluid.total_solver_iterations()

    if replicas:
        # NOT a scaling result: every rank stepped its own independent copy because the z-slab path was unavailable.
        if rank == 0:
            print(json.dumps({
                "metric": METRIC, "value": None, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "fallback-replicas", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": args.scene, "grid": [nx, ny, nz], "particles": P, "dt": dt, "solver": "tol 0.1 / 32 it / check 4", "rebinning": 60,
                           "parallelism": "FALLBACK: %d independent replicas, one per GPU -- the z-slab decomposition did not run" % world},
                "fallback_reason": os.environ.get("BLUB_BENCH_FALLBACK_REASON", "BLUB_BENCH_REPLICAS set by the caller"),
                "replica_steps_per_sec_each": round(args.steps / elapsed, 3), "roofline": None, "cpu_baseline": None}))
            sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
        return

    def timed_window(schedule, rebinning=None):
        """A fresh scene, the same warm-up and the same K steps, timed like the headline window."""
        sc_, fl_ = new_scene(schedule)
        if rebinning is not None:
            fl_.particle_rebinning_step_frequency = rebinning
        for _ in range(args.warmup):
            sc_.step(dt)
        fl_.synchronize()
        i0 = fl_.total_solver_iterations()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            sc_.step(dt)
        fl_.synchronize()
        el = time.perf_counter() - t0_
        i1 = fl_.total_solver_iterations()
        fl_.close()
        return el, i1 - i0

    # ---- the representative window beside a short one (round-4 review, hygiene): the driver passes 5 + 20, which sees only the scene's cheapest phase; the
    # no-flag default (10 + 120) costs 0.1 s of stepping, so it rides along whenever the timed window is shorter
    representative = None
    if (args.steps < 120 or args.warmup < 10) and args.scene == "corner_dams_256" and not args.no_other_schedule:
        sc_r, fl_r = new_scene(args.pcg_schedule)
        for _ in range(10):
            sc_r.step(dt)
        fl_r.synchronize()
        i0r, t0r = fl_r.total_solver_iterations(), time.perf_counter()
        for _ in range(120):
            sc_r.step(dt)
        fl_r.synchronize()
        el_r = time.perf_counter() - t0r
        representative = {"steps_per_s": round(120 / el_r, 3), "ms_per_step": round(el_r / 120 * 1e3, 4), "warmup": 10, "steps": 120,
                          "pcg_iters_per_step": round((fl_r.total_solver_iterations() - i0r) / 120, 2),
                          "note": "the same library and scene over the no-flag default window (10 warm-up + 120 timed steps: break, spread, first slosh); `value` above is the window named in config.window"}
        fl_r.close()

    # ---- the same window with the OTHER schedule (round-2 review: the cost of the literal order of operations must be visible)
    other = "reference" if headline_schedule == "single_reduction" else "single_reduction"
    el_o, it_o = timed_window(other) if not args.no_other_schedule else (float("nan"), 0)
    by_schedule = {headline_schedule: {"steps_per_s": round(args.steps / elapsed, 3), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "pcg_iters_per_step": round((it1 - it0) / args.steps, 2)},
                   other: ({"steps_per_s": round(args.steps / el_o, 3), "ms_per_step": round(el_o / args.steps * 1e3, 4), "pcg_iters_per_step": round(it_o / args.steps, 2)}
                           if not args.no_other_schedule else {"steps_per_s": None, "ms_per_step": None, "pcg_iters_per_step": None})}

    # ---- fast-forward through the native scheduler (simulation_controller.rs:96-157): the reference's own way of timing steps.
    # Same window as the timed region above (a fresh scene, the same warm-up), no Python in the stepping loop, a wait every 16 steps.
    fast_forward = None
    if not args.no_fast_forward:
        from blub_amd.simulation_controller import SimulationController
        scene_ff, fluid_ff = new_scene(args.pcg_schedule)
        sc = SimulationController()
        if args.warmup:
            sc.fast_forward_steps_fluid(fluid_ff, args.warmup * sc.simulation_delta_ns)
        n_ff = sc.fast_forward_steps_fluid(fluid_ff, args.steps * sc.simulation_delta_ns)
        fast_forward = {"steps": n_ff, "warmup": args.warmup, "steps_per_s": round(n_ff / max(sc.computation_time_last_fast_forward, 1e-9), 3),
                        "computation_time_last_fast_forward_s": round(sc.computation_time_last_fast_forward, 5), "batch": 16,
                        "driver": "blub_controller_fast_forward_steps_fluid (C-ABI; batches of 16 blub_fluid_step + blub_fluid_synchronize)"}
        sc.close()
        fluid_ff.close()

    # ---- informational: the same window with the reference's tunable `particle_rebinning_step_frequency` (hybrid_fluid.rs:20-21, GUI range
    # 0..300, default 60) at the value that suits this GPU.  NOT the headline value: BASELINE quotes the metric at the default (60).  The particle
    # kernels cost about twice as much 59 steps after a rebinning as right after it (docs/DESIGN_rounds_1-3.md 5c), and a rebinning costs ~0.17 ms here.
    rebinning_tuned = None
    if not args.no_fast_forward:
        rebinning_tuned = {"note": "informational: same scene and window with particle_rebinning_step_frequency (a tunable of the reference, default 60) changed", "runs": []}
        for freq_r in (8, 16):
            el_r, _ = timed_window(args.pcg_schedule, rebinning=freq_r)
            rebinning_tuned["runs"].append({"frequency": freq_r, "steps_per_s": round(args.steps / el_r, 3)})

    # ---- instrumented pass: per-kernel-class HIP-event timing on the engine's own stream.  The events ride inside the dispatches
    # (hipExtLaunchKernelGGL: the kernels' own start / end time stamps), so the classes sum to the GPU-busy part of a step (<= ms_per_step);
    # the per-kernel table of record is the rocprofv3 trace under profiles/.
    roofline_workload, breakdown, pcg_ms = None, None, 0.0
    if args.profile_steps > 0:
        # a fresh scene over the SAME window as the timed region (warm-up, then profile_steps steps -- by default all of them), so that the
        # classes can be held against that pass's own wall clock: sum of the kernels' durations <= its ms per step
        n_prof = args.steps if args.profile_steps >= args.steps else args.profile_steps
        scene_p, fluid_p = new_scene(args.pcg_schedule)
        for _ in range(args.warmup):
            scene_p.step(dt)
        fluid_p.synchronize()
        fluid_p.profile_enable(True)
        fluid_p.profile_reset()
        t0p = time.perf_counter()
        for _ in range(n_prof):
            scene_p.step(dt)
        fluid_p.synchronize()
        wall_p = time.perf_counter() - t0p
        prof = fluid_p.profile_read()
        fluid_p.profile_enable(False)
        F = int((fluid_p.read_volume("marker") == 1).sum())
        bc = fluid_p.brick_counts()
        fluid_p.close()
        A, Fb = bc["active"] * bc["cells_per_brick"], bc["fluid"] * bc["cells_per_brick"]
        total_ms = sum(v["total_ms"] for v in prof.values())
        dominant = max(prof, key=lambda k: prof[k]["total_ms"])
        avg_ms = prof[dominant]["total_ms"] / prof[dominant]["launches"]
        ach = algorithmic_bytes(dominant, F, P, A, Fb) / (avg_ms * 1e-3) / 1e9
        roofline_workload = {"bound": "fabric latency / launch count (see DESIGN.md 5.2, 6): a few MB per launch", "kernel": dominant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_us": round(avg_ms * 1e3, 2),
                             "share_of_step": round(prof[dominant]["total_ms"] / total_ms, 3), "fluid_cells_at_end": F, "active_brick_cells_at_end": A, "fluid_brick_cells_at_end": Fb,
                             "launches_per_step": round(prof[dominant]["launches"] / n_prof, 1)}
        pcg_ms = sum(prof[k]["total_ms"] for k in prof if k.startswith("pcg_"))
        breakdown = {"us_per_step": {k: round(v["total_ms"] / n_prof * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])},
                     "sum_us_per_step": round(total_ms / n_prof * 1e3, 1),
                     "sum_note": "GPU-busy time of THIS instrumented pass (its own clock: profiled_pass_ms_per_step, always >= the sum); it is not a share of the headline "
                                 "window's ms_per_step, which is a separate, uninstrumented pass",
                     "busy_fraction_of_profiled_pass": round(total_ms / n_prof / (wall_p / n_prof * 1e3), 3), "launches_per_step": round(sum(v["launches"] for v in prof.values()) / n_prof, 1),
                     "profiled_pass_ms_per_step": round(wall_p / n_prof * 1e3, 4),
                     "window": "steps %d..%d of a fresh scene (the timed window) with profiling on: kernel durations from events inside the dispatches; the difference to the "
                               "pass's own ms per step is idle time between dependent launches" % (args.warmup, args.warmup + n_prof)}
        pcg_iters_prof = None

    result = {
        "metric": METRIC, "value": round(args.steps / elapsed, 3), "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.scene, "grid": [nx, ny, nz], "particles": P, "dt": dt, "solver": "tol 0.1 / 32 it / check 4",
                   "rebinning": 60, "parallelism": "single GPU", "pcg_schedule": fluid.pcg_schedule(),
                   "pcg_schedule_note": ("the library's default: no schedule call was made" if args.pcg_schedule == "default" else "set by --pcg-schedule") +
                                        "; the same window with the reference's literal two-reduction order is value_reference_schedule",
                   "window": "steps %d..%d of the scene after %d warm-up steps; the dams break and spread (256 -> ~1400 fluid bricks over the first 130 steps), so steps/s depends on the window: "
                             "the no-flag default (10 + 120) is the representative figure, a 5 + 20 window sees only the cheapest phase" % (args.warmup, args.warmup + args.steps, args.warmup)},
        "value_reference_schedule": by_schedule["reference"]["steps_per_s"],
        "value_single_reduction_schedule": by_schedule["single_reduction"]["steps_per_s"],
        "by_schedule": by_schedule,
        "value_representative_window": representative,
        "pcg_iters_per_sec": round((it1 - it0) / elapsed, 1),
        "pcg_iters_per_step": round((it1 - it0) / args.steps, 2),
        "pcg_iters_per_sec_in_solver": round((it1 - it0) / args.steps * (args.steps if args.profile_steps >= args.steps else args.profile_steps) / (pcg_ms * 1e-3), 1) if pcg_ms > 0 else None,
        "fast_forward": fast_forward,
        "rebinning_tuned": rebinning_tuned,
        "roofline": None,
        "roofline_workload": roofline_workload,
        "kernel_breakdown": breakdown,
    }
    scene._fluid.close()
    scene._fluid = None
    if not args.no_dense_pcg:
        # The HBM roofline is defined on the PCG stencil at 256^3: dense fill, every byte from HBM/MALL.
        # Dominant kernel of an iteration = the update kernel (N + 20F algorithmic bytes of the 2N + 32F the fused pair moves).
        dense = dense_pcg_benchmark(256, 32)
        result["roofline_pcg_dense"] = dense
        result["roofline"] = roofline_object(dense, "same process")
        if not args.no_dense_512:
            # 256^3 is not a clean HBM number (the iteration's working set is about the size of the 256 MiB Infinity Cache): the same benchmark at
            # 512^3, where every byte comes from HBM
            dense512 = dense_pcg_benchmark(512, 32)
            result["roofline_512"] = roofline_object(dense512, "same process")
            result["roofline_pcg_dense_512"] = dense512
    if not args.no_cpu_baseline:   # rank 0 at N = 1 only
        result["cpu_baseline"] = cpu_baseline(scene_path, dt)
    print(json.dumps(result))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
