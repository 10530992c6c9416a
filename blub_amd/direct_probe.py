"""Does the DIRECT z-slab transport work between the GPUs of THIS node?  (bench.py, N > 1.)

The direct transport lets kernels store into hipIpc-mapped memory of the neighbouring ranks.  A wrong peer mapping is a
memory fault, not an error code, and the development box has one GPU -- so before a multi-GPU job relies on it, every rank runs this
probe in a CHILD process on its own GPU: the children form a small slab group of their own (real RCCL for its creation, hipIpc handles
exchanged through files), step a 32 x 32 x 48 scene whose blob straddles every slab interface three times over the direct transport,
and child 0 holds the gathered result against the single-domain engine inside the envelope of tests/test_gpu_parity.py::
test_z_slab_decomposition_matches_single_domain.  A child that faults, hangs (time-out) or disagrees only costs the probe: the
parents then stay on the RCCL transport.

Round 5 (review item 1b): a RARE stale read -- a peer's store seen late through a cache -- is a wrong dot product, not a fault, and the
free-running comparison above has an envelope it would hide in.  So the children first run a bit-exact check: one PCG problem (the divergence
of the probe scene, snapshotted) is solved with 121 launched iterations over the RCCL transport and then REPEAT times over the direct
transport from the same snapshot; the per-iteration scalars {gamma, delta, max|r|, alpha} (blub_fluid_read_scalar_log) and the final
pressure planes must agree BIT FOR BIT between the transports and between all ranks: one stale partial or ghost-plane row in any of the
>= 363 direct iterations shows up in the iteration it happens.  The probe is run per MEMORY MODE of the slabs (include/blubhip.h:
BLUB_SLAB_MEMORY_*): coarse-grained first (the fast one, outside HIP's documented cross-agent model), then fine-grained; the first that
passes is what the job uses, none -> RCCL.

usage (internal): python -m blub_amd.direct_probe RANK WORLD DEVICE WORKDIR MEMORY"""
import os
import subprocess
import sys
import time

import numpy as np

DT = 1.0 / 120.0
STEPS = 3
DIM = (32, 32, 48)
WAIT_S = float(os.environ.get("BLUB_DIRECT_PROBE_WAIT_S", "60"))      # how long a child waits for a file of another child
REPEAT = int(os.environ.get("BLUB_DIRECT_PROBE_REPEAT", "3"))         # direct-transport solves of the bit-exact check (121 iterations each)
MODES = ("coarse", "fine_grained")


def scene():
    rng = np.random.default_rng(4)
    cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
    vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)      # shear across the interfaces: particles migrate
    cfg = dict(error_tolerance=0.0, max_num_iterations=40, error_check_frequency=8)      # fixed length: no convergence DECISION to differ on
    return pos, vel, cfg


def _wait_for(path, seconds):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > seconds:
            raise SystemExit("probe: timed out waiting for %s" % os.path.basename(path))
        time.sleep(0.01)


def _publish(path, data):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.rename(path + ".tmp", path)


def _bit_exact_solve_check(group, rank):
    """-> (verdict, logs): see the module docstring.  The group comes in on the RCCL transport and leaves on the direct one."""
    import blub_amd
    f = group.local_fluid(0)
    z0, z1 = group.local_range(0)
    f.set_tuning("pcg1_max_iterations", 1000)
    f.set_tuning("pcg_scalar_log", 1)
    for w in (0, 1):
        group.set_solver_config(w, error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)
    group.run_stages(DT, "ghosts", "divergence")
    b, p0 = f.read_volume("residual"), f.read_volume("pressure_velocity")

    def solve():
        f.write_volume("residual", b)
        f.write_volume("pressure_velocity", p0)
        group.run_stages(DT, "solve_velocity", "solve_velocity")
        group.synchronize()
        return f.scalar_log(0), f.read_volume("pressure_velocity")[z0:z1].copy()
    log_h, p_h = solve()
    if log_h.shape[0] != 121:
        return "rank %d: the RCCL-transport solve logged %d iterations instead of 121" % (rank, log_h.shape[0]), {}
    group.set_transport("direct")
    logs = {"log_host": log_h}
    for rep in range(REPEAT):
        log_d, p_d = solve()
        logs["log_direct%d" % rep] = log_d
        if log_d.shape != log_h.shape or not np.array_equal(log_d.view(np.uint32), log_h.view(np.uint32)):
            n = min(len(log_d), len(log_h))
            bad = np.nonzero((log_d[:n].view(np.uint32) != log_h[:n].view(np.uint32)).any(axis=1))[0]
            first = int(bad[0]) if len(bad) else n
            return "rank %d: direct solve %d departs from the RCCL solve at iteration %d of %d (%s vs %s): a stale partial or ghost plane" % (
                rank, rep, first, len(log_h), log_d[first].tolist() if first < len(log_d) else None, log_h[first].tolist() if first < len(log_h) else None), logs
        if not np.array_equal(p_d.view(np.uint32), p_h.view(np.uint32)):
            return "rank %d: direct solve %d ends with %d pressure cells that differ from the RCCL solve's" % (rank, rep, int((p_d != p_h).sum())), logs
    return "ok", logs


def child(rank, world, device, workdir, memory="coarse"):
    import blub_amd
    pos, vel, cfg = scene()
    uid_path = os.path.join(workdir, "uid")
    if rank == 0:
        _publish(uid_path, blub_amd.SlabGroup.unique_id())
    else:
        _wait_for(uid_path, WAIT_S)
    group = blub_amd.SlabGroup(DIM, pos.shape[0], rank=rank, world=world, unique_id=open(uid_path, "rb").read(), device=device, binning="off", memory=memory)
    try:
        _publish(os.path.join(workdir, "ipc%d" % rank), group.export_handles())
        for r in range(world):
            if r != rank:
                _wait_for(os.path.join(workdir, "ipc%d" % r), WAIT_S)
                group.connect(r, open(os.path.join(workdir, "ipc%d" % r), "rb").read())
        group.set_pcg_schedule("single_reduction")
        group.set_gravity_grid((0.0, -981.0, 0.0))
        # ---- bit-exact check: RCCL transport vs direct transport on the same PCG problem
        group.set_particles(pos, *vel)
        exact, out = _bit_exact_solve_check(group, rank)
        out = dict(out)
        out["exact"] = np.frombuffer(exact.encode(), np.uint8)
        # ---- free-running steps over the direct transport against the single domain (child 0 compares)
        group.local_fluid(0).set_tuning("pcg_scalar_log", 0)
        group.set_particles(pos, *vel)
        for w in (0, 1):
            group.set_solver_config(w, **cfg)
        syncs0 = np.array(group.host_syncs())
        for step in range(STEPS):
            group.step(DT)
            group.synchronize()
            out["pos%d" % step] = group.get_particles()[0][:, :3]
        out["stats"] = np.array([group.local_fluid(0).solver_stats(0), group.local_fluid(0).solver_stats(1)], np.float64)
        out["host_syncs"] = np.array(group.host_syncs()) - syncs0
        np.savez(os.path.join(workdir, "rank%d.tmp.npz" % rank), **out)
        os.rename(os.path.join(workdir, "rank%d.tmp.npz" % rank), os.path.join(workdir, "rank%d.npz" % rank))
    finally:
        group.close()
    if rank != 0:
        return
    # child 0: the gathered result against the single-domain engine
    from scipy.spatial import cKDTree
    for r in range(world):
        _wait_for(os.path.join(workdir, "rank%d.npz" % r), 1.5 * WAIT_S)
    ranks = [np.load(os.path.join(workdir, "rank%d.npz" % r)) for r in range(world)]
    single = blub_amd.HybridFluid(DIM, pos.shape[0], device=device, binning="off")
    verdict = "ok"
    for d in ranks:      # the bit-exact check of every rank, then: every rank logged the SAME scalars
        v = bytes(d["exact"]).decode()
        if v != "ok":
            verdict = v
            break
    if verdict == "ok":
        for key in ["log_host"] + ["log_direct%d" % k for k in range(REPEAT)]:
            if not all(np.array_equal(d[key].view(np.uint32), ranks[0][key].view(np.uint32)) for d in ranks):
                verdict = "%s differs between the ranks: the slabs did not derive the same scalars" % key
                break
    try:
        if verdict != "ok":
            raise StopIteration
        single.set_pcg_schedule("single_reduction")
        single.set_gravity_grid((0.0, -981.0, 0.0))
        single.set_particles(pos, *vel)
        for w in (0, 1):
            single.set_solver_config(w, **cfg)
        for step in range(STEPS):
            single.step(DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pg = np.concatenate([d["pos%d" % step] for d in ranks]).astype(np.float64)
            if pg.shape != ps.shape:
                verdict = "step %d: %d particles in the group, %d in the single domain" % (step, pg.shape[0], ps.shape[0])
                break
            d, idx = cKDTree(ps).query(pg, k=1)
            q = (float(np.median(d)), float(np.quantile(d, 0.99)), float(np.quantile(d, 0.999)), float(d.max()))
            bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 else (2e-4, 3e-3, 3e-2, 0.1)
            if len(np.unique(idx)) != len(pg) or any(a > b for a, b in zip(q, bounds)):
                verdict = "step %d: slab group vs single domain median %.3g p99 %.3g p99.9 %.3g max %.3g cells (bounds %s)" % ((step,) + q + (bounds,))
                break
        st = [d["stats"] for d in ranks]
        if verdict == "ok" and not all(np.array_equal(x, st[0]) for x in st):
            verdict = "solver statistics differ between the ranks"
        if verdict == "ok" and any(tuple(int(v) for v in d["host_syncs"]) != (0, 0) for d in ranks):
            verdict = "the direct transport synchronised the host"
    except StopIteration:
        pass
    finally:
        single.close()
    _publish(os.path.join(workdir, "verdict"), verdict.encode())


def _run_mode(rank, world, device, timeout, memory):
    import tempfile
    import torch.distributed as dist
    box = [tempfile.mkdtemp(prefix="blub_direct_probe_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    workdir = box[0]
    env = dict(os.environ)
    if env.get("FAKE_RCCL_DIR"):      # (development box: the preloaded RCCL stand-in keeps its rings in a directory; the children need their own)
        env["FAKE_RCCL_DIR"] = workdir
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reason = "ok"
    device = int(os.environ.get("BLUB_DIRECT_PROBE_DEVICE", device))      # (test hook: an unusable ordinal makes the children fail)
    try:
        res = subprocess.run([sys.executable, "-m", "blub_amd.direct_probe", str(rank), str(world), str(device), workdir, memory],
                             cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
        if res.returncode != 0:
            reason = "rank %d: probe child exited with %d: %s" % (rank, res.returncode, (res.stderr or res.stdout)[-300:].replace("\n", " | "))
    except subprocess.TimeoutExpired:
        reason = "rank %d: probe child timed out" % rank
    reasons = [None] * world
    dist.all_gather_object(reasons, reason)
    bad = [r for r in reasons if r != "ok"]
    if not bad:
        try:
            verdict = open(os.path.join(workdir, "verdict"), "rb").read().decode() if rank == 0 else None
        except OSError:
            verdict = "no verdict from child 0"
        box = [verdict]
        dist.broadcast_object_list(box, src=0)
        if box[0] != "ok":
            bad = [box[0]]
    if rank == 0:
        import shutil
        shutil.rmtree(workdir, ignore_errors=True)
    return (not bad), (bad[0] if bad else "ok")


def run(rank, world, device, timeout=150.0, modes=None):
    """Collective over torch.distributed's default group (every rank calls it, BEFORE creating its own slab group).
    -> (ok, reason, memory): `memory` = the first slab memory mode (SlabGroup(memory=...)) whose probe passed; every rank gets the same answer."""
    if modes is None:
        modes = tuple(m for m in os.environ.get("BLUB_DIRECT_PROBE_MODES", ",".join(MODES)).split(",") if m)
    why = []
    for memory in modes:
        ok, reason = _run_mode(rank, world, device, timeout, memory)
        if ok:
            return True, ("ok" if not why else "ok with %s memory (%s)" % (memory, "; ".join(why))), memory
        why.append("%s: %s" % (memory, reason))
    return False, "; ".join(why), None


if __name__ == "__main__":
    child(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "coarse")
