"""Does the DIRECT z-slab transport work between the GPUs of THIS node?  (bench.py, N > 1.)

The direct transport lets kernels store into hipIpc-mapped memory of the neighbouring ranks.  A wrong peer mapping is a
memory fault, not an error code, and the development box has one GPU -- so before a multi-GPU job relies on it, every rank runs this
probe in a CHILD process on its own GPU: the children form a small slab group of their own (real RCCL for its creation, hipIpc handles
exchanged through files), step a 32 x 32 x 48 scene whose blob straddles every slab interface three times over the direct transport,
and child 0 holds the gathered result against the single-domain engine inside the envelope of tests/test_gpu_parity.py::
test_z_slab_decomposition_matches_single_domain.  A child that faults, hangs (time-out) or disagrees only costs the probe: the
parents then stay on the RCCL transport.

usage (internal): python -m blub_amd.direct_probe RANK WORLD DEVICE WORKDIR"""
import os
import subprocess
import sys
import time

import numpy as np

DT = 1.0 / 120.0
STEPS = 3
DIM = (32, 32, 48)
WAIT_S = float(os.environ.get("BLUB_DIRECT_PROBE_WAIT_S", "60"))      # how long a child waits for a file of another child


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


def child(rank, world, device, workdir):
    import blub_amd
    pos, vel, cfg = scene()
    uid_path = os.path.join(workdir, "uid")
    if rank == 0:
        _publish(uid_path, blub_amd.SlabGroup.unique_id())
    else:
        _wait_for(uid_path, WAIT_S)
    group = blub_amd.SlabGroup(DIM, pos.shape[0], rank=rank, world=world, unique_id=open(uid_path, "rb").read(), device=device, binning="off")
    try:
        _publish(os.path.join(workdir, "ipc%d" % rank), group.export_handles())
        for r in range(world):
            if r != rank:
                _wait_for(os.path.join(workdir, "ipc%d" % r), WAIT_S)
                group.connect(r, open(os.path.join(workdir, "ipc%d" % r), "rb").read())
        group.set_transport("direct")
        group.set_pcg_schedule("single_reduction")
        group.set_gravity_grid((0.0, -981.0, 0.0))
        group.set_particles(pos, *vel)
        for w in (0, 1):
            group.set_solver_config(w, **cfg)
        out = {}
        for step in range(STEPS):
            group.step(DT)
            group.synchronize()
            out["pos%d" % step] = group.get_particles()[0][:, :3]
        out["stats"] = np.array([group.local_fluid(0).solver_stats(0), group.local_fluid(0).solver_stats(1)], np.float64)
        out["host_syncs"] = np.array(group.host_syncs())
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
    try:
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
    finally:
        single.close()
    _publish(os.path.join(workdir, "verdict"), verdict.encode())


def run(rank, world, device, timeout=150.0):
    """Collective over torch.distributed's default group (every rank calls it, BEFORE creating its own slab group).  -> (ok, reason)"""
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
        res = subprocess.run([sys.executable, "-m", "blub_amd.direct_probe", str(rank), str(world), str(device), workdir],
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


if __name__ == "__main__":
    child(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
