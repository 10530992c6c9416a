"""One rank of a multi-process z-slab group (tests/test_gpu_multirank.py starts N of these with tests/native/libfake_rccl.so LD_PRELOADed,
all on the one GPU of the test box).  usage: multirank_worker.py MODE RANK WORLD WORKDIR SCHEDULE GATHER"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def scene():
    """The blob of tests/test_gpu_parity.py::test_z_slab_decomposition_matches_single_domain: it straddles every slab interface and shears
    across them, so every exchange carries data."""
    dim = (32, 32, 48)
    rng = np.random.default_rng(4)
    cells = np.stack(np.meshgrid(np.arange(6, 26), np.arange(8, 20), np.arange(6, 42), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    vel = [np.zeros((pos.shape[0], 4), np.float32) for _ in range(3)]
    vel[2][:, 3] = 6.0 * np.sin(pos[:, 0] * 0.4)
    cfg = dict(error_tolerance=0.0, max_num_iterations=120, error_check_frequency=8)   # far past convergence, no convergence DECISION (see the loopback test)
    return dim, pos, vel, cfg


class FileControlPlane:
    """all_gather / barrier between the workers through files (the workers have no torch.distributed): test infrastructure"""

    def __init__(self, rank, world, workdir):
        self.rank, self.world, self.dir, self.n = rank, world, workdir, 0

    def all_gather(self, obj):
        import pickle
        self.n += 1
        mine = os.path.join(self.dir, "cp%d_%d" % (self.n, self.rank))
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f)
        os.rename(mine + ".tmp", mine)
        out = []
        for r in range(self.world):
            path = os.path.join(self.dir, "cp%d_%d" % (self.n, r))
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise SystemExit("control plane: no message %d from rank %d" % (self.n, r))
                time.sleep(0.005)
            with open(path, "rb") as f:
                out.append(pickle.load(f))
        return out

    def barrier(self):
        self.all_gather(None)


def main():
    mode, rank, world, workdir, schedule, gather = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    steps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
    iterations = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    try:
        ctypes.CDLL(None).fake_rccl_marker
    except AttributeError:
        raise SystemExit("tests/native/libfake_rccl.so is not preloaded: this worker must not run against the real RCCL on a shared GPU")
    import blub_amd
    from tests import util
    uid_path = os.path.join(workdir, "uid")
    if rank == 0:
        uid = blub_amd.SlabGroup.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    dim, pos, vel, cfg = scene()
    if iterations:
        cfg = dict(cfg, max_num_iterations=iterations)
    group = blub_amd.SlabGroup(dim, pos.shape[0], rank=rank, world=world, unique_id=uid, device=0, binning="off")
    out = {"description": group.transport_description(), "range": group.local_range(0)}
    try:
        if gather == "direct":
            # DIRECT transport between processes: every rank exports the hipIpc handles of its slab, maps everybody else's, and from then on the
            # kernels store into the neighbours' memory themselves (RCCL -- here its stand-in -- only carried the group's creation)
            blob = group.export_handles()
            with open(os.path.join(workdir, "ipc%d.tmp" % rank), "wb") as f:
                f.write(blob)
            os.rename(os.path.join(workdir, "ipc%d.tmp" % rank), os.path.join(workdir, "ipc%d" % rank))
            for r in range(world):
                if r == rank:
                    continue
                path = os.path.join(workdir, "ipc%d" % r)
                t0 = time.time()
                while not os.path.exists(path):
                    if time.time() - t0 > 120:
                        raise SystemExit("no export from rank %d" % r)
                    time.sleep(0.01)
                group.connect(r, open(path, "rb").read())
            group.set_transport("direct")
            out["description"] = group.transport_description()
        elif gather != "calibrated":
            group.set_gather_mode(gather)
        group.set_pcg_schedule(schedule)
        group.local_fluid(0).set_tuning("pcg1_max_iterations", 1000)     # (120 single-reduction iterations: the engine would otherwise switch to the reference order)
        group.set_gravity_grid((0.0, -981.0, 0.0))
        group.set_particles(pos, *vel)
        for w in (0, 1):
            group.set_solver_config(w, **cfg)
        fluid = group.local_fluid(0)
        out["count0"] = fluid.num_particles()
        if mode == "stall":
            # One rank falls 10 seconds behind in the middle of the run (direct transport): its peer's bounded waits run out (~8 s), what the peer steps
            # from then on is invalid.  The ranks compare notes over the control plane after every step; on an error ALL of them recover in place
            # (blub_amd.SlabGroup.recover: back to the newest checkpoint every rank holds) and replay.
            cp = FileControlPlane(rank, world, workdir)
            group.set_checkpoint_interval(2)
            step, recoveries, stalled = 0, 0, False
            t_start = time.time()
            while step < steps:
                if rank == 1 and step == 5 and not stalled:
                    stalled = True
                    time.sleep(10.0)
                status = "ok"
                try:
                    group.step(util.DT)
                    group.synchronize()
                except blub_amd.hybrid_fluid.BlubError as e:
                    if e.status != -8:
                        raise
                    status = "error: %s" % e
                verdicts = cp.all_gather(status)
                if any(v != "ok" for v in verdicts):
                    out["first_error_step"] = step
                    out["verdicts"] = np.array(verdicts)
                    step = group.recover(cp.all_gather, cp.barrier)
                    out["restored_to"] = step
                    recoveries += 1
                    continue
                step += 1
            out["recoveries"] = recoveries
            out["seconds"] = time.time() - t_start
            out["pos_final"] = group.get_particles()[0][:, :3]
            out["stats"] = np.array([fluid.solver_stats(0), fluid.solver_stats(1)], np.float64)
            out["status"] = "ok"
            steps = 0
        for step in range(steps):
            if mode == "kill" and rank == 1 and step == 1:
                os._exit(17)            # a rank disappears in the middle of the run
            ops0 = group.transport_ops()
            group.step(util.DT)
            group.synchronize()
            out["pos%d" % step] = group.get_particles()[0][:, :3]
            out["ops%d" % step] = group.transport_ops() - ops0
        out["stats"] = np.array([fluid.solver_stats(0), fluid.solver_stats(1)], np.float64)
        out["host_syncs"] = np.array(group.host_syncs())
        out["marker"] = fluid.read_volume("marker")
        out["status"] = "ok"
    except blub_amd.hybrid_fluid.BlubError as e:
        out["status"] = "error %d: %s" % (e.status, e)
    np.savez(os.path.join(workdir, "rank%d.npz" % rank), **out)
    group.close()


if __name__ == "__main__":
    main()
