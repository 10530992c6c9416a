"""The z-slab group's RCCL transport with MORE THAN ONE RANK (round-2 review, missing item 1): N real processes, one slab each, all on
the one GPU of the test box.  RCCL itself refuses two ranks on one device, so tests/native/libfake_rccl.so -- the ~12 RCCL entry points
blub_slab.inc.hip uses, host-staged between processes; TEST INFRASTRUCTURE, see its header -- is LD_PRELOADed into the workers: the same
libblubhip.so and the same protocol code run rank sequencing, grouped send / receive, partial gathers (both spellings), calibration with
its all-reduce, migration between processes and the abort path.  Results are compared with the single-domain engine exactly like the
loopback test (tests/test_gpu_parity.py::test_z_slab_decomposition_matches_single_domain)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT, has_gpu
from tests.multirank_worker import scene
from tests.test_gpu_parity import _match_particles

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
SRC = os.path.join(ROOT, "tests", "native", "fake_rccl.cpp")
LIB = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")


def _fake_rccl():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-fPIC", "-shared", "-std=c++17", SRC, "-o", LIB])
    return LIB


def _launch(mode, world, workdir, schedule="reference", gather="calibrated", timeout=300, steps=3, iterations=0):
    env = dict(os.environ)
    if world >= 8:
        # eight processes with the runtime's default four hardware queues each oversubscribe the GPU's queue slots and the scheduler falls
        # back to rotating them on a timer (measured: ~1 s per grouped exchange instead of ~2 ms with four processes)
        env["GPU_MAX_HW_QUEUES"] = "1"
    env["LD_PRELOAD"] = _fake_rccl() + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    env["FAKE_RCCL_DIR"] = str(workdir)
    env["FAKE_RCCL_TIMEOUT_S"] = "20" if world < 8 else "180"      # (eight HIP contexts come up one after the other on the one GPU)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), mode, str(r), str(world), str(workdir), schedule, gather, str(steps), str(iterations)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung:\n" + "\n".join(outs))
    return [p.returncode for p in procs], outs


@pytest.mark.parametrize("world,schedule,gather", [(2, "reference", "calibrated"), (2, "single_reduction", "p2p"), (2, "single_reduction", "allgather"),
                                                   (4, "reference", "p2p"), (4, "single_reduction", "calibrated"), (8, "single_reduction", "p2p"),
                                                   (2, "single_reduction", "direct"), (4, "single_reduction", "direct")])
# (no 8-process run of the DIRECT transport on the one GPU: its kernels spin on words another process' kernels write, and eight processes
#  oversubscribe the GPU's hardware queues -- the producers are only scheduled when the timer rotates the queues, every bounded wait runs out
#  (~1 s each) and the run takes minutes.  With a GPU per process, which is what the transport is for, every queue is resident.)
def test_multi_process_slab_group_matches_the_single_domain_engine(world, schedule, gather, tmp_path):
    import blub_amd
    # (eight processes time-share the one GPU: two steps of 24 iterations instead of three of 120 keep the run to about a minute)
    steps, iterations = (3, 0) if world < 8 else (2, 24)
    rcs, outs = _launch("compare", world, tmp_path, schedule, gather, timeout=300, steps=steps, iterations=iterations)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)
    ranks = [np.load(os.path.join(tmp_path, "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
    if gather == "direct" and any("error -8" in str(d["status"]) for d in ranks):
        # The direct transport's kernels wait (bounded, ~8 s) on words another process' kernels write.  HERE all processes share ONE GPU -- the
        # transport is made for a GPU per process -- and once in ~10 runs of the 4-process case the device's scheduler leaves a producer's queue
        # unscheduled for longer than that while the consumers hold the CUs: BLUB_ERR_COMM, reported like any peer that fell seconds behind
        # (recoverable in place: tests below).  What THIS test is about is the result; it gets one second launch.
        print("a wait of the direct transport timed out on the shared GPU (%s); launching the ranks once more" % [str(d["status"])[:40] for d in ranks])
        retry = tmp_path / "second_launch"
        retry.mkdir()
        tmp_path = retry
        rcs, outs = _launch("compare", world, tmp_path, schedule, gather, timeout=300, steps=steps, iterations=iterations)
        assert all(rc == 0 for rc in rcs), "\n".join(outs)
        ranks = [np.load(os.path.join(tmp_path, "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
    assert all(str(d["status"]) == "ok" for d in ranks), [str(d["status"]) for d in ranks]
    dim, pos, vel, cfg = scene()
    if iterations:
        cfg = dict(cfg, max_num_iterations=iterations)
    # every rank reports the same transport (the calibration's verdict is all-reduced) and the ranges tile the domain in rank order
    desc = [str(d["description"]) for d in ranks]
    assert all(x == desc[0] for x in desc) and (("%d ranks" % world) in desc[0] or ("direct" in desc[0] and "hipIpc" in desc[0] and gather == "direct")), desc
    print("transport:", desc[0])
    rng_ = [tuple(int(v) for v in d["range"]) for d in ranks]
    assert rng_[0][0] == 0 and rng_[-1][1] == dim[2] and all(a[1] == b[0] for a, b in zip(rng_, rng_[1:]))
    counts0 = [int(d["count0"]) for d in ranks]
    assert sum(counts0) == pos.shape[0] and all(c > 0 for c in counts0[1:-1])       # (with eight slabs the outermost two are empty: the blob spans z = 6 .. 42 of 48)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    try:
        single.set_pcg_schedule(schedule)
        single.set_tuning("pcg1_max_iterations", 1000)
        single.set_gravity_grid((0.0, -981.0, 0.0))
        single.set_particles(pos, *vel)
        for w in (0, 1):
            single.set_solver_config(w, **cfg)
        for step in range(steps):
            single.step(util.DT)
            ps = single.get_particles()[0][:, :3].astype(np.float64)
            pg = np.concatenate([d["pos%d" % step] for d in ranks]).astype(np.float64)
            assert pg.shape == ps.shape                                   # no particle lost or duplicated between the processes
            for d, (z0, z1) in zip(ranks, rng_):                          # every rank only holds particles of its own z-range
                z = d["pos%d" % step][:, 2]
                assert np.all(z >= z0) and np.all(z < z1)
            dd = _match_particles(pg, ps)
            q = (np.median(dd), np.quantile(dd, 0.99), np.quantile(dd, 0.999), dd.max())
            print("step %d  %d processes vs single domain: median %.3g p99 %.3g p99.9 %.3g max %.3g" % ((step, world) + q))
            bounds = (3e-5, 4e-4, 1.5e-3, 3e-3) if step == 0 and not iterations else (2e-4, 3e-3, 3e-2, 0.1)     # (the loopback test's envelope; 24 iterations stop short of convergence)
            for a, b in zip(q, bounds):
                assert a <= b, (step, q, bounds)
            ops = [int(d["ops%d" % step]) for d in ranks]
            assert all(o == ops[0] for o in ops) and ops[0] > 0, ops      # every rank issued the same sequence of transport operations
        assert [d["pos%d" % (steps - 1)].shape[0] for d in ranks] != counts0, "no particle migrated between the processes"
        # particle exchanges synchronise the host only in the first step (no history to size the messages from): 4 of them, then none;
        # the direct transport never does, and never looks at a solve's `done` from the host either
        if gather == "direct":
            assert all(tuple(int(v) for v in d["host_syncs"]) == (0, 0) for d in ranks), [d["host_syncs"] for d in ranks]
            assert all(int(d["ops%d" % step]) == 12 for d in ranks for step in range(steps)), [int(d["ops0"]) for d in ranks]
        else:
            assert all(int(d["host_syncs"][0]) == 4 for d in ranks), [d["host_syncs"] for d in ranks]
        st = [d["stats"] for d in ranks]
        assert all(np.array_equal(x, st[0]) for x in st)                  # identical solver statistics on every rank (gathered partials, fixed order)
        m_single = single.read_volume("marker")
        m_group = np.zeros_like(m_single)
        for d, (z0, z1) in zip(ranks, rng_):
            m_group[z0:z1] = d["marker"][z0:z1]
        assert (m_group != m_single).mean() < 2e-3
    finally:
        single.close()


def test_a_ten_second_stall_of_one_rank_is_recovered_in_place(tmp_path):
    """Round-4 review, item 5: a timed-out wait of the direct transport used to poison the group (BLUB_ERR_COMM, `value: null` in the bench).  Two
    processes over hipIpc, checkpoints every 2 steps; rank 1 sleeps 10 s before its sixth step: rank 0's bounded waits run out, rank 0 reports
    BLUB_ERR_COMM at its next synchronisation; rank 1 need not notice anything by itself (rank 0's flags are all raised when it wakes up -- with invalid data).
    The ranks compare notes after every step, BOTH recover (back to the checkpoint of step 4, sequence numbers re-based), replay, and finish the eight
    steps on the single domain's trajectory as if nothing had happened."""
    import blub_amd
    steps = 8
    rcs, outs = _launch("stall", 2, tmp_path, "single_reduction", "direct", timeout=300, steps=steps, iterations=40)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)
    ranks = [np.load(os.path.join(tmp_path, "rank%d.npz" % r), allow_pickle=True) for r in range(2)]
    assert all(str(d["status"]) == "ok" for d in ranks), [str(d["status"]) for d in ranks]
    print("recovery: first error noticed at step %s, verdicts %s, restored to step %s, %d recoveries, %.1f s" % (
        ranks[0]["first_error_step"], [str(v)[:60] for v in ranks[0]["verdicts"]], ranks[0]["restored_to"], int(ranks[0]["recoveries"]), float(ranks[0]["seconds"])))
    assert all(int(d["recoveries"]) == 1 for d in ranks) and all(int(d["restored_to"]) == 4 for d in ranks)
    # (the rank that waited reports the time-out; the late rank may or may not run into one of its own -- rank 0's invalid solve can end at another
    #  iteration and leave partials unpublished -- which is why the verdict is taken collectively)
    assert "error" in str(ranks[0]["verdicts"][0])
    dim, pos, vel, cfg = scene()
    cfg = dict(cfg, max_num_iterations=40)
    single = blub_amd.HybridFluid(dim, pos.shape[0], binning="off")
    try:
        single.set_pcg_schedule("single_reduction")
        single.set_gravity_grid((0.0, -981.0, 0.0))
        single.set_particles(pos, *vel)
        for w in (0, 1):
            single.set_solver_config(w, **cfg)
        for _ in range(steps):
            single.step(util.DT)
        ps = single.get_particles()[0][:, :3].astype(np.float64)
        pg = np.concatenate([d["pos_final"] for d in ranks]).astype(np.float64)
        assert pg.shape == ps.shape
        dd = _match_particles(pg, ps)
        q = (np.median(dd), np.quantile(dd, 0.99), np.quantile(dd, 0.999), dd.max())
        print("after the recovery, step %d: group vs single domain median %.3g p99 %.3g p99.9 %.3g max %.3g" % ((steps,) + q))
        for a, b in zip(q, (2e-4, 3e-3, 3e-2, 0.1)):      # (the later-step envelope of the loopback test; measured 8e-5 / 7e-4 / 6e-3 / 0.012)
            assert a <= b, q
        st = [d["stats"] for d in ranks]
        assert np.array_equal(st[0], st[1])
    finally:
        single.close()


def test_a_rank_that_dies_gives_its_peers_a_communication_error_instead_of_a_hang(tmp_path):
    """Rank 1 of 3 exits abruptly before its second step.  Its z-neighbours are blocked in a grouped send / receive with it; they must
    come back with BLUB_ERR_COMM (and abort the communicator, so that THEIR peers fail too) rather than block forever."""
    rcs, outs = _launch("kill", 3, tmp_path, timeout=180)
    assert rcs[1] == 17, outs[1]
    for r in (0, 2):
        assert rcs[r] == 0, outs[r]
        d = np.load(os.path.join(tmp_path, "rank%d.npz" % r), allow_pickle=True)
        status = str(d["status"])
        assert status.startswith("error -8") and "communicator aborted" in status, status      # BLUB_ERR_COMM
        assert "pos0" in d.files and "pos1" not in d.files                                      # step 0 completed, step 1 failed


@pytest.mark.parametrize("transport", ["rccl", "direct", "auto", "auto_fine_grained"])
def test_bench_gpus_2_runs_the_slab_path(tmp_path, transport):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per "GPU"), control plane over gloo, data plane through
    the preloaded fake: the z-slab path must produce the JSON line itself -- not the replicas fallback.  BLUB_BENCH_TRANSPORT=direct: the opt-in
    that exchanges hipIpc handles over the control plane and lets the kernels store into the neighbour's slab.  auto (what the driver gets):
    every rank first runs blub_amd/direct_probe.py in a child process -- a small slab group of its own over the direct transport, checked against
    the single-domain engine -- and the job uses the direct transport because the probe passed."""
    import json
    env = dict(os.environ)
    fine = transport == "auto_fine_grained"      # the probe is only offered fine-grained slabs (what a node where coarse-grained memory fails it would end up with)
    if fine:
        transport = "auto"
        env["BLUB_DIRECT_PROBE_MODES"] = "fine_grained"
    env["BLUB_BENCH_TRANSPORT"] = transport
    env["LD_PRELOAD"] = _fake_rccl()
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["BLUB_BENCH_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scene", "corner_dams_128", "--no-dense-pcg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] is not None and d["value"] > 0, d
    if transport in ("direct", "auto"):
        assert "direct" in d["transport"] and "hipIpc" in d["config"]["parallelism"] and d["transport_ops_per_step"] == 12, d
        if transport == "auto":
            pr = d["direct_transport_probe"]
            assert pr["passed"] is True and pr["detail"] == "ok" and pr["slab_memory"] == ("fine_grained" if fine else "coarse"), pr
            assert ("fine-grained memory" if fine else "coarse-grained memory") in d["transport"], d["transport"]
        assert len(d["fluid_bricks_per_rank"]) == 2 and min(d["fluid_bricks_per_rank"]) > 0 and d["config"]["slab_cuts_mode"] == "dynamic", d
    else:
        assert "rccl, 2 ranks" in d["transport"] and d["transport_ops_per_step"] > 0


def test_bench_gpus_8_weak_scaling_runs_the_slab_path(tmp_path):
    """`bench.py --gpus 8 --scaling weak` as the driver would launch it on an 8-GPU node: eight ranks, eight slabs (corner_dams_128 stacked
    eight times along z), here all on one GPU with the data plane through the preloaded stand-in.  The line must be the z-slab line."""
    import json
    env = dict(os.environ)
    env["LD_PRELOAD"] = _fake_rccl()
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["BLUB_BENCH_BACKEND"] = "gloo"
    env["FAKE_RCCL_TIMEOUT_S"] = "180"
    env["GPU_MAX_HW_QUEUES"] = "1"          # (see _launch)
    env["BLUB_BENCH_TRANSPORT"] = "rccl"    # (the probe of the default "auto" would put eight more processes on the one GPU)
    env["BLUB_BENCH_SLAB_DEADLINE"] = "420"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29523",
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--scene", "corner_dams_128", "--scaling", "weak", "--no-dense-pcg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] is not None and d["value"] > 0, d
    assert "8 ranks" in d["transport"] and d["transport_ops_per_step"] > 0


def test_a_failing_direct_probe_leaves_the_job_on_rccl(tmp_path):
    """The probe's children are made to fail (an unusable device ordinal): every rank agrees on the verdict and the job runs over RCCL."""
    import json
    env = dict(os.environ)
    env["LD_PRELOAD"] = _fake_rccl()
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["BLUB_BENCH_BACKEND"] = "gloo"
    env["BLUB_DIRECT_PROBE_DEVICE"] = "63"      # test hook of blub_amd/direct_probe.py: the device ordinal the children are given
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scene", "corner_dams_128", "--no-dense-pcg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[-1])
    assert d["value"] is not None and d["value"] > 0 and "rccl, 2 ranks" in d["transport"], d
    assert d["direct_transport_probe"]["passed"] is False and "probe child" in d["direct_transport_probe"]["detail"], d["direct_transport_probe"]


def test_bench_recovers_in_place_from_a_stalled_rank(tmp_path):
    """`bench.py --gpus 2` over the direct transport with one rank falling 10 s behind in the timed window (BLUB_BENCH_STALL): the ranks agree that a wait
    timed out, recover in place (checkpoints every 4 steps here) and run the window again -- the line is a scaling result over the direct transport with
    `recovered_in_place` 1, not the RCCL second attempt and not the replicas fallback (round-4 review, item 5)."""
    import json
    env = dict(os.environ)
    env["LD_PRELOAD"] = _fake_rccl()
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["BLUB_BENCH_BACKEND"] = "gloo"
    env["BLUB_BENCH_TRANSPORT"] = "direct"
    env["BLUB_BENCH_STALL"] = "1:3:10"
    env["BLUB_BENCH_CHECKPOINT_INTERVAL"] = "4"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--scene", "corner_dams_128", "--no-dense-pcg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[-1])
    assert d["scaling"] == "strong" and d["value"] is not None and d["value"] > 0 and "direct" in d["transport"], d
    assert d["recovered_in_place"] == 1, d
    assert "recovered in place to step" in res.stderr


def test_a_direct_run_that_fails_gets_a_second_attempt_over_rccl(tmp_path):
    """The probe passes, then the run over the direct transport fails (injected after the warm-up steps): every rank re-executes itself
    over RCCL and the line says so -- a scaling result, not the replicas fallback."""
    import json
    env = dict(os.environ)
    env["LD_PRELOAD"] = _fake_rccl()
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["BLUB_BENCH_BACKEND"] = "gloo"
    env["BLUB_BENCH_FAIL_DIRECT"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scene", "corner_dams_128", "--no-dense-pcg"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[-1])
    assert d["scaling"] == "strong" and d["value"] is not None and d["value"] > 0 and "rccl, 2 ranks" in d["transport"], d
    assert "second attempt, over RCCL" in d["direct_transport_probe"]["detail"] and "injected failure" in d["direct_transport_probe"]["detail"], d["direct_transport_probe"]
