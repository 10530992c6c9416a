"""N > 1 path, host side, as two real processes over gloo (no GPU): slab ranges, the weak-scaling scene, particle
ownership, and the RCCL-id hand-out mechanism (broadcast_object_list) used by SlabGroup.from_torch_distributed."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tests.conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import blub_amd
    from blub_amd import slab_scene

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% sys.argv[1], rank=int(sys.argv[2]), world_size=2)
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = blub_amd.Scene.parse(path=os.path.join(%r, "scenes", "corner_dams_128.json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, world)
    assert dim == (128, 128, 256) and len(cubes) == 4 and maxp == 2 * cfg.max_num_particles
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    # (the upper dam of slab 0 is no longer clamped by the domain end, so slightly more than 2 x 111 600 particles)
    assert 2 * 111600 <= len(pos) <= maxp and len(pos) %% 8 == 0
    lens = [None, None]
    dist.all_gather_object(lens, (len(pos), float(pos[:, :3].sum())))
    assert lens[0] == lens[1]            # both ranks seeded the identical particle set
    own, (z0, z1) = slab_scene.partition_particles(pos, dim[2], world, rank)
    # ranges tile [0, nz) in whole 4-cell brick layers
    ranges = [None, None]
    dist.all_gather_object(ranges, (z0, z1))
    assert ranges[0][0] == 0 and ranges[1][1] == dim[2] and ranges[0][1] == ranges[1][0] and ranges[0][1] %% 4 == 0
    # every particle has exactly one owner
    counts = torch.tensor([len(own)], dtype=torch.int64)
    dist.all_reduce(counts)
    assert int(counts) == len(pos), (int(counts), len(pos))
    mask = torch.zeros(len(pos), dtype=torch.int32); mask[torch.from_numpy(own)] = 1
    dist.all_reduce(mask)
    assert int(mask.min()) == 1 and int(mask.max()) == 1
    # both dams next to the interface (upper dam of slab 0, lower dam of slab 1) supply ghost candidates to the other rank
    z = pos[:, 2]
    near = np.count_nonzero((z >= z1 - 2) & (z < z1)) if rank == 0 else np.count_nonzero((z >= z0) & (z < z0 + 2))
    assert near > 0
    # the RCCL unique id is created on rank 0 only and handed out through the process group
    payload = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0)
    assert payload[0] == bytes(range(128))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
""")


def test_two_rank_gloo_host_logic(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    port = str(29500 + os.getpid() % 2000)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_slab_ranges_are_whole_brick_layers():
    import blub_amd
    for nz, n in ((256, 1), (256, 2), (256, 8), (64, 3), (48, 5), (2048, 8)):
        r = [blub_amd.SlabGroup.slab_range(nz, n, i) for i in range(n)]
        assert r[0][0] == 0 and r[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(a[0] % 4 == 0 and a[1] > a[0] for a in r)
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 4


def test_balanced_cuts_share_out_the_fluid_bricks():
    """blub_slab_balanced_cuts (host only): the contiguous partition of the brick layers that minimises the heaviest slab.  The metric's scene keeps its
    fluid in z < 32 and z >= 224 of 256: uniform cuts into 8 leave six slabs empty (round-4 review), balanced ones give every slab an eighth."""
    import numpy as np
    import blub_amd
    from blub_amd import slab_scene
    cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "corner_dams_256.json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    for n in (1, 2, 4, 8):
        cuts, bricks = blub_amd.SlabGroup.balanced_cuts(dim, pos, n)
        assert cuts[0] == 0 and cuts[-1] == 256 and all(c % 4 == 0 for c in cuts) and all(b > a for a, b in zip(cuts, cuts[1:]))
        assert sum(bricks) == 256 and max(bricks) == 256 // n, (cuts, bricks)
        assert bricks == blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, cuts)
        # the partition agrees with how the group hands out particles (slab_scene.partition_particles with the same cuts)
        owned = [len(slab_scene.partition_particles(pos, dim[2], n, i, cuts)[0]) for i in range(n)]
        assert sum(owned) == len(pos) and min(owned) > 0
    uniform = [blub_amd.SlabGroup.slab_range(256, 8, i)[0] for i in range(8)] + [256]
    assert blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos, uniform) == [128, 0, 0, 0, 0, 0, 0, 128]
    # a minimum thickness is honoured, ties go to the most even layer counts, no fluid at all gives the uniform cuts
    cuts, _ = blub_amd.SlabGroup.balanced_cuts(dim, pos, 8, min_layers=4)
    assert min(b - a for a, b in zip(cuts, cuts[1:])) >= 16
    cuts, bricks = blub_amd.SlabGroup.balanced_cuts((64, 64, 64), np.zeros((0, 4), np.float32), 4)
    assert cuts == [0, 16, 32, 48, 64] and bricks == [0, 0, 0, 0]
    with pytest.raises(blub_amd.BlubError):
        blub_amd.SlabGroup.balanced_cuts((64, 64, 16), pos[:10], 8)          # more slabs than brick layers
    # positions outside the grid -- or not numbers at all -- carry no weight
    odd = np.array([[np.nan, 1, 1, 0], [5.5, 5.5, 5.5, 0], [1e30, 2, 2, 0], [-3, 2, 2, 0], [np.inf, 2, 2, 0]], np.float32)
    cuts, bricks = blub_amd.SlabGroup.balanced_cuts((64, 64, 64), odd, 4)
    assert sum(bricks) == 1 and cuts[0] == 0 and cuts[-1] == 64


@pytest.mark.parametrize("cuts", [(4, 32, 64), (0, 30, 64), (0, 32, 32), (0, 32, 60), (0, 40, 32)])
def test_bad_cut_planes_are_rejected_before_anything_is_allocated(cuts):
    import blub_amd
    with pytest.raises(blub_amd.BlubError) as e:
        blub_amd.SlabGroup((64, 64, 64), 1024, local=2, cuts=list(cuts))
    assert e.value.status == -1, str(e.value)          # BLUB_ERR_INVALID_ARGUMENT (not BLUB_ERR_NO_DEVICE: the cuts are checked first)


RECOVER_WORKER = textwrap.dedent("""
    import os, sys
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import blub_amd
    from blub_amd.hybrid_fluid import BlubError

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% sys.argv[1], rank=int(sys.argv[2]), world_size=2)
    rank = dist.get_rank()

    class Stub(blub_amd.SlabGroup):
        # the collective LOGIC of SlabGroup.recover without a device: which generation is restored, with which sequence base
        def __init__(self, held, seq, fail_sync):
            self.held, self.seq, self.fail_sync, self.restored = held, seq, fail_sync, None
        def synchronize(self):
            if self.fail_sync:
                self.fail_sync = False
                raise BlubError(-8, "direct transport: a wait for a peer's flag timed out")
        def checkpoints(self):
            return sorted(self.held)
        def exchange_sequence(self):
            return self.seq
        def restore(self, step, base):
            self.restored = (step, base)
        def close(self):
            pass

    # rank 0 failed at step 30 (holds the generations of steps 0 -> overwritten, 16; skipped 32), rank 1 ran on and took a generation at step 32
    g = Stub(held=[16, 0] if rank == 0 else [16, 32], seq=7012 if rank == 0 else 7345, fail_sync=(rank == 0))
    step = g.recover_over_torch_distributed()
    assert step == 16 and g.restored == (16, 7345 + 1024), (step, g.restored)      # the newest generation BOTH hold; a base above every rank's number
    # no common generation: every rank raises (and nobody restores)
    g = Stub(held=[0] if rank == 0 else [16, 32], seq=1, fail_sync=False)
    try:
        g.recover_over_torch_distributed()
        raise SystemExit("expected an error")
    except BlubError as e:
        assert e.status == -8 and g.restored is None
    # an error other than the time-out is not swallowed
    class Broken(Stub):
        def synchronize(self):
            raise BlubError(-4, "device error")
    try:
        Broken([0], 1, False).recover_over_torch_distributed()
        raise SystemExit("expected an error")
    except BlubError as e:
        assert e.status == -4
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok" %% rank)
""")


def test_recovery_picks_the_newest_generation_every_rank_holds(tmp_path):
    """blub_amd.SlabGroup.recover (in-place recovery of a z-slab group, include/blubhip.h) over a 2-rank gloo group with the device calls stubbed:
    the newest checkpoint BOTH ranks hold is restored -- a generation a rank took after its peer's failure is missing on the peer --, the sequence
    numbers restart above every rank's, no common generation is an error on every rank, other errors pass through."""
    script = tmp_path / "recover_worker.py"
    script.write_text(RECOVER_WORKER % ROOT)
    port = str(33500 + os.getpid() % 2000)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o


def test_layer_partition_properties():
    """blub_slab_balanced_cuts on random particle clouds (host only): valid cut planes, every particle-bearing brick counted once, the heaviest slab never
    heavier than under uniform cuts, min_layers honoured."""
    import numpy as np
    import blub_amd
    rng = np.random.default_rng(11)
    for trial in range(25):
        dim = (int(rng.integers(2, 9)) * 16, int(rng.integers(2, 9)) * 8, int(rng.integers(8, 40)) * 4)
        n = int(rng.integers(1, 4000))
        centre = rng.random(3) * np.array(dim)
        pos = (centre + rng.standard_normal((n, 3)) * np.array(dim) * rng.uniform(0.02, 0.4)).astype(np.float32)
        slabs = int(rng.integers(1, min(8, dim[2] // 4) + 1))
        ml = int(rng.integers(1, max(1, dim[2] // 4 // slabs) + 1))
        cuts, bricks = blub_amd.SlabGroup.balanced_cuts(dim, pos, slabs, ml)
        assert cuts[0] == 0 and cuts[-1] == dim[2] and all(c % 4 == 0 for c in cuts) and all(b - a >= 4 * ml for a, b in zip(cuts, cuts[1:])), (dim, slabs, ml, cuts)
        inside = ((pos >= 0) & (pos < np.array(dim, np.float32))).all(1)
        assert bricks == blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos[inside], cuts)
        uniform = [blub_amd.SlabGroup.slab_range(dim[2], slabs, i)[0] for i in range(slabs)] + [dim[2]]
        if ml == 1:
            assert max(bricks) <= max(blub_amd.SlabGroup.fluid_bricks_per_slab(dim, pos[inside], uniform)) + 1, (cuts, bricks, uniform)


PROBE_WORKER = textwrap.dedent("""
    import os, sys
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from blub_amd import direct_probe

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% sys.argv[1], rank=int(sys.argv[2]), world_size=2)
    rank, world = dist.get_rank(), dist.get_world_size()
    ok, why, memory = direct_probe.run(rank, world, 0, timeout=120.0)
    verdicts = [None, None]
    dist.all_gather_object(verdicts, (ok, why))
    assert verdicts[0] == verdicts[1], verdicts              # every rank holds the same verdict and the same reason
    assert ok is False and memory is None and "probe child" in why and "coarse:" in why and "fine_grained:" in why, (ok, why)      # (every memory mode was tried)    # (no GPU here: the children cannot create their slab group)
    dist.barrier()
    dist.destroy_process_group()
    print("rank %%d ok: %%s" %% (rank, why[:120]))
""")


def test_the_direct_transport_probe_reaches_one_verdict_on_all_ranks(tmp_path):
    """blub_amd/direct_probe.py (what `bench.py --gpus N` runs before it relies on the direct transport), two ranks over gloo, no GPU: the probe's
    children fail -- without a HIP device they cannot create their slab group -- and both ranks come back with the SAME negative verdict and
    reason, i.e. both would stay on the RCCL transport.  (The passing case needs GPUs: tests/test_gpu_multirank.py.)"""
    script = tmp_path / "probe_worker.py"
    script.write_text(PROBE_WORKER % ROOT)
    port = str(31500 + os.getpid() % 2000)
    env = dict(os.environ, OMP_NUM_THREADS="2", BLUB_DIRECT_PROBE_WAIT_S="5")
    procs = [subprocess.Popen([sys.executable, str(script), port, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
        assert "rank %d ok" % r in o
