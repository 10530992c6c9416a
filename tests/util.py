"""Shared helpers for the parity tests: matched oracle / HIP instances and state transfer between them."""
import numpy as np

from oracle.oracle import Oracle

DT = float(np.float32(8333333) / np.float32(1e9))   # simulation_controller.rs:33-39 -> Duration::as_secs_f32
FLOAT_VOLUMES = ["vel_x", "vel_y", "vel_z", "pressure_velocity", "pressure_density", "residual", "search"]
# R1 writes b = clamp(1 - rho/8, -0.5, 0.5) / dt: the gathered density rho (a sum of ~64 weights, |rho| ~ 8) is compared at 2e-6
# relative (a dozen ulps of the sum: only the ORDER of the additions differs between engine and oracle), which the subtraction from 1
# and the division by 8 dt = 1/15 turn into an absolute 2e-4 on b.
DENSITY_RESIDUAL_TOL = 2e-4
STEP_ORDER = ["transfer", "divergence", "solve_velocity", "binning", "project", "advect", "density_gather",
              "solve_density", "position_change", "correct"]


def make_dam(nx, ny, nz, fill=(0.45, 0.6, 1.0), seed=1, velocity_scale=3.0, max_extra=64):
    """A dam-break block of 8 jittered particles per cell with smooth random-ish APIC rows."""
    rng = np.random.default_rng(seed)
    x1, y1, z1 = max(2, int(nx * fill[0])), max(2, int(ny * fill[1])), max(2, int(nz * fill[2]) - 1)
    cells = np.stack(np.meshgrid(np.arange(1, x1), np.arange(1, y1), np.arange(1, z1), indexing="ij"), -1).reshape(-1, 3)
    # x fastest like hybrid_fluid.rs:648-650
    order = np.lexsort((cells[:, 0], cells[:, 1], cells[:, 2]))
    cells = cells[order]
    sub = np.stack(np.meshgrid([0, 1], [0, 1], [0, 1], indexing="ij"), -1).reshape(-1, 3)[:, ::-1]
    pos = (cells[:, None, :] + 0.5 * sub[None, :, :] + 0.5 * rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    n = pos.shape[0]
    vel = []
    for c in range(3):
        rows = np.zeros((n, 4), np.float32)
        rows[:, :3] = (rng.standard_normal((n, 3)) * 0.2).astype(np.float32)
        rows[:, 3] = (velocity_scale * np.sin(pos[:, (c + 1) % 3] * 0.3 + c) + rng.standard_normal(n) * 0.1).astype(np.float32)
        vel.append(rows)
    return pos, vel, n + max_extra


def new_pair(nx, ny, nz, max_particles, precond="zero", binning="off", gravity=(0.0, -981.0, 0.0), solver=None):
    import blub_amd
    o = Oracle(nx, ny, nz, max_particles)
    o.set_quirks(precond=precond, binning=binning)
    h = blub_amd.HybridFluid((nx, ny, nz), max_particles, precond=precond, binning=binning)
    o.set_gravity_grid(gravity)
    h.set_gravity_grid(gravity)
    if solver:
        for w in (0, 1):
            o.set_solver_config(w, **solver)
            h.set_solver_config(w, **solver)
    return o, h


def copy_state(o, h, volumes=("marker", "linked_list") + tuple(FLOAT_VOLUMES)):
    """oracle -> HIP: particles (with list pointers) and volumes."""
    pos, vx, vy, vz = o.get_particles()
    h.set_particles(pos, vx, vy, vz, keep_ll=True)
    for v in volumes:
        h.write_volume(v, o.read_volume(v))
    h.step_counter = o.step_counter


def max_but_three(d):
    """The largest value of `d` once its three largest are set aside: trajectory comparisons hold their MAX column with it -- a single particle at the surface or a wall that
    takes another branch of the wall handling moves by up to a cell whenever that happens (once in ~60 runs in two tests of round 6); callers hold d.max() to one cell."""
    d = np.asarray(d).ravel()
    return float(np.partition(d, len(d) - 4)[len(d) - 4]) if len(d) > 4 else float(d.max())


def assert_close_but_few(name, got, ref, rel=1e-5, few=8, factor=10.0):
    """assert_close for sums whose ORDER of additions differs between the two sides while the particles move fast (a face of the P2G gather adds up to 96
    products w * d; with |d| ~ 50 cells/s the order is worth ~1e-5 in a bad case, 3e-4 in the worst): `rel` for all but `few` values, factor * rel for all."""
    got64 = np.asarray(got, np.float64)
    ref64 = np.asarray(ref, np.float64)
    over = np.abs(got64 - ref64) > rel * np.maximum(1.0, np.abs(ref64))
    assert over.sum() <= few, "%s: %d values beyond %g (allowed: %d)" % (name, over.sum(), rel, few)
    assert_close(name, got, ref, rel=factor * rel)
    return int(over.sum())


def assert_close(name, got, ref, rel=1e-5, abs_=None):
    """|got - ref| <= rel * max(1, |ref|)  (SURVEY 8c "stated tolerances", grid fields after one kernel)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    tol = rel * np.maximum(1.0, np.abs(ref)) if abs_ is None else abs_
    bad = np.abs(got - ref) > tol
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(got - ref) - tol), got.shape)
        raise AssertionError("%s: %d / %d values differ; worst at %s got %r ref %r" % (name, bad.sum(), bad.size, i, got[i], ref[i]))


def lists_as_sets(ll, pos_ll, n):
    """Linked lists (heads volume + next pointers) -> {cell: frozenset(particles)}; order is a race in the reference."""
    nxt = pos_ll.view(np.uint32)[:, 3]
    out = {}
    heads = ll.reshape(-1)
    for cell in np.nonzero(heads)[0]:
        cur = int(heads[cell]) - 1
        members = []
        guard = 0
        while cur != 0xFFFFFFFF and guard <= n:
            members.append(cur)
            cur = int(nxt[cur])
            guard += 1
        assert guard <= n, "cycle in linked list"
        out[int(cell)] = frozenset(members)
    return out


def set_mapping(h, mapping):
    """Work mapping AND schedule of the PCG kernels for a parametrised test: "rows" / "bricks" / "bricks_staged" run the reference's
    two-reduction schedule (two kernels per iteration), "bricks_single" the single-reduction schedule (one kernel per iteration,
    staged brick tiles), "auto" leaves the engine's defaults."""
    if mapping == "auto":
        return
    if mapping == "bricks_single":
        h.set_pcg_work_mapping("bricks_staged")
        h.set_pcg_schedule("single_reduction")
    else:
        h.set_pcg_work_mapping(mapping)
        h.set_pcg_schedule("reference")
