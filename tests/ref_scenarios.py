"""Scenarios that drive the oracle and oracle/_ref (the reference's own shaders, oracle/glsl/) side by side.

Shared by tests/golden/make_ref_golden.py (writes tests/golden/ref_*.npz in the container that holds /root/reference) and
tests/test_oracle_vs_ref.py (holds the oracle to those fixtures everywhere, and to a live oracle/_ref where it exists).
`backend` is anything with the Oracle surface: oracle.oracle.Oracle, oracle.glsl.ref_fluid.RefFluid.
"""
import numpy as np

DT = float(np.float32(8333333) / np.float32(1e9))   # simulation_controller.rs:33-35: 8 333 333 ns as f32 seconds
STEP_DIM = (64, 16, 32)    # three different extents (an axis mix-up cannot cancel), N = 2 x 16384 (pressure_solver.rs:551, and see Q14)
PCG_DIM = (32, 64, 16)
STAGES = ("transfer", "divergence", "solve_velocity", "project", "advect", "density_gather", "solve_density", "position_change", "correct")
# what each stage writes -- only these are recorded / compared after the stage
STAGE_OUTPUTS = {
    "transfer": ("marker", "linked_list", "vel_x", "vel_y", "vel_z", "particles_ll"),
    "divergence": ("residual@fluid",),
    "solve_velocity": ("pressure_velocity", "residual@fluid", "search@fluid", "stats0"),
    "binning": ("particles_pos",),
    "project": ("vel_x", "vel_y", "vel_z"),
    "advect": ("marker", "linked_list", "particles",),
    "density_gather": ("residual@fluid",),
    "solve_density": ("pressure_density", "residual@fluid", "search@fluid", "stats1"),
    "position_change": ("vel_x", "vel_y", "vel_z"),
    "correct": ("particles_pos",),
}


def f16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def step_scene(seed=2024, dim=STEP_DIM, solids=True):
    """A block of fluid against two domain walls, fast enough to hit them, next to a moving solid box."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = dim
    cells = np.stack(np.meshgrid(np.arange(1, 13), np.arange(1, 11), np.arange(4, 17), indexing="ij"), -1).reshape(-1, 3)
    # 8 particles per cell, one per half-cell stratum like add_fluid_cube (hybrid_fluid.rs:648-671): every staggered dual cell then holds
    # exactly 8 of them, below the 12-entry cap of the P2G walk (transfer_gather_velocity.comp:61) -- WHICH particles a longer list
    # keeps is the order of atomics, a race in the reference, and nothing an engine could be compared on
    sub = np.stack(np.meshgrid([0, 1], [0, 1], [0, 1], indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + 0.5 * sub[None, :, :] + 0.5 * rng.random((cells.shape[0], 8, 3))).reshape(-1, 3).astype(np.float32)
    pos = np.clip(pos, 1.001, np.array(dim, np.float32) - 1.001).astype(np.float32)
    vel = []
    for c in range(3):
        rows = np.zeros((len(pos), 4), np.float32)
        rows[:, :3] = (rng.standard_normal((len(pos), 3)) * 0.2).astype(np.float32)
        rows[:, 3] = (6.0 * np.cos(pos[:, (c + 1) % 3] * 0.5) - (25.0 if c == 0 else 0.0)).astype(np.float32)   # -x: into the wall
        vel.append(rows)
    solid = np.zeros((nz, ny, nx, 4), np.float32)
    if solids:
        solid[6:11, 1:5, 11:16] = f16([0.5, -0.25, 0.125, 1.0])     # a moving box that has "eaten" two columns of the fluid block (escape path)
        solid[15:19, 1:3, 2:6] = f16([-1.5, 2.0, 0.75, 1.0])         # a second one overlapping its top layers in z, fastest along y
        solid[4:16, 6:7, 5:6] = f16([0.0, 0.0, 0.0, 1.0])            # a resting one-voxel-thick post inside the fluid (target-in-solid, stuck / push path)
    return dict(dim=np.array(dim), pos=pos, vx=vel[0], vy=vel[1], vz=vel[2], solid=solid, gravity=np.array([0.0, -981.0, 0.0], np.float32))


CAPS_COUNTS = (64, 40, 33, 32, 20, 13, 12, 5)


def caps_scene(seed=77, dim=STEP_DIM):
    """Lists LONGER than the caps of the two gathers -- 12 rounds in P2G (transfer_gather_velocity.comp:61), 32 in the density gather
    (density_projection_gather_error.comp:69) -- in an insertion order an engine can reproduce.  WHICH particles a longer list keeps is the order
    of the atomic exchanges: ascending particle index in the shim, a race on a GPU -- except for particles of ONE wavefront, which insert together.
    So the particles come in blocks of 64 consecutive indices; block k puts n_k of them (n_k cycles through 64, 40, 33, 32, 20, 13, 12, 5) into
    cell c_k of a blob and the other 64 - n_k into cell d_k of a second blob, all at c + f with f in (0.62, 0.92)^3: every one of the four dual
    cells of a particle (three staggered P2G lists, the density list) is then the same for the whole sub-block, every list lives inside one block
    of 64, and the small fall of one step (g dt^2 = 0.07 cells) keeps the density dual cell.  Small random velocities and APIC rows that differ from
    particle to particle (sigma 0.5 cells / s: 0.004 cells per step), so that WHICH 12 of a 64-entry list enter a face's average shows in the result
    at the 1e-1 level; no solids."""
    rng = np.random.default_rng(seed)
    a = np.stack(np.meshgrid(np.arange(3, 11), np.arange(3, 8), np.arange(4, 10), indexing="ij"), -1).reshape(-1, 3)        # 240 cells
    b = np.stack(np.meshgrid(np.arange(30, 38), np.arange(3, 8), np.arange(14, 20), indexing="ij"), -1).reshape(-1, 3)      # 240 cells
    pos = []
    for k in range(len(a)):
        n = CAPS_COUNTS[k % len(CAPS_COUNTS)]
        f = 0.62 + 0.30 * rng.random((64, 3))
        cell = np.where((np.arange(64) < n)[:, None], a[k][None, :], b[k][None, :])
        pos.append(cell + f)
    pos = np.concatenate(pos).astype(np.float32)
    vel = []
    for c in range(3):
        rows = np.zeros((len(pos), 4), np.float32)
        rows[:, :3] = (rng.standard_normal((len(pos), 3)) * 0.2).astype(np.float32)
        rows[:, 3] = (rng.standard_normal(len(pos)) * 0.5).astype(np.float32)
        vel.append(rows)
    return dict(dim=np.array(dim), pos=pos, vx=vel[0], vy=vel[1], vz=vel[2], solid=np.zeros(tuple(dim[::-1]) + (4,), np.float32),
                gravity=np.array([0.0, -981.0, 0.0], np.float32))


def sequential_lists(pos, offset, dim):
    """The linked lists a SEQUENTIAL insertion in ascending particle index leaves (the shim's order, transfer_build_linkedlist.comp:25:
    next = atomicExchange(head, i + 1) - 1): (heads volume [z, y, x] uint32 with index + 1, next per particle uint32, 0xFFFFFFFF = end)."""
    c = np.floor(pos[:, :3].astype(np.float32) - np.float32(offset)).astype(np.int64)
    nx, ny, nz = (int(v) for v in dim)
    inb = (c >= 0).all(1) & (c[:, 0] < nx) & (c[:, 1] < ny) & (c[:, 2] < nz)
    key = (c[:, 2] * ny + c[:, 1]) * nx + c[:, 0]
    heads = np.zeros(nx * ny * nz, np.uint32)
    nxt = np.full(len(pos), 0xFFFFFFFF, np.uint32)
    idx = np.nonzero(inb)[0]
    order = idx[np.argsort(key[idx], kind="stable")]          # by cell, ascending index inside a cell
    k = key[order]
    first = np.ones(len(order), bool); first[1:] = k[1:] != k[:-1]
    last = np.ones(len(order), bool); last[:-1] = k[1:] != k[:-1]
    nxt[order[~first]] = order[np.nonzero(~first)[0] - 1].astype(np.uint32)
    heads[k[last]] = (order[last] + 1).astype(np.uint32)
    return heads.reshape(nz, ny, nx), nxt


def lists_are_wave_local(pos, offset, dim):
    """every dual cell floor(pos - offset) holds particles of ONE block of 64 consecutive indices only"""
    c = np.floor(pos[:, :3].astype(np.float32) - np.float32(offset)).astype(np.int64)
    key = (c[:, 2] * int(dim[1]) + c[:, 1]) * int(dim[0]) + c[:, 0]
    blk = np.arange(len(pos)) // 64
    order = np.argsort(key, kind="stable")
    k, b = key[order], blk[order]
    same = k[1:] == k[:-1]
    return bool(np.all(b[1:][same] == b[:-1][same])), np.bincount(np.unique(key, return_inverse=True)[1]).max()


def configure(backend, scene, precond="zero", max_iter=32, tol=0.1, freq=4, is_ref=False):
    backend.set_gravity_grid(scene["gravity"])
    for w in (0, 1):
        backend.set_solver_config(w, error_tolerance=tol, max_num_iterations=max_iter, error_check_frequency=freq)
    if is_ref:
        backend.set_modes(precond=precond)
        backend.binning_enabled = False
    else:
        backend.set_quirks(precond=precond, binning="off")
        backend.set_dot_mode(2)
    backend.write_volume("solid", scene["solid"])
    backend.set_particles(scene["pos"], scene["vx"], scene["vy"], scene["vz"])


def capture(backend, what):
    """One recorded quantity of a backend's state as an ndarray (bit patterns preserved)."""
    if what.startswith("stats"):
        e, i = backend.solver_stats(int(what[-1]))
        return np.array([np.float32(e), np.float32(i)], np.float32)
    if what.startswith("particles"):
        p = backend.get_particles()
        if what == "particles_ll":
            return np.ascontiguousarray(p[0][:, 3]).view(np.uint32).copy()
        if what == "particles_pos":
            return np.ascontiguousarray(p[0][:, :3])
        return np.concatenate([np.ascontiguousarray(a).view(np.uint32).reshape(len(a), -1) for a in p], axis=1)   # 16 words per particle
    name, _, mask = what.partition("@")
    v = backend.read_volume(name)
    if mask == "fluid":   # scratch volumes are only defined on FLUID cells
        v = np.where(backend.read_volume("marker") == 1, v, 0).astype(v.dtype)
    return v


def run_step_recording(backend, dt=DT, stages=STAGES):
    out = {}
    for st in stages:
        backend.run_stage(st, dt)
        for what in STAGE_OUTPUTS[st]:
            out["%s/%s" % (st, what)] = capture(backend, what)
    return out


def pcg_problem(seed=7, dim=PCG_DIM):
    """Random marker field in which every diagonal d = 0..6 occurs, random right-hand side, random warm start."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = dim
    shape = (nz, ny, nx)
    marker = -np.ones(shape, np.int8)
    marker[[0, -1], :, :] = 0; marker[:, [0, -1], :] = 0; marker[:, :, [0, -1]] = 0
    blob = rng.random(shape) < 0.55
    blob[:, (2 * ny) // 3:, :] = False
    marker[(marker == -1) & blob] = 1
    marker[(rng.random(shape) < 0.08) & (marker != 0)] = 0          # scattered solids: fluid cells with 0..6 non-solid neighbours
    z0 = nz // 4
    marker[z0: z0 + 4, 2:5, 10:14] = 0
    marker[z0 + 1, 3, 11] = 1                                       # a FLUID cell walled in on all six sides: d = 0
    marker[z0, 3, 12] = 1; marker[z0 - 1, 3, 12] = -1               # one with a single opening: d = 1
    b = np.where(marker == 1, rng.standard_normal(shape), 0).astype(np.float32)
    b[z0 + 1, 3, 11] = 0      # (A has a zero row there: any other right-hand side is inconsistent and CG diverges -- D1 itself gives 0 for such a cell)
    p0 = (rng.standard_normal(shape) * 0.3).astype(np.float32)       # also non-zero outside the fluid: S0 must clear it
    return dict(dim=np.array(dim), marker=marker, b=b, p0=p0)


def run_pcg(backend, prob, k, precond="zero", tol=0.0, freq=4, warm=True, is_ref=False, dt=DT):
    if is_ref:
        backend.set_modes(precond=precond)
    else:
        backend.set_quirks(precond=precond, binning="off")
        backend.set_dot_mode(2)
    backend.write_volume("marker", prob["marker"])
    backend.write_volume("residual", prob["b"])
    backend.write_volume("pressure_velocity", prob["p0"] if warm else np.zeros_like(prob["p0"]))
    backend.reset_pressure_cleared(0, True)
    backend.set_solver_config(0, error_tolerance=tol, max_num_iterations=k, error_check_frequency=freq)
    backend.run_stage("solve_velocity", dt)
    m = prob["marker"] == 1
    e, it = backend.solver_stats(0)
    return dict(p=backend.read_volume("pressure_velocity"), r=np.where(m, backend.read_volume("residual"), 0).astype(np.float32),
                s=np.where(m, backend.read_volume("search"), 0).astype(np.float32), stats=np.array([np.float32(e), np.float32(it)], np.float32))


def binning_scene(seed=5, dim=STEP_DIM, n=3000):
    """Shuffled particles, a count that is not a multiple of 64 (Q4: the shaders have no `i < NumParticles` guard)."""
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3)) * (np.array(dim) - 2.002) + 1.001).astype(np.float32)
    pos[: n // 3] = (pos[: n // 3] * 0.25 + 4.0).astype(np.float32)   # a dense corner: many particles per cell
    return dict(dim=np.array(dim), pos=pos)
