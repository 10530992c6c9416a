"""Independent numpy restatements of four grid/particle stages, written from the shader semantics summarised in SURVEY.md 8(a)
(NOT from oracle/blub_oracle.cpp), cross-checked against the C++ oracle on random inputs.  They use a different formulation
where one exists (effective face velocities for the divergence, brute-force sums over ALL particles for the two gathers), so
an error in the oracle's literal restatement of a shader would show up here.  Volumes are indexed [z, y, x]."""
import numpy as np

from oracle.oracle import Oracle
from tests import util

DT = util.DT
FLUID, AIR, SOLID = 1, -1, 0
DIM = (20, 16, 12)   # x, y, z


def _random_marker(rng, solid_block=True):
    nx, ny, nz = DIM
    m = np.full((nz, ny, nx), AIR, np.int8)
    blob = rng.random((nz, ny, nx)) < 0.6
    blob[:, 11:, :] = False
    m[blob] = FLUID
    if solid_block:
        m[3:7, 2:6, 8:12] = SOLID
    m[[0, -1], :, :] = SOLID
    m[:, [0, -1], :] = SOLID
    m[:, :, [0, -1]] = SOLID
    return m


def _shift(a, axis, d, fill):
    """b[g] = a[g + d * e_axis] with `fill` outside (axis: 0 = x, 1 = y, 2 = z for [z, y, x] arrays)."""
    ax = 2 - axis
    out = np.full_like(a, fill)
    src = [slice(None)] * 3
    dst = [slice(None)] * 3
    if d > 0:
        src[ax], dst[ax] = slice(d, None), slice(None, -d)
    else:
        src[ax], dst[ax] = slice(None, d), slice(-d, None)
    out[tuple(dst)] = a[tuple(src)]
    return out


def test_divergence_equals_the_flux_of_effective_face_velocities():
    """divergence_compute.comp:28-87: for FLUID cells, sum_c (v+_c - v-_c) where a face shared with a SOLID neighbour carries
    the solid's own velocity component instead of the grid's (the shader adds +-(v_wall - v_solid) correction terms)."""
    rng = np.random.default_rng(1)
    nx, ny, nz = DIM
    m = _random_marker(rng)
    vel = [rng.standard_normal((nz, ny, nx)).astype(np.float32) for _ in range(3)]
    solid = np.zeros((nz, ny, nx, 4), np.float32)
    solid[..., 3] = (m == SOLID)
    solid[3:7, 2:6, 8:12, :3] = rng.standard_normal(3).astype(np.float32)          # the block moves
    o = Oracle(nx, ny, nz, 8)
    o.write_volume("solid", solid)
    o.write_volume("marker", m)
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, vel[c])
    o.write_volume("residual", np.full((nz, ny, nx), 7.0, np.float32))
    o.run_stage("divergence", DT)
    got = o.read_volume("residual")
    want = np.zeros((nz, ny, nx), np.float64)
    for c in range(3):
        vplus = vel[c].astype(np.float64)                                             # face between g and g + e_c
        vminus = _shift(vel[c], c, -1, 0.0).astype(np.float64)                        # face between g - e_c and g
        m_plus, m_minus = _shift(m, c, +1, SOLID), _shift(m, c, -1, SOLID)
        s_plus, s_minus = _shift(solid[..., c], c, +1, 0.0), _shift(solid[..., c], c, -1, 0.0)
        want += np.where(m_plus == SOLID, s_plus, vplus) - np.where(m_minus == SOLID, s_minus, vminus)
    fluid = m == FLUID
    assert fluid.sum() > 500
    assert np.abs(got[fluid] - want[fluid]).max() < 1e-5
    assert np.all(got[~fluid] == 7.0)                                                 # only FLUID cells are written


def test_pressure_gradient_subtraction_per_face():
    """divergence_remove.comp:19-49 per face (g, c): either side FLUID -> (one side SOLID ? the solid's velocity component :
    v - (p[g] - p[g + e_c]) with p = 0 outside FLUID); neither side FLUID -> 0."""
    rng = np.random.default_rng(2)
    nx, ny, nz = DIM
    m = _random_marker(rng)
    vel = [rng.standard_normal((nz, ny, nx)).astype(np.float32) for _ in range(3)]
    p = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    solid = np.zeros((nz, ny, nx, 4), np.float32)
    solid[..., 3] = (m == SOLID)
    solid[3:7, 2:6, 8:12, :3] = np.float32([0.5, -1.25, 2.0])
    o = Oracle(nx, ny, nz, 8)
    o.write_volume("solid", solid)
    o.write_volume("marker", m)
    o.write_volume("pressure_velocity", p)
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, vel[c])
    o.run_stage("project", DT)                      # divergence_remove + extrapolate; compare on faces extrapolation never writes
    pf = np.where(m == FLUID, p, 0).astype(np.float64)
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        got = o.read_volume(n)
        m_n = _shift(m, c, +1, SOLID)
        touched = (m == FLUID) | (m_n == FLUID)     # extrapolation only writes faces with NO fluid side
        want = vel[c].astype(np.float64) - (pf - _shift(pf, c, +1, 0.0))   # blub's sign: A p = +divergence with A = -Laplacian, so v -= p_c - p_nbr
        want = np.where(m_n == SOLID, _shift(solid[..., c], c, +1, 0.0), want)   # the +c side is the solid: its velocity
        want = np.where(m == SOLID, solid[..., c], want)                          # this side is the solid
        assert touched.sum() > 500
        assert np.abs(got[touched] - want[touched]).max() < 1e-5, n


def _jittered_particles(rng, per_cell=3):
    nx, ny, nz = DIM
    cells = np.stack(np.meshgrid(np.arange(2, nx - 3), np.arange(2, 9), np.arange(2, nz - 3), indexing="ij"), -1).reshape(-1, 3)
    keep = rng.random(len(cells)) < 0.7
    cells = cells[keep]
    return (cells[:, None, :] + rng.random((len(cells), per_cell, 3))).reshape(-1, 3).astype(np.float32)


def test_p2g_gather_equals_brute_force_shepard_sums():
    """transfer_gather_velocity.comp:18-26, 39-127: v(face) = sum_p w (C_row . (s - p) + v_p) / sum_p w + g_c dt with
    w = prod_k sat(1 - |s_k - p_k|), s = the face's sample point g + 0.5 + 0.5 e_c; 0 if exactly one side is SOLID; written only
    where either side is FLUID.  With fewer than 12 particles per dual cell the 8-list walk visits exactly the particles with
    w > 0, so a brute-force sum over ALL particles must agree."""
    rng = np.random.default_rng(3)
    nx, ny, nz = DIM
    pos = _jittered_particles(rng)
    rows = [np.concatenate([rng.standard_normal((len(pos), 3)) * 0.3, rng.standard_normal((len(pos), 1))], 1).astype(np.float32) for _ in range(3)]
    grav = np.float32([0.3, -9.0, 1.5])
    o = Oracle(nx, ny, nz, len(pos))
    o.set_gravity_grid(grav)
    o.set_particles(pos, *rows)
    o.run_stage("transfer", DT)
    m = o.read_volume("marker")
    P = pos.astype(np.float64)
    occupied = np.zeros((nz, ny, nx), bool)
    occupied[P[:, 2].astype(int), P[:, 1].astype(int), P[:, 0].astype(int)] = True
    assert np.array_equal(m == FLUID, occupied)                                   # FLUID = cells that hold a particle
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    for c, name in enumerate(("vel_x", "vel_y", "vel_z")):
        got = o.read_volume(name)
        e = np.eye(3)[c]
        s = np.stack([gx, gy, gz], -1).reshape(-1, 3) + 0.5 + 0.5 * e           # sample points, x-major list
        num, den = np.zeros(len(s)), np.zeros(len(s))
        R = rows[c].astype(np.float64)
        for k in range(0, len(s), 2048):                                          # blocks of faces x all particles
            d = s[k:k + 2048, None, :] - P[None, :, :]
            w = np.clip(1.0 - np.abs(d), 0.0, 1.0).prod(-1)
            val = (d * R[None, :, :3]).sum(-1) + R[None, :, 3]
            num[k:k + 2048] = (w * val).sum(1)
            den[k:k + 2048] = w.sum(1)
        v = np.where(den > 0, num / np.maximum(den, 1e-300), 0.0) + float(grav[c]) * DT
        v = v.reshape(nx, ny, nz).transpose(2, 1, 0)
        m_n = _shift(m, c, +1, SOLID)
        v = np.where((m == SOLID) | (m_n == SOLID), 0.0, v)
        written = (m == FLUID) | (m_n == FLUID)
        assert written.sum() > 300
        assert np.abs(got[written] - v[written]).max() < 2e-5, name


def test_density_error_equals_brute_force_kernel_sum():
    """density_projection_gather_error.comp:27-31, 167-196 for FLUID cells: rho = sum_p prod_k sat(1 - |c_k - p_k|) at the cell
    centre c = g + 0.5, + 0.5625 per SOLID face-neighbour, max(8, rho) if any face-neighbour is AIR, then
    b = clamp(1 - rho / 8, -0.5, 0.5) / dt.  (The lists are built by the advection stage: run it with a zero velocity field.)"""
    rng = np.random.default_rng(4)
    nx, ny, nz = DIM
    pos = _jittered_particles(rng, per_cell=9)                                    # dense enough for rho > 8 in the bulk
    o = Oracle(nx, ny, nz, len(pos))
    o.set_particles(pos)
    o.run_stage("transfer", DT)                                                    # marker
    for n in ("vel_x", "vel_y", "vel_z"):
        o.write_volume(n, np.zeros((nz, ny, nx), np.float32))
    o.run_stage("advect", DT)                                                      # zero velocity: positions stay, density lists are built
    P = o.get_particles()[0][:, :3].astype(np.float64)
    assert np.abs(P - pos).max() == 0.0
    o.run_stage("density_gather", DT)
    m = o.read_volume("marker")
    got = o.read_volume("residual")
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    cen = np.stack([gx, gy, gz], -1).reshape(-1, 3) + 0.5
    rho = np.zeros(len(cen))
    for k in range(0, len(cen), 2048):
        d = cen[k:k + 2048, None, :] - P[None, :, :]
        rho[k:k + 2048] = np.clip(1.0 - np.abs(d), 0.0, 1.0).prod(-1).sum(1)
    rho = rho.reshape(nx, ny, nz).transpose(2, 1, 0)
    any_air = np.zeros((nz, ny, nx), bool)
    for c in range(3):
        for dd in (-1, +1):
            mn = _shift(m, c, dd, SOLID)
            rho = rho + 0.5625 * (mn == SOLID)
            any_air |= mn == AIR
    rho = np.where(any_air, np.maximum(8.0, rho), rho)
    want = np.clip(1.0 - rho / 8.0, -0.5, 0.5) / DT
    fluid = m == FLUID
    assert fluid.sum() > 300 and (want[fluid] < -1.0).sum() > 20 and (np.abs(want[fluid]) < 1e-9).sum() > 20   # compressed bulk and clamped surface
    # 1 - rho / 8 cancels: an f32 rounding of rho (~1e-6 relative) is worth ~1e-6 / dt on the result
    assert np.abs(got[fluid] - want[fluid]).max() < 5e-6 / DT


def test_particle_correction_is_a_hardware_style_trilinear_fetch():
    """density_projection_correct_particles.comp:32-40: x += (T_x, T_y, T_z) with T_c = the c-th position-change volume sampled
    with a clamp-to-edge LINEAR filter at the normalised coordinate (x - 0.5 e_c) / dim, i.e. texel centres at integer + 0.5.
    Particles far from walls / solids (no step truncation, :45-69)."""
    rng = np.random.default_rng(5)
    nx, ny, nz = DIM
    o = Oracle(nx, ny, nz, 4000)
    m = np.full((nz, ny, nx), AIR, np.int8)
    m[[0, -1], :, :] = SOLID; m[:, [0, -1], :] = SOLID; m[:, :, [0, -1]] = SOLID
    o.write_volume("marker", m)
    vol = [(rng.standard_normal((nz, ny, nx)) * 0.2).astype(np.float32) for _ in range(3)]
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, vol[c])
    pos = (rng.random((3000, 3)) * (np.array(DIM) - 6.0) + 3.0).astype(np.float32)      # >= 3 cells from every wall, |delta| < 1
    o.set_particles(pos)
    o.run_stage("correct", DT)
    got = o.get_particles()[0][:, :3].astype(np.float64)
    P = pos.astype(np.float64)
    want = P.copy()
    for c in range(3):
        u = P - 0.5 * np.eye(3)[c] - 0.5                                                  # continuous texel coordinates
        i0 = np.floor(u).astype(int)
        f = u - i0
        acc = np.zeros(len(P))
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    acc += w * vol[c][i0[:, 2] + dz, i0[:, 1] + dy, i0[:, 0] + dx]
        want[:, c] += acc
    assert np.abs(got - want).max() < 5e-6
    assert np.abs(got - P).max() > 0.1


def test_g2p_apic_rows_and_rk4_for_interior_particles():
    """advect_particles.comp:74-127, 184-188 for particles that stay away from walls and solids: per component i the staggered
    cell around o_i = x - (0.5 + 0.5 e_i) gives 8 corner values; velocity = trilinear; the stored rows are
    (d/dx, d/dy, d/dz of the COMPONENT-INDEXED corner vector, v) -- i.e. transposed (Q2); RK4 adds the scalar dt * k_i to ALL
    three interpolants of component i (Q11) with saturation; x += dt / 6 (k1 + 2 k2 + 2 k3 + k4).  Vectorised f64 numpy."""
    rng = np.random.default_rng(6)
    nx, ny, nz = DIM
    o = Oracle(nx, ny, nz, 4000)
    m = np.full((nz, ny, nx), AIR, np.int8)
    m[[0, -1], :, :] = SOLID; m[:, [0, -1], :] = SOLID; m[:, :, [0, -1]] = SOLID
    o.write_volume("marker", m)
    vol = [(rng.standard_normal((nz, ny, nx)) * 8.0).astype(np.float32) for _ in range(3)]     # |v| dt ~ 0.07 cells
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, vol[c])
    pos = (rng.random((3000, 3)) * (np.array(DIM) - 6.0) + 3.0).astype(np.float32)
    o.set_particles(pos)
    o.run_stage("advect", DT)
    got = o.get_particles()
    X = pos.astype(np.float64)
    corners, t = [], []
    for i in range(3):
        off = np.full(3, 0.5); off[i] = 1.0
        oi = np.maximum(0.0, X - off)
        lo = np.floor(oi).astype(int)
        hi = np.minimum(lo + 1, np.array(DIM) - 1)
        t.append(oi - lo)
        V = vol[i].astype(np.float64)
        corners.append({(dx, dy, dz): V[(hi if dz else lo)[:, 2], (hi if dy else lo)[:, 1], (hi if dx else lo)[:, 0]] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)})
    mix = lambda a, b, w: a * (1 - w) + b * w

    def tri(i, tx, ty, tz):
        c = corners[i]
        return mix(mix(mix(c[0, 0, 0], c[1, 0, 0], tx), mix(c[0, 1, 0], c[1, 1, 0], tx), ty), mix(mix(c[0, 0, 1], c[1, 0, 1], tx), mix(c[0, 1, 1], c[1, 1, 1], tx), ty), tz)
    nv = np.stack([tri(i, t[i][:, 0], t[i][:, 1], t[i][:, 2]) for i in range(3)], 1)
    rows = np.zeros((3, len(X), 3))      # rows[axis][:, i] = d(corner vector component i) / d(axis)
    for i in range(3):
        c, (tx, ty, tz) = corners[i], (t[i][:, 0], t[i][:, 1], t[i][:, 2])
        rows[0][:, i] = mix(mix(c[1, 0, 0], c[1, 1, 0], ty), mix(c[1, 0, 1], c[1, 1, 1], ty), tz) - mix(mix(c[0, 0, 0], c[0, 1, 0], ty), mix(c[0, 0, 1], c[0, 1, 1], ty), tz)
        rows[1][:, i] = mix(mix(c[0, 1, 0], c[1, 1, 0], tx), mix(c[0, 1, 1], c[1, 1, 1], tx), tz) - mix(mix(c[0, 0, 0], c[1, 0, 0], tx), mix(c[0, 0, 1], c[1, 0, 1], tx), tz)
        rows[2][:, i] = mix(mix(c[0, 0, 1], c[1, 0, 1], tx), mix(c[0, 1, 1], c[1, 1, 1], tx), ty) - mix(mix(c[0, 0, 0], c[1, 0, 0], tx), mix(c[0, 1, 0], c[1, 1, 0], tx), ty)
    sat = lambda a: np.clip(a, 0.0, 1.0)
    step = lambda k, f: np.stack([tri(i, sat(t[i][:, 0] + f * k[:, i]), sat(t[i][:, 1] + f * k[:, i]), sat(t[i][:, 2] + f * k[:, i])) for i in range(3)], 1)
    k1 = nv
    k2 = step(k1, DT * 0.5)
    k3 = step(k2, DT * 0.5)
    k4 = step(k3, DT)
    newx = X + DT * (1.0 / 6.0) * (k1 + 2.0 * (k2 + k3) + k4)
    assert np.abs(newx - X).max() > 0.05 and np.abs(newx - X).max() < 0.9
    assert np.abs(got[0][:, :3] - newx).max() < 2e-5
    for axis in range(3):                                   # ParticleBufferVelocity{X,Y,Z} = vec4(c{x,y,z}, v_{x,y,z})  (:186-188)
        assert np.abs(got[1 + axis][:, :3] - rows[axis]).max() < 2e-4, axis
        assert np.abs(got[1 + axis][:, 3] - nv[:, axis]).max() < 2e-5, axis


def test_particle_correction_wall_truncation_is_literal():
    """density_projection_correct_particles.comp:47-68 for targets outside [1.001, dim - 1.001] or inside a SOLID cell: the step is
    cut to min(L, (d_k > 0 ? fract(x_k) : 1 - fract(x_k)) / |d_k| - 0.001) along its direction (Q12: the shader's ternary measures
    the distance to the face BEHIND the particle; followed literally), then clamped.  Uniform position-change volumes make the
    sampled step known in closed form; a SOLID block in the interior exercises the marker branch."""
    rng = np.random.default_rng(7)
    nx, ny, nz = DIM
    delta = np.float32([-0.8, 0.45, -0.3])
    o = Oracle(nx, ny, nz, 4000)
    m = np.full((nz, ny, nx), AIR, np.int8)
    m[[0, -1], :, :] = SOLID; m[:, [0, -1], :] = SOLID; m[:, :, [0, -1]] = SOLID
    m[4:8, 5:9, 6:10] = SOLID
    o.write_volume("marker", m)
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, np.full((nz, ny, nx), delta[c], np.float32))
    near_wall = np.stack([rng.uniform(1.001, 1.9, 600), rng.uniform(1.001, ny - 1.001, 600), rng.uniform(1.001, 1.4, 600)], 1)
    near_block = np.stack([rng.uniform(10.0, 10.9, 600), rng.uniform(4.2, 4.99, 600), rng.uniform(4.0, 8.0, 600)], 1)   # +x / below the block
    free = np.stack([rng.uniform(12.0, 17.0, 600), rng.uniform(9.5, 13.0, 600), rng.uniform(3.0, 9.0, 600)], 1)
    pos = np.concatenate([near_wall, near_block, free]).astype(np.float32)
    o.set_particles(pos)
    o.run_stage("correct", DT)
    got = o.get_particles()[0][:, :3].astype(np.float64)
    X = pos.astype(np.float64)
    d64 = delta.astype(np.float64)
    target = X + d64
    lo, hi = 1.001, np.array(DIM) - 1.001
    outside = np.any((target < lo) | (target > hi), axis=1)
    cell = np.clip(np.floor(target).astype(int), 0, np.array(DIM) - 1)
    in_solid = m[cell[:, 2], cell[:, 1], cell[:, 0]] == SOLID
    L = np.linalg.norm(d64) + 1e-10
    direction = d64 / L
    pic = X - np.floor(X)
    limit = np.where(direction > 0, pic, 1.0 - pic) / np.abs(direction) - 0.001
    max_step = np.minimum(L, limit.min(axis=1))
    cut = np.clip(X + direction[None, :] * max_step[:, None], lo, hi)
    want = np.where((outside | in_solid)[:, None], cut, target)
    assert outside.sum() > 300 and (in_solid & ~outside).sum() > 50 and (~outside & ~in_solid).sum() > 500
    assert np.abs(got - want).max() < 5e-6


# ---- second formulations for the kernels that had none ("weak" #1) ------------------------------------------------

def _extrapolate_numpy(m, vel):
    """extrapolate_velocity.comp:26-90, vectorised: for every non-FLUID cell g and component c whose +c neighbour is non-FLUID
    too, the face value becomes the mean of the VALID faces (one side FLUID) among the 8 neighbours in the plane normal to c,
    accumulated in f32 in the shader's order (offsets listed z-row by z-row / y-row by y-row); written only if any is valid."""
    fluid = m == FLUID
    out = [v.copy() for v in vel]
    for c in range(3):
        valid = fluid | _shift(fluid, c, +1, False)                       # isValidVelocity (:9-14); OOB marker reads SOLID
        target = (~fluid) & (~_shift(fluid, c, +1, False))
        others = [a for a in range(3) if a != c]                           # in-plane axes, the first one varies fastest in the shader's list
        num = np.zeros(m.shape, np.float32)
        acc = np.zeros(m.shape, np.float32)
        for d1 in (-1, 0, 1):                                              # slower in-plane axis (z for c = x, y; y for c = z)
            for d0 in (-1, 0, 1):
                if d0 == 0 and d1 == 0:
                    continue
                vs, vv = valid, vel[c]
                for axis, d in ((others[0], d0), (others[1], d1)):
                    if d:
                        vs, vv = _shift(vs, axis, d, False), _shift(vv, axis, d, np.float32(0))
                num = num + vs.astype(np.float32)
                acc = np.where(vs, acc + vv, acc).astype(np.float32)
        write = target & (num > 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[c] = np.where(write, (acc / num).astype(np.float32), out[c])
    return out


def test_position_change_and_extrapolation_equal_vectorised_restatements():
    """R2 (density_projection_position_change.comp:18-51): face (g, c) = (p[g + e_c] - p[g]) * dt with p = 0 outside FLUID and 0
    if either side is SOLID -- every face of the grid is written; followed by D3 (extrapolate_velocity.comp) on the result.
    Both in f32 with the shader's operation order: bit-exact against the oracle's `position_change` stage."""
    rng = np.random.default_rng(21)
    nx, ny, nz = DIM
    m = _random_marker(rng)
    p = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    o = Oracle(nx, ny, nz, 8)
    o.write_volume("marker", m)
    o.write_volume("pressure_density", p)
    for n in ("vel_x", "vel_y", "vel_z"):
        o.write_volume(n, rng.standard_normal((nz, ny, nx)).astype(np.float32))      # must be overwritten everywhere
    o.run_stage("position_change", DT)
    dt = np.float32(DT)
    pf = np.where(m == FLUID, p, np.float32(0)).astype(np.float32)
    r2 = []
    for c in range(3):
        m_n = _shift(m, c, +1, SOLID)
        d = ((_shift(pf, c, +1, np.float32(0)) - pf) * dt).astype(np.float32)
        r2.append(np.where((m == SOLID) | (m_n == SOLID), np.float32(0), d).astype(np.float32))
    want = _extrapolate_numpy(m, r2)
    changed = 0
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        got = o.read_volume(n)
        assert np.array_equal(got.view(np.uint32), want[c].view(np.uint32)), n
        changed += int((want[c] != r2[c]).sum())
    assert changed > 300                                                           # the extrapolation did write faces


def test_extrapolation_after_the_pressure_projection_equals_the_vectorised_restatement():
    """D3 on the output of D2 with p = 0 (D2 then keeps v on valid faces, puts the solid's velocity on solid-sided ones and 0 on
    faces without a FLUID side): the faces D3 may write are exactly those D2 zeroed."""
    rng = np.random.default_rng(22)
    nx, ny, nz = DIM
    m = _random_marker(rng)
    vel = [rng.standard_normal((nz, ny, nx)).astype(np.float32) for _ in range(3)]
    o = Oracle(nx, ny, nz, 8)
    o.write_volume("marker", m)
    o.write_volume("pressure_velocity", np.zeros((nz, ny, nx), np.float32))
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        o.write_volume(n, vel[c])
    o.run_stage("project", DT)
    d2 = []
    for c in range(3):
        m_n = _shift(m, c, +1, SOLID)
        touched = (m == FLUID) | (m_n == FLUID)
        v = np.where(touched, vel[c], np.float32(0))
        v = np.where(touched & ((m == SOLID) | (m_n == SOLID)), np.float32(0), v)      # no solid voxel volume: the solid's velocity is 0
        d2.append(v.astype(np.float32))
    want = _extrapolate_numpy(m, d2)
    for c, n in enumerate(("vel_x", "vel_y", "vel_z")):
        assert np.array_equal(o.read_volume(n).view(np.uint32), want[c].view(np.uint32)), n


def _apply_A(m, x):
    """pressure.glsl:34-75 in f64: (number of non-SOLID neighbours) * x - sum over FLUID neighbours, on FLUID cells."""
    fluid = m == FLUID
    xf = np.where(fluid, x, 0.0).astype(np.float64)
    diag = np.zeros(m.shape)
    nb = np.zeros(m.shape)
    for c in range(3):
        for d in (-1, 1):
            diag += _shift(m, c, d, SOLID) != SOLID
            nb += _shift(xf, c, d, 0.0) * _shift(fluid, c, d, False)
    return np.where(fluid, diag * xf - nb, 0.0)


def test_warm_start_initialisation_is_b_minus_A_p0():
    """S0 (pressure_init.comp:45-83): p := 0 outside FLUID, r := b - A p0 with last step's pressure as warm start.  With
    b = A p0 (f64, rounded to f32) the initial residual is rounding noise, so the solve must leave p0 (masked to FLUID) where it is
    and report a vanishing error -- a wrong sign / diagonal / neighbour rule in S0 would send it off by O(|b|)."""
    rng = np.random.default_rng(23)
    nx, ny, nz = DIM
    m = _random_marker(rng)
    p0 = rng.standard_normal((nz, ny, nx)).astype(np.float32)                      # also garbage outside the fluid: must be zeroed
    b = _apply_A(m, np.where(m == FLUID, p0, 0)).astype(np.float32)
    o = Oracle(nx, ny, nz, 8)
    o.write_volume("marker", m)
    o.write_volume("pressure_velocity", p0)
    o.write_volume("residual", b)
    o.reset_pressure_cleared(0, True)                                              # not the first solve: keep the warm start (pressure_solver.rs:601-603)
    o.set_solver_config(0, error_tolerance=0.0, max_num_iterations=6, error_check_frequency=2)
    o.run_stage("solve_velocity", DT)
    p = o.read_volume("pressure_velocity")
    fluid = m == FLUID
    assert np.all(p[~fluid] == 0)
    assert np.abs(p[fluid] - p0[fluid]).max() < 2e-5
    err, it = o.solver_stats(0)
    assert it == 6 and err < 1e-5 * np.abs(b).max() * DT + 1e-7
    # and the carried residual stays the true residual: r == b - A p (f64)
    r = o.read_volume("residual").astype(np.float64)
    assert np.abs(r[fluid] - (b.astype(np.float64) - _apply_A(m, p))[fluid]).max() < 1e-5


def test_literal_binning_equals_a_numpy_emulation_of_the_three_shaders():
    """Q4, from the shaders alone (particle_binning_count.comp:9-13, _prefixsum.comp:31-61, _rewrite_particles.comp:8-16,
    hybrid_fluid.rs:871-891): ceil(P / 64) * 64 threads without a bounds check bin the zero records behind the live range too,
    the destination `inclusive - slot` is 1-based (slot 0 is never written), and the WHOLE buffer is copied back.  Atomic orders
    fixed like the oracle's (ascending thread / block index).  Compared record by record."""
    dim = (24, 20, 16)
    rng = np.random.default_rng(24)
    cells = np.stack(np.meshgrid(np.arange(2, 12), np.arange(2, 10), np.arange(2, 9), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + rng.random((len(cells), 6, 3))).reshape(-1, 3).astype(np.float32)
    pos = pos[rng.permutation(len(pos))][:3003]                                    # not a multiple of 64
    n, cap = len(pos), len(pos) + 200
    o = Oracle(*dim, cap)
    o.set_quirks(binning="literal")
    o.set_particles(pos)
    o.run_stage("binning", DT)
    got = o.get_particles()[0][:, :3]
    T = (n + 63) // 64 * 64
    old = np.zeros((cap, 3), np.float32)
    old[:n] = pos
    N = dim[0] * dim[1] * dim[2]
    counts = np.zeros(N, np.int64)
    slot = np.zeros(T, np.int64)
    lin = np.zeros(T, np.int64)
    for i in range(T):
        c = old[i].astype(np.int64)                                               # ivec3(Position): truncation
        lin[i] = (c[2] * dim[1] + c[1]) * dim[0] + c[0]                            # x fastest (particle_binning_prefixsum.comp:17-22)
        slot[i] = counts[lin[i]]
        counts[lin[i]] += 1
    inclusive = np.cumsum(counts)
    new = np.zeros((cap, 3), np.float32)                                           # the tmp buffer starts zero-initialised
    for i in range(T):
        d = inclusive[lin[i]] - slot[i]
        if d < cap:                                                               # out-of-bounds stores are dropped
            new[d] = old[i]
    assert np.array_equal(got.view(np.uint32), new[:n].view(np.uint32))
    rec = lambda a: np.sort(np.ascontiguousarray(a).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).reshape(-1), order=("x", "y", "z"))
    lost = n - np.isin(rec(pos), rec(got)).sum()
    assert lost == (T - n) + 1                                                     # pad + 1 live particles are replaced by zero records


def test_reduced_precision_filter_weights_bound_the_particle_correction():
    """R3 reads the position-change volumes through the hardware's LINEAR filter (density_projection_correct_particles.comp:32-40).
    Real GPUs evaluate that filter with fixed-point weights -- 8 fractional bits is the common minimum (Vulkan: subTexelPrecisionBits
    >= 4; D3D requires 8) -- while the oracle (and the HIP path, bit-identically) use exact f32 weights.  This bounds what that
    could move: the oracle's own position-change field of a collapsing dam, sampled at its particles with exact and with
    1/256-quantised weights.  Measured: 3.0e-4 cells at most, for corrections of up to 0.23 cells (a violently collapsing block);
    bound asserted: 5e-4 cells, i.e. inside the 1e-4 .. 1e-3 cells the whole-step parity statements are made at and below the effect of
    the unconverged solver's sensitivity to dot-product rounding."""
    pos, vel, maxp = util.make_dam(32, 24, 24, seed=3)
    o = Oracle(32, 24, 24, maxp)
    o.set_quirks(binning="off")
    o.set_gravity_grid((0.0, -981.0, 0.0))
    o.set_particles(pos, *vel)
    for st in util.STEP_ORDER[:-1]:
        if st != "binning":
            o.run_stage(st, DT)
    vols = [o.read_volume(n).astype(np.float64) for n in ("vel_x", "vel_y", "vel_z")]
    P = o.get_particles()[0][:, :3].astype(np.float64)
    dim = np.array([32, 24, 24])
    worst, scale = 0.0, 0.0
    for c in range(3):
        u = P - 0.5 * np.eye(3)[c] - 0.5
        i0 = np.floor(u).astype(int)
        f = u - i0
        fq = np.round(f * 256.0) / 256.0                                          # 8 fractional bits
        lo = np.clip(i0, 0, dim - 1)
        hi = np.clip(i0 + 1, 0, dim - 1)                                          # clamp-to-edge
        res = []
        for w in (f, fq):
            acc = np.zeros(len(P))
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        wt = (w[:, 0] if dx else 1 - w[:, 0]) * (w[:, 1] if dy else 1 - w[:, 1]) * (w[:, 2] if dz else 1 - w[:, 2])
                        ix, iy, iz = (hi if dx else lo)[:, 0], (hi if dy else lo)[:, 1], (hi if dz else lo)[:, 2]
                        acc += wt * vols[c][iz, iy, ix]
            res.append(acc)
        worst = max(worst, float(np.abs(res[0] - res[1]).max()))
        scale = max(scale, float(np.abs(res[0]).max()))
    print("8-bit filter weights move the density correction by at most %.3g cells (largest correction %.3g cells)" % (worst, scale))
    assert scale > 1e-3 and worst < 5e-4
