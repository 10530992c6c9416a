"""Host-side helpers of the z-slab decomposition (no GPU needed): building the weak-scaling scene, seeding its particles
with the reference's generator, and splitting particles over slabs exactly as blub_slab_group_set_particles does."""
import numpy as np

from .hybrid_fluid import SlabGroup, seed_fluid_cube


def weak_scaling_scene(config, num_slabs):
    """N copies of the scene stacked along z: grid (nx, ny, N*nz), every fluid cube repeated with a z offset of one
    slab (world units).  Returns (grid_dimension, scale, gravity_grid, [(min_grid, max_grid), ...], max_particles)."""
    dim = [int(v) for v in config.grid_dimension]
    scale = np.float32(config.grid_to_world_scale)
    cubes = []
    for k in range(num_slabs):
        for i in range(config.num_fluid_cubes):
            mn = np.float32(list(config.cube_min[i])) / scale
            mx = np.float32(list(config.cube_max[i])) / scale
            off = np.float32([0, 0, k * dim[2]])
            cubes.append((mn + off, mx + off))
    gravity = np.float32(list(config.gravity)) / scale
    return (dim[0], dim[1], dim[2] * num_slabs), float(scale), gravity, cubes, int(config.max_num_particles) * num_slabs


def seed_scene_particles(grid_dimension, max_particles, cubes):
    """HybridFluid::add_fluid_cube for every cube (hybrid_fluid.rs:620-678), on the host; returns (n, 4) float32."""
    parts, total = [], 0
    for mn, mx in cubes:
        p = seed_fluid_cube(grid_dimension, max_particles, total, mn, mx)
        total += len(p)
        parts.append(p)
    return np.concatenate(parts) if parts else np.zeros((0, 4), np.float32)


def partition_particles(pos, nz, num_slabs, index, cuts=None):
    """Indices of the particles slab `index` owns: z in [z0, z1), the first / last slab also keep what lies outside.  `cuts`: the group's cut
    planes (SlabGroup.cuts()); None = uniform."""
    z0, z1 = SlabGroup.slab_range(nz, num_slabs, index) if cuts is None else (int(cuts[index]), min(int(cuts[index + 1]), nz))
    z = pos[:, 2]
    keep = np.ones(len(pos), bool)
    if index > 0:
        keep &= z >= np.float32(z0)
    if index + 1 < num_slabs:
        keep &= z < np.float32(z1)
    return np.nonzero(keep)[0], (z0, z1)
