"""blub_amd -- MI355X-native APIC fluid-step engine (drop-in for Wumpf/blub's `HybridFluid::step` hot path).

The product is `libblubhip.so` (hand-written HIP for gfx950 behind the C-ABI of include/blubhip.h).  This package is
only the Python mirror of the reference's host surface (src/simulation/hybrid_fluid.rs, src/scene/mod.rs) used by the
tests and the benchmark driver.  There is no CPU fallback: constructing a HybridFluid without a GPU raises.
"""
from .hybrid_fluid import (BlubError, HybridFluid, Scene, SceneConfig, SlabGroup, SolverConfig, SolverStatisticSample, STAGES,  # noqa: F401
                           VOLUMES, MeshDesc, StaticObjectConfig, default_simulation_delta, duration_nanos, lib_path, load_library, load_obj,
                           mesh_desc_at_time, seed_fluid_cube)
