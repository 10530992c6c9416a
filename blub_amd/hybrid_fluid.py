"""ctypes mirror of the reference's `HybridFluid` / `Scene` host surface over libblubhip.so.

Names, argument meaning and error behaviour follow src/simulation/hybrid_fluid.rs and src/scene/mod.rs of the
reference (cited per method) so that the parity tests read like tests of the reference type.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

VOLUMES = {"marker": 0, "linked_list": 1, "vel_x": 2, "vel_y": 3, "vel_z": 4, "pressure_velocity": 5,
           "pressure_density": 6, "residual": 7, "search": 8, "aux": 9, "aux_temp": 10, "solid": 11}
STAGES = {"transfer": 0, "divergence": 1, "solve_velocity": 2, "binning": 3, "project": 4, "advect": 5,
          "density_gather": 6, "solve_density": 7, "position_change": 8, "correct": 9}
STEP_ORDER = ["transfer", "divergence", "solve_velocity", "binning", "project", "advect", "density_gather",
              "solve_density", "position_change", "correct"]
PRECOND = {"zero": 0, "lod0": 1}
BINNING = {"fixed": 0, "literal": 1, "off": 2}
SOLVER_VELOCITY, SOLVER_DENSITY = 0, 1
MAX_CUBES = 64
MAX_STATIC_OBJECTS = 16
MAX_PATH = 256
PROF_MAX = 48


class BlubError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("blubhip error %d: %s" % (status, message))
        self.status = status


class _SolverConfig(C.Structure):
    _fields_ = [("error_tolerance", C.c_float), ("max_num_iterations", C.c_int32), ("error_check_frequency", C.c_int32)]


class _SolverStats(C.Structure):
    _fields_ = [("error", C.c_float), ("iteration_count", C.c_int32)]


class _FluidDesc(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32), ("max_num_particles", C.c_uint32),
                ("device", C.c_int32), ("precond_mode", C.c_uint32), ("binning_mode", C.c_uint32), ("volume_shift_kib", C.c_uint32)]


class StaticObjectConfig(C.Structure):
    """scene/models.rs:11-46 (StaticObjectConfig + RigidAnimation)"""
    _fields_ = [("model", C.c_char * MAX_PATH), ("world_position", C.c_float * 3), ("scale", C.c_float), ("rotation_angles_deg", C.c_float * 3),
                ("has_translation", C.c_uint32), ("translation_target", C.c_float * 3), ("translation_curve", C.c_uint32), ("translation_duration", C.c_float),
                ("has_rotation", C.c_uint32), ("rotation_axis", C.c_float * 3), ("rotation_deg_per_sec", C.c_float)]


class SceneConfig(C.Structure):
    """src/scene/mod.rs:19-43"""
    _fields_ = [("gravity", C.c_float * 3), ("world_position", C.c_float * 3), ("grid_to_world_scale", C.c_float),
                ("grid_dimension", C.c_uint32 * 3), ("max_num_particles", C.c_uint32), ("num_fluid_cubes", C.c_uint32),
                ("cube_min", (C.c_float * 3) * MAX_CUBES), ("cube_max", (C.c_float * 3) * MAX_CUBES),
                ("num_static_objects", C.c_uint32), ("static_objects", StaticObjectConfig * MAX_STATIC_OBJECTS)]


class MeshDesc(C.Structure):
    """The fields of MeshDataGpu (scene/models.rs:55-70) the voxeliser reads; include/blubhip.h blub_mesh_desc."""
    _fields_ = [("voxel_transform", (C.c_float * 4) * 3), ("fluid_space_velocity", C.c_float * 3), ("fluid_space_rotation_axis_scaled", C.c_float * 3),
                ("index_begin", C.c_uint32), ("index_end", C.c_uint32)]


class _DeviceViews(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("particles_position_ll", "particles_velocity_x", "particles_velocity_y",
                                          "particles_velocity_z", "velocity_x", "velocity_y", "velocity_z", "marker",
                                          "pressure_from_velocity", "pressure_from_density", "stream")]


class _TraceEvent(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("stage", C.c_uint32), ("step", C.c_uint32), ("start_us", C.c_double), ("duration_us", C.c_double)]


class _ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


class SolverConfig:
    """pressure_solver.rs:57-62"""

    def __init__(self, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4):
        self.error_tolerance = error_tolerance
        self.max_num_iterations = max_num_iterations
        self.error_check_frequency = error_check_frequency


class SolverStatisticSample:
    """pressure_solver.rs:64-68"""

    def __init__(self, error, iteration_count):
        self.error = error
        self.iteration_count = iteration_count

    def __repr__(self):
        return "SolverStatisticSample(error=%g, iteration_count=%d)" % (self.error, self.iteration_count)


def default_simulation_delta(steps_per_second=120):
    """simulation_controller.rs:33-39: Duration::from_nanos(1e9 / sps).as_secs_f32()"""
    nanos = 1000 * 1000 * 1000 // steps_per_second
    secs, sub = divmod(nanos, 1000 * 1000 * 1000)
    return float(np.float32(secs) + np.float32(sub) / np.float32(1e9))


def lib_path():
    """The in-tree library; BLUBHIP_LIB (development: A/B builds with other compile-time constants) names another build of it."""
    return os.environ.get("BLUBHIP_LIB") or os.path.join(_HERE, "libblubhip.so")


_lib = None


def load_library():
    """Loads libblubhip.so (built in-tree by blub_amd/build.py). Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("libblubhip.so is not built: run `python -m blub_amd.build` (there is no fallback path)")
    L = C.CDLL(path)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
    sig = {
        "blub_scene_load_json": (C.c_int, [C.c_char_p, C.POINTER(SceneConfig)]),
        "blub_scene_parse_json": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(SceneConfig)]),
        "blub_seed_fluid_cube": (C.c_int, [vp, u32, u32, vp, vp, vp, C.c_size_t, C.POINTER(u32)]),
        "blub_fluid_create": (C.c_int, [C.POINTER(_FluidDesc), C.POINTER(vp)]),
        "blub_fluid_create_from_scene": (C.c_int, [C.POINTER(SceneConfig), i32, C.POINTER(vp)]),
        "blub_fluid_destroy": (None, [vp]),
        "blub_last_error_string": (C.c_char_p, []),
        "blub_version_string": (C.c_char_p, []),
        "blub_fluid_add_fluid_cube": (C.c_int, [vp, vp, vp]),
        "blub_fluid_set_gravity_grid": (C.c_int, [vp, vp]),
        "blub_fluid_step": (C.c_int, [vp, f32]),
        "blub_fluid_update_statistics": (C.c_int, [vp]),
        "blub_fluid_synchronize": (C.c_int, [vp]),
        "blub_fluid_set_solver_config": (C.c_int, [vp, C.c_int, C.POINTER(_SolverConfig)]),
        "blub_fluid_get_solver_config": (C.c_int, [vp, C.c_int, C.POINTER(_SolverConfig)]),
        "blub_fluid_solver_stats_count": (C.c_int, [vp, C.c_int]),
        "blub_fluid_solver_stats_get": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(_SolverStats)]),
        "blub_fluid_solver_stats_latest": (C.c_int, [vp, C.c_int, C.POINTER(_SolverStats)]),
        "blub_fluid_set_rebinning_frequency": (C.c_int, [vp, u32]),
        "blub_fluid_get_rebinning_frequency": (u32, [vp]),
        "blub_fluid_num_particles": (u32, [vp]),
        "blub_fluid_max_num_particles": (u32, [vp]),
        "blub_fluid_last_add_dropped": (u32, [vp]),
        "blub_fluid_grid_dimension": (C.c_int, [vp, vp]),
        "blub_fluid_step_counter": (u32, [vp]),
        "blub_fluid_set_step_counter": (C.c_int, [vp, u32]),
        "blub_fluid_get_device_views": (C.c_int, [vp, C.POINTER(_DeviceViews)]),
        "blub_fluid_set_solid_voxels": (C.c_int, [vp, vp]),
        "blub_fluid_set_particles": (C.c_int, [vp, u32, vp, vp, vp, vp]),
        "blub_fluid_get_particles": (C.c_int, [vp, vp, vp, vp, vp]),
        "blub_fluid_volume_bytes": (C.c_size_t, [vp, C.c_int]),
        "blub_fluid_read_volume": (C.c_int, [vp, C.c_int, vp]),
        "blub_fluid_write_volume": (C.c_int, [vp, C.c_int, vp]),
        "blub_fluid_mark_pressure_initialised": (C.c_int, [vp, C.c_int, C.c_int]),
        "blub_fluid_run_stage": (C.c_int, [vp, C.c_int, f32]),
        "blub_fluid_profile_enable": (C.c_int, [vp, C.c_int]),
        "blub_fluid_profile_reset": (C.c_int, [vp]),
        "blub_fluid_profile_read": (C.c_int, [vp, C.POINTER(_ProfEntry), C.c_int, C.POINTER(C.c_int)]),
        "blub_fluid_total_solver_iterations": (C.c_uint64, [vp]),
        "blub_fluid_profile_trace": (C.c_int, [vp, C.POINTER(_TraceEvent), C.c_int, C.POINTER(C.c_int)]),
        "blub_fluid_set_pcg_work_mapping": (C.c_int, [vp, C.c_int]),
        "blub_fluid_get_brick_counts": (C.c_int, [vp, vp]),
        "blub_fluid_set_pcg_schedule": (C.c_int, [vp, C.c_int]),
        "blub_fluid_get_pcg_schedule": (C.c_int, [vp]),
        "blub_fluid_last_solve_path": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "blub_fluid_set_filter_mode": (C.c_int, [vp, C.c_int]),
        "blub_fluid_get_filter_mode": (C.c_int, [vp]),
        "blub_fluid_read_scalar_log": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]),
        "blub_fluid_read_phase_stamps": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "blub_fluid_set_max_steps_in_flight": (C.c_int, [vp, u32]),
        "blub_fluid_set_tuning": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "blub_scene_mesh_desc_at_time": (C.c_int, [C.POINTER(SceneConfig), u32, C.c_uint64, C.c_uint64, C.POINTER(MeshDesc)]),
        "blub_load_obj": (C.c_int, [C.c_char_p, vp, C.c_size_t, C.POINTER(u32), vp, C.c_size_t, C.POINTER(u32)]),
        "blub_fluid_set_meshes": (C.c_int, [vp, u32, vp, u32, vp]),
        "blub_fluid_voxelize": (C.c_int, [vp, u32, C.POINTER(MeshDesc)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError = a symbol declared in include/blubhip.h is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = None  # filled lazily by tests from include/blubhip.h


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(L, rc):
    if rc != 0:
        raise BlubError(rc, L.blub_last_error_string().decode("utf-8", "replace"))


def _vol_dtype(which):
    return {0: np.int8, 1: np.uint32}.get(which, np.float32)


def seed_fluid_cube(grid_dim, max_num_particles, num_particles_before, min_grid, max_grid):
    """Host-only: the particle generator of HybridFluid::add_fluid_cube (hybrid_fluid.rs:609-678)."""
    L = load_library()
    dim = np.asarray(grid_dim, np.uint32)
    mn = np.asarray(min_grid, np.float32)
    mx = np.asarray(max_grid, np.float32)
    cnt = C.c_uint32()
    _check(L, L.blub_seed_fluid_cube(_ptr(dim), max_num_particles, num_particles_before, _ptr(mn), _ptr(mx), None, 0, C.byref(cnt)))
    out = np.zeros((cnt.value, 4), np.float32)
    if cnt.value:
        _check(L, L.blub_seed_fluid_cube(_ptr(dim), max_num_particles, num_particles_before, _ptr(mn), _ptr(mx), _ptr(out), cnt.value, C.byref(cnt)))
    return out


def load_obj(path):
    """Host-only stand-in for tobj::load_obj(triangulate) (scene/models.rs:267-276): (positions (V,3) f32, indices (T*3,) u32)."""
    L = load_library()
    nv, ni = C.c_uint32(), C.c_uint32()
    _check(L, L.blub_load_obj(os.fsencode(path), None, 0, C.byref(nv), None, 0, C.byref(ni)))
    pos, idx = np.zeros((nv.value, 3), np.float32), np.zeros(ni.value, np.uint32)
    _check(L, L.blub_load_obj(os.fsencode(path), _ptr(pos), nv.value, C.byref(nv), _ptr(idx), ni.value, C.byref(ni)))
    return pos, idx


def duration_nanos(seconds):
    """A simulation delta given in (f32) seconds as the integer nanoseconds of the reference's `Duration`."""
    return int(round(float(seconds) * 1e9))


def mesh_desc_at_time(config, object_index, total_simulated_time_ns, simulation_delta_ns):
    """Host-only: StaticMeshData::to_gpu (scene/models.rs:190-228)."""
    L = load_library()
    d = MeshDesc()
    _check(L, L.blub_scene_mesh_desc_at_time(C.byref(config), int(object_index), int(total_simulated_time_ns), int(simulation_delta_ns), C.byref(d)))
    return d


class HybridFluid:
    """src/simulation/hybrid_fluid.rs `HybridFluid` on an MI355X."""

    PARTICLES_PER_GRID_CELL = 8  # hybrid_fluid.rs:90

    def __init__(self, grid_dimension, max_num_particles, device=-1, precond="zero", binning="fixed", _handle=None, volume_shift_kib=0):
        """HybridFluid::new (hybrid_fluid.rs:92-100). grid_dimension = (x, y, z)."""
        self._L = load_library()
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            d = _FluidDesc(int(grid_dimension[0]), int(grid_dimension[1]), int(grid_dimension[2]), int(max_num_particles),
                           int(device), PRECOND[precond], BINNING[binning], int(volume_shift_kib) & 0xFFFFFFFF)
            _check(self._L, self._L.blub_fluid_create(C.byref(d), C.byref(self._h)))
        dim = (C.c_uint32 * 3)()
        _check(self._L, self._L.blub_fluid_grid_dimension(self._h, dim))
        self.nx, self.ny, self.nz = int(dim[0]), int(dim[1]), int(dim[2])

    def close(self):
        if getattr(self, "_h", None):
            self._L.blub_fluid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface ------------------------------------------------------------------------------------
    def add_fluid_cube(self, min_grid, max_grid):
        """hybrid_fluid.rs:620 (grid space)"""
        a = np.asarray(min_grid, np.float32)
        b = np.asarray(max_grid, np.float32)
        _check(self._L, self._L.blub_fluid_add_fluid_cube(self._h, _ptr(a), _ptr(b)))

    def set_gravity_grid(self, gravity):
        """hybrid_fluid.rs:692"""
        g = np.asarray(gravity, np.float32)
        _check(self._L, self._L.blub_fluid_set_gravity_grid(self._h, _ptr(g)))

    def step(self, simulation_delta):
        """hybrid_fluid.rs:770 -- enqueues one step (asynchronous)."""
        _check(self._L, self._L.blub_fluid_step(self._h, float(simulation_delta)))

    def update_statistics(self):
        """hybrid_fluid.rs:765"""
        _check(self._L, self._L.blub_fluid_update_statistics(self._h))

    def synchronize(self):
        _check(self._L, self._L.blub_fluid_synchronize(self._h))

    def _get_cfg(self, which):
        c = _SolverConfig()
        _check(self._L, self._L.blub_fluid_get_solver_config(self._h, which, C.byref(c)))
        return SolverConfig(c.error_tolerance, c.max_num_iterations, c.error_check_frequency)

    def set_solver_config(self, which, cfg=None, **kw):
        cur = self._get_cfg(which) if cfg is None else cfg
        for k, v in kw.items():
            setattr(cur, k, v)
        c = _SolverConfig(float(cur.error_tolerance), int(cur.max_num_iterations), int(cur.error_check_frequency))
        _check(self._L, self._L.blub_fluid_set_solver_config(self._h, which, C.byref(c)))

    def pressure_solver_config_velocity(self):
        """hybrid_fluid.rs:743"""
        return self._get_cfg(SOLVER_VELOCITY)

    def pressure_solver_config_density(self):
        """hybrid_fluid.rs:747"""
        return self._get_cfg(SOLVER_DENSITY)

    def _stats(self, which):
        n = self._L.blub_fluid_solver_stats_count(self._h, which)
        out = []
        for i in range(n):
            s = _SolverStats()
            _check(self._L, self._L.blub_fluid_solver_stats_get(self._h, which, i, C.byref(s)))
            out.append(SolverStatisticSample(s.error, s.iteration_count))
        return out

    def pressure_solver_stats_velocity(self):
        """hybrid_fluid.rs:755"""
        return self._stats(SOLVER_VELOCITY)

    def pressure_solver_stats_density(self):
        """hybrid_fluid.rs:759"""
        return self._stats(SOLVER_DENSITY)

    @property
    def particle_rebinning_step_frequency(self):
        """dynamic_settings(), hybrid_fluid.rs:751"""
        return int(self._L.blub_fluid_get_rebinning_frequency(self._h))

    @particle_rebinning_step_frequency.setter
    def particle_rebinning_step_frequency(self, v):
        _check(self._L, self._L.blub_fluid_set_rebinning_frequency(self._h, int(v)))

    def num_particles(self):
        """hybrid_fluid.rs:696"""
        return int(self._L.blub_fluid_num_particles(self._h))

    def last_add_dropped(self):
        """Particles the last add_fluid_cube could not add (the reference's `error!` case, hybrid_fluid.rs:627-633)."""
        return int(self._L.blub_fluid_last_add_dropped(self._h))

    def grid_dimension(self):
        """hybrid_fluid.rs:727"""
        return (self.nx, self.ny, self.nz)

    def bind_group_renderer(self):
        """hybrid_fluid.rs:723 -- device pointers instead of a wgpu bind group."""
        v = _DeviceViews()
        _check(self._L, self._L.blub_fluid_get_device_views(self._h, C.byref(v)))
        return {n: getattr(v, n) for n, _ in _DeviceViews._fields_}

    # ---- state exchange / test hooks ---------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.nz, self.ny, self.nx)

    @property
    def step_counter(self):
        return int(self._L.blub_fluid_step_counter(self._h))

    @step_counter.setter
    def step_counter(self, v):
        _check(self._L, self._L.blub_fluid_set_step_counter(self._h, int(v)))

    def set_particles(self, pos, vx=None, vy=None, vz=None, keep_ll=False):
        pos = np.asarray(pos, np.float32)
        n = pos.shape[0]
        p4 = np.zeros((n, 4), np.float32)
        p4[:, :3] = pos[:, :3]
        if keep_ll:
            p4[:, 3] = pos[:, 3]
        else:
            p4.view(np.uint32)[:, 3] = 0xFFFFFFFF
        vs = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (vx, vy, vz)]
        _check(self._L, self._L.blub_fluid_set_particles(self._h, n, _ptr(p4), *[_ptr(v) for v in vs]))

    def get_particles(self):
        n = self.num_particles()
        out = [np.zeros((n, 4), np.float32) for _ in range(4)]
        _check(self._L, self._L.blub_fluid_get_particles(self._h, *[_ptr(o) for o in out]))
        return out

    def read_volume(self, name):
        which = VOLUMES[name]
        shape = self.shape + ((4,) if which == 11 else ())
        out = np.zeros(shape, _vol_dtype(which))
        _check(self._L, self._L.blub_fluid_read_volume(self._h, which, _ptr(out)))
        return out

    def write_volume(self, name, arr):
        which = VOLUMES[name]
        if arr is None:
            _check(self._L, self._L.blub_fluid_write_volume(self._h, which, None))
            return
        a = np.ascontiguousarray(arr, _vol_dtype(which))
        assert a.size == self.nx * self.ny * self.nz * (4 if which == 11 else 1)
        _check(self._L, self._L.blub_fluid_write_volume(self._h, which, _ptr(a)))

    def set_solid_voxels(self, vox):
        self.write_volume("solid", vox)

    def set_meshes(self, positions, indices):
        """SceneModels vertex / index buffers (scene/models.rs:354-375)."""
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        _check(self._L, self._L.blub_fluid_set_meshes(self._h, positions.shape[0], _ptr(positions), indices.shape[0], _ptr(indices)))

    def voxelize(self, mesh_descs):
        """SceneVoxelization::update (scene/voxelization.rs:116-157); enqueues on the engine's stream."""
        arr = (MeshDesc * max(1, len(mesh_descs)))(*mesh_descs)
        _check(self._L, self._L.blub_fluid_voxelize(self._h, len(mesh_descs), arr))

    def mark_pressure_initialised(self, which, initialised=True):
        _check(self._L, self._L.blub_fluid_mark_pressure_initialised(self._h, which, int(bool(initialised))))

    def run_stage(self, name, simulation_delta):
        _check(self._L, self._L.blub_fluid_run_stage(self._h, STAGES[name], float(simulation_delta)))

    def solver_stats(self, which):
        """Latest completed sample as (error, iterations); blocks until the stream is idle."""
        self.synchronize()
        s = _SolverStats()
        _check(self._L, self._L.blub_fluid_solver_stats_latest(self._h, which, C.byref(s)))
        return s.error, s.iteration_count

    def set_pcg_work_mapping(self, mode):
        """"auto" | "rows" | "bricks" ("bricks_staged": its former alias) -- performance knob, see include/blubhip.h"""
        _check(self._L, self._L.blub_fluid_set_pcg_work_mapping(self._h, {"auto": -1, "rows": 0, "bricks": 1, "bricks_staged": 2}[mode]))

    def set_pcg_schedule(self, mode):
        """"reference" (two reductions / two kernels per iteration) | "single_reduction" (one kernel per iteration on the brick
        mappings), see include/blubhip.h"""
        _check(self._L, self._L.blub_fluid_set_pcg_schedule(self._h, {"reference": 0, "single_reduction": 1}[mode]))

    def pcg_schedule(self):
        return ("reference", "single_reduction")[int(self._L.blub_fluid_get_pcg_schedule(self._h))]

    FILTER_MODES = {"separable": 0, "weighted": 1, "weighted8": 2}

    def set_filter_mode(self, mode):
        """Arithmetic of the trilinear filter in R3 / A1's push-out: "separable" (default) | "weighted" | "weighted8" (include/blubhip.h)."""
        _check(self._L, self._L.blub_fluid_set_filter_mode(self._h, self.FILTER_MODES[mode]))

    def filter_mode(self):
        return {v: k for k, v in self.FILTER_MODES.items()}[int(self._L.blub_fluid_get_filter_mode(self._h))]

    def last_solve_path(self, which):
        """(schedule, mapping) the most recently enqueued solve `which` ACTUALLY ran (include/blubhip.h: blub_fluid_last_solve_path): the selected
        schedule is a request -- dense rows, the LOD0 reading and solves beyond "pcg1_max_iterations" always run the reference's order."""
        a, b = C.c_int(), C.c_int()
        _check(self._L, self._L.blub_fluid_last_solve_path(self._h, int(which), C.byref(a), C.byref(b)))
        return ("reference", "single_reduction")[a.value], ("rows", "bricks", "lod0_literal")[b.value]

    def phase_stamps(self, which):
        """(64, 8) uint64: the time stamps workgroup 0 of K(i) left at its phase boundaries in the last single-reduction solve `which`, 10 ns ticks
        (include/blubhip.h: blub_fluid_read_phase_stamps; needs set_tuning("pcg_phase_stamps", 1))."""
        out = np.zeros((64, 8), np.uint64)
        _check(self._L, self._L.blub_fluid_read_phase_stamps(self._h, int(which), _ptr(out), 64))
        return out

    def scalar_log(self, which):
        """(iterations, 4) float32: {gamma, delta, max|r|, alpha} of every iteration of the last single-reduction solve `which`
        (include/blubhip.h: blub_fluid_read_scalar_log; needs set_tuning("pcg_scalar_log", 1))."""
        out = np.zeros((1024, 4), np.float32)
        n = C.c_int()
        _check(self._L, self._L.blub_fluid_read_scalar_log(self._h, int(which), _ptr(out), 1024, C.byref(n)))
        return out[:n.value].copy()

    def set_tuning(self, name, value):
        """Performance knobs / test hooks by name (include/blubhip.h: blub_fluid_set_tuning); the library never reads the environment."""
        _check(self._L, self._L.blub_fluid_set_tuning(self._h, name.encode(), int(value)))

    def set_max_steps_in_flight(self, n):
        _check(self._L, self._L.blub_fluid_set_max_steps_in_flight(self._h, int(n)))

    def brick_counts(self):
        out = (C.c_uint32 * 6)()
        _check(self._L, self._L.blub_fluid_get_brick_counts(self._h, out))
        return dict(zip(("fluid", "active", "reset", "stale", "total", "cells_per_brick"), [int(v) for v in out]))

    def total_solver_iterations(self):
        return int(self._L.blub_fluid_total_solver_iterations(self._h))

    def profile_enable(self, enabled=True):
        _check(self._L, self._L.blub_fluid_profile_enable(self._h, int(bool(enabled))))

    def profile_reset(self):
        _check(self._L, self._L.blub_fluid_profile_reset(self._h))

    def profile_trace(self):
        n = C.c_int()
        _check(self._L, self._L.blub_fluid_profile_trace(self._h, None, 0, C.byref(n)))
        ev = (_TraceEvent * max(n.value, 1))()
        _check(self._L, self._L.blub_fluid_profile_trace(self._h, ev, n.value, C.byref(n)))
        return [{"name": ev[i].name.decode(), "stage": int(ev[i].stage), "step": int(ev[i].step), "start_us": float(ev[i].start_us),
                 "duration_us": float(ev[i].duration_us)} for i in range(n.value)]

    def profile_read(self):
        ents = (_ProfEntry * PROF_MAX)()
        n = C.c_int()
        _check(self._L, self._L.blub_fluid_profile_read(self._h, ents, PROF_MAX, C.byref(n)))
        return {ents[i].name.decode(): {"launches": int(ents[i].launches), "total_ms": float(ents[i].total_ms)} for i in range(n.value)}


# wgpu_profiler! scope labels of HybridFluid::step (hybrid_fluid.rs:798-973), indexed by blub_stage
REFERENCE_SCOPES = ["transfer & divergence compute", "compute divergence", "primary pressure solver (divergence)", "Particle Binning",
                    "make velocity grid divergence free + extrapolate velocity grid", "advect particles & write new linked list grid",
                    "density projection: compute density error via gather", "secondary pressure solver (density)",
                    "compute position change + extrapolate velocity grid", "correct particle density error", "brick work lists"]


def write_chrome_trace(fluid, path):
    """Dump the profiled launches as a chrome-trace JSON like the reference's `simulation-trace.json` (gui/mod.rs:487-491):
    one track per simulation step, scopes named after the reference's wgpu_profiler! labels, kernels nested inside."""
    import json
    events = fluid.profile_trace()
    out = []
    for e in events:
        stage = REFERENCE_SCOPES[min(e["stage"], len(REFERENCE_SCOPES) - 1)]
        out.append({"name": e["name"], "cat": stage, "ph": "X", "ts": e["start_us"], "dur": e["duration_us"], "pid": 1, "tid": 1,
                    "args": {"step": e["step"], "scope": stage}})
    # enclosing scopes (one per stage and step)
    groups = {}
    for e in events:
        k = (e["step"], e["stage"])
        a = groups.setdefault(k, [e["start_us"], e["start_us"] + e["duration_us"]])
        a[0] = min(a[0], e["start_us"]); a[1] = max(a[1], e["start_us"] + e["duration_us"])
    for (step, stage), (t0, t1) in groups.items():
        out.append({"name": REFERENCE_SCOPES[min(stage, len(REFERENCE_SCOPES) - 1)], "cat": "scope", "ph": "X", "ts": t0, "dur": t1 - t0, "pid": 1, "tid": 0, "args": {"step": step}})
    with open(path, "w") as f:
        json.dump({"traceEvents": sorted(out, key=lambda e: e["ts"]), "displayTimeUnit": "ns"}, f)
    return len(events)


class Scene:
    """src/scene/mod.rs `Scene` reduced to what the hot path needs: JSON -> HybridFluid (:56-144) and step (:166-213)."""

    def __init__(self, path=None, text=None, device=-1, config=None):
        self._L = load_library()
        self.config = SceneConfig() if config is None else config
        if path is not None:
            _check(self._L, self._L.blub_scene_load_json(os.fsencode(path), C.byref(self.config)))
        elif text is not None:
            b = text.encode() if isinstance(text, str) else text
            _check(self._L, self._L.blub_scene_parse_json(b, len(b), C.byref(self.config)))
        self._device = device
        self._fluid = None
        self._path = path
        self.models_dir = None if path is None else os.path.join(os.path.dirname(os.path.abspath(path)), "models")
        self.total_simulated_time_ns = 0
        self._meshes = None   # [(object index, index_begin, index_end)]

    @staticmethod
    def parse(path=None, text=None):
        """Host-only parse (no GPU needed)."""
        s = Scene.__new__(Scene)
        s._L = load_library()
        s.config = SceneConfig()
        if path is not None:
            _check(s._L, s._L.blub_scene_load_json(os.fsencode(path), C.byref(s.config)))
        else:
            b = text.encode() if isinstance(text, str) else text
            _check(s._L, s._L.blub_scene_parse_json(b, len(b), C.byref(s.config)))
        s._device = -1
        s._fluid = None
        s._path = path
        s.models_dir = None if path is None else os.path.join(os.path.dirname(os.path.abspath(path)), "models")
        s.total_simulated_time_ns = 0
        s._meshes = None
        return s

    def fluid(self):
        """scene/mod.rs:216; created lazily by create_fluid_from_config (:109-144)."""
        if self._fluid is None:
            h = C.c_void_p()
            _check(self._L, self._L.blub_fluid_create_from_scene(C.byref(self.config), self._device, C.byref(h)))
            self._fluid = HybridFluid(None, None, _handle=h)
        return self._fluid

    def reset(self):
        """scene/mod.rs:146"""
        if self._fluid is not None:
            self._fluid.close()
            self._fluid = None
        self._meshes = None
        return self.fluid()

    def static_objects(self):
        return [self.config.static_objects[i] for i in range(self.config.num_static_objects)]

    def _load_models(self, fluid):
        """SceneModels::from_config (scene/models.rs:255-376): one shared vertex / index buffer, one mesh per object (the
        reference splits an OBJ by material for texturing only; the voxeliser draws all of its index ranges alike)."""
        positions, indices, meshes = [], [], []
        nv = ni = 0
        for i, o in enumerate(self.static_objects()):
            if self.models_dir is None:
                raise BlubError(-5, "scene has static objects but no models directory (Scene.models_dir)")
            p, idx = load_obj(os.path.join(self.models_dir, o.model.decode()))
            positions.append(p)
            indices.append(idx + np.uint32(nv))
            meshes.append((i, ni, ni + len(idx)))
            nv += len(p)
            ni += len(idx)
        if meshes:
            fluid.set_meshes(np.concatenate(positions), np.concatenate(indices))
        self._meshes = meshes

    def mesh_descs(self, simulation_delta_ns):
        """SceneModels::step (scene/models.rs:379-387) at the current total simulated time."""
        out = []
        for i, begin, end in self._meshes or []:
            d = mesh_desc_at_time(self.config, i, self.total_simulated_time_ns, simulation_delta_ns)
            d.index_begin, d.index_end = begin, end
            out.append(d)
        return out

    def step(self, simulation_delta):
        """scene/mod.rs:166-213: [animate models, voxelize scene,] HybridFluid::step, submit, update_statistics.  The timer has
        already advanced by the step being taken when Scene::step runs (timer.rs:124, simulation_controller.rs:213-214)."""
        f = self.fluid()
        delta_ns = duration_nanos(simulation_delta)
        self.total_simulated_time_ns += delta_ns
        if self.config.num_static_objects:
            if self._meshes is None:
                self._load_models(f)
            f.voxelize(self.mesh_descs(delta_ns))
        f.step(simulation_delta)
        f.update_statistics()


class SlabGroup:
    """z-slab domain decomposition (include/blubhip.h, blub_slab_group_*): `HybridFluid::step` over several slabs.

    local=S      : S slabs in this process on one GPU (loopback transport; validates the protocol on a single GPU)
    rank, world  : one slab per process, RCCL transport; the 128-byte RCCL id is created on rank 0 and broadcast through
                   torch.distributed by `SlabGroup.from_torch_distributed`.
    """

    MEMORY_MODES = {"coarse": 0, "fine_grained": 1, "uncached": 2}      # include/blubhip.h: BLUB_SLAB_MEMORY_*

    def __init__(self, grid_dimension, max_num_particles, local=None, rank=None, world=None, unique_id=None, device=-1, binning="fixed", cuts=None, memory="coarse",
                 movable_cuts=False):
        self._L = load_library()
        L = self._L
        vp = C.c_void_p
        for name, res, args in [
                ("blub_rccl_unique_id", C.c_int, [vp]), ("blub_slab_range", C.c_int, [C.c_uint32, C.c_int, C.c_int, vp, vp]),
                ("blub_slab_group_create_local", C.c_int, [C.POINTER(_FluidDesc), C.c_int, C.POINTER(vp)]),
                ("blub_slab_group_create_rccl", C.c_int, [C.POINTER(_FluidDesc), C.c_int, C.c_int, vp, C.POINTER(vp)]),
                ("blub_slab_group_create_local_ex", C.c_int, [C.POINTER(_FluidDesc), C.c_int, vp, C.c_uint32, C.POINTER(vp)]),
                ("blub_slab_group_create_rccl_ex", C.c_int, [C.POINTER(_FluidDesc), C.c_int, C.c_int, vp, vp, C.c_uint32, C.POINTER(vp)]),
                ("blub_slab_group_cuts", C.c_int, [vp, vp]),
                ("blub_slab_group_recut", C.c_int, [vp, vp]), ("blub_slab_group_rebalance", C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
                ("blub_slab_group_set_checkpoint_interval", C.c_int, [vp, C.c_uint32]), ("blub_slab_group_checkpoints", C.c_int, [vp, vp]),
                ("blub_slab_group_exchange_sequence", C.c_int, [vp, C.POINTER(C.c_uint32)]), ("blub_slab_group_restore", C.c_int, [vp, C.c_uint32, C.c_uint32]),
                ("blub_slab_group_destroy", None, [vp]), ("blub_slab_group_num_local", C.c_int, [vp]),
                ("blub_slab_group_local_fluid", vp, [vp, C.c_int]), ("blub_slab_group_local_range", C.c_int, [vp, C.c_int, vp, vp]),
                ("blub_slab_group_set_particles", C.c_int, [vp, C.c_uint32, vp, vp, vp, vp]), ("blub_slab_group_num_particles", C.c_uint32, [vp]),
                ("blub_slab_group_get_particles", C.c_int, [vp, vp, vp, vp, vp]), ("blub_slab_group_set_gravity_grid", C.c_int, [vp, vp]),
                ("blub_slab_group_set_solver_config", C.c_int, [vp, C.c_int, C.POINTER(_SolverConfig)]),
                ("blub_slab_group_set_rebinning_frequency", C.c_int, [vp, C.c_uint32]),
                ("blub_slab_group_set_pcg_schedule", C.c_int, [vp, C.c_int]), ("blub_slab_group_set_gather_mode", C.c_int, [vp, C.c_int]),
                ("blub_slab_group_set_async_exchange", C.c_int, [vp, C.c_int]),
                ("blub_slab_group_host_syncs", C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
                ("blub_slab_group_held_back", C.c_uint64, [vp]),
                ("blub_slab_group_run_stages", C.c_int, [vp, C.c_float, C.c_int, C.c_int]),
                ("blub_slab_group_set_transport", C.c_int, [vp, C.c_int]), ("blub_slab_group_get_transport", C.c_int, [vp]),
                ("blub_slab_group_export_size", C.c_int, [vp]), ("blub_slab_group_export", C.c_int, [vp, vp, C.c_int]),
                ("blub_slab_group_connect", C.c_int, [vp, C.c_int, vp, C.c_int]),
                ("blub_slab_group_step", C.c_int, [vp, C.c_float]), ("blub_slab_group_synchronize", C.c_int, [vp]),
                ("blub_slab_group_transport_ops", C.c_uint64, [vp]), ("blub_slab_group_transport_description", C.c_char_p, [vp]),
                ("blub_slab_group_set_meshes", C.c_int, [vp, C.c_uint32, vp, C.c_uint32, vp]),
                ("blub_slab_group_voxelize", C.c_int, [vp, C.c_uint32, C.POINTER(MeshDesc)])]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        self.grid = tuple(int(v) for v in grid_dimension)
        d = _FluidDesc(self.grid[0], self.grid[1], self.grid[2], int(max_num_particles), int(device), 0, BINNING[binning], 0)
        self._g = vp()
        self.num_slabs = int(local if local is not None else world)
        cut_arr = None if cuts is None else np.ascontiguousarray(cuts, np.int32)      # cut planes (include/blubhip.h: blub_slab_group_create_*_cuts); None = uniform
        if cut_arr is not None and cut_arr.shape != (self.num_slabs + 1,):
            raise ValueError("cuts must hold num_slabs + 1 planes")
        if local is not None:
            _check(L, L.blub_slab_group_create_local_ex(C.byref(d), int(local), _ptr(cut_arr), self.MEMORY_MODES[memory] | (0x100 if movable_cuts else 0), C.byref(self._g)))
        else:
            buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
            _check(L, L.blub_slab_group_create_rccl_ex(C.byref(d), int(rank), int(world), buf, _ptr(cut_arr), self.MEMORY_MODES[memory] | (0x100 if movable_cuts else 0), C.byref(self._g)))

    @staticmethod
    def unique_id():
        L = load_library()
        L.blub_rccl_unique_id.restype, L.blub_rccl_unique_id.argtypes = C.c_int, [C.c_void_p]
        buf = (C.c_uint8 * 128)()
        _check(L, L.blub_rccl_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def slab_range(nz, num_slabs, index):
        """Host-only: the z-range [z0, z1) of slab `index` (whole 4-cell brick layers, as even as possible)."""
        L = load_library()
        L.blub_slab_range.restype, L.blub_slab_range.argtypes = C.c_int, [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        a, b = C.c_int32(), C.c_int32()
        _check(L, L.blub_slab_range(int(nz), int(num_slabs), int(index), C.byref(a), C.byref(b)))
        return a.value, b.value

    @staticmethod
    def balanced_cuts(grid_dimension, pos, num_slabs, min_layers=1):
        """Host-only: (cuts, fluid_bricks_per_slab) -- cut planes that give every slab about the same number of FLUID bricks for these particle
        positions (include/blubhip.h: blub_slab_balanced_cuts)."""
        L = load_library()
        L.blub_slab_balanced_cuts.restype = C.c_int
        L.blub_slab_balanced_cuts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        dim = np.asarray(grid_dimension, np.uint32)
        pos = np.asarray(pos, np.float32)
        p4 = pos if (pos.ndim == 2 and pos.shape[1] == 4 and pos.flags.c_contiguous) else np.concatenate([pos[:, :3], np.zeros((len(pos), 1), np.float32)], axis=1)
        p4 = np.ascontiguousarray(p4, np.float32)
        cuts, bricks = np.zeros(num_slabs + 1, np.int32), np.zeros(num_slabs, np.uint32)
        _check(L, L.blub_slab_balanced_cuts(_ptr(dim), len(p4), _ptr(p4), int(num_slabs), int(min_layers), _ptr(cuts), _ptr(bricks)))
        return [int(c) for c in cuts], [int(b) for b in bricks]

    @staticmethod
    def fluid_bricks_per_slab(grid_dimension, pos, cuts):
        """Host-only: FLUID 16x8x4 bricks (bricks that hold a particle) inside each of the slabs the cut planes define."""
        pos = np.asarray(pos, np.float32)
        nx, ny, nz = (int(v) for v in grid_dimension)
        b = (pos[:, :3].astype(np.int64) // np.array([16, 8, 4]))
        key = np.unique((b[:, 2] * ((ny + 7) // 8) + b[:, 1]) * ((nx + 15) // 16) + b[:, 0])
        layer = key // (((ny + 7) // 8) * ((nx + 15) // 16))
        edges = np.asarray(cuts, np.int64) // 4
        edges[-1] = (nz + 3) // 4
        return [int(((layer >= edges[r]) & (layer < edges[r + 1])).sum()) for r in range(len(cuts) - 1)]

    def recut(self, cuts):
        """COLLECTIVE, between steps, groups created with movable_cuts=True: move the cut planes (each strictly between its old neighbours);
        include/blubhip.h: blub_slab_group_recut"""
        a = np.ascontiguousarray(cuts, np.int32)
        if a.shape != (self.num_slabs + 1,):
            raise ValueError("cuts must hold num_slabs + 1 planes")
        _check(self._L, self._L.blub_slab_group_recut(self._g, _ptr(a)))

    def rebalance(self, min_layers=1):
        """COLLECTIVE: re-cut towards equal FLUID bricks per slab if that helps (include/blubhip.h: blub_slab_group_rebalance); True if the cuts moved"""
        ch = C.c_int(0)
        _check(self._L, self._L.blub_slab_group_rebalance(self._g, int(min_layers), C.byref(ch)))
        return bool(ch.value)

    def cuts(self):
        out = np.zeros(self.num_slabs + 1, np.int32)
        _check(self._L, self._L.blub_slab_group_cuts(self._g, _ptr(out)))
        return [int(c) for c in out]

    @staticmethod
    def from_torch_distributed(grid_dimension, max_num_particles, device=-1, binning="fixed", cuts=None, memory="coarse", movable_cuts=False):
        """One slab per rank of the default process group; rank 0's RCCL id is broadcast (works over gloo or nccl)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        payload = [SlabGroup.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(payload, src=0)
        return SlabGroup(grid_dimension, max_num_particles, rank=rank, world=world, unique_id=payload[0], device=device, binning=binning, cuts=cuts, memory=memory, movable_cuts=movable_cuts)

    def connect_direct_over_torch_distributed(self):
        """DIRECT transport between the ranks of the default process group: every rank's hipIpc handles are all-gathered over the control plane,
        every rank maps everybody else's slab, and all ranks switch together -- or none does (returns False; the RCCL transport stays)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        try:
            blob = self.export_handles()
        except BlubError:
            blob = None
        blobs = [None] * world
        dist.all_gather_object(blobs, blob)
        ok = all(b is not None for b in blobs)
        if ok:
            try:
                for r in range(world):
                    if r != rank:
                        self.connect(r, blobs[r])
            except BlubError:
                ok = False
        verdict = [None] * world
        dist.all_gather_object(verdict, ok)
        if not all(verdict):
            return False
        self.set_transport("direct")
        return True

    # ---- checkpoints and in-place recovery (include/blubhip.h: blub_slab_group_set_checkpoint_interval ... _restore) ----
    def set_checkpoint_interval(self, every_n_steps):
        _check(self._L, self._L.blub_slab_group_set_checkpoint_interval(self._g, int(every_n_steps)))

    def checkpoints(self):
        """step numbers of the (up to two) checkpoint generations this rank holds; blocks"""
        out = (C.c_uint32 * 2)()
        _check(self._L, self._L.blub_slab_group_checkpoints(self._g, out))
        return sorted(int(v) for v in out if int(v) != 0xFFFFFFFF)

    def exchange_sequence(self):
        v = C.c_uint32()
        _check(self._L, self._L.blub_slab_group_exchange_sequence(self._g, C.byref(v)))
        return int(v.value)

    def restore(self, step, sequence_base):
        _check(self._L, self._L.blub_slab_group_restore(self._g, int(step), int(sequence_base)))

    def recover(self, all_gather, barrier):
        """In-place recovery after a timed-out wait of the direct transport (a peer seconds late): COLLECTIVE -- every rank calls it, whether or not it saw
        the error itself.  `all_gather(obj) -> [obj of rank 0, ...]` and `barrier()` are the caller's control plane.  Returns the step the group is back
        at (the newest checkpoint every rank holds): the caller steps on from there.  Raises if no common checkpoint exists."""
        failure = None
        try:
            self.synchronize()          # drains the stream (every wait is bounded); reports the time-out once and clears the mark
        except BlubError as e:
            if e.status != -8:          # BLUB_ERR_COMM is what we are here for; anything else is raised COLLECTIVELY below (round-5 ADVICE: a rank
                failure = "%d: %s" % (e.status, e)      # that raised here left its peers blocked in the all-gather until the caller's watchdog fired)
        info = all_gather((self.checkpoints() if failure is None else [], self.exchange_sequence() if failure is None else 0, failure))
        failed = [(r, i[2]) for r, i in enumerate(info) if i[2] is not None]
        if failed:
            raise BlubError(-4, "in-place recovery impossible: rank(s) %s report an error that is not a transport time-out" % failed)
        common = set(info[0][0])
        for steps, _, _ in info[1:]:
            common &= set(steps)
        if not common:
            raise BlubError(-8, "no checkpoint generation is held by every rank: %s" % [i[0] for i in info])
        step = max(common)
        # above every rank's number, over the full 32 bits (waits compare by signed DIFFERENCE and the counter skips 0 on wrap: a 31-bit mask made the new
        # base SMALLER than stale flags once a counter had passed 2^31 -- round-5 ADVICE)
        base = (max(seq for _, seq, _ in info) + 1024) & 0xFFFFFFFF or 1
        barrier()                       # nobody rewrites its state while a peer's kernels may still be storing into it
        self.restore(step, base)
        barrier()
        return step

    def recover_over_torch_distributed(self):
        import torch.distributed as dist

        def all_gather(obj):
            out = [None] * dist.get_world_size()
            dist.all_gather_object(out, obj)
            return out
        return self.recover(all_gather, dist.barrier)

    def close(self):
        if getattr(self, "_g", None):
            self._L.blub_slab_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_local(self):
        return int(self._L.blub_slab_group_num_local(self._g))

    def local_range(self, i):
        a, b = C.c_int32(), C.c_int32()
        _check(self._L, self._L.blub_slab_group_local_range(self._g, i, C.byref(a), C.byref(b)))
        return a.value, b.value

    def local_fluid(self, i):
        """Borrowed HybridFluid view of local slab i (do not close it)."""
        h = self._L.blub_slab_group_local_fluid(self._g, i)
        f = HybridFluid(None, None, _handle=C.c_void_p(h))
        f.close = lambda: None
        return f

    def set_particles(self, pos, vx=None, vy=None, vz=None):
        pos = np.asarray(pos, np.float32)
        n = pos.shape[0]
        p4 = np.zeros((n, 4), np.float32)
        p4[:, :3] = pos[:, :3]
        p4.view(np.uint32)[:, 3] = 0xFFFFFFFF
        vs = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (vx, vy, vz)]
        _check(self._L, self._L.blub_slab_group_set_particles(self._g, n, _ptr(p4), *[_ptr(v) for v in vs]))

    def num_particles(self):
        return int(self._L.blub_slab_group_num_particles(self._g))

    def get_particles(self):
        self.synchronize()
        n = self.num_particles()
        out = [np.zeros((n, 4), np.float32) for _ in range(4)]
        _check(self._L, self._L.blub_slab_group_get_particles(self._g, *[_ptr(o) for o in out]))
        return out

    def set_gravity_grid(self, g):
        a = np.asarray(g, np.float32)
        _check(self._L, self._L.blub_slab_group_set_gravity_grid(self._g, _ptr(a)))

    def set_solver_config(self, which, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4):
        c = _SolverConfig(float(error_tolerance), int(max_num_iterations), int(error_check_frequency))
        _check(self._L, self._L.blub_slab_group_set_solver_config(self._g, which, C.byref(c)))

    def set_rebinning_frequency(self, f):
        _check(self._L, self._L.blub_slab_group_set_rebinning_frequency(self._g, int(f)))

    def set_pcg_schedule(self, mode):
        """"reference" | "single_reduction" on every local slab (all ranks must pass the same mode)"""
        _check(self._L, self._L.blub_slab_group_set_pcg_schedule(self._g, {"reference": 0, "single_reduction": 1}[mode]))

    def set_async_exchange(self, enabled):
        """Particle exchanges without host synchronisation (default on), see include/blubhip.h; all ranks must agree."""
        _check(self._L, self._L.blub_slab_group_set_async_exchange(self._g, 1 if enabled else 0))

    def host_syncs(self):
        """(stream synchronisations by particle exchanges, by looks at a solve's `done`) issued inside step() so far."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self._L, self._L.blub_slab_group_host_syncs(self._g, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    SLAB_STAGES = {"ghosts": 0, "transfer": 1, "divergence": 2, "solve_velocity": 3, "binning": 4, "project": 5, "advect": 6, "migrate": 7,
                   "density_gather": 8, "solve_density": 9, "position_change": 10, "correct": 11, "migrate_b": 12, "finish": 13}

    def set_transport(self, kind):
        """"host": host-issued transport operations between the kernels (device copies in a local group, RCCL between processes);
        "direct": peer-mapped stores + flags, no host in the loop (include/blubhip.h: blub_slab_group_set_transport).  Between steps; all ranks agree."""
        _check(self._L, self._L.blub_slab_group_set_transport(self._g, {"host": 0, "direct": 1}[kind]))

    def transport(self):
        return ("host", "direct")[int(self._L.blub_slab_group_get_transport(self._g))]

    def export_handles(self):
        """bytes for the other ranks' connect(): one hipIpc handle + size per exportable allocation of the local slab"""
        n = int(self._L.blub_slab_group_export_size(self._g))
        buf = (C.c_uint8 * n)()
        _check(self._L, self._L.blub_slab_group_export(self._g, buf, n))
        return bytes(buf)

    def connect(self, rank, blob):
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        _check(self._L, self._L.blub_slab_group_connect(self._g, int(rank), buf, len(blob)))

    def run_stages(self, simulation_delta, first, last):
        """TEST HOOK (include/blubhip.h: blub_slab_group_run_stages): segments first..last (names of SLAB_STAGES) of one step."""
        _check(self._L, self._L.blub_slab_group_run_stages(self._g, float(simulation_delta), self.SLAB_STAGES[first], self.SLAB_STAGES[last]))

    def held_back(self):
        """Particles a migration held back for one exchange + ghost copies left out for one step because a message was sized too
        small from the previous step's count (include/blubhip.h: blub_slab_group_held_back); blocks."""
        return int(self._L.blub_slab_group_held_back(self._g))

    def set_gather_mode(self, mode):
        """RCCL transport of the PCG partials: "p2p" (grouped send/recv fused with the halo) | "allgather"; all ranks must agree."""
        _check(self._L, self._L.blub_slab_group_set_gather_mode(self._g, {"p2p": 0, "allgather": 1}[mode]))

    def step(self, simulation_delta):
        _check(self._L, self._L.blub_slab_group_step(self._g, float(simulation_delta)))

    def synchronize(self):
        _check(self._L, self._L.blub_slab_group_synchronize(self._g))

    def set_meshes(self, positions, indices):
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        _check(self._L, self._L.blub_slab_group_set_meshes(self._g, positions.shape[0], _ptr(positions), indices.shape[0], _ptr(indices)))

    def voxelize(self, mesh_descs):
        arr = (MeshDesc * max(1, len(mesh_descs)))(*mesh_descs)
        _check(self._L, self._L.blub_slab_group_voxelize(self._g, len(mesh_descs), arr))

    def transport_description(self):
        return self._L.blub_slab_group_transport_description(self._g).decode()

    def transport_ops(self):
        """Grouped transport operations (halo / partial / particle exchanges) issued by this process so far."""
        return int(self._L.blub_slab_group_transport_ops(self._g))
