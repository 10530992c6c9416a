"""ctypes binding of oracle/libbluboracle.so -- the CPU restatement of blub's HybridFluid::step.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Parity unpinned (the reference has no tests / cannot run here): see blub_oracle.cpp's header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbluboracle.so")

VOLUMES = {"marker": 0, "linked_list": 1, "vel_x": 2, "vel_y": 3, "vel_z": 4, "pressure_velocity": 5,
           "pressure_density": 6, "residual": 7, "search": 8, "aux": 9, "aux_temp": 10, "solid": 11}
STAGES = {"transfer": 0, "divergence": 1, "solve_velocity": 2, "binning": 3, "project": 4, "advect": 5,
          "density_gather": 6, "solve_density": 7, "position_change": 8, "correct": 9}
PRECOND = {"zero": 0, "lod0": 1}
BINNING = {"fixed": 0, "literal": 1, "off": 2}


def build_oracle(force=False):
    src = os.path.join(_HERE, "blub_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libbluboracle.so"])
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build_oracle()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32]
        for name in ("orc_destroy", "orc_set_gravity_grid", "orc_set_solver_config", "orc_set_quirks",
                     "orc_set_rebinning_frequency", "orc_reset_pressure_cleared", "orc_get_particles", "orc_run_stage",
                     "orc_step", "orc_get_solver_stats", "orc_get_solver_totals", "orc_set_step_counter",
                     "orc_rng_from_seed", "orc_rng_seed_from_u64"):
            getattr(L, name).restype = None
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_add_fluid_cube.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_gravity_grid.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_solver_config.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_set_quirks.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_set_rebinning_frequency.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_reset_pressure_cleared.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_num_particles.restype = C.c_uint32
        L.orc_num_particles.argtypes = [C.c_void_p]
        L.orc_step_counter.restype = C.c_uint32
        L.orc_step_counter.argtypes = [C.c_void_p]
        L.orc_set_step_counter.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_set_particles.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_get_particles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_read_volume.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_write_volume.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_run_stage.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_step.argtypes = [C.c_void_p, C.c_float]
        L.orc_get_solver_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_solver_totals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rng_from_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_rng_seed_from_u64.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_voxelize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_f16_round.restype = C.c_float
        L.orc_f16_round.argtypes = [C.c_float]
        L.orc_set_dot_mode.restype = None
        L.orc_set_dot_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_num_threads.restype = None
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_max_threads.restype = C.c_int
        L.orc_set_num_threads(usable_cpus())
        _lib = L
    return _lib


def usable_cpus(cap=32):
    """CPUs this process may really use: affinity mask and cgroup quota (OpenMP's default = all logical CPUs of the host
    oversubscribes badly inside a CPU-limited container)."""
    if "OMP_NUM_THREADS" in os.environ:
        try:
            return max(1, int(os.environ["OMP_NUM_THREADS"]))
        except ValueError:
            pass
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def num_threads():
    return int(_load().orc_get_max_threads())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _vol_dtype(which):
    return {0: np.int8, 1: np.uint32, 11: np.float32}.get(which, np.float32)


class Oracle:
    """Same surface as blub_amd.HybridFluid (which mirrors src/simulation/hybrid_fluid.rs), on the CPU."""

    def __init__(self, nx, ny, nz, max_num_particles):
        self._L = _load()
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.max_num_particles = int(max_num_particles)
        self._h = C.c_void_p(self._L.orc_create(self.nx, self.ny, self.nz, self.max_num_particles))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    @property
    def shape(self):
        return (self.nz, self.ny, self.nx)

    def add_fluid_cube(self, min_grid, max_grid):
        a = np.asarray(min_grid, np.float32)
        b = np.asarray(max_grid, np.float32)
        return self._L.orc_add_fluid_cube(self._h, _ptr(a), _ptr(b))

    def set_gravity_grid(self, g):
        a = np.asarray(g, np.float32)
        self._L.orc_set_gravity_grid(self._h, _ptr(a))

    def set_solver_config(self, which, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4):
        self._L.orc_set_solver_config(self._h, int(which), float(error_tolerance), int(max_num_iterations), int(error_check_frequency))

    def set_quirks(self, precond="zero", binning="fixed"):
        self._L.orc_set_quirks(self._h, PRECOND[precond], BINNING[binning])

    def set_dot_mode(self, mode):
        """0: PCG dot products accumulated in f64 (default), 1: in f32 (rows -> planes -> total; a sensitivity probe), 2: the
        reference's own reduction, literally (WG-linear reduce buffer, 16 strided reads per thread, 1024 -> 1 tree:
        pressure_reduce.comp:35-61, pressure_solver.rs:543-589) -- bit-exact against oracle/_ref (tests/test_oracle_vs_ref.py)."""
        self._L.orc_set_dot_mode(self._h, int(mode))

    def set_rebinning_frequency(self, f):
        self._L.orc_set_rebinning_frequency(self._h, int(f))

    def reset_pressure_cleared(self, which, cleared):
        self._L.orc_reset_pressure_cleared(self._h, int(which), int(bool(cleared)))

    @property
    def num_particles(self):
        return int(self._L.orc_num_particles(self._h))

    @property
    def step_counter(self):
        return int(self._L.orc_step_counter(self._h))

    @step_counter.setter
    def step_counter(self, v):
        self._L.orc_set_step_counter(self._h, int(v))

    def set_particles(self, pos, vx=None, vy=None, vz=None, keep_ll=False):
        """pos: (n,3) or (n,4) float32 (4th column = linked-list bits, kept only with keep_ll)."""
        pos = np.asarray(pos, np.float32)
        n = pos.shape[0]
        p4 = np.zeros((n, 4), np.float32)
        p4[:, :3] = pos[:, :3]
        if keep_ll:
            p4[:, 3] = pos[:, 3]
        else:
            p4.view(np.uint32)[:, 3] = 0xFFFFFFFF
        vs = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (vx, vy, vz)]
        rc = self._L.orc_set_particles(self._h, n, _ptr(p4), *[_ptr(v) for v in vs])
        if rc != 0:
            raise ValueError("too many particles")

    def get_particles(self):
        n = self.num_particles
        out = [np.zeros((n, 4), np.float32) for _ in range(4)]
        self._L.orc_get_particles(self._h, *[_ptr(o) for o in out])
        return out  # pos_ll, vx, vy, vz

    def read_volume(self, name):
        which = VOLUMES[name]
        shape = self.shape + ((4,) if which == 11 else ())
        out = np.zeros(shape, _vol_dtype(which))
        if self._L.orc_read_volume(self._h, which, _ptr(out)) != 0:
            raise ValueError("volume %s unavailable" % name)
        return out

    def write_volume(self, name, arr):
        which = VOLUMES[name]
        if arr is None:
            self._L.orc_write_volume(self._h, which, None)
            return
        a = np.ascontiguousarray(arr, _vol_dtype(which))
        assert a.size == self.nx * self.ny * self.nz * (4 if which == 11 else 1)
        if self._L.orc_write_volume(self._h, which, _ptr(a)) != 0:
            raise ValueError("volume %s unavailable" % name)

    def voxelize(self, positions, indices, mesh_descs):
        """scene/voxelization.rs:116-157 on the oracle's solid volume.  mesh_descs: (M, 20) float32-viewable array or a list of
        20-word records laid out like include/blubhip.h blub_mesh_desc (12 transform floats, 3 velocity, 3 axis, 2 u32)."""
        pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        d = np.ascontiguousarray(mesh_descs).view(np.uint32).reshape(-1, 20)
        self._L.orc_voxelize(self._h, _ptr(pos), _ptr(idx), d.shape[0], _ptr(d))

    def run_stage(self, name, dt):
        self._L.orc_run_stage(self._h, STAGES[name], float(dt))

    def step(self, dt):
        self._L.orc_step(self._h, float(dt))

    def solver_stats(self, which):
        e = C.c_float()
        i = C.c_int()
        self._L.orc_get_solver_stats(self._h, int(which), C.byref(e), C.byref(i))
        return e.value, i.value

    def solver_totals(self):
        it = C.c_uint64()
        s = C.c_double()
        self._L.orc_get_solver_totals(self._h, C.byref(it), C.byref(s))
        return it.value, s.value


def f16_round(v):
    return float(_load().orc_f16_round(float(v)))


def pack_mesh_desc(voxel_transform, velocity=(0, 0, 0), rotation_axis_scaled=(0, 0, 0), index_begin=0, index_end=0):
    """One blub_mesh_desc record (include/blubhip.h) as 20 uint32 words."""
    f = np.concatenate([np.asarray(voxel_transform, np.float32).reshape(12), np.asarray(velocity, np.float32), np.asarray(rotation_axis_scaled, np.float32)])
    return np.concatenate([f.view(np.uint32), np.asarray([index_begin, index_end], np.uint32)])


def rng_from_seed(seed32: bytes, n: int):
    L = _load()
    out = np.zeros(n, np.uint64)
    buf = (C.c_uint8 * 32).from_buffer_copy(seed32)
    L.orc_rng_from_seed(buf, _ptr(out), n)
    return out


def rng_seed_from_u64(seed: int, n: int):
    L = _load()
    state = np.zeros(4, np.uint64)
    out = np.zeros(n, np.float32)
    L.orc_rng_seed_from_u64(seed, _ptr(state), _ptr(out), n)
    return state, out
