"""ref_fluid.py -- HybridFluid::step (src/simulation/hybrid_fluid.rs:770-977) and PressureSolver::solve
(src/simulation/pressure_solver.rs:543-729) as a sequence of dispatches of the reference's OWN compute shaders, compiled by
oracle/glsl/build_ref.sh into oracle/_ref/libblubref.so.

TEST INFRASTRUCTURE.  What is restated here is only the host side: which pipeline is dispatched with which bind group,
push constants and group counts (each call cites the Rust line it follows).  Every arithmetic operation is executed by the
shader text itself.  `RefFluid` has the surface of oracle.oracle.Oracle (set_particles / run_stage / step / read_volume /
solver_stats) so that tests can drive both side by side.

Used (a) in this container, where /root/reference exists, to generate tests/golden/ref_*.npz (tests/golden/make_ref_golden.py)
and to hold the oracle to the reference live (tests/test_oracle_vs_ref.py), (b) nowhere else.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.normpath(os.path.join(_HERE, "..", "_ref"))
_LIB = os.path.join(_REF_DIR, "libblubref.so")

FMT_R32F, FMT_R8_SNORM, FMT_R32UI, FMT_RGBA32F = 0, 1, 2, 3
STAGES = ("transfer", "divergence", "solve_velocity", "binning", "project", "advect", "density_gather", "solve_density",
          "position_change", "correct")

REDUCE_READS_PER_THREAD = 16          # pressure_solver.rs:225
COMPUTE_LOCAL_SIZE_REDUCE = 1024      # :224
REDUCE_REDUCTION_PER_STEP = REDUCE_READS_PER_THREAD * COMPUTE_LOCAL_SIZE_REDUCE   # :226
RESULTMODE_REDUCE, RESULTMODE_INIT, RESULTMODE_ALPHA, RESULTMODE_BETA, RESULTMODE_MAX_ERROR = 0, 1, 2, 3, 4   # :213-217


def available(reference_root="/root/reference"):
    return os.path.exists(_LIB) or os.path.isdir(os.path.join(reference_root, "shader", "simulation"))


def build(reference_root="/root/reference", force=False):
    """Compiles oracle/_ref/libblubref.so from the reference's shader sources where they lie.  Returns the path or None
    when neither the library nor the reference is present (e.g. on the GPU box without a prebuilt copy)."""
    if os.path.exists(_LIB) and not force:
        srcs = [os.path.join(_HERE, f) for f in ("glsl_shim.h", "glsl2cpp.py", "ref_runtime.cpp", "build_ref.sh")]
        if all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
            return _LIB
    if not os.path.isdir(os.path.join(reference_root, "shader", "simulation")):
        return _LIB if os.path.exists(_LIB) else None
    subprocess.check_call([os.path.join(_HERE, "build_ref.sh"), reference_root], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libblubref.so is absent and /root/reference is not available to build it")
        L = C.CDLL(path)
        L.ref_bind_volume.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_bind_buffer.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_ulonglong]
        L.ref_set.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.ref_dispatch.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint]
        L.ref_set_mode.argtypes = [C.c_int, C.c_int]
        L.ref_set_mode.restype = None
        L.ref_shader_name.restype = C.c_char_p
        L.ref_shader_name.argtypes = [C.c_int]
        _lib = L
    return _lib


def shader_names():
    L = _load()
    return [L.ref_shader_name(i).decode() for i in range(L.ref_num_shaders())]


def _groups1d(n, local):   # wgpu_utils/mod.rs:18-20
    return (n + local - 1) // local


class RefFluid:
    def __init__(self, nx, ny, nz, max_num_particles):
        self.L = _load()
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.N = self.nx * self.ny * self.nz
        assert self.N > REDUCE_REDUCTION_PER_STEP, "pressure_solver.rs:551 asserts N > 16384"
        self.max_num_particles = int(max_num_particles)
        shp = (self.nz, self.ny, self.nx)
        f32 = lambda: np.zeros(shp, np.float32)
        # hybrid_fluid.rs:105-154, pressure_solver.rs:104-108, 332-371 (wgpu zero-initialises)
        self.marker = np.zeros(shp, np.int8)
        self.ll = np.zeros(shp, np.uint32)
        self.vel = [f32(), f32(), f32()]
        self.pressure = [f32(), f32()]
        self.residual, self.search, self.aux, self.aux_temp = f32(), f32(), f32(), f32()
        self.solid = np.zeros(shp + (4,), np.float32)
        self.reduce = [np.zeros(self.N, np.float32), np.zeros(max(1, self.N // REDUCE_REDUCTION_PER_STEP), np.float32)]
        self.ctrl = np.zeros(16, np.float32)       # dotproduct_reduce_result_and_dispatch_buffer, :366-371
        self.pos = np.zeros((self.max_num_particles, 4), np.float32)
        self.pos_tmp = np.zeros((self.max_num_particles, 4), np.float32)
        self.pvel = [np.zeros((self.max_num_particles, 4), np.float32) for _ in range(3)]
        self.bin_counter = np.zeros(1, np.uint32)
        self.num_particles = 0
        self.gravity = np.zeros(3, np.float32)
        self.cfg = [dict(error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4) for _ in range(2)]   # hybrid_fluid.rs:253-257
        self.pressure_cleared = [False, False]
        self.last_stats = [(0.0, 0), (0.0, 0)]
        self.rebin_freq = 60
        self.step_counter = 0
        self.binning_enabled = True
        self.push = np.zeros(2, np.uint32)          # push-constant bytes persist between dispatches (Vulkan)
        self.dispatch_log = None                    # set to a list to record (shader, groups)
        self._precond, self._filter = "zero", "weighted"
        self.set_modes()

    # ---- configuration --------------------------------------------------------------------------------
    def set_modes(self, precond=None, filter=None):
        """precond: 'zero' -> a texelFetch at LOD 1 of a one-level texture returns 0 (SURVEY Q1 reading A); 'lod0' -> clamps to level 0.
        filter: how SamplerTrilinearClamp evaluates the trilinear interpolant -- 'weighted' (Vulkan's formula, f32 weights),
        'weighted8' (weights quantised to 8 fractional bits, as real samplers do), 'separable' (lerp in x, then y, then z)."""
        if precond is not None:
            self._precond = precond
        if filter is not None:
            self._filter = filter
        self.L.ref_set_mode({"zero": 0, "lod0": 1}[self._precond], {"weighted": 0, "weighted8": 1, "separable": 2}[self._filter])

    @property
    def shape(self):
        return (self.nz, self.ny, self.nx)

    def set_gravity_grid(self, g):
        self.gravity = np.asarray(g, np.float32).copy()

    def set_solver_config(self, which, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4):
        self.cfg[which] = dict(error_tolerance=float(error_tolerance), max_num_iterations=int(max_num_iterations),
                               error_check_frequency=int(error_check_frequency))

    def set_rebinning_frequency(self, f):
        self.rebin_freq = int(f)

    def reset_pressure_cleared(self, which, cleared):
        self.pressure_cleared[which] = bool(cleared)

    def set_particles(self, pos, vx=None, vy=None, vz=None):
        pos = np.asarray(pos, np.float32)
        n = pos.shape[0]
        assert n <= self.max_num_particles
        self.pos[:] = 0
        self.pos[:n, :3] = pos[:, :3]
        self.pos.view(np.uint32)[:n, 3] = 0xFFFFFFFF
        for c, v in enumerate((vx, vy, vz)):
            self.pvel[c][:] = 0
            if v is not None:
                self.pvel[c][:n] = np.asarray(v, np.float32)
        self.num_particles = n

    def get_particles(self):
        n = self.num_particles
        return [self.pos[:n].copy()] + [v[:n].copy() for v in self.pvel]

    _VOLS = {"marker": "marker", "linked_list": "ll", "residual": "residual", "search": "search", "aux": "aux",
             "aux_temp": "aux_temp", "solid": "solid"}

    def _vol(self, name):
        if name in ("vel_x", "vel_y", "vel_z"):
            return self.vel["xyz".index(name[-1])]
        if name == "pressure_velocity":
            return self.pressure[0]
        if name == "pressure_density":
            return self.pressure[1]
        return getattr(self, self._VOLS[name])

    def read_volume(self, name):
        if name == "marker":   # R8Snorm texels -> the oracle's {-1, 0, +1} (hybrid_fluid.glsl:20-22)
            return np.sign(self.marker).astype(np.int8)
        return self._vol(name).copy()

    def write_volume(self, name, arr):
        v = self._vol(name)
        a = np.asarray(arr).reshape(v.shape)
        v[...] = (a.astype(np.int16) * 127).astype(np.int8) if name == "marker" else a

    def solver_stats(self, which):
        return self.last_stats[which]

    # ---- plumbing -------------------------------------------------------------------------------------
    def _vb(self, shader, name, arr, fmt):
        rc = self.L.ref_bind_volume(shader.encode(), name.encode(), arr.ctypes.data, fmt, self.nx, self.ny, self.nz)
        assert rc == 0, (shader, name)

    def _bb(self, shader, name, arr, count=None):
        rc = self.L.ref_bind_buffer(shader.encode(), name.encode(), arr.ctypes.data, int(arr.shape[0] if count is None else count))
        assert rc == 0, (shader, name)

    def _set(self, shader, name, value):
        a = np.ascontiguousarray(value)
        rc = self.L.ref_set(shader.encode(), name.encode(), a.ctypes.data, a.nbytes)
        assert rc == 0, (shader, name)

    def _dispatch(self, shader, gx, gy=1, gz=1):
        if self.dispatch_log is not None:
            self.dispatch_log.append((shader, int(gx), int(gy), int(gz)))
        if gx == 0 or gy == 0 or gz == 0:
            return
        rc = self.L.ref_dispatch(shader.encode(), int(gx), int(gy), int(gz))
        assert rc == 0, shader

    def _global_bindings(self, shader, dt):
        """bind group 0 (global_bindings.glsl: PerFrameConstants, samplers) -- only what the simulation shaders read."""
        time = np.array([0.0, 0.0, 0.0, dt], np.float32)   # TimerData.SimulationDelta = Duration::as_secs_f32 (simulation_controller.rs)
        self._set(shader, "Time", time)
        rend = np.zeros(12, np.float32)                    # GlobalRenderingSettings: FluidGridResolution at float offset 8
        rend.view(np.uint32)[8:11] = (self.nx, self.ny, self.nz)
        self._set(shader, "Rendering", rend)
        self._set(shader, "SamplerTrilinearClamp", np.array([1], np.int32))
        self._set(shader, "SamplerPointClamp", np.array([0], np.int32))

    def _general(self, shader):
        """bind group 1 (hybrid_fluid.rs:263-272): SimulationProperties + SceneVoxelization."""
        self._set(shader, "GravityGridSpace", self.gravity)
        self._set(shader, "NumParticles", np.array([self.num_particles], np.uint32))
        self._vb(shader, "SceneVoxelization", self.solid, FMT_RGBA32F)

    def _grid_groups(self):      # hybrid_fluid.rs:786 (COMPUTE_LOCAL_SIZE_FLUID 8x8x8, :735-739)
        return ((self.nx + 7) // 8, (self.ny + 7) // 8, (self.nz + 7) // 8)

    def _particle_groups(self):  # :787
        return _groups1d(self.num_particles, 64)

    def _transfer_bindings(self, shader, comp, dt):   # bind_group_transfer_velocity[i], :274-296
        self._global_bindings(shader, dt)
        self._general(shader)
        self._bb(shader, "Particles", self.pos)
        self._bb(shader, "ParticleBufferVelocityComponent", self.pvel[comp])
        self._vb(shader, "LinkedListDualGrid", self.ll, FMT_R32UI)
        self._vb(shader, "MarkerVolume", self.marker, FMT_R8_SNORM)
        self._vb(shader, "VelocityComponentVolume", self.vel[comp], FMT_R32F)
        self._set(shader, "VelocityTransferComponent", np.array([comp], np.uint32))

    def _write_volume_bindings(self, shader, pressure, dt):   # group_layout_write_velocity_volume, :304-317
        self._global_bindings(shader, dt)
        self._general(shader)
        self._vb(shader, "MarkerVolume", self.marker, FMT_R8_SNORM)
        for c, n in enumerate(("VelocityVolumeX", "VelocityVolumeY", "VelocityVolumeZ")):
            self._vb(shader, n, self.vel[c], FMT_R32F)
        self._vb(shader, "PressureVolume", pressure, FMT_R32F)

    # ---- stages (hybrid_fluid.rs:770-977) ---------------------------------------------------------------
    def transfer_clear(self, comp, dt):            # :810-814 / :916-921
        self._transfer_bindings("transfer_clear", comp, dt)
        self._dispatch("transfer_clear", *self._grid_groups())

    def set_boundary_marker(self, dt):             # :821-826 / :928-932
        self._transfer_bindings("transfer_set_boundary_marker", 0, dt)
        self._dispatch("transfer_set_boundary_marker", *self._grid_groups())

    def stage_transfer(self, dt):                  # :806-833
        for i in range(3):
            self.transfer_clear(i, dt)
            self._transfer_bindings("transfer_build_linkedlist", i, dt)
            self._dispatch("transfer_build_linkedlist", self._particle_groups())
            if i == 0:
                self.set_boundary_marker(dt)
            self._transfer_bindings("transfer_gather_velocity", i, dt)
            self._dispatch("transfer_gather_velocity", *self._grid_groups())

    def stage_divergence(self, dt):                # :836-840, bind group :297-303
        s = "divergence_compute"
        self._general(s)
        self._vb(s, "MarkerVolume", self.marker, FMT_R8_SNORM)
        for c, n in enumerate(("VelocityVolumeX", "VelocityVolumeY", "VelocityVolumeZ")):
            self._vb(s, n, self.vel[c], FMT_R32F)
        self._vb(s, "Divergence", self.residual, FMT_R32F)
        self._dispatch(s, *self._grid_groups())

    def stage_binning(self, dt):                   # :857-892
        self.bin_counter[...] = 0                  # encoder.clear_buffer(particle_binning_atomic_counter) at the top of step(), :793
        self.ll[...] = 0                           # clear_texture(volume_linked_lists), :859
        gb = self._global_bindings
        for s in ("particle_binning_count", "particle_binning_prefixsum", "particle_binning_rewrite_particles"):
            if s != "particle_binning_rewrite_particles":
                gb(s, dt)
            self._bb(s, "Old_Particles", self.pos)
            self._bb(s, "New_Particles", self.pos_tmp)
            self._vb(s, "ParticleBinningVolume", self.ll, FMT_R32UI)
            self._bb(s, "ParticleBinningAtomicCounter_", self.bin_counter)
        self._dispatch("particle_binning_count", self._particle_groups())
        self._dispatch("particle_binning_prefixsum", _groups1d(self.N, 1024))         # scan_work_groups, :788-791
        self._dispatch("particle_binning_rewrite_particles", self._particle_groups())
        self.pos[...] = self.pos_tmp               # copy_buffer_to_buffer of the whole buffer, :883-890

    def stage_project(self, dt):                   # :904-914
        self._write_volume_bindings("divergence_remove", self.pressure[0], dt)
        self._dispatch("divergence_remove", *self._grid_groups())
        self._write_volume_bindings("extrapolate_velocity", self.pressure[0], dt)
        self._dispatch("extrapolate_velocity", *self._grid_groups())

    def stage_advect(self, dt):                    # :915-932
        self.transfer_clear(0, dt)
        s = "advect_particles"                     # bind group :319-329
        self._global_bindings(s, dt)
        self._general(s)
        for c, n in enumerate(("VelocityVolumeX", "VelocityVolumeY", "VelocityVolumeZ")):
            self._vb(s, n, self.vel[c], FMT_R32F)
        self._vb(s, "MarkerVolume", self.marker, FMT_R8_SNORM)
        self._vb(s, "LinkedListDualGrid", self.ll, FMT_R32UI)
        self._bb(s, "Particles", self.pos)
        for c, n in enumerate(("ParticleBufferVelocityX", "ParticleBufferVelocityY", "ParticleBufferVelocityZ")):
            self._bb(s, n, self.pvel[c])
        self._dispatch(s, self._particle_groups())
        self.set_boundary_marker(dt)

    def stage_density_gather(self, dt):            # :933-937, bind group :338-343 (binding 4 is never bound: SURVEY Q9)
        s = "density_projection_gather_error"
        self._global_bindings(s, dt)
        self._general(s)
        self._bb(s, "Particles", self.pos)
        self._vb(s, "LinkedListDualGrid", self.ll, FMT_R32UI)
        self._vb(s, "MarkerVolume", self.marker, FMT_R8_SNORM)
        self._vb(s, "DensityVolume", self.residual, FMT_R32F)
        self._dispatch(s, *self._grid_groups())

    def stage_position_change(self, dt):           # :958-967
        self._write_volume_bindings("density_projection_position_change", self.pressure[1], dt)
        self._dispatch("density_projection_position_change", *self._grid_groups())
        self._write_volume_bindings("extrapolate_velocity", self.pressure[1], dt)
        self._dispatch("extrapolate_velocity", *self._grid_groups())

    def stage_correct(self, dt):                   # :969-973, bind group :344-350
        s = "density_projection_correct_particles"
        self._global_bindings(s, dt)
        self._general(s)
        self._bb(s, "Particles", self.pos)
        self._vb(s, "MarkerVolume", self.marker, FMT_R8_SNORM)
        for c, n in enumerate(("VelocityVolumeX", "VelocityVolumeY", "VelocityVolumeZ")):
            self._vb(s, n, self.vel[c], FMT_R32F)
        self._dispatch(s, self._particle_groups())

    # ---- PressureSolver (pressure_solver.rs) ----------------------------------------------------------------
    def _ctrl_u32(self):
        return self.ctrl.view(np.uint32)

    def _push(self, shader, *words):
        """set_push_constants(0, words): only the bytes given are overwritten."""
        for k, w in enumerate(words):
            self.push[k] = w
        self._set(shader, "PushConstants.Mode", self.push[0:1])
        self._set(shader, "PushConstants.SourceBufferSize", self.push[1:2])

    def _pressure_common(self, shader, which, dt):
        """bind group 0 = marker (:375-377), bind group 1 = pressure volume + Config (:112-115, 193-200)."""
        self._vb(shader, "MarkerVolume", self.marker, FMT_R8_SNORM)
        self._vb(shader, "Pressure", self.pressure[which], FMT_R32F)
        c = self.cfg[which]
        self._set(shader, "ErrorTolerance", np.array([np.float32(c["error_tolerance"]) / np.float32(dt)], np.float32))   # :197
        self._set(shader, "MaxNumSolverIterations", np.array([c["max_num_iterations"]], np.uint32))

    def _dispatch_indirect(self, shader, byte_offset):
        w = self._ctrl_u32()[byte_offset // 4: byte_offset // 4 + 3]
        self._dispatch(shader, int(w[0]), int(w[1]), int(w[2]))

    def _reduce(self, which, dt, result_mode, shader):   # :543-589
        num_entries_remaining = self.N
        assert num_entries_remaining > REDUCE_REDUCTION_PER_STEP
        source = 0
        DISPATCH_BUFFER_OFFSETS = (32, 32)         # :555 (both levels point at DispatchCommandReduce0: SURVEY Q5)
        self._pressure_common(shader, which, dt)
        step = 0
        while num_entries_remaining > REDUCE_REDUCTION_PER_STEP:
            # bind_group_dotproduct_reduce[source]: source buffer -> the other buffer (:417-426)
            self._bb(shader, "DotProductSource", self.reduce[source])
            self._bb(shader, "DotProductDest", self.reduce[1 - source])
            self._push(shader, RESULTMODE_REDUCE, num_entries_remaining)
            if step < len(DISPATCH_BUFFER_OFFSETS):
                self._dispatch_indirect(shader, DISPATCH_BUFFER_OFFSETS[step])
            else:
                self._dispatch(shader, _groups1d(num_entries_remaining // REDUCE_READS_PER_THREAD, COMPUTE_LOCAL_SIZE_REDUCE))
            source = 1 - source
            num_entries_remaining //= REDUCE_REDUCTION_PER_STEP
            step += 1
        # final (:582-588): bind_group_dotproduct_final[source] (:427-436): source buffer -> the control buffer
        self._bb(shader, "DotProductSource", self.reduce[source])
        self._bb(shader, "DotProductDest", self.ctrl)
        self._push(shader, result_mode, num_entries_remaining)
        self._dispatch(shader, 1)

    def _preconditioner(self, which, dt, pass_mode, reduce_size, out_vol, in_vol, indirect):
        s = "pressure_apply_preconditioner"        # bind groups :386-405
        self._pressure_common(s, which, dt)
        self._bb(s, "ReduceBuffer", self.reduce[0])
        self._vb(s, "Residual", self.residual, FMT_R32F)
        self._vb(s, "AuxiliaryOrTemp", out_vol, FMT_R32F)
        self._vb(s, "ResidualOrTemp", in_vol, FMT_R32F)
        if reduce_size is None:
            self._push(s, pass_mode)
        else:
            self._push(s, pass_mode, reduce_size)
        if indirect:
            self._dispatch_indirect(s, 16)
        else:
            self._dispatch(s, (self.nx + 7) // 8, (self.ny + 7) // 8, self.nz)

    def solve(self, which, dt):                    # :591-729
        if not self.pressure_cleared[which]:       # :601-603
            self.pressure[which][...] = 0
            self.pressure_cleared[which] = True
        c = self.cfg[which]
        max_iter, freq = c["max_num_iterations"], c["error_check_frequency"]
        reduce_pass_initial_group_size = _groups1d(self.N // REDUCE_READS_PER_THREAD, COMPUTE_LOCAL_SIZE_REDUCE)   # :614-618
        # init (:625-648)
        s = "pressure_init"
        self._pressure_common(s, which, dt)
        self._vb(s, "Residual", self.residual, FMT_R32F)
        self._bb(s, "ReduceResultAndMainDispatchBuffer", self.ctrl)
        self._dispatch(s, (self.nx + 7) // 8, (self.ny + 7) // 8, self.nz)   # COMPUTE_LOCAL_SIZE_VOLUME 8x8x1, :219-223, 626
        self._push("pressure_apply_preconditioner", 0)                                   # :640
        self._preconditioner(which, dt, 0, None, self.aux_temp, self.residual, False)    # bind_group_preconditioner[0]
        self._preconditioner(which, dt, 1, reduce_pass_initial_group_size, self.search, self.aux_temp, False)   # [2]: to search
        self._reduce(which, dt, RESULTMODE_INIT, "pressure_reduce_sum")
        i = 0
        while True:                                # :654-723
            s = "pressure_apply_coeff"             # :661-666, bind group :381-385
            self._pressure_common(s, which, dt)
            self._bb(s, "ReduceBuffer", self.reduce[0])
            self._vb(s, "Search", self.search, FMT_R32F)
            self._push(s, 0, reduce_pass_initial_group_size)
            self._dispatch_indirect(s, 16)
            self._reduce(which, dt, RESULTMODE_ALPHA, "pressure_reduce_sum")
            check = (max_iter == i) or (i > 0 and i % freq == 0)     # :672-673
            s = "pressure_update_pressure_and_residual"             # :675-685, bind group :438-443
            self._pressure_common(s, which, dt)
            self._bb(s, "ReduceBuffer", self.reduce[0])
            self._vb(s, "Residual", self.residual, FMT_R32F)
            self._vb(s, "Search", self.search, FMT_R32F)
            self._set(s, "Scalars", self.ctrl[0:4])                  # the control buffer bound as the PcgScalars uniform
            if check:
                self._push(s, 1, reduce_pass_initial_group_size)
            else:
                self._push(s, 0)
            self._dispatch_indirect(s, 16)
            if check:
                self._reduce(which, dt, RESULTMODE_MAX_ERROR + i, "pressure_reduce_max")   # :691-693
                if max_iter == i:
                    break
            self._preconditioner(which, dt, 0, None, self.aux_temp, self.residual, True)                            # [0]
            self._preconditioner(which, dt, 1, reduce_pass_initial_group_size, self.aux, self.aux_temp, True)       # [1]
            self._reduce(which, dt, RESULTMODE_BETA, "pressure_reduce_sum")
            s = "pressure_update_search"           # :714-718, bind group :444-449
            self._pressure_common(s, which, dt)
            self._vb(s, "Search", self.search, FMT_R32F)
            self._vb(s, "Auxillary", self.aux, FMT_R32F)
            self._set(s, "Scalars", self.ctrl[0:4])
            self._dispatch_indirect(s, 16)
            i += 1
        # enqueue_error_buffer_read copies bytes 8..16 (:176); retrieve_new_error_samples: error = max_error * dt (:162)
        self.last_stats[which] = (float(np.float32(self.ctrl[2]) * np.float32(dt)), int(self.ctrl[3]))

    # ---- driver -------------------------------------------------------------------------------------------
    def run_stage(self, name, dt):
        dt = float(np.float32(dt))
        if name == "solve_velocity":
            self.solve(0, dt)
        elif name == "solve_density":
            self.solve(1, dt)
        else:
            getattr(self, "stage_" + name)(dt)

    def step(self, dt):
        self.bin_counter[...] = 0                  # encoder.clear_buffer(particle_binning_atomic_counter), :793
        for name in ("transfer", "divergence", "solve_velocity"):
            self.run_stage(name, dt)
        if self.binning_enabled and self.rebin_freq != 0 and self.step_counter % self.rebin_freq == 0:   # :854-856
            self.run_stage("binning", dt)
        for name in ("project", "advect", "density_gather", "solve_density", "position_change", "correct"):
            self.run_stage(name, dt)
        self.step_counter += 1
