"""ctypes binding of the native step scheduler in libblubhip.so (blub_controller_*, blub_amd/csrc/scheduler_host.cpp): the
reference's `SimulationController` (src/simulation_controller.rs) on top of `Timer` (src/timer.rs), integer-nanosecond `Duration`
arithmetic.  Method names follow the reference; durations are nanoseconds (ints) unless a name says seconds.

* default 120 simulation steps per second => delta = Duration::from_nanos(1e9 / 120) = 8 333 333 ns (:33-39)
* `frame_steps`: step while the simulated time lags the render clock; real-time mode gives up (accepts lag) once the steps of one
  frame cover more than MAX_STEP_COMPUTATION_PER_FRAME = 1/50 s (:31, 159-217; timer.rs:94-130)
* `fast_forward_steps`: batches of 16 steps followed by a wait for the GPU (:105-140); jumps at least one step (:119-121); the wall
  clock of the whole jump is `computation_time_last_fast_forward` (:147) -- the only place the reference measures time per step,
  which is why bench.py's steps/s is defined the same way (enqueue K steps, wait, divide).
"""
import ctypes as C

from .hybrid_fluid import BlubError, load_library

MAX_STEP_COMPUTATION_PER_FRAME = 1.0 / 50.0      # simulation_controller.rs:31
FAST_FORWARD_BATCH = 16                          # :112
NS = 1000 * 1000 * 1000

_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_float, C.c_uint64)
_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class _Callbacks(C.Structure):
    _fields_ = [("step", _STEP_FN), ("wait", _WAIT_FN), ("user", C.c_void_p)]


def _lib():
    L = load_library()
    if getattr(L, "_controller_bound", False):
        return L
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    sig = {
        "blub_controller_create": (C.c_int, [u64, C.POINTER(vp)]), "blub_controller_destroy": (None, [vp]),
        "blub_controller_set_simulation_steps_per_second": (C.c_int, [vp, u64]), "blub_controller_simulation_steps_per_second": (u64, [vp]),
        "blub_controller_simulation_delta_ns": (u64, [vp]), "blub_controller_total_simulated_time_ns": (u64, [vp]),
        "blub_controller_total_render_time_ns": (u64, [vp]), "blub_controller_num_simulation_steps_performed": (u32, [vp]),
        "blub_controller_num_simulation_steps_performed_for_current_frame": (u32, [vp]),
        "blub_controller_computation_time_last_fast_forward_ns": (u64, [vp]), "blub_controller_get_status": (C.c_int, [vp]),
        "blub_controller_set_simulation_stop_time_ns": (C.c_int, [vp, u64]), "blub_controller_simulation_stop_time_ns": (u64, [vp]),
        "blub_controller_set_time_scale": (C.c_int, [vp, C.c_float]), "blub_controller_pause_or_resume": (C.c_int, [vp]),
        "blub_controller_start_recording_with_fixed_frame_length": (C.c_int, [vp, C.c_double]), "blub_controller_restart": (C.c_int, [vp]),
        "blub_controller_on_frame_submitted": (C.c_int, [vp, C.c_int64]),
        "blub_controller_frame_steps": (C.c_int, [vp, C.POINTER(_Callbacks), C.POINTER(u32)]),
        "blub_controller_fast_forward_steps": (C.c_int, [vp, u64, C.POINTER(_Callbacks), C.POINTER(u32)]),
        "blub_controller_frame_steps_fluid": (C.c_int, [vp, vp, C.POINTER(u32)]),
        "blub_controller_fast_forward_steps_fluid": (C.c_int, [vp, u64, vp, C.POINTER(u32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L._controller_bound = True
    return L


def _check(L, rc):
    if rc != 0:
        raise BlubError(rc, L.blub_last_error_string().decode("utf-8", "replace"))


class SimulationController:
    REALTIME, RECORD, FAST_FORWARD, PAUSED = "Realtime", "RecordingWithFixedFrameLength", "FastForward", "Paused"
    _STATUS = (REALTIME, RECORD, FAST_FORWARD, PAUSED)

    def __init__(self, steps_per_second=120):
        self._L = _lib()
        self._c = C.c_void_p()
        _check(self._L, self._L.blub_controller_create(int(steps_per_second), C.byref(self._c)))

    def close(self):
        if getattr(self, "_c", None):
            self._L.blub_controller_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the reference's accessors -------------------------------------------------------------------------------
    @property
    def simulation_steps_per_second(self):
        return int(self._L.blub_controller_simulation_steps_per_second(self._c))

    @simulation_steps_per_second.setter
    def simulation_steps_per_second(self, v):
        _check(self._L, self._L.blub_controller_set_simulation_steps_per_second(self._c, int(v)))

    @property
    def simulation_delta_ns(self):
        return int(self._L.blub_controller_simulation_delta_ns(self._c))

    @property
    def total_simulated_time_ns(self):
        return int(self._L.blub_controller_total_simulated_time_ns(self._c))

    @property
    def total_render_time_ns(self):
        return int(self._L.blub_controller_total_render_time_ns(self._c))

    @property
    def num_simulation_steps_performed(self):
        return int(self._L.blub_controller_num_simulation_steps_performed(self._c))

    @property
    def computation_time_last_fast_forward_ns(self):
        return int(self._L.blub_controller_computation_time_last_fast_forward_ns(self._c))

    @property
    def computation_time_last_fast_forward(self):
        """seconds"""
        return self.computation_time_last_fast_forward_ns * 1e-9

    @property
    def status(self):
        return self._STATUS[int(self._L.blub_controller_get_status(self._c))]

    @property
    def simulation_stop_time_ns(self):
        return int(self._L.blub_controller_simulation_stop_time_ns(self._c))

    @simulation_stop_time_ns.setter
    def simulation_stop_time_ns(self, v):
        _check(self._L, self._L.blub_controller_set_simulation_stop_time_ns(self._c, int(v)))

    def set_time_scale(self, s):
        _check(self._L, self._L.blub_controller_set_time_scale(self._c, float(s)))

    def pause_or_resume(self):
        _check(self._L, self._L.blub_controller_pause_or_resume(self._c))

    def start_recording_with_fixed_frame_length(self, frames_per_second):
        _check(self._L, self._L.blub_controller_start_recording_with_fixed_frame_length(self._c, float(frames_per_second)))

    def restart(self):
        _check(self._L, self._L.blub_controller_restart(self._c))

    def on_frame_submitted(self, measured_frame_duration_ns=-1):
        """Timer::on_frame_submitted; a negative duration measures the real time since the previous call."""
        _check(self._L, self._L.blub_controller_on_frame_submitted(self._c, int(measured_frame_duration_ns)))

    # ---- stepping ---------------------------------------------------------------------------------------------------
    def _callbacks(self, scene):
        """`scene` needs step(dt_seconds) and fluid().synchronize() like blub_amd.Scene (Scene::step, scene/mod.rs:166-213)."""
        errors = []

        def step(_user, dt, _total_ns):
            try:
                scene.step(dt)
                return 0
            except Exception as e:   # noqa: BLE001 -- must not unwind through the C frame
                errors.append(e)
                return -4

        def wait(_user):
            try:
                scene.fluid().synchronize()
                return 0
            except Exception as e:   # noqa: BLE001
                errors.append(e)
                return -4
        cb = _Callbacks(_STEP_FN(step), _WAIT_FN(wait), None)
        return cb, errors

    def frame_steps(self, scene):
        """One rendered frame (:159-173): call on_frame_submitted() first, like the reference's event loop. Returns the step count."""
        cb, errors = self._callbacks(scene)
        n = C.c_uint32()
        rc = self._L.blub_controller_frame_steps(self._c, C.byref(cb), C.byref(n))
        if errors:
            raise errors[0]
        _check(self._L, rc)
        return n.value

    def fast_forward_steps(self, scene, simulation_jump_length_ns):
        """:96-157. Returns the number of steps performed; computation_time_last_fast_forward holds the wall clock."""
        cb, errors = self._callbacks(scene)
        n = C.c_uint32()
        rc = self._L.blub_controller_fast_forward_steps(self._c, int(simulation_jump_length_ns), C.byref(cb), C.byref(n))
        if errors:
            raise errors[0]
        _check(self._L, rc)
        return n.value

    def fast_forward_steps_fluid(self, fluid, simulation_jump_length_ns):
        """The same for a bare HybridFluid (no Python in the loop): blub_fluid_step + update_statistics per step, synchronize per batch."""
        n = C.c_uint32()
        _check(self._L, self._L.blub_controller_fast_forward_steps_fluid(self._c, int(simulation_jump_length_ns), fluid._h, C.byref(n)))
        return n.value

    def frame_steps_fluid(self, fluid):
        n = C.c_uint32()
        _check(self._L, self._L.blub_controller_frame_steps_fluid(self._c, fluid._h, C.byref(n)))
        return n.value
