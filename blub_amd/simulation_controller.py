"""Fixed-step scheduler with the semantics of the reference's `SimulationController` (src/simulation_controller.rs) and
`Timer::simulation_frame_loop` (src/timer.rs:94-130), reduced to what drives `HybridFluid::step`.

* default 120 simulation steps per second => dt = Duration::from_nanos(1e9 / 120).as_secs_f32() (:33-39)
* `frame_steps`: step while the simulated time lags the render time, but give up on real time once the steps of one
  frame took longer than MAX_STEP_COMPUTATION_PER_FRAME = 1/50 s (:31, 159-211)
* `fast_forward_steps`: batches of 16 steps followed by a wait for the GPU (TDR avoidance, :105-112, 131-140); the wall
  clock of the whole fast-forward is kept as `computation_time_last_fast_forward` (:147) -- the only place the reference
  measures time per step, which is why bench.py's steps/s is defined the same way (enqueue K steps, wait, divide).
"""
import time

from .hybrid_fluid import default_simulation_delta

MAX_STEP_COMPUTATION_PER_FRAME = 1.0 / 50.0      # simulation_controller.rs:31
FAST_FORWARD_BATCH = 16                          # :105-112
MIN_STEPS_PER_SECOND, MAX_STEPS_PER_SECOND = 20, 1200   # gui/mod.rs:288-292


class SimulationController:
    REALTIME, RECORD, FAST_FORWARD, PAUSED = "Realtime", "Record", "FastForward", "Paused"

    def __init__(self, steps_per_second=120):
        self.status = self.REALTIME
        self.simulation_stop_time = 60.0 * 60.0           # an hour (:41)
        self.time_scale = 1.0
        self.simulation_steps_per_second = steps_per_second
        self.total_simulated_time = 0.0
        self.total_render_time = 0.0
        self.num_simulation_steps_performed = 0
        self.computation_time_last_fast_forward = 0.0

    @property
    def simulation_steps_per_second(self):
        return self._sps

    @simulation_steps_per_second.setter
    def simulation_steps_per_second(self, v):
        self._sps = int(min(max(v, MIN_STEPS_PER_SECOND), MAX_STEPS_PER_SECOND))
        self.simulation_delta = default_simulation_delta(self._sps)

    def _single_step(self, scene):
        scene.step(self.simulation_delta)            # Scene::step -> HybridFluid::step + update_statistics
        self.total_simulated_time += self.simulation_delta
        self.num_simulation_steps_performed += 1

    def frame_steps(self, scene, frame_delta, clock=time.perf_counter):
        """One rendered frame (:159-211). Returns the number of simulation steps performed."""
        if self.status == self.PAUSED:
            return 0
        self.total_render_time += frame_delta * self.time_scale
        start, steps = clock(), 0
        while self.total_simulated_time + self.simulation_delta <= min(self.total_render_time, self.simulation_stop_time):
            self._single_step(scene)
            steps += 1
            if clock() - start > MAX_STEP_COMPUTATION_PER_FRAME:    # give up on real time: drop the backlog (timer.rs:110-118)
                self.total_render_time = self.total_simulated_time
                break
        return steps

    def fast_forward_steps(self, scene, duration, clock=time.perf_counter):
        """Simulate `duration` seconds as fast as possible (:96-157). Returns the number of steps."""
        previous, self.status = self.status, self.FAST_FORWARD
        start = clock()
        target = min(self.total_simulated_time + duration, self.simulation_stop_time)
        steps = 0
        while self.total_simulated_time + self.simulation_delta <= target + 1e-9:
            for _ in range(FAST_FORWARD_BATCH):
                if self.total_simulated_time + self.simulation_delta > target + 1e-9:
                    break
                self._single_step(scene)
                steps += 1
            scene.fluid().synchronize()                  # device.poll(Wait) every 16 steps (:140)
        self.computation_time_last_fast_forward = clock() - start
        self.total_render_time = self.total_simulated_time
        self.status = previous
        return steps
