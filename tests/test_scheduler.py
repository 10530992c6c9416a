"""SimulationController semantics (src/simulation_controller.rs) against a fake scene -- no GPU needed."""
import numpy as np

from blub_amd.simulation_controller import FAST_FORWARD_BATCH, SimulationController


class FakeFluid:
    def __init__(self):
        self.syncs = 0

    def synchronize(self):
        self.syncs += 1


class FakeScene:
    def __init__(self, cost=0.0):
        self.steps, self.cost, self.now, self._f = [], cost, 0.0, FakeFluid()

    def step(self, dt):
        self.steps.append(dt)
        self.now += self.cost

    def fluid(self):
        return self._f


def test_default_delta_and_clamp():
    c = SimulationController()
    assert c.simulation_delta == float(np.float32(8333333) / np.float32(1e9))      # 120 Hz (:39)
    c.simulation_steps_per_second = 5
    assert c.simulation_steps_per_second == 20
    c.simulation_steps_per_second = 5000
    assert c.simulation_steps_per_second == 1200


def test_frame_steps_follow_render_time():
    c, s = SimulationController(), FakeScene()
    n = c.frame_steps(s, 1.0 / 60.0, clock=lambda: s.now)     # a 60 Hz frame holds two 120 Hz steps
    assert n == 2 and len(s.steps) == 2
    assert c.frame_steps(s, 1.0 / 240.0, clock=lambda: s.now) == 0
    assert c.frame_steps(s, 1.0 / 240.0, clock=lambda: s.now) == 1


def test_frame_steps_give_up_on_realtime():
    c, s = SimulationController(), FakeScene(cost=0.015)      # every step "takes" 15 ms: budget 1/50 s is hit after 2 steps
    n = c.frame_steps(s, 0.5, clock=lambda: s.now)
    assert n == 2
    assert abs(c.total_render_time - c.total_simulated_time) < 1e-12     # backlog dropped
    assert c.frame_steps(s, 0.0, clock=lambda: s.now) == 0


def test_fast_forward_batches_of_16():
    c, s = SimulationController(), FakeScene(cost=0.001)
    n = c.fast_forward_steps(s, 0.5, clock=lambda: s.now)     # 0.5 s = 60 steps
    assert n == 60 and len(s.steps) == 60
    assert s.fluid().syncs == (60 + FAST_FORWARD_BATCH - 1) // FAST_FORWARD_BATCH
    assert abs(c.computation_time_last_fast_forward - 0.060) < 1e-9
    assert c.status == SimulationController.REALTIME
