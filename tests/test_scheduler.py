"""The native step scheduler (blub_controller_*, include/blubhip.h) against the reference's semantics
(src/simulation_controller.rs, src/timer.rs), driven through the C-ABI with a fake scene -- no GPU needed."""
import numpy as np
import pytest

from blub_amd.simulation_controller import FAST_FORWARD_BATCH, NS, SimulationController

DELTA = NS // 120     # 8 333 333 ns


class FakeFluid:
    def __init__(self):
        self.syncs = 0

    def synchronize(self):
        self.syncs += 1


class FakeScene:
    def __init__(self):
        self.steps, self._f = [], FakeFluid()

    def step(self, dt):
        self.steps.append(dt)

    def fluid(self):
        return self._f


def test_default_delta_is_the_reference_duration():
    c = SimulationController()
    assert c.simulation_steps_per_second == 120 and c.simulation_delta_ns == 8333333      # Duration::from_nanos(1e9 / 120), :33-39
    s = FakeScene()
    c.on_frame_submitted(DELTA + 1000)      # (frame deltas pass through Duration::mul_f32, i.e. f32 seconds: +-1 ns, hence the margin)
    assert c.frame_steps(s) == 1
    assert s.steps == [float(np.float32(8333333) / np.float32(1e9))]                     # Duration::as_secs_f32
    c.simulation_steps_per_second = 60
    assert c.simulation_delta_ns == 16666666
    assert c.status == SimulationController.REALTIME and c.simulation_stop_time_ns == 3600 * NS


def test_frame_steps_follow_the_render_clock():
    c, s = SimulationController(), FakeScene()
    c.on_frame_submitted(NS // 60 + 1000)            # a 60 Hz frame holds two 120 Hz steps
    assert c.frame_steps(s) == 2 and c.total_simulated_time_ns == 2 * DELTA
    c.on_frame_submitted(NS // 240)                  # residual ~4.17 ms < delta
    assert c.frame_steps(s) == 0
    c.on_frame_submitted(NS // 240 - 2000)           # ~8.332 ms: still short of one step
    assert c.frame_steps(s) == 0
    c.on_frame_submitted(4000)
    assert c.frame_steps(s) == 1
    assert c.num_simulation_steps_performed == 3 and len(s.steps) == 3


def test_realtime_gives_up_after_a_fiftieth_of_a_second_of_steps():
    """timer.rs:110-118: once num_steps_this_frame * delta > 1/50 s the rest of the backlog is (90 %) accepted as lag."""
    c, s = SimulationController(), FakeScene()
    c.on_frame_submitted(NS // 2)                    # half a second behind
    n = c.frame_steps(s)
    assert n == 3                                    # 0, 1, 2 steps * delta <= 20 ms; with 3 steps done 25 ms > 20 ms stops the loop
    backlog = NS // 2 - 3 * DELTA
    c.on_frame_submitted(0)
    again = c.frame_steps(s)                         # 10 % of the backlog is still owed: 47.5 ms -> again capped at 3 steps
    assert again == 3
    assert c.total_simulated_time_ns == 6 * DELTA and backlog > 0


def test_time_scale_and_pause():
    c, s = SimulationController(), FakeScene()
    c.set_time_scale(0.5)
    c.on_frame_submitted(4 * DELTA + 4000)
    assert c.frame_steps(s) == 2                     # Duration::mul_f32(0.5)
    c.pause_or_resume()
    assert c.status == SimulationController.PAUSED
    c.on_frame_submitted(NS)
    assert c.frame_steps(s) == 0                     # skip_simulation_frame: the frame's time is accepted as lag ...
    c.pause_or_resume()
    c.on_frame_submitted(0)
    assert c.frame_steps(s) == 0                     # ... so nothing is owed afterwards


def test_recording_forces_the_frame_delta():
    c, s = SimulationController(), FakeScene()
    c.start_recording_with_fixed_frame_length(30.0)
    c.on_frame_submitted(123)                        # the measured duration is replaced by 1/30 s
    assert c.frame_steps(s) == 4                     # 33 333 333 ns = 4 steps (no real-time cap while recording, :197-201)


def test_fast_forward_batches_of_16_and_stop_time_mechanism():
    c, s = SimulationController(), FakeScene()
    n = c.fast_forward_steps(s, NS // 2)             # 0.5 s = 60 steps
    assert n == 60 and len(s.steps) == 60
    assert s.fluid().syncs == (60 + FAST_FORWARD_BATCH - 1) // FAST_FORWARD_BATCH
    assert c.total_simulated_time_ns == 60 * DELTA
    assert c.computation_time_last_fast_forward_ns > 0
    assert c.status == SimulationController.PAUSED              # the jump ends through the stop-time mechanism (:204-207)
    assert c.simulation_stop_time_ns == 3600 * NS               # restored (:150)
    c.pause_or_resume()
    c.on_frame_submitted(0)
    assert c.frame_steps(s) == 0                                # force_frame_delta(0): nothing is owed after the jump


def test_fast_forward_jumps_at_least_one_step():
    c, s = SimulationController(), FakeScene()
    assert c.fast_forward_steps(s, 1000) == 1                   # simulation_jump_length.max(simulation_delta), :119-121
    assert c.total_simulated_time_ns == DELTA


def test_fast_forward_respects_an_exact_multiple():
    c, s = SimulationController(), FakeScene()
    assert c.fast_forward_steps(s, 32 * DELTA) == 32 and s.fluid().syncs == 3     # 16 + 16 + the empty batch that detects the stop


def test_step_errors_end_the_jump_and_are_reported():
    class Broken(FakeScene):
        def step(self, dt):
            if len(self.steps) == 5:
                raise RuntimeError("boom")
            super().step(dt)
    c, s = SimulationController(), Broken()
    with pytest.raises(RuntimeError):
        c.fast_forward_steps(s, NS)
    assert len(s.steps) == 5 and c.status == SimulationController.PAUSED


def _rust_mul_f32(ns, scale):
    """Duration::mul_f32 = from_secs_f32(rhs * self.as_secs_f32()): f32 arithmetic, then the exact value of the f32 rounded to the nearest
    nanosecond, ties to even (core::time::Duration::try_from_secs_f32)."""
    from fractions import Fraction
    secs = np.float32(ns // NS) + np.float32(ns % NS) / np.float32(1e9)
    v = np.float32(scale) * secs
    exact = Fraction(float(v)) * NS
    q, r = divmod(exact.numerator, exact.denominator)
    twice = 2 * r
    if twice > exact.denominator or (twice == exact.denominator and q % 2 == 1):
        q += 1
    return q


def test_frame_deltas_round_like_duration_from_secs_f32():
    """Round-2 ADVICE: the frame delta (and the accepted lag) used to be TRUNCATED to nanoseconds where the reference rounds to nearest; one
    nanosecond off for about half of all inputs."""
    rng = np.random.default_rng(12)
    durations = [1, 999, 4166666, 8333333, 16666667, NS // 3, NS - 1, NS + 1, 7 * NS + 123456789] + [int(v) for v in rng.integers(1, 3 * NS, 40)]
    off_by_truncation = 0
    for scale in (1.0, 0.5, 0.9, 1.7):
        for d in durations:
            c = SimulationController()
            c.set_time_scale(scale)
            c.on_frame_submitted(d)
            want = _rust_mul_f32(d, scale)
            assert c.total_render_time_ns == want, (d, scale, c.total_render_time_ns, want)
            secs = np.float32(d // NS) + np.float32(d % NS) / np.float32(1e9)
            off_by_truncation += int(int(float(np.float32(scale) * secs) * 1e9) != want)
            c.close()
    assert off_by_truncation > 20       # the inputs do distinguish rounding from truncation


def test_a_failing_step_is_not_counted():
    """A step callback that fails ends the frame; the step it stood for did not happen and must not stay on the clocks."""
    class Failing(FakeScene):
        def step(self, dt):
            if len(self.steps) == 1:
                raise RuntimeError("device lost")
            super().step(dt)
    c, s = SimulationController(), Failing()
    c.on_frame_submitted(NS // 30 + 1000)            # four 120 Hz steps wanted
    with pytest.raises(RuntimeError):
        c.frame_steps(s)
    assert len(s.steps) == 1 and c.num_simulation_steps_performed == 1 and c.total_simulated_time_ns == DELTA
