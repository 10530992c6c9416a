// Host-only: the reference's step scheduler -- `SimulationController` (src/simulation_controller.rs) on top of `Timer`
// (src/timer.rs) -- restated with the same integer-nanosecond `Duration` arithmetic, so that a host above the C-ABI drives
// `HybridFluid::step` exactly as the reference's event loop does: fixed simulation delta (120 steps/s by default), frames that
// step while the simulation lags the render clock and give up on real time after 1/50 s of simulated steps per frame, and
// fast-forward in batches of 16 steps followed by a wait for the GPU, whose wall-clock time is kept as
// `computation_time_last_fast_forward` (the only time-per-step measurement the reference has, simulation_controller.rs:128-147).
// No device code, no HIP calls: stepping goes through callbacks (default: blub_fluid_step / blub_fluid_synchronize).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <new>

#include "blub_internal.h"

namespace {

using Clock = std::chrono::steady_clock;
constexpr uint64_t NS = 1000ull * 1000ull * 1000ull;
constexpr uint64_t MAX_STEP_COMPUTATION_PER_FRAME_NS = NS / 50;    // simulation_controller.rs:31 (Duration::from_secs_f64(1.0 / 50.0))
constexpr uint64_t NO_LIMIT = ~0ull;                              // Duration::from_secs(u64::MAX), :191
constexpr int MAX_FAST_FORWARD_SIMULATION_BATCH_SIZE = 16;         // :112

// Duration::from_secs_f32 (core::time, try_from_secs_f32): the f32 is converted EXACTLY -- v = m 2^e with a 24-bit m -- and the nanoseconds are
// rounded to nearest, ties to even (round 2 truncated: 1 ns off the reference for about half of all inputs).  Negative / NaN input and values
// beyond u64 seconds make the reference panic; they saturate here (0 / "no limit").
uint64_t duration_from_secs_f32(float v) {
    if (!(v > 0.0f)) return 0;
    if (!(v < 1.8446744e19f)) return ~0ull;
    int e = 0;
    const float fr = std::frexp(v, &e);                        // v = fr 2^e, 0.5 <= fr < 1
    const uint64_t m = (uint64_t)std::ldexp(fr, 24);            // exact: 24-bit significand
    e -= 24;                                                    // v = m 2^e
    const unsigned __int128 num = (unsigned __int128)m * (unsigned __int128)NS;   // < 2^24 x 2^30
    unsigned __int128 ns128;
    if (e >= 0) {
        if (e > 64) return ~0ull;
        ns128 = num << e;
    } else {
        const int sh = -e;
        if (sh >= 100) return 0;
        const unsigned __int128 q = num >> sh, rem = num - (q << sh), half = (unsigned __int128)1 << (sh - 1);
        ns128 = q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0);
    }
    return ns128 > (unsigned __int128)~0ull ? ~0ull : (uint64_t)ns128;
}
// Duration::mul_f32: from_secs_f32(rhs * self.as_secs_f32())
uint64_t mul_f32(uint64_t ns, float rhs) {
    const float secs = (float)(ns / NS) + (float)(ns % NS) / 1e9f;
    return duration_from_secs_f32(rhs * secs);
}

enum StepResult { PERFORM_STEP_AND_CALL_AGAIN, CAUGHT_UP_WITH_RENDER_TIME, DROPPING_SIMULATION_STEPS };   // timer.rs:37-43

struct Timer {                                   // timer.rs:18-35
    Clock::time_point timestamp_last_frame = Clock::now();
    uint64_t duration_last_frame = 0;
    uint64_t total_rendered_time = 0, current_frame_delta = 0;
    uint32_t num_frames_rendered = 0;
    uint64_t simulation_delta = 0;
    uint32_t num_simulation_steps = 0, num_simulation_steps_this_frame = 0;
    uint64_t total_simulated_time = 0, accepted_simulation_to_render_lag = 0;

    void force_frame_delta(uint64_t delta) {     // :70-74
        total_rendered_time -= current_frame_delta;
        current_frame_delta = delta;
        total_rendered_time += current_frame_delta;
    }
    void on_frame_submitted(float time_scale, uint64_t measured_ns) {   // :76-88 (measured_ns replaces timestamp_last_frame.elapsed())
        duration_last_frame = measured_ns;
        current_frame_delta = mul_f32(duration_last_frame, time_scale);
        total_rendered_time += current_frame_delta;
        timestamp_last_frame = Clock::now();
        num_simulation_steps_this_frame = 0;
        num_frames_rendered += 1;
    }
    void skip_simulation_frame() { accepted_simulation_to_render_lag += current_frame_delta; }   // :90-92
    StepResult simulation_frame_loop(uint64_t max_total_step_per_frame) {                        // :94-130
        const uint64_t behind = total_simulated_time + accepted_simulation_to_render_lag;
        const uint64_t residual = total_rendered_time > behind ? total_rendered_time - behind : 0;   // (the reference's checked_sub().unwrap() panics instead)
        if (residual < simulation_delta) return CAUGHT_UP_WITH_RENDER_TIME;
        const unsigned __int128 stepped = (unsigned __int128)num_simulation_steps_this_frame * simulation_delta;
        if (stepped > (unsigned __int128)max_total_step_per_frame) {
            accepted_simulation_to_render_lag += mul_f32(residual, 0.9f);
            return DROPPING_SIMULATION_STEPS;
        }
        num_simulation_steps_this_frame += 1;
        num_simulation_steps += 1;
        total_simulated_time += simulation_delta;
        return PERFORM_STEP_AND_CALL_AGAIN;
    }
};

uint64_t delta_from_steps_per_second(uint64_t sps) { return NS / sps; }   // simulation_controller.rs:33-35

}  // namespace

struct blub_controller {                          // simulation_controller.rs:19-26
    Timer timer;
    uint64_t computation_time_last_fast_forward = 0;
    uint64_t simulation_steps_per_second = 120;
    int status = BLUB_CONTROLLER_REALTIME;
    uint64_t status_duration = 0;                 // payload of RecordingWithFixedFrameLength / FastForward
    uint64_t simulation_stop_time = 60ull * 60ull * NS;   // an hour (:44)
    float time_scale = 1.0f;
};

namespace {

bool start_simulation_frame(blub_controller* c) {   // :175-190
    switch (c->status) {
    case BLUB_CONTROLLER_REALTIME: break;
    case BLUB_CONTROLLER_RECORDING: case BLUB_CONTROLLER_FAST_FORWARD: c->timer.force_frame_delta(c->status_duration); break;
    case BLUB_CONTROLLER_PAUSED: c->timer.skip_simulation_frame(); return false;
    }
    return true;
}

// :192-217; *rc receives the callback's status (a failing step ends the frame)
bool single_step(blub_controller* c, const blub_step_callbacks* cb, int* rc) {
    const uint64_t max_total = c->status == BLUB_CONTROLLER_REALTIME ? MAX_STEP_COMPUTATION_PER_FRAME_NS : NO_LIMIT;
    if (c->timer.total_simulated_time + c->timer.simulation_delta > c->simulation_stop_time) {
        c->status = BLUB_CONTROLLER_PAUSED;
        return false;
    }
    if (c->timer.simulation_frame_loop(max_total) == PERFORM_STEP_AND_CALL_AGAIN) {
        // Scene::step sees a timer that has already advanced by this step (timer.rs:124); dt = Duration::as_secs_f32
        const uint64_t d = c->timer.simulation_delta;
        const float dt = (float)(d / NS) + (float)(d % NS) / 1e9f;
        *rc = cb->step(cb->user, dt, c->timer.total_simulated_time);
        if (*rc != BLUB_OK) {      // the step did not happen: take it off the clocks again (round-2 ADVICE)
            c->timer.num_simulation_steps_this_frame -= 1; c->timer.num_simulation_steps -= 1; c->timer.total_simulated_time -= d;
        }
        return *rc == BLUB_OK;
    }
    return false;
}

int fluid_step_cb(void* user, float dt, uint64_t) {
    int rc = blub_fluid_step((blub_fluid*)user, dt);
    return rc != BLUB_OK ? rc : blub_fluid_update_statistics((blub_fluid*)user);   // Scene::step, scene/mod.rs:199-213
}
int fluid_wait_cb(void* user) { return blub_fluid_synchronize((blub_fluid*)user); }

}  // namespace

extern "C" {

int blub_controller_create(uint64_t steps_per_second, blub_controller** out) {   // SimulationController::new, :38-50
    if (!out) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (steps_per_second == 0) steps_per_second = 120;                           // DEFAULT_SIMULATION_STEPS_PER_SECOND
    if (steps_per_second > NS) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "steps per second out of range");
    blub_controller* c = new (std::nothrow) blub_controller();
    if (!c) return blub::set_error(BLUB_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->simulation_steps_per_second = steps_per_second;
    c->timer.simulation_delta = delta_from_steps_per_second(steps_per_second);
    *out = c;
    return BLUB_OK;
}
void blub_controller_destroy(blub_controller* c) { delete c; }
int blub_controller_set_simulation_steps_per_second(blub_controller* c, uint64_t sps) {   // :88-92
    if (!c || sps == 0 || sps > NS) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    c->simulation_steps_per_second = sps;
    c->timer.simulation_delta = delta_from_steps_per_second(sps);
    return BLUB_OK;
}
uint64_t blub_controller_simulation_steps_per_second(const blub_controller* c) { return c ? c->simulation_steps_per_second : 0; }
uint64_t blub_controller_simulation_delta_ns(const blub_controller* c) { return c ? c->timer.simulation_delta : 0; }
uint64_t blub_controller_total_simulated_time_ns(const blub_controller* c) { return c ? c->timer.total_simulated_time : 0; }
uint64_t blub_controller_total_render_time_ns(const blub_controller* c) { return c ? c->timer.total_rendered_time : 0; }
uint32_t blub_controller_num_simulation_steps_performed(const blub_controller* c) { return c ? c->timer.num_simulation_steps : 0; }
uint32_t blub_controller_num_simulation_steps_performed_for_current_frame(const blub_controller* c) { return c ? c->timer.num_simulation_steps_this_frame : 0; }
uint64_t blub_controller_computation_time_last_fast_forward_ns(const blub_controller* c) { return c ? c->computation_time_last_fast_forward : 0; }
int blub_controller_get_status(const blub_controller* c) { return c ? c->status : BLUB_ERR_INVALID_ARGUMENT; }
int blub_controller_set_simulation_stop_time_ns(blub_controller* c, uint64_t t) { if (!c) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle"); c->simulation_stop_time = t; return BLUB_OK; }
uint64_t blub_controller_simulation_stop_time_ns(const blub_controller* c) { return c ? c->simulation_stop_time : 0; }
int blub_controller_set_time_scale(blub_controller* c, float s) { if (!c || !(s >= 0.0f)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument"); c->time_scale = s; return BLUB_OK; }
int blub_controller_pause_or_resume(blub_controller* c) {   // :72-78
    if (!c) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    c->status = c->status == BLUB_CONTROLLER_PAUSED ? BLUB_CONTROLLER_REALTIME : BLUB_CONTROLLER_PAUSED;
    return BLUB_OK;
}
int blub_controller_start_recording_with_fixed_frame_length(blub_controller* c, double frames_per_second) {   // :80-82
    if (!c || !(frames_per_second > 0.0)) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "bad argument");
    c->status = BLUB_CONTROLLER_RECORDING;
    c->status_duration = (uint64_t)(1.0 / frames_per_second * 1e9);
    return BLUB_OK;
}
int blub_controller_restart(blub_controller* c) {   // :94-96
    if (!c) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    c->timer = Timer();
    c->timer.simulation_delta = delta_from_steps_per_second(c->simulation_steps_per_second);
    return BLUB_OK;
}
int blub_controller_on_frame_submitted(blub_controller* c, int64_t measured_frame_duration_ns) {   // :56-58 -> timer.rs:76-88
    if (!c) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    uint64_t d = measured_frame_duration_ns >= 0 ? (uint64_t)measured_frame_duration_ns
                                                  : (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - c->timer.timestamp_last_frame).count();
    c->timer.on_frame_submitted(c->time_scale, d);
    return BLUB_OK;
}

int blub_controller_frame_steps(blub_controller* c, const blub_step_callbacks* cb, uint32_t* steps_out) {   // :159-173
    if (!c || !cb || !cb->step) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    if (steps_out) *steps_out = 0;
    if (!start_simulation_frame(c)) return BLUB_OK;
    int rc = BLUB_OK;
    uint32_t n = 0;
    while (single_step(c, cb, &rc)) ++n;
    if (steps_out) *steps_out = n;
    return rc;
}

int blub_controller_fast_forward_steps(blub_controller* c, uint64_t simulation_jump_length_ns, const blub_step_callbacks* cb, uint32_t* steps_out) {   // :96-157
    if (!c || !cb || !cb->step) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null argument");
    if (steps_out) *steps_out = 0;
    c->status = BLUB_CONTROLLER_FAST_FORWARD;
    const uint64_t jump = std::max(simulation_jump_length_ns, c->timer.simulation_delta);   // "jump at least one simulation step", :119-121
    // (the reference forces the UN-maxed length as frame delta, :177-185, and would then spin in its batch loop without ever stepping
    //  when the jump is shorter than one step; the intent stated at :119 is one step)
    c->status_duration = jump;
    const uint64_t previous_simulation_end = c->simulation_stop_time;
    c->simulation_stop_time = c->timer.total_simulated_time + jump;
    start_simulation_frame(c);
    int rc = BLUB_OK;
    uint32_t finished = 0;
    const auto start_time = Clock::now();
    while (c->status == BLUB_CONTROLLER_FAST_FORWARD) {
        int batch = MAX_FAST_FORWARD_SIMULATION_BATCH_SIZE;
        for (int i = 0; i < MAX_FAST_FORWARD_SIMULATION_BATCH_SIZE; ++i)
            if (!single_step(c, cb, &rc)) { batch = i; break; }
        if (cb->wait) { int rw = cb->wait(cb->user); if (rc == BLUB_OK) rc = rw; }   // device.poll(Maintain::Wait), :140
        finished += (uint32_t)batch;
        // (the reference would spin forever here if the render clock lagged the simulation clock by more than the jump; it never does
        //  because frames only add render time.  A failing step callback must not hang either.)
        if (rc != BLUB_OK || (batch == 0 && c->status == BLUB_CONTROLLER_FAST_FORWARD)) { c->status = BLUB_CONTROLLER_PAUSED; break; }
    }
    c->computation_time_last_fast_forward = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - start_time).count();   // :147
    c->timer.on_frame_submitted(1.0f, (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - c->timer.timestamp_last_frame).count());
    c->timer.force_frame_delta(0);
    c->simulation_stop_time = previous_simulation_end;
    if (steps_out) *steps_out = finished;
    return rc;
}

int blub_controller_frame_steps_fluid(blub_controller* c, blub_fluid* h, uint32_t* steps_out) {
    if (!h) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    blub_step_callbacks cb{fluid_step_cb, fluid_wait_cb, h};
    return blub_controller_frame_steps(c, &cb, steps_out);
}
int blub_controller_fast_forward_steps_fluid(blub_controller* c, uint64_t jump_ns, blub_fluid* h, uint32_t* steps_out) {
    if (!h) return blub::set_error(BLUB_ERR_INVALID_ARGUMENT, "null handle");
    blub_step_callbacks cb{fluid_step_cb, fluid_wait_cb, h};
    return blub_controller_fast_forward_steps(c, jump_ns, &cb, steps_out);
}

}  // extern "C"
