"""Generates tests/golden/ref_*.npz from the REFERENCE'S OWN compute shaders.

The shaders under /root/reference/shader/simulation are compiled with g++ through oracle/glsl/ (glsl_shim.h + glsl2cpp.py ->
oracle/_ref/libblubref.so) and dispatched in the order HybridFluid::step / PressureSolver::solve record them
(oracle/glsl/ref_fluid.py).  These fixtures are therefore reference outputs, not oracle outputs: they are what pins the CPU
oracle (tests/test_oracle_vs_ref.py) and, through it and directly, the HIP engine (tests/test_gpu_vs_ref.py).

Needs /root/reference (this container; the GPU box only sees the committed .npz files).  Run from the repo root:
    python tests/golden/make_ref_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_scenarios as S  # noqa: E402
from oracle.glsl import ref_fluid  # noqa: E402
from oracle.glsl.ref_fluid import RefFluid  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def provenance():
    with open(os.path.join(ROOT, "oracle", "_ref", "shader_sha256.txt")) as f:
        shaders = f.read()
    recipe = "".join(hashlib.sha256(open(os.path.join(ROOT, "oracle", "glsl", n), "rb").read()).hexdigest() + "  " + n + "\n"
                     for n in ("glsl_shim.h", "glsl2cpp.py", "ref_runtime.cpp", "ref_fluid.py"))
    return dict(shader_sha256=np.array(shaders), recipe_sha256=np.array(recipe))


def step_fixture():
    sc = S.step_scene(seed=2024, dim=S.STEP_DIM)
    n = len(sc["pos"])
    out = dict(provenance(), dim=sc["dim"], dt=np.float32(S.DT), gravity=sc["gravity"], solid=sc["solid"], pos_in=sc["pos"], vx_in=sc["vx"], vy_in=sc["vy"], vz_in=sc["vz"])
    # (a) the separable evaluation of the trilinear filter: every recorded array of step 0 + the particles after steps 1 and 2
    r = RefFluid(*sc["dim"], n + 64)
    S.configure(r, sc, is_ref=True)
    r.set_modes(filter="separable")
    rec = S.run_step_recording(r)
    for k, v in rec.items():
        out["s0/" + k] = v
    for step in (1, 2):
        rec = S.run_step_recording(r)
        out["s%d/particles" % step] = S.capture(r, "particles")
        out["s%d/stats" % step] = np.stack([rec["solve_velocity/stats0"], rec["solve_density/stats1"]])
        out["s%d/sha" % step] = np.array(["%s %s" % (k, sha(v)) for k, v in sorted(rec.items())])
    # (b) the filter-dependent outputs of step 0 under the two other evaluations (Vulkan's weighted sum; 8-bit weights)
    for flt in ("weighted", "weighted8"):
        r = RefFluid(*sc["dim"], n + 64)
        S.configure(r, sc, is_ref=True)
        r.set_modes(filter=flt)
        rec = S.run_step_recording(r)
        out["s0_%s/advect/particles" % flt] = rec["advect/particles"]
        out["s0_%s/correct/particles_pos" % flt] = rec["correct/particles_pos"]
        # (round 5: what `correct` reads in this mode -- the marker after advection and the position-change volumes -- so that an engine with the same
        #  filter arithmetic can be held to correct/particles_pos bit for bit, tests/test_gpu_vs_ref.py)
        out["s0_%s/advect/marker" % flt] = rec["advect/marker"]
        for v in ("vel_x", "vel_y", "vel_z"):
            out["s0_%s/position_change/%s" % (flt, v)] = rec["position_change/" + v]
    np.savez_compressed(os.path.join(HERE, "ref_step_64x16x32.npz"), **out)


PCG_CASES = [("zero", 0, True), ("zero", 1, True), ("zero", 4, True), ("zero", 7, True), ("zero", 8, True), ("zero", 8, False), ("zero", 32, True),
             ("lod0", 1, True), ("lod0", 8, True)]
PCG_FULL = {("zero", 4, True), ("zero", 8, True), ("zero", 32, True), ("lod0", 8, True)}


def pcg_fixture():
    prob = S.pcg_problem(seed=7, dim=S.PCG_DIM)
    out = dict(provenance(), dim=prob["dim"], dt=np.float32(S.DT), marker=prob["marker"], b=prob["b"], p0=prob["p0"])
    fluid = prob["marker"] == 1
    cases = []
    for precond, k, warm in PCG_CASES:
        r = RefFluid(*prob["dim"], 8)
        res = S.run_pcg(r, prob, k, precond, warm=warm, is_ref=True)
        tag = "%s_k%d_%s" % (precond, k, "warm" if warm else "cold")
        cases.append(tag)
        out[tag + "/stats"] = res["stats"]
        out[tag + "/sha"] = np.array([sha(res[q]) for q in ("p", "r", "s")])
        if (precond, k, warm) in PCG_FULL:
            for q in ("p", "r", "s"):
                out[tag + "/" + q] = res[q][fluid]          # FLUID cells in memory order; p is 0 elsewhere (asserted by the hash)
    # the reference's defaults on the same problem: tolerance 0.1, <= 32 iterations, check every 4
    r = RefFluid(*prob["dim"], 8)
    res = S.run_pcg(r, prob, 32, "zero", tol=0.1, warm=False, is_ref=True)
    cases.append("default")
    out["default/stats"] = res["stats"]
    out["default/sha"] = np.array([sha(res[q]) for q in ("p", "r", "s")])
    for q in ("p", "r", "s"):
        out["default/" + q] = res[q][fluid]
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "ref_pcg_32x64x16.npz"), **out)


def binning_fixture():
    bs = S.binning_scene(seed=5, dim=S.STEP_DIM)
    r = RefFluid(*bs["dim"], len(bs["pos"]) + 100)
    r.set_particles(bs["pos"])
    r.run_stage("binning", S.DT)
    np.savez_compressed(os.path.join(HERE, "ref_binning_64x16x32.npz"), **dict(provenance(), dim=bs["dim"], pos_in=bs["pos"], max_num_particles=np.int64(len(bs["pos"]) + 100),
                        pos_out=r.get_particles()[0], buffer_out=r.pos.copy(), counters=r.read_volume("linked_list")))


FREERUN_STEPS, FREERUN_SOLVER = 8, dict(max_iter=200, tol=1e-4, freq=4)


def freerun_fixture():
    """Eight FREE-RUNNING steps of the step scene with CONVERGED solves (tolerance 1e-4: ~44 iterations per solve): the particles after every
    step.  With the reference's loose default (0.1) two runs that differ in the rounding of a dot product drift apart by construction (the CG is
    stopped unconverged); converged, the trajectory is a property of the algorithm and an implementation can be held to it step after step."""
    sc = S.step_scene(seed=2024, dim=S.STEP_DIM)
    n = len(sc["pos"])
    out = dict(provenance(), dim=sc["dim"], dt=np.float32(S.DT), gravity=sc["gravity"], solid=sc["solid"], pos_in=sc["pos"], vx_in=sc["vx"], vy_in=sc["vy"], vz_in=sc["vz"],
               solver=np.array([FREERUN_SOLVER["max_iter"], FREERUN_SOLVER["tol"], FREERUN_SOLVER["freq"]], np.float64))
    r = RefFluid(*sc["dim"], n + 64)
    S.configure(r, sc, is_ref=True, **FREERUN_SOLVER)
    r.set_modes(filter="separable")
    for step in range(FREERUN_STEPS):
        r.step(S.DT)
        out["s%d/pos" % step] = S.capture(r, "particles_pos")
        out["s%d/stats" % step] = np.array([r.solver_stats(0), r.solver_stats(1)], np.float64)
        print("free run step", step, r.solver_stats(0), r.solver_stats(1), flush=True)
    np.savez_compressed(os.path.join(HERE, "ref_freerun_64x16x32.npz"), **out)


def caps_fixture():
    """Lists beyond the 12- / 32-entry caps of the gathers in a reproducible insertion order (tests/ref_scenarios.py: caps_scene): every recorded array of
    step 0 up to the density gather."""
    sc = S.caps_scene()
    n = len(sc["pos"])
    ok_p2g = [S.lists_are_wave_local(sc["pos"], off, sc["dim"]) for off in ((1.0, 0.5, 0.5), (0.5, 1.0, 0.5), (0.5, 0.5, 1.0))]
    assert all(o[0] for o in ok_p2g) and max(o[1] for o in ok_p2g) == 64, ok_p2g
    out = dict(provenance(), dim=sc["dim"], dt=np.float32(S.DT), gravity=sc["gravity"], solid=sc["solid"], pos_in=sc["pos"], vx_in=sc["vx"], vy_in=sc["vy"], vz_in=sc["vz"])
    r = RefFluid(*sc["dim"], n + 64)
    S.configure(r, sc, is_ref=True)
    r.set_modes(filter="separable")
    rec = S.run_step_recording(r, stages=S.STAGES[:S.STAGES.index("density_gather") + 1])
    adv = rec["advect/particles"].view(np.float32)
    ok_den = S.lists_are_wave_local(adv[:, :3], (0.5, 0.5, 0.5), sc["dim"])
    assert ok_den[0] and ok_den[1] == 64, ok_den      # the density lists (built from the ADVECTED positions) are still wave-local, the longest holds 64
    for k, v in rec.items():
        out["s0/" + k] = v
    np.savez_compressed(os.path.join(HERE, "ref_caps_64x16x32.npz"), **out)


def full_size_scene(name="dam_halfhalf"):
    """A BASELINE scene at its full size as the backends take it: grid dimension, seeded particles (the oracle's restatement of add_fluid_cube; the
    product's generator equals it bit for bit, tests/test_host_abi.py), gravity in grid units.  No solids (the shipped fluid-only scenes)."""
    import json
    from oracle.oracle import Oracle
    cfg = json.load(open(os.path.join(ROOT, "scenes", name + ".json")))
    fl = cfg["fluid"]
    dim = (int(fl["grid_dimension"]["x"]), int(fl["grid_dimension"]["y"]), int(fl["grid_dimension"]["z"]))
    scale = np.float32(fl["grid_to_world_scale"])
    o = Oracle(dim[0], dim[1], dim[2], int(fl["max_num_particles"]))
    for c in fl["fluid_cubes"]:
        mn = np.float32([c["min"][k] for k in "xyz"]) / scale
        mx = np.float32([c["max"][k] for k in "xyz"]) / scale
        o.add_fluid_cube(mn, mx)
    pos = o.get_particles()[0][:, :3].copy()
    g = np.float32([cfg["gravity"][k] for k in "xyz"]) / scale      # scene/mod.rs:139
    return dim, pos, g


def fullsize_fixture(name="dam_halfhalf"):
    """Round-4 review, item 4a: a reference-shader fixture at a BASELINE size.  Step 0 of scenes/dam_halfhalf.json (128 x 64 x 64, 1 218 672 particles,
    the reference's default solver configuration) through the reference's own shaders, stage by stage -- minutes of CPU, so only the SHA-256 of
    every recorded array is kept (plus the solver statistics and two planes of the final pressure for eyeballing).  The oracle must reproduce every
    hash (tests/test_oracle_vs_ref.py, slow), which carries the pin to the full-size arrays the engine is compared with on the GPU."""
    import time
    dim, pos, g = full_size_scene(name)
    n = len(pos)
    out = dict(provenance(), scene=np.array(name), dim=np.array(dim), dt=np.float32(S.DT), gravity=g, num_particles=np.int64(n), pos_in_sha=np.array(sha(pos)))
    r = RefFluid(dim[0], dim[1], dim[2], n + 64)
    r.set_gravity_grid(g)
    for w in (0, 1):
        r.set_solver_config(w, error_tolerance=0.1, max_num_iterations=32, error_check_frequency=4)
    r.set_modes(precond="zero", filter="separable")
    r.binning_enabled = False
    r.set_particles(pos)
    for step in range(2):      # step 0 stage by stage, step 1 chained behind it
        keys, hashes = [], []
        for st in S.STAGES:
            t0 = time.time()
            r.run_stage(st, S.DT)
            for what in S.STAGE_OUTPUTS[st]:
                a = S.capture(r, what)
                keys.append("%s/%s" % (st, what)); hashes.append(sha(a))
                if what.startswith("stats"):
                    out["s%d/%s/%s" % (step, st, what)] = a
            print("step %d %s: %.1f s" % (step, st, time.time() - t0), flush=True)
        r.step_counter += 1
        out["s%d/keys" % step] = np.array(keys)
        out["s%d/sha" % step] = np.array(hashes)
    pv = r.read_volume("pressure_velocity")
    out["s0/pressure_velocity_planes"] = np.stack([pv[dim[2] // 4], pv[dim[2] // 2]])
    np.savez_compressed(os.path.join(HERE, "ref_fullsize_%s.npz" % name), **out)


if __name__ == "__main__":
    if "fullsize" in sys.argv[1:] or "caps" in sys.argv[1:] or "step" in sys.argv[1:]:      # (single fixtures: fullsize = the BASELINE-size one -- minutes)
        if ref_fluid.build(force=False) is None:
            sys.exit("the reference checkout is not available: cannot regenerate the reference fixtures")
        if "fullsize" in sys.argv[1:]:
            fullsize_fixture()
        if "caps" in sys.argv[1:]:
            caps_fixture()
        if "step" in sys.argv[1:]:
            step_fixture()
        sys.exit(0)
    if ref_fluid.build(force="freerun" not in sys.argv[1:]) is None:
        sys.exit("the reference checkout is not available: cannot regenerate the reference fixtures")
    if "freerun" not in sys.argv[1:]:      # (`make_ref_golden.py freerun`: only the free-running fixture)
        step_fixture()
        pcg_fixture()
        binning_fixture()
    freerun_fixture()
    for n in sorted(os.listdir(HERE)):
        if n.startswith("ref_"):
            print(n, os.path.getsize(os.path.join(HERE, n)))
