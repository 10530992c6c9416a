"""The CPU oracle against the REFERENCE'S OWN compute shaders.

tests/golden/ref_*.npz were produced by compiling /root/reference/shader/simulation/**/*.comp -- unmodified -- with g++ through
oracle/glsl/ (glsl_shim.h, glsl2cpp.py) and dispatching them as HybridFluid::step / PressureSolver::solve record them
(oracle/glsl/ref_fluid.py; generator: tests/golden/make_ref_golden.py).  They are reference outputs.  The oracle, with its
reductions in the reference's own order (`set_dot_mode(2)`), has to reproduce them BIT FOR BIT:
every stage of a step (P2G with its 24 barriers, divergence with moving solids, both PCG solves incl. the statistics, pressure
projection, extrapolation, G2P with wall truncation / escape / push, density gather, position change), the PCG for both readings
of SURVEY Q1 at fixed iteration counts, the literal Q4 binning.  The one place Vulkan leaves the arithmetic to the
implementation -- the hardware trilinear filter of density_projection_correct_particles.comp:32-40 and of the push term of
advect_particles.comp:155-163 -- is pinned in its separable evaluation bit for bit, and bounded for the two others
(Vulkan's weighted-sum formula: 4e-6 cells; 8-bit filter weights: 3e-3 cells for 99.9 % of the particles, 3e-2 for all).

Where oracle/_ref can be built (this container: /root/reference is present) the same comparison also runs LIVE on other
seeds and grid shapes, including a grid whose cell count is not a multiple of 16384 (Q14).
"""
import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_scenarios as S  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype.itemsize == b.dtype.itemsize and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def assert_bits(tag, got, want):
    if not bits_equal(got, want):
        g, w = np.asarray(got), np.asarray(want)
        if g.shape == w.shape and g.dtype.kind == "f":
            d = np.abs(g.astype(np.float64) - w.astype(np.float64))
            raise AssertionError("%s: %d of %d values differ, max |d| = %.3e" % (tag, int((g.view(np.uint32) != w.view(np.uint32)).sum()), g.size, d.max()))
        raise AssertionError("%s differs (%s %s vs %s %s)" % (tag, g.shape, g.dtype, w.shape, w.dtype))


def scene_from_fixture(fx):
    return dict(dim=fx["dim"], pos=fx["pos_in"], vx=fx["vx_in"], vy=fx["vy_in"], vz=fx["vz_in"], solid=fx["solid"], gravity=fx["gravity"])


@pytest.fixture(scope="module")
def step_fx():
    return dict(np.load(os.path.join(GOLD, "ref_step_64x16x32.npz")))


def test_reference_fixtures_are_current():
    """The fixtures name the shader sources and the recipe they were made from; where the reference is present they must still match."""
    ref_shaders = "/root/reference/shader"
    for name in ("ref_step_64x16x32.npz", "ref_pcg_32x64x16.npz", "ref_binning_64x16x32.npz", "ref_freerun_64x16x32.npz", "ref_caps_64x16x32.npz", "ref_fullsize_dam_halfhalf.npz"):
        fx = np.load(os.path.join(GOLD, name))
        listed = dict(line.split("  ")[::-1] for line in str(fx["shader_sha256"]).strip().split("\n"))
        assert len(listed) >= 28 and "simulation/transfer_gather_velocity.comp" in listed
        if os.path.isdir(ref_shaders):
            for rel, digest in listed.items():
                with open(os.path.join(ref_shaders, rel), "rb") as f:
                    assert hashlib.sha256(f.read()).hexdigest() == digest, "%s changed since %s was generated" % (rel, name)
        for line in str(fx["recipe_sha256"]).strip().split("\n"):
            digest, rel = line.split("  ")
            with open(os.path.join(ROOT, "oracle", "glsl", rel), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == digest, "oracle/glsl/%s changed: regenerate with tests/golden/make_ref_golden.py" % rel


def test_every_stage_of_a_step_matches_the_reference_shaders_bit_for_bit(step_fx):
    sc = scene_from_fixture(step_fx)
    o = Oracle(*sc["dim"], len(sc["pos"]) + 64)
    S.configure(o, sc)
    rec = S.run_step_recording(o, float(step_fx["dt"]))
    assert set("s0/" + k for k in rec) <= set(step_fx.keys())
    for k, v in rec.items():
        assert_bits(k, v, step_fx["s0/" + k])
    # the scene reaches the branches it was built for
    adv = rec["advect/particles"].view(np.float32)
    hi = np.array(sc["dim"], np.float32) - np.float32(1.001)
    assert ((adv[:, :3] == np.float32(1.001)) | (adv[:, :3] == hi)).any(), "no particle ended on the domain clamp"
    c = sc["pos"].astype(int)
    assert (sc["solid"][c[:, 2], c[:, 1], c[:, 0], 3] > 0).sum() > 100, "no particle starts inside a solid voxel (escape path)"
    assert rec["solve_velocity/stats0"][1] >= 4 and rec["solve_density/stats1"][1] >= 4


def test_lists_beyond_the_gather_caps_match_the_reference_shaders_bit_for_bit():
    """Round-4 review, item 4b: P2G lists of up to 64 entries against the 12-round cap (transfer_gather_velocity.comp:61), density lists of up to 64
    against the 32-round cap (density_projection_gather_error.comp:69) -- the reference's shaders decide which particles take part (the first 12 / 32
    from the head: the LAST inserted), the oracle has to keep the same ones."""
    fx = dict(np.load(os.path.join(GOLD, "ref_caps_64x16x32.npz")))
    sc = scene_from_fixture(fx)
    o = Oracle(*sc["dim"], len(sc["pos"]) + 64)
    S.configure(o, sc)
    stages = S.STAGES[:S.STAGES.index("density_gather") + 1]
    rec = S.run_step_recording(o, float(fx["dt"]), stages=stages)
    assert set("s0/" + k for k in rec) == set(k for k in fx if k.startswith("s0/"))
    for k, v in rec.items():
        assert_bits(k, v, fx["s0/" + k])
    # the scene does what it was built for: list lengths beyond both caps, every list inside one block of 64 consecutive particles
    for off in ((1.0, 0.5, 0.5), (0.5, 1.0, 0.5), (0.5, 0.5, 1.0)):
        local, longest = S.lists_are_wave_local(sc["pos"], off, sc["dim"])
        assert local and longest == 64
    adv = rec["advect/particles"].view(np.float32)
    local, longest = S.lists_are_wave_local(adv[:, :3], (0.5, 0.5, 0.5), sc["dim"])
    assert local and longest == 64
    # ... and the caps bite: the density of a cell whose list holds 64 particles counts 32 of them (rho <= 32 there; all 64 would give ~40)
    assert np.abs(rec["density_gather/residual@fluid"]).max() > 0


@pytest.mark.slow
def test_a_full_size_step_matches_the_reference_shaders_hash_for_hash():
    """Round-4 review, item 4a: scenes/dam_halfhalf.json (128 x 64 x 64, 1 218 672 particles -- BASELINE configs[1]) through the reference's own shaders,
    two chained steps, every recorded array as a SHA-256 (tests/golden/ref_fullsize_dam_halfhalf.npz; make_ref_golden.py fullsize).  The oracle in the
    reference's literal reduction order reproduces every hash: the pin reaches the full-size arrays the engine is compared with on the GPU
    (tests/test_gpu_vs_ref.py::test_full_size_step...)."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_ref_golden as G
    fx = dict(np.load(os.path.join(GOLD, "ref_fullsize_dam_halfhalf.npz")))
    dim, pos, g = G.full_size_scene(str(fx["scene"]))
    assert tuple(fx["dim"]) == dim and len(pos) == int(fx["num_particles"]) and sha(pos) == str(fx["pos_in_sha"])
    o = Oracle(dim[0], dim[1], dim[2], len(pos) + 64)
    o.set_gravity_grid(g)
    o.set_quirks(precond="zero", binning="off")
    o.set_dot_mode(2)
    o.set_particles(pos)
    for step in range(2):
        want = dict(zip(fx["s%d/keys" % step], fx["s%d/sha" % step]))
        for st in S.STAGES:
            o.run_stage(st, float(fx["dt"]))
            for what in S.STAGE_OUTPUTS[st]:
                assert sha(S.capture(o, what)) == want["%s/%s" % (st, what)], "step %d %s/%s differs from the reference's shaders" % (step, st, what)
        o.step_counter = o.step_counter + 1


def test_three_chained_steps_match_the_reference_shaders_bit_for_bit(step_fx):
    sc = scene_from_fixture(step_fx)
    o = Oracle(*sc["dim"], len(sc["pos"]) + 64)
    S.configure(o, sc)
    S.run_step_recording(o, float(step_fx["dt"]))
    for step in (1, 2):
        rec = S.run_step_recording(o, float(step_fx["dt"]))
        want = dict(line.split(" ") for line in step_fx["s%d/sha" % step])
        bad = [k for k, v in sorted(rec.items()) if sha(v) != want[k]]
        assert not bad, "step %d: %s differ from the reference" % (step, bad)
        assert_bits("particles after step %d" % step, S.capture(o, "particles"), step_fx["s%d/particles" % step])


def test_eight_free_running_steps_with_converged_solves_match_the_reference_shaders_bit_for_bit():
    """Free-running, not stage by stage: eight steps with solves converged to 1e-4 (36 - 52 iterations each).  The oracle in the reference's
    literal reduction order reproduces every particle position of every step and the solver statistics exactly."""
    fx = dict(np.load(os.path.join(GOLD, "ref_freerun_64x16x32.npz")))
    sc = scene_from_fixture(fx)
    max_iter, tol, freq = fx["solver"]
    o = Oracle(*sc["dim"], len(sc["pos"]) + 64)
    S.configure(o, sc, max_iter=int(max_iter), tol=float(tol), freq=int(freq))
    steps = sum(1 for k in fx if k.endswith("/pos"))
    assert steps == 8
    for step in range(steps):
        o.step(float(fx["dt"]))
        assert_bits("particle positions after step %d" % step, S.capture(o, "particles_pos"), fx["s%d/pos" % step])
        stats = np.array([o.solver_stats(0), o.solver_stats(1)], np.float64)
        assert np.array_equal(stats[:, 1], fx["s%d/stats" % step][:, 1]) and np.allclose(stats[:, 0], fx["s%d/stats" % step][:, 0], rtol=1e-6, atol=0), (step, stats, fx["s%d/stats" % step])


def test_the_hardware_filter_is_the_only_implementation_defined_arithmetic(step_fx):
    """Same step with Vulkan's weighted-sum form of the trilinear filter and with 8-bit filter weights: only the particles the
    filter touches move, by at most the stated bounds."""
    sep_adv = step_fx["s0/advect/particles"].view(np.float32)
    sep_pos = step_fx["s0/correct/particles_pos"]
    for flt, bound_adv, bound_pos, bound_p999 in (("weighted", 4e-6, 4e-6, 2e-6), ("weighted8", 5e-3, 3e-2, 3e-3)):
        adv = step_fx["s0_%s/advect/particles" % flt].view(np.float32)
        pos = step_fx["s0_%s/correct/particles_pos" % flt]
        moved = np.abs(adv[:, :3] - sep_adv[:, :3]).max(1)
        assert 0 < (moved > 0).sum() < 0.05 * len(adv)        # only pushed particles (stuck inside a solid) see the filter in A1
        assert moved.max() <= bound_adv
        assert np.array_equal(adv[:, 3:].view(np.uint32), sep_adv[:, 3:].view(np.uint32))   # links and APIC rows never do
        d = np.abs(pos - sep_pos).max(1)
        # (8-bit weights: a handful of particles sit where the filtered displacement decides whether the wall branch is taken)
        assert 0 < d.max() <= bound_pos and np.percentile(d, 99.9) <= bound_p999, (flt, d.max(), np.percentile(d, 99.9))


def test_pcg_matches_the_reference_shaders_bit_for_bit():
    fx = dict(np.load(os.path.join(GOLD, "ref_pcg_32x64x16.npz")))
    prob = dict(dim=fx["dim"], marker=fx["marker"], b=fx["b"], p0=fx["p0"])
    fluid = prob["marker"] == 1
    nb = np.zeros(fluid.shape, int)
    m = np.pad(prob["marker"], 1)
    for ax in (0, 1, 2):
        for sh_ in (-1, 1):
            nb += (np.roll(m, sh_, ax)[1:-1, 1:-1, 1:-1] != 0)
    assert set(np.unique(nb[fluid])) == set(range(7)), "the problem must exercise every diagonal d = 0..6"
    for tag in fx["cases"]:
        tag = str(tag)
        if tag == "default":
            precond, k, warm, tol = "zero", 32, False, 0.1
        else:
            precond, kk, w = tag.split("_")
            k, warm, tol = int(kk[1:]), w == "warm", 0.0
        o = Oracle(*prob["dim"], 8)
        res = S.run_pcg(o, prob, k, precond, tol=tol, warm=warm, dt=float(fx["dt"]))
        assert_bits(tag + " statistics", res["stats"], fx[tag + "/stats"])
        for q, want in zip(("p", "r", "s"), fx[tag + "/sha"]):
            if tag + "/" + q in fx:
                assert_bits("%s %s (FLUID cells)" % (tag, q), res[q][fluid], fx[tag + "/" + q])
            assert sha(res[q]) == str(want), "%s: %s differs from the reference" % (tag, q)
    assert fx["default/stats"][1] < 32, "the default-configuration case is meant to converge before the iteration cap"
    # reading B of Q1 does not converge: the reference's own shaders say so
    assert fx["lod0_k8_warm/stats"][0] > 3 * fx["zero_k8_warm/stats"][0]   # (the walled-in d = 0 cell keeps its |b| in both)


def test_literal_binning_matches_the_reference_shaders_bit_for_bit():
    fx = np.load(os.path.join(GOLD, "ref_binning_64x16x32.npz"))
    o = Oracle(*fx["dim"], int(fx["max_num_particles"]))
    o.set_quirks(binning="literal")
    o.set_particles(fx["pos_in"])
    o.run_stage("binning", S.DT)
    got = o.get_particles()[0]
    assert_bits("binned particles", got, fx["pos_out"])
    # Q4 as the shaders run it: the records are NOT a permutation of the input (slot 0 is never written, padding threads bin stale rows)
    a = np.sort(fx["pos_in"].view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    b = np.sort(np.ascontiguousarray(fx["pos_out"][:, :3]).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    assert not np.array_equal(a, b)


# ---- live: oracle/_ref rebuilt from /root/reference (skipped where the reference is absent, e.g. on the GPU box) -------------

def _ref_fluid():
    from oracle.glsl import ref_fluid
    if not ref_fluid.available():
        pytest.skip("oracle/_ref/libblubref.so is absent and /root/reference is not available to build it")
    return ref_fluid


def test_the_shim_compiles_every_simulation_shader():
    rf = _ref_fluid()
    names = set(rf.shader_names())
    assert len(names) == 21 and {"transfer_gather_velocity", "density_projection_gather_error", "pressure_reduce_sum", "pressure_reduce_max",
                                 "particle_binning_prefixsum", "advect_particles"} <= names


@pytest.mark.parametrize("seed,dim,solids", [(11, (32, 32, 32), True), (12, (16, 64, 32), False), (13, (48, 24, 32), True)])
def test_live_step_against_the_reference_shaders(seed, dim, solids):
    """Other seeds and shapes; (48, 24, 32) has N = 36864, not a multiple of 16384: the reference's last reduction level then reads
    only floor(N / 16384) of its ceil(N / 16384) partial sums (pressure_solver.rs:578, 587) -- Q14 -- and the oracle's literal
    reduction has to drop the same terms."""
    rf = _ref_fluid()
    sc = S.step_scene(seed, dim, solids)
    if dim[1] < 16:
        pytest.skip("scene does not fit")
    n = len(sc["pos"])
    o = Oracle(*dim, n + 64)
    r = rf.RefFluid(*dim, n + 64)
    S.configure(o, sc)
    S.configure(r, sc, is_ref=True)
    r.set_modes(filter="separable")
    for step in range(2):
        ro, rr = S.run_step_recording(o), S.run_step_recording(r)
        for k in ro:
            assert_bits("step %d %s" % (step, k), ro[k], rr[k])


def test_live_free_running_steps_with_literal_binning():
    """Whole steps through HybridFluid::step's own sequence (binning on step 0, Q13) instead of stage by stage."""
    rf = _ref_fluid()
    sc = S.step_scene(21, (32, 32, 32), True)
    n = len(sc["pos"])
    o = Oracle(32, 32, 32, n + 64)
    r = rf.RefFluid(32, 32, 32, n + 64)
    S.configure(o, sc)
    S.configure(r, sc, is_ref=True)
    o.set_quirks(binning="literal")
    r.binning_enabled = True
    r.set_modes(filter="separable")
    for w in (0, 1):
        o.set_solver_config(w, 0.1, 8, 4)
        r.set_solver_config(w, 0.1, 8, 4)
    for step in range(2):
        o.step(S.DT)
        r.step(S.DT)
        assert_bits("particles after step %d" % step, S.capture(o, "particles"), S.capture(r, "particles"))
        assert o.solver_stats(0) == pytest.approx(r.solver_stats(0), abs=0) and o.solver_stats(1) == pytest.approx(r.solver_stats(1), abs=0)


def test_live_pcg_on_random_problems():
    rf = _ref_fluid()
    for seed, dim in ((101, (32, 32, 32)), (102, (64, 32, 16))):
        prob = S.pcg_problem(seed, dim)
        for precond, k, warm in (("zero", 5, True), ("lod0", 3, False), ("zero", 12, False)):
            a = S.run_pcg(Oracle(*dim, 8), prob, k, precond, warm=warm)
            b = S.run_pcg(rf.RefFluid(*dim, 8), prob, k, precond, warm=warm, is_ref=True)
            for q in a:
                assert_bits("seed %d %s k=%d %s" % (seed, precond, k, q), a[q], b[q])
