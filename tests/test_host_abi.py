"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol include/blubhip.h declares,
and its host-only logic (scene JSON, particle seeding, error behaviour) matches the reference's rules. No GPU calls."""
import json
import os
import re

import numpy as np
import pytest

import blub_amd
from blub_amd.hybrid_fluid import BlubError
from oracle.oracle import Oracle
from tests.conftest import ROOT, has_gpu


def declared_functions():
    text = open(os.path.join(ROOT, "include", "blubhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blub_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    lib = ctypes.CDLL(blub_amd.lib_path())
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libblubhip.so does not export %s" % n
    blub_amd.load_library()


def test_integration_md_ffi_block_is_generated_from_the_header():
    """INTEGRATION.md section 2 is the output of tools/gen_rust_ffi.py for today's include/blubhip.h: every declared function has
    a Rust binding line, nothing is stale."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    block, names = gen.generate()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a, b = doc.index(gen.BEGIN) + len(gen.BEGIN), doc.index(gen.END)
    assert doc[a:b].strip() == block.strip(), "run `python tools/gen_rust_ffi.py --write`"
    assert sorted(names) == declared_functions()


def test_no_reference_shader_text_is_tracked():
    """oracle/_ref/ (the reference's shader text turned into C++ by oracle/glsl/build_ref.sh, and the library built from it) is a build product: git-ignored,
    never in history (round-4 review, hygiene).  Where this is a git checkout: nothing under oracle/_ref is tracked and no tracked file carries a line that
    only the reference's shaders have."""
    import subprocess
    try:
        tracked = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True, timeout=30)
    except (OSError, subprocess.TimeoutExpired):
        pytest.skip("git is not available")
    if tracked.returncode != 0:
        pytest.skip("not a git checkout")
    files = tracked.stdout.split()
    assert files and not [f for f in files if f.startswith("oracle/_ref/")]
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read()
    ref = "/root/reference/shader/simulation/transfer_gather_velocity.comp"
    if os.path.exists(ref):
        lines = [l.strip() for l in open(ref) if len(l.strip()) > 60][:8]      # long, characteristic lines of one of the reference's shaders
        assert lines
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".md", ".sh", ".c")):
                text = open(os.path.join(ROOT, f), errors="replace").read()
                assert not any(l in text for l in lines), "%s contains shader text of the reference" % f


def test_scene_json_matches_reference_schema(tmp_path):
    s = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "double_dam.json")).config
    assert list(s.grid_dimension) == [128, 64, 64] and s.max_num_particles == 2000000 and s.num_fluid_cubes == 2
    assert np.float32(s.grid_to_world_scale) == np.float32(0.01)
    assert np.allclose(list(s.gravity), [0, -9.81, 0]) and np.allclose(list(s.cube_min[1]), [0.96, 0, 0])
    # static_objects is #[serde(default)]; unknown fields are ignored like serde does (nested junk exercises the JSON reader)
    obj = {"model": "m.obj", "world_position": {"x": 1, "y": 2, "z": 3}, "scale": 0.5, "rotation_angles": {"x": 0, "y": 90, "z": 0}}
    txt = json.dumps({"gravity": {"x": 0, "y": -1, "z": 0}, "static_objects": [dict(obj, a=1), dict(obj, b=[1, 2, {"c": "d\\n"}])],
                      "fluid": {"world_position": {"x": 0, "y": 0, "z": 0}, "grid_to_world_scale": 1e-2, "max_num_particles": 10,
                                "grid_dimension": {"x": 32, "y": 32, "z": 32}, "fluid_cubes": []}})
    c = blub_amd.Scene.parse(text=txt).config
    assert c.num_static_objects == 2 and c.num_fluid_cubes == 0 and c.static_objects[1].model == b"m.obj"
    assert blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "double_dam.json")).config.num_static_objects == 0
    w = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", "wavegenerator_cube.json")).config
    assert w.num_static_objects == 1 and w.static_objects[0].has_translation == 1 and w.static_objects[0].model == b"unit_cube.obj"


@pytest.mark.parametrize("mutate,status", [
    (lambda d: d.pop("gravity"), -6), (lambda d: d["fluid"].pop("grid_dimension"), -6),
    (lambda d: d["fluid"].__setitem__("max_num_particles", -3), -6), (lambda d: d["fluid"]["fluid_cubes"].append({"min": {"x": 0}}), -6)])
def test_scene_json_errors(mutate, status):
    d = json.load(open(os.path.join(ROOT, "scenes", "dam_halfhalf.json")))
    mutate(d)
    with pytest.raises(BlubError) as e:
        blub_amd.Scene.parse(text=json.dumps(d))
    assert e.value.status == status
    with pytest.raises(BlubError) as e:
        blub_amd.Scene.parse(text="{ not json")
    assert e.value.status == -6
    with pytest.raises(BlubError) as e:
        blub_amd.Scene.parse(path="/nonexistent/scene.json")
    assert e.value.status == -5


@pytest.mark.parametrize("scene,expect", [("single_cell_debug", 8), ("dam_halfhalf", 1218672), ("double_dam", 1199328),
                                          ("corner_dams_128", 111600), ("corner_dams_256", 968688)])
def test_product_seeding_equals_oracle_bitwise(scene, expect):
    c = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", scene + ".json")).config
    dim = list(c.grid_dimension)
    o = Oracle(dim[0], dim[1], dim[2], c.max_num_particles)
    scale = np.float32(c.grid_to_world_scale)
    parts, total = [], 0
    for i in range(c.num_fluid_cubes):
        mn, mx = np.float32(list(c.cube_min[i])) / scale, np.float32(list(c.cube_max[i])) / scale
        p = blub_amd.seed_fluid_cube(dim, c.max_num_particles, total, mn, mx)
        total += len(p)
        parts.append(p)
        o.add_fluid_cube(mn, mx)
    assert total == expect == o.num_particles
    got = np.concatenate(parts)
    assert np.array_equal(got.view(np.uint32), o.get_particles()[0].view(np.uint32))


def test_seeding_truncates_like_the_reference():
    p = blub_amd.seed_fluid_cube((32, 32, 32), 100, 40, (1, 1, 1), (9, 9, 9))   # hybrid_fluid.rs:627-633
    assert len(p) == 60
    p = blub_amd.seed_fluid_cube((32, 32, 32), 100, 0, (-5, 40, 3), (2, 50, 3))  # clamps to [1, dim-1]; empty extent
    assert len(p) == 0


def test_default_simulation_delta():
    assert blub_amd.default_simulation_delta() == float(np.float32(8333333) / np.float32(1e9))


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(BlubError) as e:
        blub_amd.HybridFluid((32, 32, 32), 16)
    assert e.value.status == -7


def test_constant_divisor_division_is_correctly_rounded(tmp_path):
    """The single-reduction PCG kernels divide by d = m 2^k (m in {1, 3, 5}) with a multiply and one fma correction step
    (blub_pcg1.hip.h, precond_exact).  tests/native/div_const_check.c checks q == y / m for EVERY f32 significand and both signs,
    plus a sweep over the binades, with the host's fused multiply-add (same IEEE operation as v_fma_f32)."""
    import shutil
    import subprocess
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU has no FMA instruction")
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = tmp_path / "div_const_check"
    subprocess.check_call([gcc, "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "div_const_check.c"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr
