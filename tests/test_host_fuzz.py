"""The host-only entry points of the C-ABI never crash on malformed input: a mutation fuzzer over the scene JSON reader, the
static-object animation and the OBJ reader, compiled with AddressSanitizer + UBSan (no GPU, no HIP)."""
import json
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_parsers_survive_mutated_inputs(tmp_path):
    exe = tmp_path / "fuzz"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "blub_amd", "csrc"), os.path.join(ROOT, "tests", "native", "fuzz_host_parsers.cpp"),
           os.path.join(ROOT, "blub_amd", "csrc", "scene_host.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not available: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    scene = json.load(open(os.path.join(ROOT, "scenes", "wavegenerator_cube.json")))
    scene["static_objects"][0]["animation"]["rotation"] = {"axis": {"x": 0, "y": 1, "z": 0}, "deg_per_sec": 90.0}
    base = tmp_path / "base.json"
    base.write_text(json.dumps(scene))
    run = subprocess.run([str(exe), str(base), os.path.join(ROOT, "scenes", "models", "unit_cube.obj"), str(tmp_path / "m.obj"), "12000"],
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    accepted, rejected = [int(v) for v in run.stdout.split() if v.isdigit()]
    assert accepted > 1000 and rejected > 1000      # both the accepting and the rejecting paths were exercised
