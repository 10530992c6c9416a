import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import blub_amd
from blub_amd.simulation_controller import SimulationController
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dt = blub_amd.default_simulation_delta()
for mode in ("python", "native", "python", "native"):
    scene = blub_amd.Scene(path=os.path.join(ROOT, "scenes", "corner_dams_256.json"))
    f = scene.fluid()
    for _ in range(10): scene.step(dt)
    f.synchronize()
    it0 = f.total_solver_iterations()
    t0 = time.perf_counter()
    if mode == "python":
        for _ in range(120): scene.step(dt)
        f.synchronize()
    else:
        c = SimulationController(); c.fast_forward_steps_fluid(f, 120 * c.simulation_delta_ns); c.close()
    el = time.perf_counter() - t0
    print(mode, "%.1f steps/s" % (120 / el), "iters/step %.1f" % ((f.total_solver_iterations() - it0) / 120))
    f.close()
