"""Development probe: where do the free-running engine and the oracle part ways?  A = engine.step() (sparse brick lists),
C = engine through run_stage (every brick active, the reference's dense semantics), B = oracle.step()."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blub_amd
from oracle.oracle import Oracle
from tests import util

name = sys.argv[1] if len(sys.argv) > 1 else "dam_halfhalf"
binning = sys.argv[2] if len(sys.argv) > 2 else "fixed"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
path = os.path.join(ROOT, "scenes", name + ".json")
sA = blub_amd.Scene(path=path); A = sA.fluid()
sC = blub_amd.Scene(path=path); C = sC.fluid()
nx, ny, nz = A.grid_dimension()
B = Oracle(nx, ny, nz, A.num_particles() + 64)
B.set_quirks(binning=binning)
g = np.float32(list(sA.config.gravity)) / np.float32(sA.config.grid_to_world_scale)
B.set_gravity_grid(g)
B.set_particles(A.get_particles()[0])
if binning == "off":
    A.particle_rebinning_step_frequency = 0; C.particle_rebinning_step_frequency = 0
def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
for step in range(steps):
    A.step(util.DT); A.synchronize()
    B.step(util.DT)
    for st in util.STEP_ORDER:
        if st == "binning" and (binning == "off" or C.step_counter % 60 != 0):
            continue
        C.run_stage(st, util.DT)
    C.step_counter = C.step_counter + 1
    C.synchronize()
    for w in (0, 1):
        print("step %d solver %d: oracle %s | engine.step %s | engine.run_stage %s" % (step, w, B.solver_stats(w), A.solver_stats(w), C.solver_stats(w)))
    for vol in ("pressure_velocity", "pressure_density", "vel_y"):
        b = B.read_volume(vol)
        print("   %-18s rel L2 vs oracle: step() %.3g  run_stage %.3g   | step() vs run_stage %.3g" % (vol, rel(A.read_volume(vol), b), rel(C.read_volume(vol), b), rel(A.read_volume(vol), C.read_volume(vol))))
    mb = B.read_volume("marker")
    print("   marker mismatches: step() %d run_stage %d" % ((A.read_volume("marker") != mb).sum(), (C.read_volume("marker") != mb).sum()))
    if binning == "off":
        pb = B.get_particles()[0][:, :3]
        for lab, f in (("step()", A), ("run_stage", C)):
            d = np.abs(f.get_particles()[0][:, :3] - pb).max(axis=1)
            print("   positions %-9s: median %.3g p99 %.3g max %.3g" % (lab, np.median(d), np.quantile(d, 0.99), d.max()))
