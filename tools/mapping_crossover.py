"""Which work mapping of the PCG kernels wins where: time of one 32-iteration solve (convergence exit disabled) on an all-FLUID slab of a grid,
for the dense 2.5-D mapping, the brick mapping with the reference's two-reduction schedule and the brick mapping with the single-reduction
schedule.  usage (GPU box): python tools/mapping_crossover.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import blub_amd  # noqa: E402


def run(shape, fill, mapping, schedule, iterations=32):
    nx, ny, nz = shape
    h = blub_amd.HybridFluid(shape, 16, binning="off")
    try:
        h.set_pcg_work_mapping(mapping)
        h.set_pcg_schedule(schedule)
        marker = np.full((nz, ny, nx), -1, np.int8) if False else None
        vol = h.read_volume("marker")
        m = np.full(vol.shape, -1, np.int8)
        top = max(2, int(round(fill * (vol.shape[1] - 2))) + 1)            # fluid in the lower part (y)
        idx = [slice(1, -1)] * 3
        m[0, :, :] = 0; m[-1, :, :] = 0; m[:, 0, :] = 0; m[:, -1, :] = 0; m[:, :, 0] = 0; m[:, :, -1] = 0
        m[1:-1, 1:top, 1:-1] = 1
        rng = np.random.default_rng(1)
        b = (rng.standard_normal(vol.shape) * (m == 1)).astype(np.float32)
        h.write_volume("marker", m)
        h.set_solver_config(0, error_tolerance=0.0, max_num_iterations=iterations, error_check_frequency=4)
        dt = blub_amd.default_simulation_delta()
        best = 1e30
        for rep in range(4):
            h.write_volume("residual", b)
            h.mark_pressure_initialised(0, False)
            h.synchronize()
            t0 = time.perf_counter()
            h.run_stage("solve_velocity", dt)
            h.synchronize()
            if rep:
                best = min(best, time.perf_counter() - t0)
        return best * 1e6 / iterations, int((m == 1).sum())
    finally:
        h.close()


if __name__ == "__main__":
    for shape in ((256, 128, 128), (256, 256, 256), (384, 256, 256)):
        for fill in (1.0, 0.5, 0.25):
            row = []
            for mapping, schedule in (("rows", "reference"), ("bricks_staged", "reference"), ("bricks_staged", "single_reduction")):
                us, F = run(shape, fill, mapping, schedule)
                row.append("%s/%s %.1f us" % (mapping, schedule, us))
            print("%dx%dx%d fill %.2f (F = %.1f M): " % (shape + (fill, F / 1e6)) + " | ".join(row), flush=True)
