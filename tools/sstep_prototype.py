"""Round-4 review, item 2 -- PROTOTYPE (numpy, f32) of an s-step form of the default PCG, s = 4 = the reference's check interval
(pressure_solver.rs:654-723 looks at convergence every 4th iteration only): would ONE launch per four iterations keep the numbers the engine
is held to?  Test infrastructure / design study only: nothing here is product code.

Three solvers on the same f32 operator (A of pressure.glsl:34-75, M^-1 = 1/d^2 -- quirk Q1 reading "zero"):
  ref   : the reference's two-reduction order,
  cg1   : the single-reduction (Chronopoulos-Gear) recurrence the library runs by default (blub_pcg1.hip.h),
  ca<s> : (ca4_cheb: f32 vectors and f32 Gram sums; ca4_cheb_f64gram: the 45 Gram sums accumulated in f64; ca4_mono: monomial basis)
          communication-avoiding CG (Carson & Demmel 2014, "CA-CG") on the symmetrically preconditioned operator Ahat = S A S, S = 1/d:
          per outer iteration ONE basis of 2s + 1 vectors [rho_0..rho_s (Ahat) p, rho_0..rho_{s-1} (Ahat) r] (monomial or Chebyshev on
          [0, lambda_max], lambda_max from a Gershgorin bound), ONE Gram matrix G = V^T V (the only global reduction: (2s+1)(2s+2)/2 = 45
          sums for s = 4 instead of 2 per iteration), then s iterations in coordinates of the basis and one update x, r, p <- V x', V r', V p'.
Measured here:
  (1) tests/golden/ref_pcg_32x64x16.npz: |p - p_reference_shaders| after k = 4, 8, 32 iterations in units of the field scale (the engine's
      fixed-k bound is 3e-4, tests/test_gpu_vs_ref.py);
  (2) a sweep of dam-break-like problems with the reference's default stopping rule (0.1 / dt on max|r|, <= 32 iterations, check every 4):
      iteration count and reported error of every solver on the same inputs;
  (3) PCG problems captured from the CPU oracle stepping a scene (both solves of every step; no feedback: every solver sees the same inputs).

usage: python tools/sstep_prototype.py [fixture] [sweep] [oracle SCENE STEPS]      (no arguments: fixture + sweep)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


class Operator:
    """A x = d x - sum over FLUID neighbours (pressure.glsl:34-75); d = number of non-SOLID neighbours, out of bounds = SOLID."""

    def __init__(self, marker):
        m = np.asarray(marker, np.int8)
        self.shape = m.shape
        self.fluid = m == 1
        pad = np.pad(m, 1, constant_values=0)
        nons = (pad != 0).astype(np.float32)
        c = (slice(1, -1),) * 3
        self.d = np.zeros(m.shape, np.float32)
        self.nb = []
        for ax in range(3):
            for s in (-1, 1):
                sl = list(c)
                sl[ax] = slice(1 + s, pad.shape[ax] - 1 + s)
                self.d += nons[tuple(sl)]
                self.nb.append((ax, s))
        self.d = np.where(self.fluid, self.d, f32(1.0)).astype(np.float32)
        self.dsafe = np.maximum(self.d, f32(1.0))

    def shift(self, x, ax, s):
        out = np.zeros_like(x)
        src = [slice(None)] * 3
        dst = [slice(None)] * 3
        if s == 1:
            src[ax], dst[ax] = slice(1, None), slice(0, -1)
        else:
            src[ax], dst[ax] = slice(0, -1), slice(1, None)
        out[tuple(dst)] = x[tuple(src)]
        return out

    def A(self, x):
        x = np.where(self.fluid, x, f32(0.0)).astype(np.float32)
        acc = np.zeros_like(x)
        for ax, s in self.nb:
            acc = (acc + self.shift(x, ax, s)).astype(np.float32)
        return np.where(self.fluid, self.d * x - acc, f32(0.0)).astype(np.float32)

    def Minv(self, r):
        return np.where(self.fluid, (r / self.dsafe) / self.dsafe, f32(0.0)).astype(np.float32)

    def gershgorin_lambda_max_hat(self):
        """bound on the spectrum of Ahat = S A S, S = 1/d: max_i (1/d_i) (1 + sum over FLUID neighbours 1/d_j)"""
        inv = np.where(self.fluid, f32(1.0) / self.dsafe, f32(0.0)).astype(np.float32)
        acc = np.zeros_like(inv)
        for ax, s in self.nb:
            acc += self.shift(inv, ax, s)
        return float((inv * (1.0 + acc))[self.fluid].max())


def dot(a, b):
    return f32(np.dot(a.ravel().astype(np.float32), b.ravel().astype(np.float32)))


def eps_div(a, b):
    return f32(a) / (f32(b) + (f32(1e-10) if b >= 0 else f32(-1e-10)))      # pressure_reduce.comp: sigma / (x +- 1e-10)


def stop_rule(i, maxit, freq):
    return i == maxit or (i > 0 and freq > 0 and i % freq == 0)


def solve_ref(op, b, p0, maxit, tol, freq, history=None):
    """SURVEY Appendix D literally (f32)."""
    p = np.where(op.fluid, p0, f32(0.0)).astype(np.float32)
    r = np.where(op.fluid, b - op.A(p), f32(0.0)).astype(np.float32)
    s = op.Minv(r)
    sigma = dot(s, r)
    for i in range(maxit + 1):
        As = op.A(s)
        alpha = eps_div(sigma, dot(s, As))
        p = (p + alpha * s).astype(np.float32)
        r = (r - alpha * As).astype(np.float32)
        if history is not None:
            history.append((i, p.copy()))
        if stop_rule(i, maxit, freq):
            e = float(np.abs(r[op.fluid]).max())
            if i == maxit or e < tol:
                return p, r, i, e
        z = op.Minv(r)
        sigma2 = dot(z, r)
        beta = eps_div(sigma2, sigma)
        sigma = sigma2
        s = (z + beta * s).astype(np.float32)


def solve_cg1(op, b, p0, maxit, tol, freq, history=None):
    """blub_pcg1.hip.h: one reduction per iteration."""
    p = np.where(op.fluid, p0, f32(0.0)).astype(np.float32)
    r = np.where(op.fluid, b - op.A(p), f32(0.0)).astype(np.float32)
    u = op.Minv(r)
    w = op.A(u)
    d = np.zeros_like(r)
    q = np.zeros_like(r)
    g_prev = a_prev = f32(0.0)
    for i in range(maxit + 1):
        gamma, delta = dot(r, u), dot(w, u)
        if i == 0:
            beta, alpha = f32(0.0), eps_div(gamma, delta)
        else:
            beta = eps_div(gamma, g_prev)
            alpha = eps_div(gamma, delta - (beta * gamma) / a_prev)
        d = (u + beta * d).astype(np.float32)
        q = (w + beta * q).astype(np.float32)
        p = (p + alpha * d).astype(np.float32)
        r = (r - alpha * q).astype(np.float32)
        if history is not None:
            history.append((i, p.copy()))
        g_prev, a_prev = gamma, alpha
        if stop_rule(i, maxit, freq):
            e = float(np.abs(r[op.fluid]).max())
            if i == maxit or e < tol:
                return p, r, i, e
        u = op.Minv(r)
        w = op.A(u)


def solve_ca(op, b, p0, maxit, tol, freq, s=4, basis="chebyshev", coord_dtype=np.float64, gram_dtype=np.float32, history=None, lam_scale=1.0):
    """CA-CG on Ahat = S A S (S = 1/d), xhat = x / S.  The check cadence is the outer iteration (freq must equal s for the reference's rule; the
    iteration count reported is that of the last completed inner iteration, like the reference's numIter).  The first outer iteration holds
    iteration 0 alone so that the blocks end where the reference looks at max|r| (i = 4, 8, ...): 1 + maxit / s outer iterations per solve."""
    S = np.where(op.fluid, f32(1.0) / op.dsafe, f32(0.0)).astype(np.float32)
    fl = op.fluid

    def Ahat(v):
        return (S * op.A(S * v)).astype(np.float32)
    p_true = np.where(fl, p0, f32(0.0)).astype(np.float32)
    r_true = np.where(fl, b - op.A(p_true), f32(0.0)).astype(np.float32)
    x = np.zeros_like(r_true)               # correction in hat coordinates: p = p0 + S x
    r = (S * r_true).astype(np.float32)
    pd = r.copy()
    lam = op.gershgorin_lambda_max_hat() * lam_scale
    theta, dl = f32(lam / 2.0), f32(lam / 2.0)
    n = 2 * s + 1
    # change of basis: Ahat V[:, j] = sum_k B[k, j] V[:, k]   (columns s and 2s have no image inside the basis: never needed)
    B = np.zeros((n, n), np.float64)
    for blk, width in ((0, s + 1), (s + 1, s)):
        for j in range(width - 1):
            c = blk + j
            if basis == "monomial":
                B[c + 1, c] = 1.0
            else:
                if j == 0:
                    B[c, c], B[c + 1, c] = theta, dl
                else:
                    B[c, c], B[c + 1, c], B[c - 1, c] = theta, dl / 2.0, dl / 2.0
    it = 0
    done = False
    while not done:
        V = []
        for v0, width in ((pd, s + 1), (r, s)):
            blk = [v0]
            for j in range(1, width):
                if basis == "monomial":
                    blk.append(Ahat(blk[-1]))
                elif j == 1:
                    blk.append(((Ahat(blk[0]) - theta * blk[0]) / dl).astype(np.float32))
                else:
                    blk.append((f32(2.0) / dl * (Ahat(blk[-1]) - theta * blk[-1]) - blk[-2]).astype(np.float32))
            V += blk
        Vm = np.stack([v[fl] for v in V], axis=1).astype(gram_dtype)
        G = (Vm.T @ Vm).astype(coord_dtype)       # the ONE global reduction of the outer iteration
        Bc = B.astype(coord_dtype)
        pc = np.zeros(n, coord_dtype); pc[0] = 1
        rc = np.zeros(n, coord_dtype); rc[s + 1] = 1
        xc = np.zeros(n, coord_dtype)
        rGr = rc @ G @ rc
        inner = 0
        for j in range(1 if it == 0 else s):      # (iteration 0 alone, then blocks of s: the blocks end at i = s, 2s, ... -- where the reference checks)
            if it > maxit:
                break
            Bp = Bc @ pc
            den = pc @ G @ Bp
            alpha = rGr / (den + (1e-10 if den >= 0 else -1e-10))
            xc = xc + alpha * pc
            rc = rc - alpha * Bp
            rGr2 = rc @ G @ rc
            beta = rGr2 / (rGr + (1e-10 if rGr >= 0 else -1e-10))
            pc = rc + beta * pc
            rGr = rGr2
            it += 1
            inner += 1
            if history is not None:
                xx = (x + sum(f32(xc[k]) * V[k] for k in range(n))).astype(np.float32)
                history.append((it - 1, (p_true + S * xx).astype(np.float32)))
        x = (x + sum(f32(xc[k]) * V[k] for k in range(n))).astype(np.float32)
        r = sum(f32(rc[k]) * V[k] for k in range(n)).astype(np.float32)
        pd = sum(f32(pc[k]) * V[k] for k in range(n)).astype(np.float32)
        i_last = it - 1
        e = float(np.abs((r / np.where(fl, S, f32(1.0)))[fl]).max())       # true residual r = rhat / S
        if i_last >= maxit or (stop_rule(i_last, maxit, freq) and e < tol) or not np.isfinite(e):
            done = True
    return (p_true + S * x).astype(np.float32), (r / np.where(fl, S, f32(1.0))).astype(np.float32), i_last, e


SOLVERS = {
    "ref": solve_ref,
    "cg1": solve_cg1,
    "ca4_cheb": lambda *a, **k: solve_ca(*a, s=4, basis="chebyshev", **k),
    "ca4_cheb_f64gram": lambda *a, **k: solve_ca(*a, s=4, basis="chebyshev", gram_dtype=np.float64, **k),
    "ca4_cheb_f32coords": lambda *a, **k: solve_ca(*a, s=4, basis="chebyshev", coord_dtype=np.float32, **k),
    "ca4_mono": lambda *a, **k: solve_ca(*a, s=4, basis="monomial", **k),
}


def fixture_check():
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_pcg_32x64x16.npz"))
    op = Operator(fx["marker"])
    fl = op.fluid
    b, p0 = fx["b"], fx["p0"]
    out = {}
    for name, fn in SOLVERS.items():
        hist = []
        fn(op, b, p0, 32, 0.0, 4, history=hist)
        row = {}
        for k in (4, 8, 32):
            ref = fx["zero_k%d_warm/p" % k]
            scale = float(np.abs(ref).max())
            got = [p for (i, p) in hist if i == k][0][fl]
            row["k%d" % k] = float(np.abs(got - ref).max() / scale)
        out[name] = row
    return {"fixture": "tests/golden/ref_pcg_32x64x16.npz (the reference's own shaders, warm start, k + 1 pressure updates)", "bound_of_the_engine_tests": 3e-4,
            "max_abs_deviation_over_field_scale": out}


def dam_problem(rng, dim=(48, 48, 48)):
    """a sloshing body of water: FLUID below a smooth random surface inside the SOLID shell, a few SOLID pillars, random smooth divergence"""
    nz, ny, nx = dim
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    h = ny * (0.3 + 0.25 * rng.random()) + 4.0 * np.sin(x * rng.uniform(0.05, 0.3) + rng.uniform(0, 6)) + 4.0 * np.cos(z * rng.uniform(0.05, 0.3) + rng.uniform(0, 6))
    m = np.where(y < h, 1, -1).astype(np.int8)
    for _ in range(rng.integers(0, 4)):
        cx, cz, rad = rng.integers(6, nx - 6), rng.integers(6, nz - 6), rng.integers(2, 5)
        m[(x - cx) ** 2 + (z - cz) ** 2 < rad * rad] = 0
    m[0], m[-1], m[:, 0], m[:, -1], m[:, :, 0], m[:, :, -1] = 0, 0, 0, 0, 0, 0
    op = Operator(m)
    amp = rng.uniform(0.1, 0.6)      # (so that the stopping rule fires between 4 and 32 iterations, like the solves of a running scene)
    b = (amp * (rng.standard_normal(dim) * 20.0 + 60.0 * np.sin(x * 0.2 + rng.uniform(0, 6)) * np.cos(y * 0.15))).astype(np.float32)
    b = np.where(op.fluid, b, 0).astype(np.float32)
    p0 = np.zeros(dim, np.float32)
    return op, b, p0


def summarise(rows):
    """rows: list of {solver: (iterations, error)}"""
    out = {}
    for name in rows[0]:
        it = np.array([r[name][0] for r in rows], np.float64)
        er = np.array([r[name][1] for r in rows], np.float64)
        ok = np.isfinite(er)
        out[name] = {"mean_iterations": float(it.mean()), "share_at_cap": float((it >= 32).mean()), "geomean_error": float(np.exp(np.log(np.maximum(er[ok], 1e-30)).mean())) if ok.any() else None,
                     "diverged": int((~ok).sum()), "same_iteration_count_as_ref": float((it == np.array([r["ref"][0] for r in rows])).mean())}
    return out


def sweep(n=60, seed=7):
    rng = np.random.default_rng(seed)
    dt = 1.0 / 120.0
    rows = []
    for _ in range(n):
        op, b, p0 = dam_problem(rng)
        # warm start like a running simulation: the previous step's pressure = the solution of a slightly different right-hand side
        pw = solve_ref(op, (b * f32(0.9)).astype(np.float32), p0, 32, 0.1 / dt, 4)[0]
        rows.append({name: fn(op, b, pw, 32, 0.1 / dt, 4)[2:4] for name, fn in SOLVERS.items()})
    return {"problems": n, "grid": [48, 48, 48], "rule": "tolerance 0.1 / dt on max|r|, <= 32 iterations, check every 4 (hybrid_fluid.rs:253-257)", "solvers": summarise(rows)}


def oracle_capture(scene="dam_halfhalf", steps=40):
    """PCG problems of a scene stepped by the CPU oracle: (marker, b, p0) in front of both solves of every step."""
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    import blub_amd
    from blub_amd import slab_scene
    cfg = blub_amd.Scene.parse(path=os.path.join(ROOT, "scenes", scene + ".json")).config
    dim, scale, gravity, cubes, maxp = slab_scene.weak_scaling_scene(cfg, 1)
    pos = slab_scene.seed_scene_particles(dim, maxp, cubes)
    o = Oracle(dim[0], dim[1], dim[2], len(pos) + 64)
    o.set_gravity_grid(gravity)
    o.set_particles(pos)
    dt = blub_amd.default_simulation_delta()
    rows = []
    order = ["transfer", "divergence", "solve_velocity", "binning", "project", "advect", "density_gather", "solve_density", "position_change", "correct"]
    for step in range(steps):
        for st in order:
            if st in ("solve_velocity", "solve_density"):
                op = Operator(o.read_volume("marker"))
                b = o.read_volume("residual").astype(np.float32)
                p0 = o.read_volume("pressure_velocity" if st == "solve_velocity" else "pressure_density").astype(np.float32)
                if op.fluid.any():
                    rows.append({name: fn(op, b, p0, 32, 0.1 / dt, 4)[2:4] for name, fn in SOLVERS.items()})
            if st == "binning" and (o.step_counter % 60) != 0:
                continue
            o.run_stage(st, dt)
        o.step_counter = o.step_counter + 1
    return {"scene": scene, "steps": steps, "solves": len(rows), "solvers": summarise(rows)}


if __name__ == "__main__":
    a = sys.argv[1:] or ["fixture", "sweep"]
    res = {}
    if "fixture" in a:
        res["fixture"] = fixture_check()
    if "sweep" in a:
        res["sweep"] = sweep()
    if "oracle" in a:
        k = a.index("oracle")
        res["oracle"] = oracle_capture(a[k + 1] if len(a) > k + 1 else "dam_halfhalf", int(a[k + 2]) if len(a) > k + 2 else 40)
    print(json.dumps(res, indent=1))
