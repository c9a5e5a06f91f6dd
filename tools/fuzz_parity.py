"""Randomised parity sweep of the HIP rasterizer against the C oracle (GPU box).  usage: fuzz_parity.py [n_cases] [seed0]"""
import os, sys, time, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "gaussian-mesh-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import _util as U
from games_hip import synthetic as syn

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
worst = {}
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.choice([1, 7, 100, 1500, 6000, 20000]))
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 220))
    deg = int(rng.integers(0, 4))
    aa = bool(rng.integers(0, 2))
    lo = float(rng.choice([0.002, 0.01, 0.05])); hi = lo * float(rng.choice([2, 10, 40]))
    op_lo = float(rng.choice([0.01, 0.1, 0.6])); op_hi = min(0.999, op_lo + float(rng.choice([0.05, 0.4])))
    sc = syn.random_scene(P, seed=seed0 + case, scale_lo=lo, scale_hi=hi, opacity_lo=op_lo, opacity_hi=op_hi)
    cam = syn.orbit_camera(int(rng.integers(0, 8)), width=W, height=H, radius=float(rng.choice([1.5, 3.0, 6.0])))
    bg = torch.tensor(rng.uniform(0, 1, 3), dtype=torch.float32)
    kw = U.settings_kwargs(cam, bg, antialiasing=aa, sh_degree=deg, scale_modifier=float(rng.choice([1.0, 0.6, 1.8])))
    inputs = dict(means3D=sc.means3D, opacities=sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    tag = f"case {case}: P={P} {W}x{H} deg={deg} aa={aa} scale=[{lo},{hi}] op=[{op_lo},{op_hi}]"
    try:
        o = U.oracle_render(inputs, kw)
        gc = syn.upstream_grad(torch.from_numpy(o["color"])).numpy() * 1000.0
        gd = np.full((1, H, W), 1e-3, np.float32) if rng.integers(0, 2) else None
        o = U.oracle_render(inputs, kw, gc, gd)
        o64 = U.oracle_render(inputs, kw, gc, gd, precision="f64")

        def grad_bad(hg):
            """The criterion of the test-suite (tests/_util.py::grad_report): every entry within 1e-3 of the float32 oracle, or
            explained by float32 conditioning against the float64 oracle (K x the larger error of the two float32 oracle
            builds on the same row), or on a Gaussian with a pixel-level decision inside exp() rounding, or -- for Gaussians
            sharing a pixel with such a decision -- in agreement with the oracle under one of its two forced outcomes."""
            oacc = U.oracle_render(inputs, kw, gc, gd, precision="f32acc")
            rows, alt_fn = U.alt_oracles(inputs, kw, gc, gd, o["details"])
            rep = U.grad_report(hg, o["grads"], go64=o64["grads"], go32acc=oacc["grads"], excuse=U.excused_rows(o["details"]),
                                alt_rows=rows if rows.any() else None, alts=alt_fn() if rows.any() else None)
            def fails(v):       # the suite's assertions (tests/_util.py::assert_grads), same constants
                return (v["zero_violation"] or v["outliers"] > max(U.RARE_MIN, int(U.RARE_FRAC * v["size"]))
                        or v.get("excused", 0) + v.get("alt_explained", 0) > max(U.RARE_MIN, int(U.ROW_FRAC * v["size"]))
                        or not (v["q_rel"] <= U.GRAD_REL or v["size"] < U.Q_MIN_SIZE or (v["size"] < 8000 and v.get("q_rel_clean", v["q_rel"]) <= U.GRAD_REL))
                        or v["unexplained"] > int(U.UNEXPLAINED_PER_MILLION * 1e-6 * v["size"]))
            worst.update({k: max(worst.get(k, 0.0), v["worst_ratio"]) for k, v in rep.items()})
            return {k: (v["max_rel"], v["q_rel"], v["outliers"], v["unexplained"], round(v["worst_ratio"], 1), v.get("excused", 0), v.get("alt_explained", 0), v["size"])
                    for k, v in rep.items() if fails(v)}

        for rep_i in range(2):                       # twice: second call takes the capacity-hint path
            h = U.hip_render(inputs, kw, grad_color=gc, grad_invdepth=gd)
            rep = U.forward_report(h, o, W, H)
            ok = rep["radii_unexplained"] == 0 and rep["max_clean"] <= 1e-4 and rep["max_invdepth_clean"] <= 1e-4 and rep["max_amb"] <= 0.02
            gbad = grad_bad(h["grads"])
            if not ok or gbad:
                bad += 1
                print("MISMATCH", tag, "call", rep_i, {k: rep[k] for k in ("radii_unexplained", "max_clean", "max_invdepth_clean", "max_amb", "amb_frac")}, gbad, flush=True)
                break
        else:
            print("ok", tag, f"N={o['N']}", flush=True)
    except Exception:
        bad += 1
        print("ERROR", tag, flush=True); traceback.print_exc()
print("worst adjudication ratio per tensor (K = %g):" % U.ADJUDICATE_K, {k: round(v, 2) for k, v in worst.items()})
print(f"{n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
