"""P2G transfer in isolation on a scene state: step `--steps` times, then time the stand-alone transfer stage per kernel class (events inside the
dispatches) for each `--tune` set.  The particles do not move between the timings, so ablations and list forms see the SAME state.
usage: python tools/p2g_probe.py [--scene corner_dams_256] [--steps 40] [--reps 5] [--tunes "p2g_compact=0,p2g_own=1;p2g_compact=0,p2g_own=0;p2g_compact=1"]
(round 6 used it with the tunings of the run-list experiments, which were removed with them: profiles/r06_run_lists_rejected.txt)"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import blub_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="corner_dams_256")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tunes", default="p2g_compact=0,p2g_own=1;p2g_compact=0,p2g_own=0;p2g_compact=1")
args = ap.parse_args()
dt = blub_amd.default_simulation_delta()
sc = blub_amd.Scene(path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scenes", args.scene + ".json"))
f = sc.fluid()
for _ in range(args.steps):
    sc.step(dt)
f.synchronize()
out = {"scene": args.scene, "steps": args.steps, "lib": os.environ.get("BLUBHIP_LIB", "default"), "runs": []}
for tune in args.tunes.split(";"):
    for kv in tune.split(","):
        k, v = kv.split("=")
        f.set_tuning(k, int(v))
    f.run_stage("transfer", dt)   # warm
    f.synchronize()
    f.profile_enable(True)
    f.profile_reset()
    for _ in range(args.reps):
        f.run_stage("transfer", dt)
    f.synchronize()
    prof = f.profile_read()
    f.profile_enable(False)
    row = {"tune": tune}
    for k in ("build_lists", "gather_velocity", "reset_bricks", "brick_lists", "copy"):
        if k in prof:
            row[k] = round(prof[k]["total_ms"] / args.reps * 1e3, 1)
    out["runs"].append(row)
    for k, v in (("p2g_compact", -1), ("p2g_own", 1)):      # back to the defaults
        f.set_tuning(k, v)
print(json.dumps(out))
