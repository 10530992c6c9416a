"""Writes the scene JSON files used by tests / bench (schema: src/scene/mod.rs:19-43 of the reference).

The first four restate the parameter values of the reference's shipped scenes of the same name (BASELINE.json configs);
corner_dams_{128,256,512} are the synthetic family of SURVEY.md 8(d) ("1M particles @ 256^3", "8M @ 512^3").
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def scene(dim, scale, max_particles, cubes, gravity=(0.0, -9.81, 0.0), static_objects=None):
    v3 = lambda t: {"x": t[0], "y": t[1], "z": t[2]}
    s = {"gravity": v3(gravity),
         "fluid": {"world_position": v3((0.0, 0.0, 0.0)), "max_num_particles": max_particles, "grid_to_world_scale": scale,
                   "grid_dimension": v3(dim), "fluid_cubes": [{"min": v3(a), "max": v3(b)} for a, b in cubes]}}
    if static_objects:
        s["static_objects"] = static_objects
    return s


def moving_cube(position, target, scale, duration):
    """StaticObjectConfig (scene/models.rs:11-46) of a translating scenes/models/unit_cube.obj."""
    v3 = lambda t: {"x": t[0], "y": t[1], "z": t[2]}
    return {"model": "unit_cube.obj", "world_position": v3(position), "scale": scale, "rotation_angles": v3((0.0, 0.0, 0.0)),
            "animation": {"translation": {"target": v3(target), "curve": "Linear", "duration": duration}}}


CORNERS = [((0.0, 0.0, 0.0), (0.16, 0.32, 0.16)), ((1.12, 0.0, 1.12), (1.28, 0.32, 1.28))]
SCENES = {
    "single_cell_debug": scene((64, 64, 128), 0.01, 1238328, [((0.319, 0.319, 0.639), (0.32, 0.32, 0.64))]),
    "dam_halfhalf": scene((128, 64, 64), 0.01, 1238328, [((0.0, 0.0, 0.0), (0.64, 0.4, 0.64))]),
    "double_dam": scene((128, 64, 64), 0.01, 2000000, [((0.0, 0.0, 0.0), (0.32, 0.4, 0.64)), ((0.96, 0.0, 0.0), (1.28, 0.4, 0.64))]),
    "dam_halfhalf_highres": scene((256, 128, 128), 0.005, 10193528, [((0.0, 0.0, 0.0), (0.64, 0.4, 0.64))]),
    "corner_dams_128": scene((128, 128, 128), 0.01, 111600 + 64, CORNERS),
    "corner_dams_256": scene((256, 256, 256), 0.005, 968688 + 64, CORNERS),
    "corner_dams_512": scene((512, 512, 512), 0.0025, 8065008 + 64, CORNERS),
    # the reference's scenes/wavegenerator.json with this repository's unit cube (its models/cube.obj is a git-lfs pointer without
    # data): a 0.64 m block enters the domain through the +x wall and retreats, period 1.6 s
    "wavegenerator_cube": scene((128, 64, 64), 0.01, 1238328, [((0.0, 0.0, 0.0), (0.64, 0.4, 0.64))],
                                static_objects=[moving_cube((1.6, 0.32, 0.32), (1.32, 0.32, 0.32), 0.64, 0.8)]),
}

if __name__ == "__main__":
    for name, s in SCENES.items():
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(s, f, separators=(",", ":"))
            f.write("\n")
    print("wrote", len(SCENES), "scenes")
