#!/usr/bin/env python3
"""Compile the reference's MJCF scenes to flat JSON (mopa_rl_amd/scenes/*.json).

Run in the build container, where /root/reference is mounted.  The GPU box has
no asset tree; it loads the JSON.  Only numbers and names are emitted (the
kinematic tree, primitive geoms, candidate pairs) -- no XML text is copied.

    python tools/compile_scenes.py [--xml-dir /root/reference/env/assets/xml]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mopa_rl_amd.mjcf import compile_mjcf, pair_type_histogram  # noqa: E402
from mopa_rl_amd.scene import SCENE_DIR  # noqa: E402

SCENES = ["sawyer_push_obstacle", "sawyer_lift_obstacle", "sawyer_assembly_obstacle", "pusher_obstacle"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--xml-dir", default="/root/reference/env/assets/xml")
    args = ap.parse_args()
    os.makedirs(SCENE_DIR, exist_ok=True)
    for s in SCENES:
        m = compile_mjcf(os.path.join(args.xml_dir, s + ".xml"))
        out = os.path.join(SCENE_DIR, s + ".json")
        if os.path.exists(out):      # keep what later tools wrote into the scene's meta (tools/prove_separated_pairs.py)
            from mopa_rl_amd.mjcf import CompiledModel
            m.meta = {**CompiledModel.load(out).meta, **m.meta}
            if m.drop_stale_proofs():      # the MJCF changed under the proofs: they go, tools/prove_separated_pairs.py has to run again
                print(f"    {s}: geometry changed -- separation proofs dropped from the meta; rerun tools/prove_separated_pairs.py")
        m.save(out)
        print(f"{s}: nq={m.nq} bodies={len(m.body_names)} geoms={len(m.all_geom_names)} "
              f"collidable={len(m.geom_type)} pairs={len(m.pair_geom)} -> {out} ({os.path.getsize(out)} B)")
        print("   ", pair_type_histogram(m))


if __name__ == "__main__":
    main()
