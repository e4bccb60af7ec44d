"""Times TriMesh::init's BVH build with the host recursion and with mipt_build_bvh on the GPU, on the meshes of the
bench workloads, and checks that both give the same tree.  `python tools/bvh_build_bench.py [grid ...]`
(grid 1120 = 2.5 M triangles (configs[2]), 3444 = 23.7 M (configs[4])); the compiled reference's own build is timed by tests/tools/ref_build_time.py."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes   # noqa: E402


def main():
    grids = [int(a) for a in sys.argv[1:] if not a.startswith('-')] or [256, 1120]
    cfg = scenes.config_c1(16, 16, 1)
    for g in grids:
        mesh = scenes.blob_mesh(g, fine_detail=True)
        row = {"grid": g, "triangles": mesh.ntri}
        dumps = {}
        for mode in ("gpu", "gpu", "host"):          # first GPU call pays the code-object load; the second is the figure
            capi.set_bvh_builder(mode)
            H = capi.HostRaytracer()
            H.apply_config(cfg)
            t0 = time.time()
            obj = H.add_mesh(mesh)
            t1 = time.time()
            who, secs, dev = H.mesh_bvh_builder(obj)
            assert who == mode
            row[mode] = {"build_bvh_s": round(secs, 4), "device_s": round(dev, 4), "add_mesh_s": round(t1 - t0, 3)}
            d = H.mesh_dump(obj)
            dumps[mode] = (d["perm"], d["nodes_i"], d["nodes_bb"])
            del H
        same = all(np.array_equal(a, b) for a, b in zip(dumps["gpu"], dumps["host"]))
        row["same_tree"] = bool(same)
        row["nodes"] = int(dumps["gpu"][1].shape[0])
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
