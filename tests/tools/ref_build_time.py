"""Times the compiled reference's own TriMesh::init (serial build_bvh_recur) on the mesh of a bench workload — the figure the
GPU BVH build (tools/bvh_build_bench.py) is compared with.  Test infrastructure: uses oracle/_ref.
usage: python tests/tools/ref_build_time.py [grid]   (1120 = 2.5 M triangles)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding                 # noqa: E402
from pathtracer_amd import scenes          # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1120
mesh = scenes.blob_mesh(g, fine_detail=True)
R = binding.Ref()
R.apply_config(scenes.config_c1(16, 16, 1))
t0 = time.time()
R.add_mesh(mesh)
print(json.dumps({"grid": g, "triangles": mesh.ntri, "reference_add_mesh_s": round(time.time() - t0, 2)}))
