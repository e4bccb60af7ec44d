"""Time of TriMesh::init (host mirror) for a workload's mesh, by phase (MIPT_BUILD_TRACE=1 prints the phases of the GPU build).
usage: python tools/init_time.py [c2|c4]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MIPT_BUILD_TRACE", "1")
from pathtracer_amd import capi, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
mesh, cfg, mat, text = scenes.workload(wl, spp=4)
for rep in range(3):
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    os.environ['MIPT_CTOR_TRACE'] = '1'
    t0 = time.time(); oid = rt.add_mesh(mesh); t1 = time.time()
    who, s, dev = rt.mesh_bvh_builder(oid)
    t2 = time.time(); rt.prepare(); t3 = time.time()
    print(f"{text}: add_mesh (TriMesh::init) {t1 - t0:.3f} s [builder {who}: {s:.3f} s, device {dev:.3f} s], prepare (upload) {t3 - t2:.3f} s", flush=True)
    rt.close()
