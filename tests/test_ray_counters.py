"""The numerator of bench.py's `value`: mipt_stats.rays_closest / rays_shadow are the calls of Scene::intersection /
Scene::intersection_shadow the reference's loop makes — counted by the oracle's restatement of that loop (o_render_omp: one
count per call, oracle/pt_oracle.c) on the same scene, frame and seeds.  (VERDICT r5 weak #1b: until round 6 the claim
"counted like the oracle counts them" rested on a printed ratio, not on an assertion.)"""
import copy

import numpy as np
import pytest

from helpers import setup_scene
from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def oracle_counts(setup):
    from oracle.binding import Oracle
    O = Oracle()
    cfg = setup(O)
    O.prepare()
    t, img, cnt, rays = O.render_omp(O.cdll.o_max_threads())
    return int(rays[0]), int(rays[1]), cfg


@pytest.mark.parametrize("name", ["cornell", "blob32", "glass", "textured", "merl"])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_golden_scene_ray_counts_equal_the_oracles(name, pipeline):
    oc, osh, cfg = oracle_counts(lambda O: setup_scene(O, name)[1])
    rt = capi.HostRaytracer(device=0)
    setup_scene(rt, name)
    rt.set_option("pipeline", pipeline)
    rt.render()
    st = rt.stats()
    assert st["paths"] == cfg.W * cfg.H * cfg.spp
    assert (st["rays_closest"], st["rays_shadow"]) == (oc, osh)


@pytest.mark.parametrize("wl", ["c2", "c3"])
def test_bench_workload_ray_counts_equal_the_oracles(wl):
    """scenes.workload(wl, grid=200) at 240 x 135 x 2 spp: bench.py's scene, camera, materials and depth in small."""
    mesh, cfg, mat, text = scenes.workload(wl, 240, 135, 2, 200)

    def setup(O):
        O.apply_config(cfg)
        scenes.install(O, mesh, mat)
        return cfg
    oc, osh, _ = oracle_counts(setup)
    rt = capi.HostRaytracer(device=0)
    rt.apply_config(cfg)
    oid = rt.add_mesh(mesh)
    scenes.install_material(rt, oid, mat)
    rt.prepare()
    rt.render()
    st = rt.stats()
    assert st["pipeline"] == 1 and st["paths"] == cfg.W * cfg.H * cfg.spp
    assert (st["rays_closest"], st["rays_shadow"]) == (oc, osh)
    # the same through the device-buffer entry bench.py times, sample range by sample range (its `step`)
    import torch
    acc = torch.zeros(cfg.W * cfg.H * 4, dtype=torch.float32, device="cuda:0")
    P = rt.params
    tot = [0, 0]
    for k in range(cfg.spp):
        P.sample_begin, P.sample_end = k, k + 1
        rt.render_device(acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        s2 = rt.stats()
        tot[0] += s2["rays_closest"]; tot[1] += s2["rays_shadow"]
    assert tuple(tot) == (oc, osh)
