"""Raytracer::render_image (Raytracer.cpp:1444-1531): one publish per sample index — `imagedouble` / `sample_count` hold the sums through
sample k when the caller is told about k (the GUI thread reads them while the loop runs), `stopped` ends the loop between two samples.
mipt_render pipelines the publishes (the download of sample k travels while sample k + 1 is splatted) and renders several samples per
pass; what the caller sees must not depend on either."""
import numpy as np
import pytest

from helpers import assert_bits, setup_scene
from pathtracer_amd import capi

pytestmark = pytest.mark.gpu


def progressive(rt, lookahead, cancel_after=None):
    rt.set_option("samples_per_pass", 1)
    rt.set_option("progressive_lookahead", lookahead)
    seen = []
    rc, img, cnt, calls = rt.render_progressive(on_pass=lambda done, total, i, c: seen.append((done, i.copy(), c.copy())), cancel_after=cancel_after)
    rt.set_option("samples_per_pass", 0)
    return rc, img, cnt, calls, seen


@pytest.mark.parametrize("name", ["blob32", "glass"])
def test_every_publish_holds_the_sums_through_its_sample(name):
    rt = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(rt, name)
    rc1, img1, cnt1, calls1, seen1 = progressive(rt, 1)          # one sample per pass, as the reference's loop
    assert rc1 == capi.MIPT_OK and [c[0] for c in calls1] == list(range(1, cfg.spp + 1))
    for look in (4, 3, 64, 0):            # 0: the library sizes the pass (default since round 6)
        rc, img, cnt, calls, seen = progressive(rt, look)
        assert rc == capi.MIPT_OK and [c[0] for c in calls] == list(range(1, cfg.spp + 1))
        for (d1, i1, c1), (d2, i2, c2) in zip(seen1, seen):
            assert_bits(i2, i1, f"lookahead {look}: imagedouble at the publish of sample {d1}")
            assert_bits(c2, c1, f"lookahead {look}: sample_count at the publish of sample {d1}")
        assert_bits(img, img1, "final imagedouble"); assert_bits(cnt, cnt1, "final sample_count")
    # the sums of sample k are the sums of a render of samples [0, k)
    k = 3
    rt.params.sample_begin, rt.params.sample_end = 0, k
    rt.set_option("samples_per_pass", 1)
    img_k, cnt_k = rt.render()
    rt.set_option("samples_per_pass", 0)
    rt.params.sample_begin, rt.params.sample_end = 0, 0
    assert_bits(seen1[k - 1][1], img_k, "publish k == render of the first k samples")


def test_stop_between_two_samples():
    rt = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(rt, "blob32")
    full = progressive(rt, 4)
    for stop_at in (1, 2, 5):
        rc, img, cnt, calls, seen = progressive(rt, 4, cancel_after=stop_at)
        assert rc == capi.MIPT_ERR_CANCELLED
        assert [c[0] for c in calls] == list(range(1, stop_at + 1)), "no publish after the stop"
        # the buffers are the ones of the last publish: samples rendered ahead of it are not in them
        assert_bits(img, full[4][stop_at - 1][1], f"imagedouble after a stop at sample {stop_at}")
        assert_bits(cnt, full[4][stop_at - 1][2], f"sample_count after a stop at sample {stop_at}")
    rc, img, cnt, calls, seen = progressive(rt, 4, cancel_after=cfg.spp)      # raised in the last publish: the render is complete
    assert rc == capi.MIPT_OK
    assert_bits(img, full[1], "complete render")


def test_publishes_of_the_contribution_queue_pipeline():
    """A ghost floor over a background photo (pipeline 2: getColor's contribution queue as wavefront stages, its any-hit requests on the
    order-free kernel): the publishes do not depend on the lookahead either."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden import compositing_scene
    rt = capi.HostRaytracer(device=0)
    cfg = compositing_scene(rt, "both")
    rc1, img1, cnt1, calls1, seen1 = progressive(rt, 1)
    assert rc1 == capi.MIPT_OK and [c[0] for c in calls1] == list(range(1, cfg.spp + 1)) and rt.stats()["pipeline"] == 2
    rc4, img4, cnt4, calls4, seen4 = progressive(rt, 4)
    assert rc4 == capi.MIPT_OK and [c[0] for c in calls4] == list(range(1, cfg.spp + 1))
    for (d1, i1, c1), (d2, i2, c2) in zip(seen1, seen4):
        assert_bits(i2, i1, f"imagedouble at the publish of sample {d1}")
    assert_bits(img4, img1, "final imagedouble"); assert_bits(cnt4, cnt1, "final sample_count")
