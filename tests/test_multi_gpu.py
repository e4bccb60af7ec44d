"""Several GPUs, both ways the product offers, exercised on ONE device so that the tests run on a one-GPU box:

* inside the library — mipt_create(device_ids, n > 1): worker threads, per-device streams and partial framebuffers, the
  reduce into member 0 (RCCL between distinct devices; device copies + adds when a device is listed twice, as here; the
  RCCL entry points themselves are checked by a single-rank self test);
* one process per GPU — world size 2 over gloo, each rank rendering its tiles with the real HIP path
  (mipt_render_device) and ONE all-reduce of the partial framebuffers, as bench.py does over RCCL.

Both must reproduce the single-device frame up to the order of the float additions (1e-5 relative to white).

On a box with two or more devices the tests at the end enable themselves: the group [0, 1] must sum with RCCL's ncclReduce,
and the one-process-per-GPU form runs over backend nccl (= RCCL) on two distinct devices."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import WHITE, load_golden, setup_scene, spawn_ranks
from pathtracer_amd import capi, scenes

pytestmark = pytest.mark.gpu


def normalised(img, cnt):
    return img / np.maximum(cnt, 1e-30)[..., None] / WHITE


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_group_context_matches_single_device(devices):
    g = load_golden("scene_blob32.npz")
    one = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(one, "blob32")
    img1, cnt1 = one.render()
    grp = capi.HostRaytracer(device=devices)
    setup_scene(grp, "blob32")
    assert grp.group_size() == len(devices) and one.group_size() == 1
    assert grp.group_reduce_kind().startswith("copy reduce") and one.group_reduce_kind() == ""
    img, cnt = grp.render()
    st, st1 = grp.stats(), one.stats()
    assert st["paths"] == st1["paths"] == cfg.W * cfg.H * cfg.spp
    assert st["rays_closest"] == st1["rays_closest"] and st["rays_shadow"] == st1["rays_shadow"]      # the same samples, dealt to the members
    assert np.abs(normalised(img, cnt) - normalised(img1, cnt1)).max() < 1e-5
    assert np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max() < 1e-5
    np.testing.assert_allclose(cnt, g["count"], rtol=1e-5)
    # a second render on the same group (buffers reused), through the reference's entry point of the host mirror
    a, _, u8a = grp.render_image_nopreviz()
    b, _, u8b = one.render_image_nopreviz()
    assert np.abs(a - b).max() / WHITE < 1e-5 and np.abs(u8a.astype(int) - u8b.astype(int)).max() <= 1
    # the caller's own partition is refined by the group's: two "processes" of two devices each = four ranks
    acc_i, acc_c = np.zeros_like(img), np.zeros_like(cnt)
    for rank in range(2):
        r = capi.HostRaytracer(device=[0, 0])
        r.set_partition(16, rank, 2)
        setup_scene(r, "blob32")
        i2, c2 = r.render()
        acc_i += i2; acc_c += c2
    assert np.abs(normalised(acc_i, acc_c) - normalised(img1, cnt1)).max() < 1e-5


def test_group_progress_and_cancel():
    rt = capi.HostRaytracer(device=[0, 0])
    mesh, cfg, oid = setup_scene(rt, "blob32")
    slots = cfg.W * cfg.H // 2
    rt.set_option("paths_per_pass", 2 * slots)                       # about two samples per chunk and device
    rc, img, cnt, calls = rt.render_progressive()
    assert rc == capi.MIPT_OK and len(calls) > 1 and calls[-1][0] == cfg.spp
    sums = [c[2] for c in calls]
    assert all(b > a for a, b in zip(sums, sums[1:]))                # the caller's buffers hold the complete sums at every call
    g = load_golden("scene_blob32.npz")
    assert np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max() < 1e-4
    rc2, img2, cnt2, calls2 = rt.render_progressive(cancel_after=1)
    assert rc2 == capi.MIPT_ERR_CANCELLED and len(calls2) == 1
    done = calls2[0][0]
    rt.params.sample_begin, rt.params.sample_end = 0, done
    img_ref, cnt_ref = rt.render()
    np.testing.assert_allclose(cnt2, cnt_ref, rtol=1e-6)
    assert np.abs(normalised(img2, cnt2) - normalised(img_ref, cnt_ref)).max() < 1e-6
    with pytest.raises(capi.MiptError, match="RCCL"):
        rt.set_option("reduce", 1)                                   # a device listed twice has no communicator


def test_group_chunks_without_a_callback_do_not_race():
    """A cancel flag without a progress callback cuts the range into chunks that are NOT separated by a host
    synchronisation: member i's stream must not clear its partial framebuffer for chunk k + 1 while member 0's stream still
    copies chunk k out of it (copy reduce).  Same chunks with the callback (host-synchronised) = the same bits."""
    rt = capi.HostRaytracer(device=[0, 0, 0])
    mesh, cfg, oid = setup_scene(rt, "blob32")
    slots = cfg.W * cfg.H // 3
    rt.set_option("paths_per_pass", max(64, slots))                  # one sample per chunk and device: cfg.spp chunks
    assert cfg.spp >= 3
    rc, img_cb, cnt_cb, calls = rt.render_progressive()
    assert rc == capi.MIPT_OK and len(calls) >= 3
    for _ in range(3):
        img, cnt = rt.render_cancellable()
        assert (img.view(np.uint32) == img_cb.view(np.uint32)).all() and (cnt.view(np.uint32) == cnt_cb.view(np.uint32)).all()
    img1, cnt1 = rt.render()                                         # the whole range at once: other addition order
    np.testing.assert_allclose(cnt, cnt1, rtol=1e-6)
    assert np.abs(normalised(img, cnt) - normalised(img1, cnt1)).max() < 1e-6


def test_rccl_is_loadable_and_callable():
    """dlopen(librccl.so.1), ncclCommInitAll, ncclGroupStart / ncclReduce / ncclGroupEnd, ncclCommDestroy with the
    signatures the group path uses, on one rank."""
    rt = capi.HostRaytracer(device=0)
    rt.rccl_selftest()


def _rank(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt = capi.HostRaytracer(device=0)                                # --share-gpu: both ranks on GPU 0
    rt.set_partition(16, rank, world)
    setup_scene(rt, "blob32")
    accum = torch.zeros(rt.W * rt.H * 4, dtype=torch.float32, device="cuda:0")
    rt.render_device(accum.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = accum.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM)                      # the framebuffer reduce (bench.py: RCCL on the device buffer)
    if rank == 0:
        np.save(out, host.numpy())
    dist.destroy_process_group()


def test_two_processes_reduce_real_partial_frames(tmp_path):
    out = str(tmp_path / "sum.npy")
    spawn_ranks(_rank, 2, out)
    one = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(one, "blob32")
    img1, cnt1 = one.render()
    acc = np.load(out)
    npx = cfg.W * cfg.H
    img, cnt = acc[: 3 * npx].reshape(cfg.H, cfg.W, 3), acc[3 * npx:].reshape(cfg.H, cfg.W)
    np.testing.assert_allclose(cnt, cnt1, rtol=1e-5)
    assert np.abs(normalised(img, cnt) - normalised(img1, cnt1)).max() < 1e-5


# ---- two or more devices: skipped on a one-GPU box, enabled by themselves elsewhere ---------------------------------------

def _n_devices():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="needs two devices")


@needs_two
@pytest.mark.parametrize("n", [2, 4, 8])
def test_group_of_distinct_devices_reduces_with_rccl(n):
    if _n_devices() < n:
        pytest.skip(f"needs {n} devices")
    g = load_golden("scene_blob32.npz")
    one = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(one, "blob32")
    img1, cnt1 = one.render()
    grp = capi.HostRaytracer(device=list(range(n)))
    setup_scene(grp, "blob32")
    assert grp.group_size() == n
    assert grp.group_reduce_kind() == "RCCL ncclReduce(sum, fp32, root 0)"
    grp.set_option("reduce", 1)                                      # RCCL or fail
    for _ in range(2):                                               # the second render reuses buffers and communicators
        img, cnt = grp.render()
        st, st1 = grp.stats(), one.stats()
        assert st["rays_closest"] == st1["rays_closest"] and st["rays_shadow"] == st1["rays_shadow"]
        np.testing.assert_allclose(cnt, cnt1, rtol=1e-5)
        assert np.abs(normalised(img, cnt) - normalised(img1, cnt1)).max() < 1e-5
        assert np.abs(normalised(img, cnt) - normalised(g["image"], g["count"])).max() < 1e-5
    img_c, cnt_c = grp.render_cancellable()                          # chunked, a reduce per chunk, no host synchronisation
    assert np.abs(normalised(img_c, cnt_c) - normalised(img1, cnt1)).max() < 1e-5
    grp.set_option("reduce", 2)                                      # peer copies + adds between distinct devices
    img2, cnt2 = grp.render()
    assert np.abs(normalised(img2, cnt2) - normalised(img1, cnt1)).max() < 1e-5


def _rank_nccl(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    rt = capi.HostRaytracer(device=rank)
    rt.set_partition(16, rank, world)
    setup_scene(rt, "blob32")
    accum = torch.zeros(rt.W * rt.H * 4, dtype=torch.float32, device=f"cuda:{rank}")
    rt.render_device(accum.data_ptr(), torch.cuda.current_stream().cuda_stream)
    dist.all_reduce(accum, op=dist.ReduceOp.SUM)                     # RCCL, on the device buffers, as bench.py does
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out, accum.cpu().numpy())
    dist.destroy_process_group()


@needs_two
def test_two_processes_on_two_devices_reduce_over_rccl(tmp_path):
    out = str(tmp_path / "sum.npy")
    spawn_ranks(_rank_nccl, 2, out)
    one = capi.HostRaytracer(device=0)
    mesh, cfg, oid = setup_scene(one, "blob32")
    img1, cnt1 = one.render()
    acc = np.load(out)
    npx = cfg.W * cfg.H
    img, cnt = acc[: 3 * npx].reshape(cfg.H, cfg.W, 3), acc[3 * npx:].reshape(cfg.H, cfg.W)
    np.testing.assert_allclose(cnt, cnt1, rtol=1e-5)
    assert np.abs(normalised(img, cnt) - normalised(img1, cnt1)).max() < 1e-5


def test_bench_refuses_to_measure_fewer_gpus_than_asked_for():
    """`python bench.py --gpus N` without the launcher must use N devices (the in-library group) or exit non-zero."""
    n = _n_devices()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to measure fewer GPUs" in (r.stderr + r.stdout)
