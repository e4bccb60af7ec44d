"""What the ranks of an N-rank tile partition cost, measured on ONE GPU (a step of the workload: 1080p x 1024 spp; configs[4]: 4K x 256 spp).

For N = 1, 2, 4, 8 EVERY rank r = 0 .. N-1 renders its share of the frame, one after the other on this GPU (same scene, same options, the
partition of mipt_render_params: tiles of tile_size x tile_size pixels, tile t -> rank mipt_tile_owner(t)).  A strong-scaling step on N GPUs
lasts max_r t_r + the reduce, so
    predicted_speedup(N) = t_1 / (max_r t_r + t_reduce(N)),
with t_reduce estimated from the partial frame buffer (W x H x 4 floats) over xGMI (see REDUCE_GBPS below) — the part that cannot be
measured without N devices.  Also printed: mean_r t_r, max / mean (the load balance of the tile deal), rays per rank, and N x mean / t_1
(what the smaller launches of a rank cost).

usage: python tools/rank_probe.py [c2|c3|c4] [tile_size=32] [ranks=1,2,4,8] [option=value ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import capi, scenes

# one ncclReduce of W*H*4 floats to rank 0 over xGMI: ring reduce, per-link bound (7 links x ~153 GB/s per GPU, MI355X_MICROARCH.md); a
# ring moves (N-1)/N of the buffer over the slowest link at ~60 % of its peak in practice -> ~90 GB/s effective; plus launch latency
REDUCE_GBPS = 90.0
REDUCE_LATENCY_MS = 0.05

wl = "c2"
tile = 32
ranks = [1, 2, 4, 8]
opts = []
for a in sys.argv[1:]:
    if a in ("c1", "c2", "c3", "c4"):
        wl = a
    elif a.startswith("tile_size="):
        tile = int(a.split("=")[1])
    elif a.startswith("ranks="):
        ranks = [int(v) for v in a.split("=")[1].split(",")]
    elif "=" in a:
        opts.append(a.split("="))
dims = (3840, 2160, 256) if wl == "c4" else (1920, 1080, 1024)
mesh, cfg, mat, text = scenes.workload(wl, dims[0], dims[1], dims[2], None)
H = capi.HostRaytracer(device=0)
H.apply_config(cfg); scenes.install(H, mesh, mat); H.prepare()
for k, v in opts:
    H.set_option(k, int(v))
t1 = None
for nr in ranks:
    per = []
    for r in range(nr):
        pr = H.params
        pr.tile_size, pr.tile_rank, pr.tile_nranks = tile, r, nr
        if r == 0:
            H.render()          # warm-up of this partition's buffers
        H.render()
        st = H.stats()
        per.append({"rank": r, "render_ms": round(st["render_ms"], 2), "extend": round(st["traverse_ms"], 1), "shadow": round(st["shadow_ms"], 1), "shade": round(st["shade_ms"], 1),
                    "resolve": round(st["resolve_ms"], 1), "rays": st["rays_closest"] + st["rays_shadow"]})
    ts = [p["render_ms"] for p in per]
    if nr == 1:
        t1 = ts[0]
    def reduce_at(gbps):
        return 0.0 if nr == 1 else REDUCE_LATENCY_MS + (dims[0] * dims[1] * 16 * (nr - 1) / nr) / (gbps * 1e6)
    reduce_ms = reduce_at(REDUCE_GBPS)
    out = {"workload": wl, "tile_size": tile, "nranks": nr, "t_max_ms": max(ts), "t_mean_ms": round(sum(ts) / nr, 2), "max_over_mean": round(max(ts) / (sum(ts) / nr), 4),
           "rays_max_over_mean": round(max(p["rays"] for p in per) / (sum(p["rays"] for p in per) / nr), 4),
           "n_x_mean_over_t1": round(nr * (sum(ts) / nr) / t1, 4) if t1 else None, "reduce_estimate_ms": round(reduce_ms, 3),
           "predicted_speedup": round(t1 / (max(ts) + reduce_ms), 3) if t1 else None,
           # the reduce is an ASSUMPTION (never measured: no multi-GPU box in this pool; mipt_rccl_selftest has only run on one device) — its constants, and
           # what the prediction becomes at half and at 5/3 of the assumed rate
           "reduce_assumption": {"effective_gb_per_s": REDUCE_GBPS, "latency_ms": REDUCE_LATENCY_MS, "bytes": dims[0] * dims[1] * 16, "measured": False},
           "predicted_speedup_at_45_and_150_gb_per_s": [round(t1 / (max(ts) + reduce_at(45.0)), 3), round(t1 / (max(ts) + reduce_at(150.0)), 3)] if t1 else None,
           "ranks": per}
    print(json.dumps(out), flush=True)
