"""What rocprofv3's FETCH_SIZE reports per byte actually requested, for the two access patterns of this code base:
a wide streaming read (16 B per lane, consecutive) and 64-byte records gathered at random offsets (the traversal kernels'
node fetches).  MI355X_MICROARCH.md gives the streaming factor (FETCH_SIZE = 1/2 of the bytes); the gather factor resolves
the [1x, 2x] bracket profiles/hbm_traffic.json had to carry in round 1.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/fetch_calibration.py run     (on the GPU box)
    python3 tools/fetch_calibration.py report <dir>  ->  JSON with bytes requested, KB reported, factor per kernel
"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BUF = 16 << 30                 # far beyond the 256 MiB Infinity Cache
STREAM_REPEATS = 3             # + 1 warm-up launch of the same size
GATHER_RECORDS = 1 << 28       # 64-byte records per timed launch (16 GiB requested)
GATHER_REPEATS = 2             # + 1 short warm-up launch


def run():
    from pathtracer_amd import capi
    rt = capi.HostRaytracer(device=0)
    s = rt.measure_stream_read(BUF, STREAM_REPEATS)
    g = rt.measure_gather_read(BUF, GATHER_RECORDS, GATHER_REPEATS)
    print(json.dumps({"stream_gb_per_s": s, "gather_gb_per_s_of_64B_records": g}))


def report(root):
    rows = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    out = {}
    grid_threads = None
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if k not in ("k_stream_read", "k_gather_read"):
            continue
        out.setdefault(k, []).append(float(r["Counter_Value"]))
        grid_threads = int(r["Grid_Size"]) if "Grid_Size" in r else grid_threads
    res = {}
    if "k_stream_read" in out:
        v = out["k_stream_read"]
        res["stream"] = {"launches": len(v), "bytes_requested_per_launch": BUF, "fetch_size_kb_per_launch": sum(v) / len(v),
                         "reported_over_requested": sum(v) / len(v) * 1024 / BUF}
    if "k_gather_read" in out:
        v = sorted(out["k_gather_read"])[-GATHER_REPEATS:]          # the warm-up launch is the small one
        threads = grid_threads or 256 * 8 * 256
        iters = max(1, GATHER_RECORDS // threads)
        req = threads * iters * 64
        res["gather64"] = {"launches": len(v), "bytes_requested_per_launch": req, "fetch_size_kb_per_launch": sum(v) / len(v),
                           "reported_over_requested": sum(v) / len(v) * 1024 / req}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
