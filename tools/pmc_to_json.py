"""profiles/r2_pmc_counters.json from PMC summaries (tools/pmc.sh) and the bench line of one of the profiled runs:
per kernel and launch the counters, and per RAY of that launch what bench.py scales to its own run (TCP line lookups,
HBM bytes).  FETCH_SIZE is corrected as tools/fetch_calibration.py measured it on this device: the traversal kernels read
by 64-byte gathers (factor `gather64`), the other kernels mostly by wide streaming accesses (factor `stream`, the 1/2 of
MI355X_MICROARCH.md).

usage: python tools/pmc_to_json.py profiles/r2_fetch_calibration.json c2=profiles/r2_c2_pmc_summary.txt:gpurun_out/pmc_c2_p1.log [c1=...]"""
import json, re, sys

cal = json.load(open(sys.argv[1]))
f_gather = cal["gather64"]["reported_over_requested"]
f_stream = cal["stream"]["reported_over_requested"]
out = {}
for arg in sys.argv[2:]:
    wl, rest = arg.split("=")
    path, benchlog = rest.split(":")
    bench = None
    for line in open(benchlog):
        if line.startswith("{"):
            bench = json.loads(line)
    ls = bench["launch_stats"]
    cur, ks = None, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); ks[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)", line)
            if m: ks[cur][m.group(1)] = float(m.group(2))
    kernels = {}
    for k, v in ks.items():
        if "FETCH_SIZE" not in v:
            continue
        trav = k.startswith("k_wf_traverse")
        factor = f_gather if trav else f_stream
        rays = {"k_wf_traverse<0>": ls["rays_closest"] / max(1, ls["extend_launches"]), "k_wf_traverse<1>": ls["rays_shadow"] / max(1, ls["shadow_launches"])}.get(k)
        e = {"fetch_size_kb_per_launch": v["FETCH_SIZE"], "write_size_kb_per_launch": v.get("WRITE_SIZE", 0.0), "fetch_size_factor": factor,
             "hbm_bytes_per_launch": v["FETCH_SIZE"] * 1024 / factor + v.get("WRITE_SIZE", 0.0) * 1024,
             "tcp_accesses_per_launch": v.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "vmem_read_instructions_per_launch": v.get("SQ_INSTS_VMEM_RD"),
             "valu_instructions_per_launch": v.get("SQ_INSTS_VALU"), "lds_instructions_per_launch": v.get("SQ_INSTS_LDS"),
             "gpu_cycles_per_launch": v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum")}
        if rays:
            e["rays_per_launch"] = rays
            e["tcp_accesses_per_ray"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays
            e["hbm_bytes_per_ray"] = e["hbm_bytes_per_launch"] / rays
            e["l1_lookups_per_cu_cycle"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * e["gpu_cycles_per_launch"])
        kernels[k] = e
    out[wl] = {"source": path + " (rocprofv3 --pmc, one counter group per pass, bench.py --steps 1 --warmup 1 --pmc: mean over the launches of both passes); "
                                "FETCH_SIZE factors from " + sys.argv[1], "kernels": kernels}
json.dump(out, open("profiles/r2_pmc_counters.json", "w"), indent=1)
for wl in out:
    for k, v in out[wl]["kernels"].items():
        print(wl, k, "HBM %.1f GB/launch" % (v["hbm_bytes_per_launch"] / 1e9), "TCP/ray %s" % (("%.1f" % v["tcp_accesses_per_ray"]) if "tcp_accesses_per_ray" in v else "-"),
              "L1 lookups/CU-cycle %s" % (("%.3f" % v["l1_lookups_per_cu_cycle"]) if "l1_lookups_per_cu_cycle" in v else "-"))
