"""profiles/hbm_traffic.json from PMC summaries (tools/pmc.sh): HBM bytes per launch of each kernel.
usage: python tools/pmc_traffic.py c1=profiles/r1_g_c1_pmc_summary.txt c2=profiles/r1_g_c2_pmc_summary.txt"""
import json, re, sys
out = {}
for arg in sys.argv[1:]:
    wl, path = arg.split("=")
    cur, ks = None, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); ks[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)", line)
            if m: ks[cur][m.group(1)] = float(m.group(2))
    kernels = {}
    for k, v in ks.items():
        if "FETCH_SIZE" in v:
            kernels[k] = {"fetch_kb_per_launch": v["FETCH_SIZE"], "write_kb_per_launch": v.get("WRITE_SIZE", 0.0),
                          "hbm_bytes_per_launch_low": (v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024,
                          "hbm_bytes_per_launch_high": (2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024}
    out[wl] = {"source": path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --steps 1 --warmup 1, default samples per step: 256 at 1080p)",
               "units": "FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide reads by up to 2x (MI355X_MICROARCH.md, HBM section): read bytes are bracketed [1x, 2x]",
               "kernels": kernels}
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
for wl in out:
    for k, v in out[wl]["kernels"].items():
        print(wl, k, "%.2f .. %.2f GB per launch" % (v["hbm_bytes_per_launch_low"] / 1e9, v["hbm_bytes_per_launch_high"] / 1e9))
