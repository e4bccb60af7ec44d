"""profiles/pmc_counters.json from PMC summaries (tools/pmc.sh) and the bench line of one of the profiled runs: per kernel and
launch the counters, and per RAY of that launch what bench.py scales to its own run (L2 misses = fabric line fetches, TCP line
lookups, HBM bytes, instructions by kind, mean vector-memory latency, wait share).  FETCH_SIZE is corrected as
tools/fetch_calibration.py measured it on this device: the traversal kernels read by 64-byte gathers (factor `gather64`), the
other kernels mostly by wide streaming accesses (factor `stream`, the 1/2 of MI355X_MICROARCH.md).  The file records the git
commit and the hash of the SOURCES + compiler flags of the library that was profiled (__graft_entry__.source_hash(); the binary is
not bit-reproducible): bench.py marks what it derives from this file as stale when it runs another build.

usage: python tools/pmc_to_json.py profiles/r2_fetch_calibration.json c2=<summary>:<log with the bench line>[:<path shown as source>] [c1=...]
(on the GPU box there is no .git: MIPT_GIT_COMMIT names the commit of the snapshot)"""
import hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = json.load(open(sys.argv[1]))
f_gather = cal["gather64"]["reported_over_requested"]
f_stream = cal["stream"]["reported_over_requested"]


def canon(k):          # "k_wf_traverse<0, false>" -> "k_wf_traverse<0>"; "k_wf_shade<1, true>" -> "k_wf_shade<1>[depth0]"
    # the second template argument of k_wf_shade / k_wf_traverse-era kernels is INITIAL (csrc/mipt_wavefront.h: the build that runs at
    # depth 0 only, where a path's starting state is recomputed instead of fetched) — NOT the "[quad]" of a round-3 experiment, which
    # this function called it until round 6
    return re.sub(r"<(\d+), (false|true)>", lambda m: "<%s>%s" % (m.group(1), "" if m.group(2) == "false" else "[depth0]"), k)


STEPS_IN_PMC_RUN = 2   # tools/pmc.sh: bench.py --steps 1 --warmup 1 --pmc, every step profiled


out = {}
for arg in sys.argv[2:]:
    wl, rest = arg.split("=")
    path, benchlog, *shown = rest.split(":")          # optional third field: the path the summary is committed under
    shown = shown[0] if shown else path
    bench = None
    for line in open(benchlog):
        if line.startswith("{"):
            bench = json.loads(line)
    ls = bench["launch_stats"]
    cur, ks, nd = None, {}, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = canon(line.strip()); ks[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)(?:\s+\((\d+) dispatches\))?", line)
            if m:
                ks[cur][m.group(1)] = float(m.group(2))
                if m.group(3) and m.group(1) == "FETCH_SIZE": nd[cur] = int(m.group(3))
    kernels = {}
    for k, v in ks.items():
        if "FETCH_SIZE" not in v:
            continue
        trav = k.startswith("k_wf_traverse") or k.startswith("k_wf_anyhit")
        factor = f_gather if trav else f_stream
        rays = {"k_wf_traverse<0>": ls["rays_closest"] / max(1, ls["extend_launches"]), "k_wf_traverse<1>": ls["rays_shadow"] / max(1, ls["shadow_launches"]), "k_wf_anyhit": ls["rays_shadow"] / max(1, ls["shadow_launches"])}.get(k)
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = {"fetch_size_kb_per_launch": v["FETCH_SIZE"], "write_size_kb_per_launch": v.get("WRITE_SIZE", 0.0), "fetch_size_factor": factor,
             "hbm_bytes_per_launch": v["FETCH_SIZE"] * 1024 / factor + v.get("WRITE_SIZE", 0.0) * 1024,
             "tcp_accesses_per_launch": v.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "vmem_read_instructions_per_launch": v.get("SQ_INSTS_VMEM_RD"),
             "valu_instructions_per_launch": v.get("SQ_INSTS_VALU"), "salu_instructions_per_launch": v.get("SQ_INSTS_SALU"), "lds_instructions_per_launch": v.get("SQ_INSTS_LDS"),
             "gpu_cycles_per_launch": cyc, "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum"), "waves": v.get("SQ_WAVES"),
             "wait_share_of_wave_cycles": v.get("SQ_WAIT_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0)),
             "mean_vmem_latency_cycles": v.get("TCP_TCP_LATENCY_sum", 0.0) / max(1.0, v.get("TCP_TA_TCP_STATE_READ_sum", 1.0)),
             # mean active lanes of a vector instruction (thread-cycles / instruction-cycles): what bench.py prices a vector-memory instruction at
             "active_lanes_per_vector_instruction": v.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, v.get("SQ_ACTIVE_INST_VALU", 1.0))}
        if k in nd:    # how often a step launches THIS build (bench.py charges every build with its own count, not with the stage's)
            e["launches_per_step"] = nd[k] / STEPS_IN_PMC_RUN
        if rays:
            e["rays_per_launch"] = rays
            e["tcp_accesses_per_ray"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays
            e["hbm_bytes_per_ray"] = e["hbm_bytes_per_launch"] / rays
            e["l2_misses_per_ray"] = v.get("TCC_MISS_sum", 0.0) / rays
            e["valu_per_ray"] = v.get("SQ_INSTS_VALU", 0.0) / rays
            e["salu_per_ray"] = v.get("SQ_INSTS_SALU", 0.0) / rays
            e["vmem_per_ray"] = v.get("SQ_INSTS_VMEM_RD", 0.0) / rays
            e["l1_lookups_per_cu_cycle"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * cyc)
        kernels[k] = e
    # the generate + shade stage as a whole: every build charged with its own launches per step, per shade vertex (= closest-hit ray) of the profiled step
    stage = sum(v["hbm_bytes_per_launch"] * v.get("launches_per_step", 0.0) for k, v in kernels.items() if k.startswith("k_wf_shade") or k.startswith("k_wf_generate") or k.startswith("k_wf_merl_eval"))
    stage_obj = {"hbm_bytes_per_step": stage, "vertices_per_step": ls["rays_closest"], "hbm_bytes_per_vertex": stage / max(1, ls["rays_closest"]),
                 "note": "sum over k_wf_generate, every k_wf_shade<tier>[depth0 or not] build and (measured-BRDF scenes) k_wf_merl_eval of HBM bytes per launch x launches per step of that build; vertices = closest-hit rays of the profiled step"}
    out[wl] = {"stage_generate_shade": stage_obj, "source": shown + " (rocprofv3 --pmc, one counter group per pass, bench.py --steps 1 --warmup 1 --pmc: mean over the launches of both passes); "
                                "FETCH_SIZE factors from " + sys.argv[1], "kernels": kernels}
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
out["_build"] = {"git_commit": os.environ.get("MIPT_GIT_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
                 "source_sha256_16": ge.source_hash(),
                 "flags": " ".join(ge.HIPCC_FLAGS) + " (default library)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_counters.json"), "w"), indent=1)
for wl in out:
    if wl.startswith("_"):
        continue
    for k, v in out[wl]["kernels"].items():
        print(wl, k, "HBM %.1f GB/launch" % (v["hbm_bytes_per_launch"] / 1e9), "TCP/ray %s" % (("%.1f" % v["tcp_accesses_per_ray"]) if "tcp_accesses_per_ray" in v else "-"),
              "L2 misses/ray %s" % (("%.2f" % v["l2_misses_per_ray"]) if "l2_misses_per_ray" in v else "-"), "VMEM latency %.0f cycles, wait %.2f" % (v["mean_vmem_latency_cycles"], v["wait_share_of_wave_cycles"]))
