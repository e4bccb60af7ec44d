"""profiles/fp64_counters.json from the summary of tools/pmc_fp64.sh: per kernel build of the generate + shade stage the floating-point wave-instructions by
precision and kind, the vector pipe's busy share and the mean active lanes; and for the stage as a whole the fp64 operations per shade vertex.
bench.py --workload c4 turns them into roofline_shade_kernel {bound: "fp64"}: the measured-BRDF tiers of configs[4] spend their time in fp64 arithmetic (glibc's
acos / atan2 / sincos restated, csrc/mipt_libm64.h), not in bytes.  Keyed on the library's source hash like profiles/pmc_counters.json.

usage: python tools/fp64_to_json.py c4=<summary>:<log with the bench line>[:<path shown as source>] [...]"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS_IN_PMC_RUN = 2      # tools/pmc_fp64.sh: bench.py --steps 1 --warmup 1 --pmc


def canon(k):
    return re.sub(r"<(\d+), (false|true)>", lambda m: "<%s>%s" % (m.group(1), "" if m.group(2) == "false" else "[depth0]"), k)


out = {}
for arg in sys.argv[1:]:
    wl, rest = arg.split("=")
    path, benchlog, *shown = rest.split(":")
    bench = [json.loads(l) for l in open(benchlog) if l.startswith("{")][-1]
    ls = bench["launch_stats"]
    cur, ks, nd = None, {}, {}
    for line in open(path):
        if not line.startswith(" "):
            cur = canon(line.strip()); ks[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)\s+\((\d+) dispatches\)", line)
            if m:
                ks[cur][m.group(1)] = float(m.group(2)); nd[cur] = int(m.group(3))
    kernels, stage_flop, stage_slot_flop = {}, 0.0, 0.0
    for k, v in ks.items():
        if not (k.startswith("k_wf_shade") or k.startswith("k_wf_generate") or k.startswith("k_wf_merl_eval")) or "SQ_INSTS_VALU" not in v:     # the stage = generate + shade tiers + (tier 5) the evaluation stage
            continue
        lanes = v["SQ_THREAD_CYCLES_VALU"] / max(1.0, v["SQ_ACTIVE_INST_VALU"])
        f64 = {x: v.get("SQ_INSTS_VALU_%s_F64" % x, 0.0) for x in ("ADD", "MUL", "FMA", "TRANS")}
        f32 = {x: v.get("SQ_INSTS_VALU_%s_F32" % x, 0.0) for x in ("ADD", "MUL", "FMA", "TRANS")}
        ops64 = f64["ADD"] + f64["MUL"] + 2 * f64["FMA"] + f64["TRANS"]              # operations per lane of a wave-instruction
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0                                                # shader cycles of the dispatch (the counter sums the 8 XCDs)
        e = {"launches_per_step": nd[k] / STEPS_IN_PMC_RUN, "valu_instructions_per_launch": v["SQ_INSTS_VALU"], "fp64_instructions_per_launch": f64, "fp32_instructions_per_launch": f32,
             "int_instructions_per_launch": v.get("SQ_INSTS_VALU_INT32", 0.0) + v.get("SQ_INSTS_VALU_INT64", 0.0),
             "fp64_share_of_vector_instructions": sum(f64.values()) / v["SQ_INSTS_VALU"], "active_lanes_per_vector_instruction": lanes,
             "fp64_flop_per_launch": ops64 * lanes, "fp64_flop_per_launch_if_all_64_lanes": ops64 * 64.0,
             "vector_instructions_per_simd_and_cycle": v["SQ_INSTS_VALU"] / (1024.0 * cyc),      # (a SIMD-32 issues a wave64 fp32 instruction in 2 cycles at best, an fp64 one in 4)
             "wait_share_of_wave_cycles": v.get("SQ_WAIT_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0)),
             "gpu_cycles_per_launch": cyc}
        kernels[k] = e
        stage_flop += e["fp64_flop_per_launch"] * e["launches_per_step"]
        stage_slot_flop += e["fp64_flop_per_launch_if_all_64_lanes"] * e["launches_per_step"]
    out[wl] = {"source": (shown[0] if shown else path) + " (rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F{64,32} ..., one counter group per pass, bench.py --steps 1 --warmup 1 --pmc)",
               "kernels": kernels,
               "stage_generate_shade": {"fp64_flop_per_step": stage_flop, "fp64_flop_per_step_if_all_64_lanes": stage_slot_flop, "vertices_per_step": ls["rays_closest"],
                                        "fp64_flop_per_vertex": stage_flop / max(1, ls["rays_closest"]), "fp64_issue_slot_flop_per_vertex": stage_slot_flop / max(1, ls["rays_closest"]),
                                        "note": "fp64 operations = (ADD + MUL + 2 FMA + TRANS wave-instructions) x mean active lanes, summed over the builds of the stage with each build's launches per step; vertices = closest-hit rays of the profiled step"}}
import __graft_entry__ as ge
out["_build"] = {"git_commit": os.environ.get("MIPT_GIT_COMMIT") or subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
                 "source_sha256_16": ge.source_hash()}
json.dump(out, open(os.path.join(ROOT, "profiles", "fp64_counters.json"), "w"), indent=1)
for wl in out:
    if wl.startswith("_"): continue
    for k, v in out[wl]["kernels"].items():
        print(wl, "%-24s fp64 share %.2f  lanes %.1f  vector instructions per SIMD-cycle %.3f  wait %.2f  fp64 Gflop/launch %.0f" % (k, v["fp64_share_of_vector_instructions"], v["active_lanes_per_vector_instruction"], v["vector_instructions_per_simd_and_cycle"], v["wait_share_of_wave_cycles"], v["fp64_flop_per_launch"] / 1e9))
    print(wl, "fp64 flop per shade vertex: %.0f" % out[wl]["stage_generate_shade"]["fp64_flop_per_vertex"])
