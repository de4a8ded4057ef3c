#!/usr/bin/env python3
"""Is the product kernel's vector-ALU issue capacity in use?  From hardware counters alone.

Inputs: two rocprofv3 --pmc passes collected by tools/profile_round.sh with the same counter set
(SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
GRBM_GUI_ACTIVE): <prof>/valu (bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras) and <prof>/valu_ref
(tools/valu_busy: a pure v_xor_b32 loop, a pure v_bcnt_u32_b32 loop, the inner-loop mix without memory traffic).

What the counters say on gfx950 (checked against the reference loops' known instruction counts): SQ_ACTIVE_INST_VALU
advances by ONE per wave64 VALU instruction whatever its class (it equals SQ_INSTS_VALU), so by itself it is an
instruction count, not a busy time.  Busy time comes from the two pure loops: they are issue-bound by construction
(8 waves per SIMD, independent registers, nothing but that one instruction), so
    SIMD-cycles per instruction of a class = (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) / SQ_ACTIVE_INST_VALU
is that class's issue cost, measured by counters.  The product kernel's VALU-busy fraction is then
    (n_xor x cost_xor + n_other x cost_bcnt) / (SIMD-cycles the kernel ran),
n_xor = 8 per 64 distances (the XORs), n_other = every other VALU instruction (popcounts, min3, shifts, compares: the
4-cycle class of profiles/r01_valu_class.txt).
    valu_busy_summary.py <prof_dir> <out_json> [distances per step, default cfg2's]"""
import collections
import csv
import json
import os
import sys

prof, out_path = sys.argv[1], sys.argv[2]
N_SIMD = 1024


def load(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(os.path.join(prof, sub, "p_counter_collection.csv"))):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[r["Kernel_Name"]] += 1
    return agg, n


def derived(c):
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD          # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs
    return {"valu_instructions": c["SQ_ACTIVE_INST_VALU"], "simd_cycles": simd_cycles,
            "simd_cycles_per_valu_instruction": simd_cycles / max(c["SQ_ACTIVE_INST_VALU"], 1.0),
            "active_lanes_per_valu_instruction": c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1.0),
            "wave_cycles_waiting_to_issue_frac": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
            "wave_cycles_parked_on_waitcnt_frac": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
            "wave_cycles_issuing_frac": c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
            "raw": dict(c)}


ref, _ = load("valu_ref")
prod, launches = load("valu")
refs = {}
for name, c in ref.items():
    for key in ("k_pure_xor", "k_pure_bcnt", "k_pair_mix", "k_prio_mix"):
        if key in name:
            refs[key] = derived(c)
cost_xor = refs["k_pure_xor"]["simd_cycles_per_valu_instruction"]
cost_bcnt = refs["k_pure_bcnt"]["simd_cycles_per_valu_instruction"]
out = {"source": "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras and of tools/valu_busy "
                 "(tools/profile_round.sh), summarised by tools/valu_busy_summary.py",
       "reference_loops": refs,
       "issue_cost_simd_cycles": {"v_xor_b32 (pure loop)": cost_xor, "v_bcnt_u32_b32 (pure loop)": cost_bcnt,
                                  "ratio": cost_bcnt / cost_xor},
       "kernels": {}}
if "k_pair_mix" in refs:
    # the inner-loop body alone (16 xor + 17 four-cycle-class instructions per two distances): what the two class costs predict
    pred = (16 * cost_xor + 17 * cost_bcnt) / 33.0
    refs["k_pair_mix"]["predicted_simd_cycles_per_valu_instruction"] = pred
    refs["k_pair_mix"]["valu_busy_frac"] = pred / refs["k_pair_mix"]["simd_cycles_per_valu_instruction"]
if "k_prio_mix" in refs:
    refs["k_prio_mix"]["serial_issue_cost_over_cycles"] = (16 * cost_xor + 17 * cost_bcnt) / 33.0 / refs["k_prio_mix"]["simd_cycles_per_valu_instruction"]
DIST_PER_STEP = float(sys.argv[3]) if len(sys.argv) > 3 else 1.88374e12      # cfg2
for name, c in prod.items():
    if "k_score_rowlane" not in name:
        continue
    d = derived(c)
    steps = launches[name] / 9.0                             # cfg2 at 1 GiB of scratch: 9 chunk launches per step
    rows64 = DIST_PER_STEP / 64.0 * steps
    n_xor = 8.0 * rows64
    n_other = d["valu_instructions"] - n_xor
    busy = n_xor * cost_xor + n_other * cost_bcnt
    d.update({"launches": launches[name], "steps": steps, "valu_instructions_per_64_distances": d["valu_instructions"] / rows64,
              "valu_busy_frac": busy / d["simd_cycles"],
              "valu_busy_model": "SERIAL-issue pricing: 8 v_xor_b32 per 64 distances at the pure-xor cost + every other VALU instruction at "
                                 "the pure-bcnt cost, over the SIMD-cycles of the kernel (GRBM_GUI_ACTIVE / 8 x 1024).  1.00 = the "
                                 "instructions took as long as if each had the SIMD to itself (rounds 1-2); above 1 = half-rate "
                                 "instructions issued BESIDE quarter-rate ones (round 3's priority-steered order)"})
    out["kernels"][name] = d
head = [k for k in out["kernels"] if "8, 1, false, true" in k]
if head:
    out["valu_busy_frac"] = out["kernels"][head[0]]["valu_busy_frac"]
    out["kernel"] = head[0]
# ---- optional third pair of passes (valu2 / valu2_ref): SQ_ACTIVE_INST_VALU2, a gfx950 counter: quad-cycles in which TWO
# VALU instructions were issued on a SIMD.  A quad-cycle holds either one 4-cycle-class instruction or up to two 2-cycle-
# class ones, so  quads with a VALU issue = SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2, and that over all quad-cycles of
# the kernel (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / 4) is the fraction of issue slots in use: a DIRECT busy measure.
if os.path.exists(os.path.join(prof, "valu2", "p_counter_collection.csv")):
    dual = {}
    for sub in ("valu2_ref", "valu2"):
        if not os.path.exists(os.path.join(prof, sub, "p_counter_collection.csv")):
            continue
        agg, nl = load(sub)
        for name, c in agg.items():
            if "k_score_rowlane" not in name and "k_p" not in name:
                continue
            quads = c["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD / 4.0
            inst, two = c["SQ_ACTIVE_INST_VALU"], c.get("SQ_ACTIVE_INST_VALU2", 0.0)
            dual[name] = {"valu_instructions_per_quad_cycle": inst / quads, "quad_cycles_with_two_valu_issued_frac": two / quads,
                          "quad_cycles_with_a_valu_issue_frac": (inst - two) / quads,
                          "raw": {k: c[k] for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "GRBM_GUI_ACTIVE")}}
            if "k_score_rowlane" in name:
                # the quarter-rate pipe: every VALU instruction of the kernel except its 8 v_xor_b32 per distance is a
                # quarter-rate one (v_bcnt_u32_b32, v_min3_u32, v_lshl_or_b32, ...: profiles/r01_valu_class.txt), and a SIMD
                # issues at most one of those per quad-cycle
                rows64 = DIST_PER_STEP / 64.0 * (nl[name] / 9.0)
                slow = inst - 8.0 * rows64
                dual[name].update({"valu_instructions_per_64_distances": inst / rows64,
                                   "quad_cycles_per_64_distances": quads / rows64, "simd_cycles_per_64_distances": 4.0 * quads / rows64,
                                   "quarter_rate_instructions_per_64_distances": slow / rows64,
                                   "quarter_rate_pipe_busy_frac": slow / quads})
    out["dual_issue"] = {"counter": "SQ_ACTIVE_INST_VALU2 (gfx950: quad-cycles in which two VALU instructions are issued, per SIMD)",
                         "kernels": dual}
    for name, d in dual.items():
        if "8, 1, false, true" in name:
            out["valu_issue_slots_busy_frac"] = d["quad_cycles_with_a_valu_issue_frac"]
            out["quad_cycles_with_two_valu_issued_frac"] = d["quad_cycles_with_two_valu_issued_frac"]
            out["quarter_rate_pipe_busy_frac"] = d.get("quarter_rate_pipe_busy_frac")
            out["simd_cycles_per_64_distances"] = d.get("simd_cycles_per_64_distances")
json.dump(out, open(out_path, "w"), indent=1)
if "dual_issue" in out:
    print("SQ_ACTIVE_INST_VALU2 (quad-cycles with two VALU instructions issued):")
    for name, d in out["dual_issue"]["kernels"].items():
        print("  %-62s %.3f instr per quad, dual quads %.3f, quads with a VALU issue %.3f%s" % (
            name[:62], d["valu_instructions_per_quad_cycle"], d["quad_cycles_with_two_valu_issued_frac"], d["quad_cycles_with_a_valu_issue_frac"],
            "; %.2f SIMD-cycles per 64 distances, quarter-rate pipe busy %.3f" % (d["simd_cycles_per_64_distances"], d["quarter_rate_pipe_busy_frac"])
            if "quarter_rate_pipe_busy_frac" in d else ""))
print("issue cost, SIMD-cycles per wave64 instruction: v_xor_b32 %.3f   v_bcnt_u32_b32 %.3f   ratio %.2f" % (cost_xor, cost_bcnt, cost_bcnt / cost_xor))
for key, d in refs.items():
    print("  %-12s %.3f cycles per VALU instruction, %.1f active lanes, waiting to issue %.0f %% of wave-cycles%s" % (
        key, d["simd_cycles_per_valu_instruction"], d["active_lanes_per_valu_instruction"], 100 * d["wave_cycles_waiting_to_issue_frac"],
        ", VALU busy %.3f" % d["valu_busy_frac"] if "valu_busy_frac" in d else ""))
for name, d in out["kernels"].items():
    print("%s\n   %.2f VALU instructions per 64 distances, %.3f SIMD-cycles per VALU instruction, VALU busy %.3f, lanes %.1f, "
          "wave-cycles: waiting to issue %.0f %%, parked on s_waitcnt %.0f %%" % (
              name, d["valu_instructions_per_64_distances"], d["simd_cycles_per_valu_instruction"], d["valu_busy_frac"],
              d["active_lanes_per_valu_instruction"], 100 * d["wave_cycles_waiting_to_issue_frac"], 100 * d["wave_cycles_parked_on_waitcnt_frac"]))
