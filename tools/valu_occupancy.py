#!/usr/bin/env python3
"""VALU issue occupancy per kernel from the committed SQ counter summaries (tools/sq_counters.sh -> profiles/*_sq_counters.txt).

    python tools/valu_occupancy.py profiles/r02_v23_sq_counters.txt [--json profiles/r02_valu_occupancy.json]

occupancy = SQ_ACTIVE_INST_VALU x 4 / (1 024 SIMDs x kernel duration x shader clock)

* SQ_ACTIVE_INST_VALU counts QUAD-cycles (MI355X_MICROARCH.md, cycle-constants table) in which a SIMD has a vector
  instruction in flight, summed over the chip's 256 CUs x 4 SIMDs; x 4 turns it into shader cycles.
* the clock is GRBM_GUI_ACTIVE / 8 / duration when that counter was collected in the same pass (the guide's DVFS
  recipe: rocprofv3 sums it over the 8 XCDs), else the nominal 2.4 GHz -- which can only UNDER-state the occupancy.
* cross-check printed beside it: cycles per VALU instruction = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU.  The issue table
  assumed for the instruction mix (guide, "vector-instruction ISSUE cost"): plain and packed fp32 / integer VALU = 4 cycles
  per wave-instruction, transcendentals (v_exp_f32, v_rcp_f32) = 8.  A kernel with a fraction t of transcendentals
  should therefore show 4 (1 + t) cycles per instruction.
"""
import argparse
import json
import re

N_SIMD = 1024
NOMINAL_GHZ = 2.4


def parse(path):
    k, out = None, {}
    for line in open(path):
        if not line.startswith(" "):
            k = line.strip()
            out.setdefault(k, {})
        else:
            m = re.match(r"\s+(\S+)\s+(\S+)", line)
            if m and k:
                # the file holds one block per collection pass: later passes add counters, durations are kept per pass
                name, val = m.group(1), float(m.group(2))
                if name == "_dur_ns":
                    out[k].setdefault("_dur_ns_passes", []).append(val)
                out[k][name] = val
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--json")
    a = ap.parse_args()
    res = {"source": a.summary, "formula": "SQ_ACTIVE_INST_VALU*4 / (1024 SIMD * duration * clock)", "kernels": {}}
    for k, d in parse(a.summary).items():
        if "SQ_ACTIVE_INST_VALU" not in d:
            continue
        dur = d["_dur_ns"] * 1e-9                         # the pass that collected SQ_ACTIVE_INST_VALU comes last
        ghz = d["GRBM_GUI_ACTIVE"] / 8.0 / d["_dur_ns"] if "GRBM_GUI_ACTIVE" in d else NOMINAL_GHZ
        clock_src = "GRBM_GUI_ACTIVE/8/duration" if "GRBM_GUI_ACTIVE" in d else "nominal 2.4 GHz"
        if ghz > 2.45:                                    # the quotient reads high on dispatches well under 0.3 ms (guide)
            ghz, clock_src = NOMINAL_GHZ, "nominal 2.4 GHz (GRBM quotient unreliable on a dispatch this short)"
        occ = d["SQ_ACTIVE_INST_VALU"] * 4 / (N_SIMD * dur * ghz * 1e9)
        cpi = 4 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"] if d.get("SQ_INSTS_VALU") else None
        wave_res = d["SQ_WAVE_CYCLES"] * 4 / d["SQ_WAVES"] / (dur * ghz * 1e9) if d.get("SQ_WAVES") and d.get("SQ_WAVE_CYCLES") else None
        res["kernels"][k] = {"valu_issue_occupancy": round(occ, 3), "duration_us": round(dur * 1e6, 2), "clock_GHz": round(ghz, 3),
                             "clock_source": clock_src, "cycles_per_valu_instruction": round(cpi, 2) if cpi else None,
                             "mean_wave_residency_of_kernel": round(wave_res, 3) if wave_res else None,
                             "valu_wave_instructions": d.get("SQ_INSTS_VALU")}
        print("%-28s VALU issue occupancy %.3f  (%.1f us, %.2f GHz [%s], %.2f cycles per VALU instruction)"
              % (k, occ, dur * 1e6, ghz, clock_src, cpi or 0))
    if a.json:
        flat = {"source": res["source"], "formula": res["formula"]}
        flat.update(res["kernels"])
        json.dump(flat, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
