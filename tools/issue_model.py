#!/usr/bin/env python3
"""The update kernel against its ISSUE roofline ("a counter-backed floor"): from the rocprofv3 counter passes of the C2
command on a caller's stream (one launch per step) --
    tools/prof_pmc2.sh  -> <pmc.json>      SQ_WAVES, SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*, SQ_INSTS_*, SQ_BUSY_CU_CYCLES ...
    tools/prof_pmc3.sh  -> <pmc_mix.json>  SQ_INSTS_VALU by class (ADD / MUL / FMA F32, TRANS, F64, CVT, INT32, INT64)
-- and the measured cost of a vector instruction by class (tools/ubench/valu_rate.hip, MI355X_MICROARCH.md: a wave64 instruction occupies
the SIMD-32 for 2 cycles; packed FP32, binary64 and conversions to / from binary64 for 4; transcendentals for 8), the cycles the vector
unit of one SIMD is busy per wave, and from them the time the launch would take if the vector units never idled:
    valu_floor_ms = waves x valu_cycles_per_wave / (SIMDs x shader clock)
What is between that floor and the measured time is not arithmetic: it is the rate at which the THREE resident waves of a SIMD get
through their own dependent instruction streams (a wave's lifetime inside the full launch over its lifetime alone, the wait shares).

    python3 tools/issue_model.py profiles/r05_c2_pmc.json profiles/r05_c2_pmc_mix.json <kernel_stats_caller_stream.csv> profiles/issue_latest.json
"""
import csv
import json
import sys

PK_PER_WAVE = 1536         # v_pk_mul / v_pk_add of the Hilbert FIR per wave and block: 8 trips x 192 (tools/isa_blocks.py on the loop-free kernel)
SIMDS = 1024               # 256 CUs x 4
CYC = {"f32": 2, "pk": 4, "f64": 4, "trans": 8, "cvt": 2, "int": 2, "other": 2}


def main():
    pmc, mix, stats, out = sys.argv[1:5]
    p = json.load(open(pmc)); m = json.load(open(mix))
    kern = p["kernel"]
    c = p["kernels"][kern]
    cm = m["kernels"].get(kern) or m["kernels"][m["kernel"]]
    waves = c["SQ_WAVES"]
    dur_ns = None
    with open(stats) as f:
        for row in csv.DictReader(f):
            if row["Name"].strip('"') == kern:
                dur_ns = float(row["AverageNs"]); calls = int(row["Calls"])
    per = lambda k: cm.get(k, 0.0) / waves
    n_valu = c["SQ_INSTS_VALU"] / waves
    f32 = per("SQ_INSTS_VALU_ADD_F32") + per("SQ_INSTS_VALU_MUL_F32") + per("SQ_INSTS_VALU_FMA_F32")
    f64 = per("SQ_INSTS_VALU_ADD_F64") + per("SQ_INSTS_VALU_MUL_F64") + per("SQ_INSTS_VALU_FMA_F64") + per("SQ_INSTS_VALU_TRANS_F64")
    trans = per("SQ_INSTS_VALU_TRANS_F32")
    cvt = per("SQ_INSTS_VALU_CVT")
    ints = per("SQ_INSTS_VALU_INT32") + per("SQ_INSTS_VALU_INT64")
    other = max(0.0, n_valu - (f32 + f64 + trans + cvt + ints))
    pk = min(PK_PER_WAVE, f32)
    classes = {"f32_plain": f32 - pk, "f32_packed": pk, "f64": f64, "transcendental": trans, "convert": cvt, "integer": ints, "moves_selects_dpp_other": other}
    cycles = ((f32 - pk) * CYC["f32"] + pk * CYC["pk"] + f64 * CYC["f64"] + trans * CYC["trans"] + cvt * CYC["cvt"] + ints * CYC["int"] + other * CYC["other"])
    busy_cu_cycles = c.get("SQ_BUSY_CU_CYCLES", 0.0) / 256.0                      # cycles a CU was busy during the launch
    clock_ghz = busy_cu_cycles / dur_ns if dur_ns else None                        # cycles per ns
    floor_ms = waves * cycles / (SIMDS * clock_ghz * 1e9) * 1e3 if clock_ghz else None
    wave_cycles = c["SQ_WAVE_CYCLES"] * 4.0 / waves                                # (SQ_WAVE_CYCLES counts quad-cycles)
    res = {
        "source_sha256": p.get("source_sha256"), "library_sha256": p.get("library_sha256"), "kernel": kern, "workload": p.get("workload"),
        "waves_per_launch": waves, "valu_instructions_per_wave": round(n_valu, 1),
        "valu_instructions_per_wave_by_class": {k: round(v, 1) for k, v in classes.items()},
        "cycles_per_instruction_by_class": {"f32_plain": 2, "f32_packed": 4, "f64": 4, "transcendental": 8, "convert": 2, "integer": 2, "moves_selects_dpp_other": 2},
        "valu_cycles_per_wave": round(cycles, 0),
        "salu_instructions_per_wave": round(c.get("SQ_INSTS_SALU", 0.0) / waves, 1), "lds_instructions_per_wave": round(c.get("SQ_INSTS_LDS", 0.0) / waves, 1),
        "wave_lifetime_cycles_in_the_full_launch": round(wave_cycles, 0),
        "wave_cycles_issuing_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "wave_cycles_issue_stalled_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "wave_cycles_parked_at_a_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
        "kernel_avg_ms_rocprofv3": round(dur_ns / 1e6, 5) if dur_ns else None, "kernel_calls_in_that_trace": calls if dur_ns else None,
        "shader_clock_ghz_during_the_launch": round(clock_ghz, 3) if clock_ghz else None,
        "valu_floor_ms": round(floor_ms, 5) if floor_ms else None,
        "valu_busy_frac": round(floor_ms / (dur_ns / 1e6), 3) if floor_ms else None,
        "resident_waves_per_simd": 3,
    }
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, sort_keys=True))


if __name__ == "__main__":
    main()
