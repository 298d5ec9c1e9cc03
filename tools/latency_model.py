#!/usr/bin/env python3
"""The update kernel against its LATENCY roofline ("dependent-chain length x measured per-instruction latency at 3 waves per
SIMD, per phase, summed, within 10 % of the measured wave lifetime, reproducible by a committed tool").

What is measured (GPU box, one run of this tool):
  1. tools/ubench/issue_rate  -- for the instruction mix of every phase family (dependent chain, biquad pipeline step, packed FIR trip, pointwise
     envelope pass, AGC chunk): cycles per instruction of ONE wave when W = 1, 2, 3 waves per SIMD run that mix: t1, t2, t3.
  2. tools/timeline.py mw     -- asdr_update_kernel_mw's marks (clock64 at 16 phase boundaries + either side of the six workgroup barriers)
     (a) as a LONE workgroup (32 channels: one wave per SIMD, idle memory system) and (b) inside the full C2 launch (65,536 channels, steady bank).
The model:
  * A phase's dependent instruction stream is what the LONE workgroup needs for it: lone_p cycles = N_p instructions x t1(mix_p).  That is the
    latency floor of this mapping: nothing shortens a wave's own stream.
  * With three resident waves per SIMD all active in that mix, the same stream takes shared_p = lone_p x t3 / t1 (the ubench's ratio for the mix).
  * Waits that are not instructions: the first HBM round trip (prologue) is taken as measured in the full launch; barrier parking is NOT added --
    a parked wave's duty is somebody's stream, which the workgroup's critical path already contains once (the duty sections).
  * Workgroup critical path = every "c" phase (mean over the four waves) + the duty sections (the duty wave's time) + the prologue.
  lifetime_lone      = the lone workgroup's critical path                (floor: 8,192 waves / 3,072 resident x lifetime / clock)
  lifetime_3_active  = sum of shared_p + prologue under load             (what three ALWAYS-active waves per SIMD would give)
  lifetime_measured  = the full launch's wave lifetime                   (sits between the two: on average fewer than three waves are active)
Output: a JSON object (bench.py puts it on its line as `roofline_latency` when its source hash matches the library's).

  python3 tools/latency_model.py run  [out.json]      GPU box (needs tools/ubench/issue_rate and audiosdr_amd/variants/libasdr_timeline.so: build both here)
  python3 tools/latency_model.py build                 here: the two binaries
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

# phase (name prefix of tools/timeline.py MW_PHASES) -> instruction mix of tools/ubench/issue_rate
MIX_OF = [("NB: envelopes", "pointwise"), ("NB: publish", "pointwise"), ("NB chain duty", "chain2"), ("NB: threshold", "pointwise"), ("NB: mask", "pointwise"),
          ("IF pipeline", "biquad"), ("mixer", "pointwise"), ("Hilbert: stage", "pointwise"), ("Hilbert: FIR", "fir"), ("sideband", "pointwise"),
          ("AGC table", "pointwise"), ("audio cascade duty", "biquad"), ("(to the next", "pointwise"), ("AGC: |x|", "pointwise"), ("AGC: publish", "pointwise"),
          ("AGC chain duty", "agc"), ("AGC: apply", "pointwise"), ("output", "pointwise")]
WAVES, RESIDENT = 8192, 3072   # C2: 65,536 channels / 8 per wave; 256 CUs x 12 waves


def build():
    from audiosdr_amd import build as b
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-I", b.CSRC,
                           os.path.join(ROOT, "tools", "ubench", "issue_rate.hip"), "-o", os.path.join(ROOT, "tools", "ubench", "issue_rate")])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), "build"])


def mix_for(name):
    for prefix, mix in MIX_OF:
        if name.startswith(prefix):
            return mix
    return None


def run(out_path=None):
    from audiosdr_amd import build as b
    rate = json.loads(subprocess.check_output([os.path.join(ROOT, "tools", "ubench", "issue_rate")], text=True))
    mixes = {k: v["cycles_per_instruction_of_a_wave"] for k, v in rate["mixes"].items()}
    tmp = tempfile.mkdtemp()
    env = dict(os.environ)
    tl = {}
    for key, n_ch, extra in (("lone", 32, {"ASDR_MW_MIN_WAVES": "1"}), ("full", 65536, {})):
        e = dict(env); e.update(extra); e["TIMELINE_JSON"] = os.path.join(tmp, key + ".json")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), "mw", "300", str(n_ch)], env=e, stdout=subprocess.DEVNULL)
        tl[key] = json.load(open(e["TIMELINE_JSON"]))[0]
    phases, lone_sum, shared_sum, prologue_full, prologue_lone = [], 0.0, 0.0, 0.0, 0.0
    for pl, pf in zip(tl["lone"]["phases"], tl["full"]["phases"]):
        kind = pl["kind"]
        if kind == "b":
            continue
        # a compute phase: the mean over the four waves; a duty section: the duty wave's time (the largest of the four columns)
        lone_c = max(pl["cycles_by_duty"]) if kind == "d" else pl["cycles_mean"]
        full_c = max(pf["cycles_by_duty"]) if kind == "d" else pf["cycles_mean"]
        if kind == "w":
            prologue_lone, prologue_full = lone_c, full_c
            phases.append({"phase": pl["name"], "lone_cycles": round(lone_c), "full_launch_cycles": round(full_c), "what": "memory wait, not instructions"})
            continue
        mix = mix_for(pl["name"])
        t1, t2, t3 = mixes[mix][0], mixes[mix][1], mixes[mix][2]
        shared = lone_c * t3 / t1
        lone_sum += lone_c; shared_sum += shared
        phases.append({"phase": pl["name"], "mix": mix, "lone_cycles": round(lone_c), "instructions": round(lone_c / t1),
                       "cycles_per_instruction_at_1_2_3_waves_per_simd": [t1, t2, t3], "three_active_waves_cycles": round(shared),
                       "full_launch_cycles": round(full_c)})
    measured = tl["full"]["wave_lifetime_cycles"]
    barriers_full = sum(p["cycles_mean"] for p in tl["full"]["phases"] if p["kind"] == "b")
    life_lone = lone_sum + prologue_lone
    life_3 = shared_sum + prologue_full
    # shader clock of the full launch: the library's own kernel time is not available here; bench.py converts with the clock it measures.  2.13 GHz = profiles/issue_latest.json
    clock_ghz = float(os.environ.get("ASDR_LATENCY_CLOCK_GHZ", "2.13"))
    to_ms = lambda cycles: WAVES / RESIDENT * cycles / (clock_ghz * 1e9) * 1e3
    res = {
        "kernel": "asdr_update_kernel_mw", "workload": "bench.py C2: 65,536 channels x 1 block per launch, steady bank",
        "source_sha256": b.source_sha256(),
        "model": "workgroup critical path = compute phases + duty sections (+ the first HBM round trip); lone = a lone workgroup (one wave per SIMD): N_p "
                 "instructions x t1; three_active = the same streams at the measured issue interval of a wave that shares its SIMD with two others "
                 "running the same mix (tools/ubench/issue_rate.hip); barrier parking is not added (a duty is in the path once)",
        "phases": phases,
        "wave_lifetime_cycles": {"lone_workgroup_critical_path": round(life_lone), "lone_workgroup_measured": round(tl["lone"]["wave_lifetime_cycles"]),
                                 "three_always_active_waves_per_simd": round(life_3), "full_launch_measured": round(measured),
                                 "full_launch_parked_at_barriers_mean": round(barriers_full)},
        "model_over_measured": round(life_3 / measured, 3),
        "shader_clock_ghz_assumed": clock_ghz, "waves": WAVES, "resident_waves": RESIDENT,
        "ms_per_step": {"floor_lone_stream": round(to_ms(life_lone), 4), "three_always_active_waves": round(to_ms(life_3), 4),
                        "from_the_measured_lifetime": round(to_ms(measured), 4), "north_star_target": 0.0891,
                        "lifetime_the_target_needs_cycles": round(0.0891e-3 * clock_ghz * 1e9 * RESIDENT / WAVES)},
        "issue_rate_ubench": rate,
    }
    s = json.dumps(res, indent=1)
    if out_path:
        open(out_path, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else None)
