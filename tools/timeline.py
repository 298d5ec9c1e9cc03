#!/usr/bin/env python3
"""Phase timeline of single waves of asdr_update_kernel inside a full C2 launch (65,536 channels): a profiling build
(-DASDR_TIMELINE) makes lane 0 of waves 0 / 2731 / 5461 / 8191 write clock64() at 16 phase boundaries into the taps buffer.
Also runs a lone-wave launch (8 channels).  Prints microseconds per phase (s_memtime ticks at 100 MHz).  (GPU box.)

  python tools/timeline.py build   # here
  python tools/timeline.py run     # GPU box
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
LIB = os.environ.get("TIMELINE_LIB") or os.path.join(ROOT, "audiosdr_amd", "variants", "libasdr_timeline.so")
NAMES = ["prologue+load", "NB: store ring, envelopes", "NB: sequential average/threshold", "NB: mask, ramp, carry, output",
         "IF pipeline", "mixer phase recurrence", "mixer multiply", "Hilbert: stage history", "Hilbert: FIR", "sideband combine",
         "audio pipeline", "AGC: table to LDS", "AGC: sequential envelope", "AGC: apply", "output + status"]


def build():
    from audiosdr_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    b.build(force=True, extra_flags=["-DASDR_TIMELINE"], out=LIB)
    print("built", LIB)


def run():
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    L = A.binding.load_library(LIB)
    for n_ch in (65536, 8):
        uniq = min(n_ch, 2048)
        I, Q = make_iq(uniq, 6, fc=6290.0, A=0.25)
        I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
        h = L.asdr_create(n_ch, 0)
        L.asdr_setDemodMode(h, -1, 1); L.asdr_enableAudioFilter(h, -1); L.asdr_enable_taps(h, 1)
        dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
        rows = []
        dIs = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(6)]
        dQs = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(6)]
        warm = int(os.environ.get("TIMELINE_WARM", "300"))   # the AGC envelope has converged, as in bench.py's steady state
        for b in range(warm):
            L.asdr_update_device(h, C.c_void_p(dIs[b % 6].data_ptr()), C.c_void_p(dQs[b % 6].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
        for b in range(6):
            dI, dQ = dIs[(warm + b) % 6], dQs[(warm + b) % 6]
            L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
            taps = np.zeros((12, n_ch, 128), dtype=np.float32)
            L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
            tl = taps.reshape(-1).view(np.uint64)[:128].reshape(4, 32)[:, :16].astype(np.int64)
            if b >= 2:
                rows.append(tl)
        tl = np.stack(rows)                      # [launch][wave][16]
        d = np.diff(tl, axis=2).astype(np.float64)   # ticks per phase
        nw = 4 if n_ch >= 65536 else 1
        d = d[:, :nw]
        tick_us = 0.01                            # s_memtime: 100 MHz on gfx9
        med = np.median(d.reshape(-1, 15), axis=0) * tick_us
        print("== %d channels (%s): wave lifetime %.1f us (median over %d waves x launches)" %
              (n_ch, "full launch" if n_ch > 8 else "lone wave", med.sum(), d.shape[0] * nw))
        for nm, v in zip(NAMES, med):
            print("   %-36s %7.2f us  %5.1f %%" % (nm, v, 100 * v / med.sum()))
        L.asdr_destroy(h)


MW_SEQ = [0, 1, 2, 17, 18, 19, 20, 3, 4, 5, 6, 7, 8, 9, 10, 21, 22, 23, 24, 11, 12, 25, 26, 27, 28, 13, 14, 15]
# the 27 intervals between consecutive marks of MW_SEQ; kind: c = the wave computes, b = parked at a workgroup barrier, d = a duty section (one or two
# waves of the workgroup work, the others pass through), w = dominated by a memory wait
MW_PHASES = [("prologue + loads issued (first HBM round trip)", "w"), ("NB: envelopes, ring store", "c"), ("NB: publish chain inputs", "c"),
             ("  barrier 1 (chains may start): parked", "b"), ("  NB chain duty (rel 0 works, others pass)", "d"), ("  barrier 2 (chains done): parked", "b"),
             ("NB: threshold test", "c"), ("NB: mask / output rows", "c"), ("IF pipeline (+ ring prefetch issue)", "c"), ("mixer phase / uniform test", "c"),
             ("mixer multiply, AF / AGC state loads", "c"), ("Hilbert: stage history", "c"), ("Hilbert: FIR", "c"), ("sideband combine (+ delayed I wait)", "c"),
             ("AGC table request", "c"), ("  barrier 3 (audio rows ready): parked", "b"), ("  audio cascade duty (rel 1, 2 work)", "d"),
             ("  barrier 4 (cascades done): parked", "b"), ("(to the next mark)", "c"), ("AGC: |x|, block maximum, quiet test", "c"), ("AGC: publish", "c"),
             ("  barrier 5 (AGC inputs): parked", "b"), ("  AGC chain duty (rel 3 works)", "d"), ("  barrier 6 (AGC chains done): parked", "b"),
             ("(to the next mark)", "c"), ("AGC: apply", "c"), ("output + status", "c")]
MW_WGS = (0, 683, 1365, 2047)


def mw_measure(warm, n_ch=65536, launches=12):
    """Raw marks [launch][workgroup][wave][32] of the four-wave form: all four waves of workgroups 0, 683, 1365, 2047 (a lone workgroup: 0 only)."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    L = A.binding.load_library(LIB)
    uniq = min(2048, n_ch)
    I, Q = make_iq(uniq, 6, fc=6290.0, A=0.25)
    I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
    h = L.asdr_create(n_ch, 0)
    L.asdr_setDemodMode(h, -1, 1); L.asdr_enableAudioFilter(h, -1); L.asdr_enable_taps(h, 1)
    dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
    dIs = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(6)]
    dQs = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(6)]
    for b in range(warm):
        L.asdr_update_device(h, C.c_void_p(dIs[b % 6].data_ptr()), C.c_void_p(dQs[b % 6].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
    rows = []
    n_wg = 4 if n_ch >= 65536 else 1
    for b in range(launches):
        dI, dQ = dIs[(warm + b) % 6], dQs[(warm + b) % 6]
        L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
        taps = np.zeros((12, n_ch, 128), dtype=np.float32)
        L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
        rows.append(taps.reshape(-1).view(np.uint64)[:16 * 32].reshape(4, 4, 32)[:n_wg].astype(np.int64))
    L.asdr_destroy(h)
    return np.stack(rows)


def mw_table(raw, n_ch, warm):
    """Per phase and duty: median cycles.  Returns the dict tools/latency_model.py works on."""
    n_wg = raw.shape[1]
    tl = raw[:, :, :, MW_SEQ]
    d = np.diff(tl, axis=3).astype(np.float64)      # 27 intervals
    rel = np.array([[(w - wg) & 3 for w in range(4)] for wg in MW_WGS[:n_wg]])   # duty of wave w of workgroup wg
    life = (tl[..., -1] - tl[..., 0]).astype(np.float64)
    wg_span = (tl[..., -1].max(axis=2) - tl[..., 0].min(axis=2)).astype(np.float64)
    phases = []
    for k, (nm, kind) in enumerate(MW_PHASES):
        v = [float(np.median(d[:, rel == r, k])) for r in range(4)]
        phases.append({"name": nm.strip(), "kind": kind, "cycles_by_duty": v, "cycles_mean": float(np.mean(v))})
    return {"channels": n_ch, "warm_up_blocks": warm, "launches": int(raw.shape[0]), "workgroups_sampled": list(MW_WGS[:n_wg]),
            "workgroup_span_cycles": float(np.median(wg_span)), "wave_lifetime_cycles_by_duty": [float(np.median(life[:, rel == r])) for r in range(4)],
            "wave_lifetime_cycles": float(np.median(life)), "phases": phases}


def mw_print(t):
    print("== asdr_update_kernel_mw, %d channels%s, %d warm-up blocks (%s): shader-clock cycles, medians over %d launches x %d workgroup(s)" %
          (t["channels"], " (a LONE workgroup: nothing else on the chip, one wave per SIMD)" if t["channels"] <= 32 else "", t["warm_up_blocks"],
           "steady bank: AGC quiet path" if t["warm_up_blocks"] >= 200 else "fresh bank: AGC attacking", t["launches"], len(t["workgroups_sampled"])))
    print("   workgroup span (first wave in .. last wave out) %.0f; wave lifetime by duty:" % t["workgroup_span_cycles"],
          "  ".join("rel %d %.0f" % (r, v) for r, v in enumerate(t["wave_lifetime_cycles_by_duty"])))
    print("   %-52s" % "phase" + "".join("%9s" % ("rel %d" % r) for r in range(4)) + "%9s" % "mean")
    tot = np.zeros(4); park = np.zeros(4)
    for ph in t["phases"]:
        v = np.array(ph["cycles_by_duty"]); tot += v
        if ph["kind"] == "b":
            park += v
        print("   %-52s" % (("  " if ph["kind"] in "bd" else "") + ph["name"]) + "".join("%9.0f" % x for x in v) + "%9.0f" % v.mean())
    print("   %-52s" % "SUM" + "".join("%9.0f" % x for x in tot) + "%9.0f" % tot.mean())
    print("   %-52s" % "parked at the six barriers" + "".join("%9.0f" % x for x in park) + "%9.0f  (%.1f %% of the lifetime)" % (park.mean(), 100 * park.mean() / tot.mean()))


def mw(warm=None, n_ch=65536):
    """The four-wave workgroup form (asdr_update_kernel_mw) inside a full C2 launch (or, n_ch = 32 with ASDR_MW_MIN_WAVES=1, as a LONE workgroup):
    clock64() at the 16 phase boundaries and on either side of the six workgroup barriers.  Prints, per duty (rel 0 = blanker / phase chains,
    1 and 2 = audio cascades, 3 = AGC chain), shader-clock cycles per phase, the cycles parked at each barrier, and the sums.
    TIMELINE_JSON=<path>: the tables as JSON as well (tools/latency_model.py)."""
    n_ch = int(n_ch)
    out = []
    for w in ([int(warm)] if warm is not None else [300, 8]):
        raw = mw_measure(w, n_ch)
        if os.environ.get("TIMELINE_RAW"):
            hw = raw[-1, :, :, 29]
            for wi in range(raw.shape[1]):
                t0 = raw[-1, wi, :, 0].min()
                for wv in range(4):
                    t = raw[-1, wi, wv]
                    print("   wg %4d wave %d rel %d simd %d slot %d cu %2d | start %7d  envelopes %6d  IF %6d  mixer %6d  FIR %6d" % (
                        MW_WGS[wi], wv, (wv - MW_WGS[wi]) & 3, (hw[wi, wv] >> 4) & 3, hw[wi, wv] & 15, (hw[wi, wv] >> 8) & 15, t[0] - t0, t[2] - t[1], t[5] - t[4], t[7] - t[6], t[9] - t[8]))
        t = mw_table(raw, n_ch, w)
        mw_print(t)
        out.append(t)
    if os.environ.get("TIMELINE_JSON"):
        import json
        json.dump(out, open(os.environ["TIMELINE_JSON"], "w"), indent=1)
    return out


def stream(T=256, n_ch=512):
    """The block pipeline's three roles of channel group 0 over the last two blocks of a T-block call (WSPR settings, as C5):
    time between two consecutive block starts of each role (its cycle time = the pipeline's time per block if it is the slowest)
    and where inside the block it goes: waits for the neighbours' counters | every phase boundary of the chain."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    import bench
    L = A.binding.load_library(LIB)
    I, Q = make_iq(n_ch, T, fc=6890.0, A=0.02, noise=0.05)
    dI, dQ = torch.from_numpy(I).cuda(), torch.from_numpy(Q).cuda()
    dO = torch.empty((n_ch, T, 128), dtype=torch.int16, device="cuda")
    h = L.asdr_create(n_ch, 0)
    class H:   # the bench's C5 settings through the raw handle
        def __getattr__(self, name):
            return lambda *a: getattr(L, "asdr_" + name)(h, -1, *a)
    bench.configure_c5(H())
    L.asdr_enable_taps(h, 1)
    names = ["waits for the neighbours"] + NAMES
    acc = []
    for rep in range(8):
        L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dO.data_ptr()), T, None)
        L.asdr_synchronize(h)
        taps = np.zeros((12, n_ch, 128), dtype=np.float32)
        L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
        tl = taps.reshape(-1).view(np.uint64)[:32 * 14].reshape(14, 32).astype(np.int64)
        if rep >= 2:
            acc.append(tl)
    print("pipeline launches:", L.asdr_stream_pipeline_launches(h))
    tl = np.stack(acc)   # [rep][slot][32]
    last, prev = (T - 1) & 1, (T - 2) & 1
    for role in (1, 2, 3):
        a_, p_ = tl[:, 2 * (3 + role) + last], tl[:, 2 * (3 + role) + prev]
        cyc = np.median(a_[:, 16] - p_[:, 16])
        print("== role %d: %.0f cycles between two block starts" % (role, cyc))
        seq = np.concatenate([a_[:, 16:17], a_[:, :16]], axis=1)    # pre-wait, TL0..TL15
        d = np.diff(seq, axis=1)
        med = np.median(d, axis=0)
        for nm, v in zip(names, med):
            if v > 0 and v < 10 * cyc:
                print("   %-36s %8.0f cycles" % (nm, v))
        if os.environ.get("TIMELINE_RAW"):
            # the marks this role passed, in time order: cycles since the block start (marks a role does not pass hold another block's clock)
            off = np.median(a_[:, :16] - a_[:, 16:17], axis=0)
            print("   marks in time order (TL index: cycles since the block start): " + "  ".join("%d:%d" % (i, v) for v, i in sorted((v, i) for i, v in enumerate(off) if 0 <= v < 2 * cyc)))
    L.asdr_destroy(h)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    elif sys.argv[1:2] == ["mw"]:
        mw(*sys.argv[2:4])
    elif sys.argv[1:2] == ["stream"]:
        stream(*[int(x) for x in sys.argv[2:]])
    else:
        run()
