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
LIB = os.path.join(ROOT, "audiosdr_amd", "variants", "libasdr_timeline.so")
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
MW_NAMES = ["prologue + loads issued", "NB: envelopes, ring store", "NB: publish chain inputs",
            "  barrier 1 (chains may start): parked", "  NB chain duty (rel 0 works, others pass)", "  barrier 2 (chains done): parked",
            "NB: threshold test", "NB: mask / output rows (quiet path)", "IF pipeline (+ ring prefetch issue)", "mixer phase / uniform test",
            "mixer multiply, AF / AGC state loads", "Hilbert: stage history", "Hilbert: FIR", "sideband combine (+ delayed I wait)",
            "  barrier 3 (audio rows ready): parked", "  audio cascade duty (rel 1, 2 work)", "  barrier 4 (cascades done): parked",
            "(tap row)", "AGC: |x|, block maximum, quiet test", "  barrier 5 (AGC inputs): parked", "  AGC chain duty (rel 3 works)",
            "  barrier 6 (AGC chains done): parked", "(status)", "AGC: apply", "output + status"]


def mw(warm=None):
    """The four-wave workgroup form (asdr_update_kernel_mw) inside a full C2 launch: all four waves of workgroups 0, 683, 1365, 2047 write
    clock64() at the 16 phase boundaries and on either side of the six workgroup barriers.  Prints, per duty (rel 0 = blanker / phase chains,
    1 and 2 = audio cascades, 3 = AGC chain), shader-clock cycles per phase, the cycles parked at each barrier, and the sums."""
    import audiosdr_amd as A
    from audiosdr_amd.synth import make_iq
    import torch
    L = A.binding.load_library(LIB)
    n_ch = 65536
    uniq = 2048
    I, Q = make_iq(uniq, 6, fc=6290.0, A=0.25)
    I = np.tile(I, (n_ch // uniq, 1, 1)); Q = np.tile(Q, (n_ch // uniq, 1, 1))
    for warm in ([int(warm)] if warm is not None else [300, 8]):
        h = L.asdr_create(n_ch, 0)
        L.asdr_setDemodMode(h, -1, 1); L.asdr_enableAudioFilter(h, -1); L.asdr_enable_taps(h, 1)
        dOut = torch.empty((n_ch, 128), dtype=torch.int16, device="cuda")
        dIs = [torch.from_numpy(np.ascontiguousarray(I[:, b])).cuda() for b in range(6)]
        dQs = [torch.from_numpy(np.ascontiguousarray(Q[:, b])).cuda() for b in range(6)]
        for b in range(warm):
            L.asdr_update_device(h, C.c_void_p(dIs[b % 6].data_ptr()), C.c_void_p(dQs[b % 6].data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
        rows = []
        for b in range(12):
            dI, dQ = dIs[(warm + b) % 6], dQs[(warm + b) % 6]
            L.asdr_update_device(h, C.c_void_p(dI.data_ptr()), C.c_void_p(dQ.data_ptr()), C.c_void_p(dOut.data_ptr()), 1, None)
            taps = np.zeros((12, n_ch, 128), dtype=np.float32)
            L.asdr_read_taps(h, taps.ctypes.data_as(C.POINTER(C.c_float)))
            tl = taps.reshape(-1).view(np.uint64)[:16 * 32].reshape(4, 4, 32).astype(np.int64)   # [workgroup][wave][entry]
            rows.append(tl)
        L.asdr_destroy(h)
        raw = np.stack(rows)                            # [launch][wg][wave][32]
        if os.environ.get("TIMELINE_RAW"):
            # per workgroup and wave (last launch): duty, SIMD, wave slot, and the cycles of a few phases
            hw = raw[-1, :, :, 29]
            for wi, wg in enumerate((0, 683, 1365, 2047)):
                for w in range(4):
                    t = raw[-1, wi, w]
                    print("   wg %4d wave %d rel %d simd %d slot %d cu %2d | start %7d  envelopes %6d  IF %6d  mixer %6d  FIR %6d  TL4 %7d TL5 %7d TL8 %7d TL9 %7d" % (
                        wg, w, (w - wg) & 3, (hw[wi, w] >> 4) & 3, hw[wi, w] & 15, (hw[wi, w] >> 8) & 15, t[0] - raw[-1, wi, :, 0].min(), t[2] - t[1], t[5] - t[4], t[7] - t[6], t[9] - t[8],
                        t[4] - raw[-1, wi, :, 0].min(), t[5] - raw[-1, wi, :, 0].min(), t[8] - raw[-1, wi, :, 0].min(), t[9] - raw[-1, wi, :, 0].min()))
        tl = raw[:, :, :, MW_SEQ]            # [launch][wg][wave][28]
        d = np.diff(tl, axis=3).astype(np.float64)      # 27 intervals
        # fold "(tap row)" and "(status)" style slivers: keep all 27, name by MW_NAMES (25) + two joins
        names = MW_NAMES[:]
        # intervals: 0-1,1-2,2-17,17-18,18-19,19-20,20-3,3-4,4-5,5-6,6-7,7-8,8-9,9-10,10-21,21-22,22-23,23-24,24-11,11-12,12-25,25-26,26-27,27-28,28-13,13-14,14-15
        names = ["prologue + loads issued", "NB: envelopes, ring store", "NB: publish chain inputs",
                 "  barrier 1 (chains may start): parked", "  NB chain duty (rel 0 works, others pass)", "  barrier 2 (chains done): parked",
                 "NB: threshold test", "NB: mask / output rows", "IF pipeline (+ ring prefetch issue)", "mixer phase / uniform test",
                 "mixer multiply, AF / AGC state loads", "Hilbert: stage history", "Hilbert: FIR", "sideband combine (+ delayed I wait)", "AGC table request",
                 "  barrier 3 (audio rows ready): parked", "  audio cascade duty (rel 1, 2 work)", "  barrier 4 (cascades done): parked",
                 "(to TL11)", "AGC: |x|, block maximum, quiet test", "AGC: publish",
                 "  barrier 5 (AGC inputs): parked", "  AGC chain duty (rel 3 works)", "  barrier 6 (AGC chains done): parked",
                 "(to TL13)", "AGC: apply", "output + status"]
        rel = np.array([[(w - wg) & 3 for w in range(4)] for wg in (0, 683, 1365, 2047)])   # duty of wave w of workgroup wg
        print("== asdr_update_kernel_mw, 65536 channels, %d warm-up blocks (%s): shader-clock cycles, medians over %d launches x 4 workgroups" %
              (warm, "steady bank: AGC quiet path" if warm >= 200 else "fresh bank: AGC attacking", d.shape[0]))
        life = (tl[..., -1] - tl[..., 0]).astype(np.float64)
        wg_span = (tl[..., -1].max(axis=2) - tl[..., 0].min(axis=2)).astype(np.float64)
        print("   workgroup span (first wave in .. last wave out) %.0f; wave lifetime by duty:" % np.median(wg_span),
              "  ".join("rel %d %.0f" % (r, np.median(life[:, rel == r])) for r in range(4)))
        hdr = "   %-46s" % "phase" + "".join("%9s" % ("rel %d" % r) for r in range(4)) + "%9s" % "mean"
        print(hdr)
        tot_park = np.zeros(4); tot = np.zeros(4)
        for k, nm in enumerate(names):
            v = [np.median(d[:, rel == r, k]) for r in range(4)]
            tot += v
            if "parked" in nm:
                tot_park += v
            print("   %-46s" % nm + "".join("%9.0f" % x for x in v) + "%9.0f" % np.mean(v))
        print("   %-46s" % "SUM" + "".join("%9.0f" % x for x in tot) + "%9.0f" % tot.mean())
        print("   %-46s" % "parked at the six barriers" + "".join("%9.0f" % x for x in tot_park) + "%9.0f  (%.1f %% of the lifetime)" % (tot_park.mean(), 100 * tot_park.mean() / tot.mean()))


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
    L.asdr_destroy(h)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build()
    elif sys.argv[1:2] == ["mw"]:
        mw(*sys.argv[2:3])
    elif sys.argv[1:2] == ["stream"]:
        stream(*[int(x) for x in sys.argv[2:]])
    else:
        run()
