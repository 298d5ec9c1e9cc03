"""Setter scripts + synthetic-signal parameters shared by the GPU parity tests, the golden fixtures and
tools/gpu_check.py.  Mode/filter ids are the reference's (AudioSDR.h:44-71)."""
from helpers import S

LSB, USB, CW_LSB, CW_USB, AM, SAM, WSPR = range(7)
tone = dict(fc=6290.0, A=0.25)
am = dict(fc=6890.0, A=0.3, m=0.5, fm=400.0)
imp = dict(fc=6290.0, A=0.25, impulse_every=900)
two = dict(fc=6290.0, A=0.25, f2=7290.0, a2=0.125)

# name -> (n_channels, n_blocks, setters, signal)
CASES = {
    "usb_nb_off_agc_off": (3, 6, [S("setDemodMode", USB), S("disableNoiseBlanker"), S("disableAGC")], tone),
    "usb_nb_off": (3, 6, [S("setDemodMode", USB), S("disableNoiseBlanker")], tone),
    "usb_default_nb": (3, 8, [S("setDemodMode", USB)], tone),
    "usb_c2": (9, 10, [S("setDemodMode", USB), S("enableAudioFilter")], imp),            # BASELINE config 2 settings
    "lsb_c2": (2, 8, [S("setDemodMode", LSB), S("enableAudioFilter")], imp),
    "lsb_defaults_only": (2, 6, [], dict(fc=8390.0 - 700, A=0.2)),                       # power-on state: LSB
    "cw_usb": (2, 8, [S("setDemodMode", CW_USB), S("enableAudioFilter"), S("setAudioFilter", 1)], dict(fc=6390 + 700.0, A=0.2)),
    "cw_lsb": (2, 8, [S("setDemodMode", CW_LSB), S("setNoiseBlankerThresholdDb", 10.0)], dict(fc=7390 - 700.0, A=0.2)),
    "wspr_sketch": (2, 8, [S("enableAGC"), S("setAGCmode", 2), S("disableALSfilter"), S("disableNoiseBlanker"),
                           S("setNoiseBlankerThresholdDb", 10.0), S("setInputGain", 1.0), S("setOutputGain", 0.5),
                           S("setIQgainBalance", 1.020), S("setAudioFilter", 2), S("setDemodMode", WSPR), S("setMute", 0)],
                    dict(fc=6890.0, A=0.02, noise=0.05)),                                   # BareBonesWSPR.ino:87-102,129
    "wspr_audio_filter_typo_table": (2, 8, [S("setDemodMode", WSPR), S("setAudioFilter", 2), S("enableAudioFilter"),
                                            S("disableNoiseBlanker")], dict(fc=6890.0, A=0.05, noise=0.02)),
    "am_default": (2, 8, [S("setDemodMode", AM)], am),                                      # BASELINE config 1 settings
    "am_nb10": (2, 10, [S("setDemodMode", AM), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", 0)], am),
    "sam_c3": (4, 12, [S("setDemodMode", SAM), S("setNoiseBlankerThresholdDb", 10.0), S("enableAudioFilter"), S("setAudioFilter", 0)], am),
    "sam_offset_carriers": (7, 14, [S("setDemodMode", SAM), S("setNoiseBlankerThresholdDb", 10.0)],
                            dict(fc=[6890.0 + (c % 7 - 3) * 50.0 for c in range(7)], A=0.3, m=0.5, fm=400.0)),
    "sam_default_unlocked": (2, 8, [S("setDemodMode", SAM)], am),
    "sam_far_carrier_never_locks": (2, 10, [S("setDemodMode", SAM), S("disableNoiseBlanker")], dict(fc=9500.0, A=0.3, m=0.5)),
    "usb_als_notch": (2, 8, [S("setDemodMode", USB), S("setNoiseBlankerThresholdDb", 10.0), S("enableALSfilter")], two),
    "usb_als_peak_static": (2, 6, [S("setDemodMode", USB), S("disableNoiseBlanker"), S("enableALSfilter"), S("setALSfilterPeak"),
                                   S("setALSfilterStatic")], tone),
    "usb_als_peak_adaptive": (2, 8, [S("setDemodMode", USB), S("disableNoiseBlanker"), S("enableALSfilter"), S("setALSfilterPeak")], two),
    "usb_als_params": (2, 8, [S("setDemodMode", USB), S("disableNoiseBlanker"), S("enableALSfilter"),
                              S("setALSfilterParams", 100, 0.25, 7.0)], two),
    "usb_als_m_plus_delay_over_128": (2, 6, [S("setDemodMode", USB), S("disableNoiseBlanker"), S("enableALSfilter"),
                                             S("setALSfilterParams", 128, 0.05, 20.0)], two),
    "mixed_modes_als_c4": (21, 10, [S("setNoiseBlankerThresholdDb", 10.0), S("enableALSfilter")] +
                           [S("setDemodMode", m, sel=(lambda c, m=m: c % 7 == m)) for m in range(7)],
                           dict(fc=6890.0 - 300, A=0.3, m=0.4, f2=7500.0, a2=0.1)),     # BASELINE config 4 settings
    "muted": (2, 4, [S("setDemodMode", USB), S("setMute", 1)], tone),
    "gains": (2, 6, [S("setDemodMode", USB), S("setInputGain", 3.3), S("setOutputGain", 0.9), S("setAGCstaticGain", 25.0)], tone),
    "int16_wrap_no_agc": (2, 6, [S("setDemodMode", USB), S("disableAGC"), S("disableNoiseBlanker"), S("setInputGain", 10.0),
                                 S("setOutputGain", 4.0)], dict(fc=6290.0, A=0.9)),       # (int) -> int16 wraps, AudioSDR.cpp:160
    "agc_fast_thresh": (2, 8, [S("setDemodMode", USB), S("setAGCmode", 1), S("setAGCthreshold", -40.0), S("setAGCslope", 0.3),
                               S("setAGCkneeWidth", 6.0)], tone),
    "agc_slow_hang": (2, 12, [S("setDemodMode", USB), S("setAGCmode", 3), S("disableNoiseBlanker")], dict(fc=6290.0, A=0.3, m=0.9, fm=120.0)),
    "nb_ratio_threshold_impulses": (3, 12, [S("setDemodMode", USB), S("setNoiseBlankerThreshold", 3.0)], dict(fc=6290.0, A=0.1, impulse_every=333)),
    "silence": (2, 5, [S("setDemodMode", USB)], dict(fc=6290.0, A=0.0, noise=0.0)),        # x = 0 path of fast_sqrt, atan2(0,0)
    "sam_silence": (2, 5, [S("setDemodMode", SAM)], dict(fc=6290.0, A=0.0, noise=0.0)),
    "full_scale": (2, 5, [S("setDemodMode", AM)], dict(fc=6890.0, A=0.99, noise=0.0)),
}
