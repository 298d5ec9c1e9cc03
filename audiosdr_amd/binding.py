"""ctypes binding of libasdr_hip.so (include/asdr.h).  Host-side mirror of the reference's
`class AudioSDR` (SRC/AudioSDRlib/AudioSDR.h:75-156) for a batch of channels.

Every reference method `sdr.foo(args)` becomes `batch.foo(args, ch=ALL)`; getters take `ch`.
"""
import ctypes as C
import os

import numpy as np

BLOCK = 128
ALL = -1
STREAM_BATCH = (1 << 64) - 1     # ASDR_STREAM_BATCH ((void *)-1): the batch's own streams (include/asdr.h)
LSBmode, USBmode, CW_LSBmode, CW_USBmode, AMmode, SAMmode, WSPRmode = range(7)       # AudioSDR.h:44-50
(audioAM, audioCW, audioWSPR, audio2100, audio2300, audio2500, audio2700, audio2900, audio3100, audio3300,
 audioBypass) = range(11)                                                             # AudioSDR.h:56-66
AGCoff, AGCfast, AGCmedium, AGCslow = range(4)                                        # AudioSDR.h:68-71
TAPS = ["SCALED_I", "SCALED_Q", "NB_I", "NB_Q", "IF_I", "IF_Q", "MIX_I", "MIX_Q", "DEMOD", "AUDIO_FILT", "AGC", "ALS"]

_HERE = os.path.dirname(os.path.abspath(__file__))


class AsdrError(RuntimeError):
    pass


def library_path():
    return os.path.join(_HERE, "libasdr_hip.so")


def library_sha256(path=None):
    """sha256 of the built library: stamps measurement files (profiles/pmc_latest.json) with the build they were taken from."""
    import hashlib
    with open(path or library_path(), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


# name -> (extra argtypes after (batch, ch), restype)
_f, _i, _u = C.c_float, C.c_int, C.c_uint
_SETTERS_VOID = ["init", "enableAudioFilter", "disableAudioFilter", "enableALSfilter", "disableALSfilter",
                 "setALSfilterNotch", "setALSfilterPeak", "setALSfilterAdaptive", "setALSfilterStatic", "enableAGC",
                 "disableAGC", "enableNoiseBlanker", "disableNoiseBlanker"]
_SETTERS_F = ["setInputGain", "setIQgainBalance", "setOutputGain", "setAGCthreshold", "setAGCslope", "setAGCkneeWidth",
              "setAGCattackTime", "setAGCreleaseTime", "setAGChangTime", "setAGCstaticGain", "setNoiseBlankerThreshold",
              "setNoiseBlankerThresholdDb"]
_SETTERS_I = ["setMute", "setAudioFilter", "setAGCmode"]
_GETTERS_F = ["getTuningOffset", "getBPFlower", "getBPFupper", "getAGCthreshold", "getAGCslope", "getAGCkneeWidth",
              "getAGCattack", "getAGCrelease", "getAAGalphaAttack", "getAGCbetaAttack", "getAGCalphaRelease",
              "getAGCbetaRelease", "getAGCstaticGain", "getAMcarrierLevel", "getSAMfrequency"]
_GETTERS_I = ["getMute", "getAudioFilter", "ALSfilterIsEnabled", "ALSfilterIsNotch", "ALSfilterIsPeak",
              "ALSfilterIsAdaptive", "AGCisEnabled", "AGCisActive", "NoiseBlankerisEnabled", "NoiseBlankerDetection",
              "getSAMphaseLockStatus"]

# every symbol include/asdr.h declares (checked by the CPU test-suite against the built library)
EXPORTS = (["asdr_create", "asdr_destroy", "asdr_last_error", "asdr_n_channels", "asdr_update", "asdr_update_device",
            "asdr_synchronize", "asdr_setDemodMode", "asdr_getDemodMode", "asdr_setALSfilterParams", "asdr_getAGClookup",
            "asdr_read_status", "asdr_control_plane_flush", "asdr_get_chain_constants", "asdr_stream_pipeline_launches", "asdr_schedule_layout", "asdr_set_exact_unknown_mode", "asdr_get_exact_unknown_mode", "asdr_stream_pipeline_recoveries", "asdr_stream_pipeline_max_groups", "asdr_set_stream_fir_helpers", "asdr_stream_pipeline_h3_calls", "asdr_set_stream_pipeline", "asdr_set_sam_launch_form", "asdr_set_als_launch_form", "asdr_debug_set_stream_spin_limit", "asdr_debug_set_stream_max_groups", "asdr_enable_taps", "asdr_read_taps", "asdr_last_kernel_ms", "asdr_version",
            "asdr_kernel_timing_begin", "asdr_kernel_timing_end", "asdr_set_launch_timing", "asdr_region_timing_begin", "asdr_region_timing_end", "asdr_update_device_strided", "asdr_capture_open",
            "asdr_capture_close", "asdr_capture_capacity", "asdr_capture_position", "asdr_capture_rewind",
            "asdr_capture_device_ptr", "asdr_capture_update_device", "asdr_capture_read",
            "asdr_create_sharded", "asdr_n_shards", "asdr_shard", "asdr_shard_first_channel", "asdr_shard_device",
            "asdr_host_alloc", "asdr_host_free", "asdr_host_register", "asdr_host_unregister", "asdr_set_host_chunks",
            "asdr_host_path_info", "asdr_stream_pipeline_alloc_failures", "asdr_set_launch_split", "asdr_order_after", "asdr_order_before", "asdr_lane_calls", "asdr_sam_role_calls", "asdr_sam_chunk_calls", "asdr_als_role_calls", "asdr_set_lanes"] +
           ["asdr_" + n for n in _SETTERS_VOID + _SETTERS_F + _SETTERS_I + _GETTERS_F + _GETTERS_I])

_lib = None


def host_autopin(on=-1):
    """Switch the auto-pinning of recurring pageable caller buffers (include/asdr.h) on / off for the process; returns the previous setting."""
    return int(load_library().asdr_host_autopin(int(on)))


def host_autopin_clear():
    load_library().asdr_host_autopin_clear()


def host_autopin_info():
    out = (C.c_long * 4)()
    load_library().asdr_host_autopin_info(out)
    return {"registered_now": int(out[0]), "registrations": int(out[1]), "released": int(out[2]), "refused": int(out[3])}


def kernels_launched(reset=False):
    """{kernel name: launches since the last reset} for the kernels that were launched at all (process-wide census, include/asdr.h)."""
    L = load_library()
    out = {}
    for i in range(L.asdr_kernels_count()):
        n = int(L.asdr_kernels_launches(i))
        if n:
            out[L.asdr_kernels_name(i).decode()] = n
    if reset:
        L.asdr_kernels_launches_reset()
    return out


def load_library(path=None):
    """dlopen libasdr_hip.so.  Raises AsdrError (loudly) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or library_path()
    if not os.path.exists(p):
        raise AsdrError("%s not found: build it with `python -m audiosdr_amd.build` (hipcc, gfx950). "
                        "There is no CPU fallback." % p)
    L = C.CDLL(p)
    vp, i16p = C.c_void_p, C.POINTER(C.c_int16)
    L.asdr_create.argtypes = [_i, _i]; L.asdr_create.restype = vp
    L.asdr_destroy.argtypes = [vp]; L.asdr_destroy.restype = None
    L.asdr_last_error.restype = C.c_char_p
    L.asdr_version.restype = C.c_char_p
    L.asdr_set_pool_priority.argtypes = [_i]; L.asdr_set_pool_priority.restype = _i
    L.asdr_host_autopin.argtypes = [_i]; L.asdr_host_autopin.restype = _i
    L.asdr_host_autopin_clear.restype = None
    L.asdr_host_autopin_info.argtypes = [C.POINTER(C.c_long)]; L.asdr_host_autopin_info.restype = _i
    L.asdr_kernels_count.restype = _i
    L.asdr_kernels_name.argtypes = [_i]; L.asdr_kernels_name.restype = C.c_char_p
    L.asdr_kernels_launches.argtypes = [_i]; L.asdr_kernels_launches.restype = C.c_ulonglong
    L.asdr_kernels_launches_reset.restype = None
    L.asdr_n_channels.argtypes = [vp]; L.asdr_n_channels.restype = _i
    L.asdr_update.argtypes = [vp, i16p, i16p, i16p, _i]; L.asdr_update.restype = _i
    L.asdr_update_device.argtypes = [vp, vp, vp, vp, _i, vp]; L.asdr_update_device.restype = _i
    L.asdr_synchronize.argtypes = [vp]; L.asdr_synchronize.restype = _i
    L.asdr_update_device_strided.argtypes = [vp, vp, vp, vp, _i, C.c_long, C.c_long, vp]; L.asdr_update_device_strided.restype = _i
    L.asdr_capture_open.argtypes = [vp, C.c_long]; L.asdr_capture_open.restype = _i
    for n in ("asdr_capture_close", "asdr_capture_rewind"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = _i
    for n in ("asdr_capture_capacity", "asdr_capture_position"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = C.c_long
    L.asdr_capture_device_ptr.argtypes = [vp]; L.asdr_capture_device_ptr.restype = vp
    L.asdr_capture_update_device.argtypes = [vp, vp, vp, _i, C.c_long, vp]; L.asdr_capture_update_device.restype = _i
    L.asdr_capture_read.argtypes = [vp, _i, C.c_long, C.c_long, i16p]; L.asdr_capture_read.restype = _i
    L.asdr_last_kernel_ms.argtypes = [vp]; L.asdr_last_kernel_ms.restype = _f
    for n in _SETTERS_VOID:
        fn = getattr(L, "asdr_" + n); fn.argtypes = [vp, _i]; fn.restype = None
    for n in _SETTERS_F:
        fn = getattr(L, "asdr_" + n); fn.argtypes = [vp, _i, _f]; fn.restype = None
    for n in _SETTERS_I:
        fn = getattr(L, "asdr_" + n); fn.argtypes = [vp, _i, _i]; fn.restype = None
    for n in _GETTERS_F:
        fn = getattr(L, "asdr_" + n); fn.argtypes = [vp, _i]; fn.restype = _f
    for n in _GETTERS_I:
        fn = getattr(L, "asdr_" + n); fn.argtypes = [vp, _i]; fn.restype = _i
    L.asdr_setDemodMode.argtypes = [vp, _i, _i]; L.asdr_setDemodMode.restype = _f
    L.asdr_getDemodMode.argtypes = [vp, _i]; L.asdr_getDemodMode.restype = C.c_int16
    L.asdr_setALSfilterParams.argtypes = [vp, _i, _u, _f, _f]; L.asdr_setALSfilterParams.restype = None
    L.asdr_getAGClookup.argtypes = [vp, _i, _i]; L.asdr_getAGClookup.restype = _f
    i32p, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.asdr_read_status.argtypes = [vp, i32p, i32p, i32p, fp, fp]; L.asdr_read_status.restype = _i
    if path is None or hasattr(L, "asdr_get_chain_constants"):
        L.asdr_get_chain_constants.argtypes = [vp, _i, C.POINTER(C.c_float)]; L.asdr_get_chain_constants.restype = C.c_uint
    if path is None or hasattr(L, "asdr_control_plane_flush"):   # older builds timed by tools/ablate.py lack it
        L.asdr_control_plane_flush.argtypes = [vp, C.POINTER(C.c_longlong)]; L.asdr_control_plane_flush.restype = _i
    L.asdr_enable_taps.argtypes = [vp, _i]; L.asdr_enable_taps.restype = _i
    L.asdr_read_taps.argtypes = [vp, fp]; L.asdr_read_taps.restype = _i
    L.asdr_kernel_timing_begin.argtypes = [vp, _i]; L.asdr_kernel_timing_begin.restype = _i
    if path is None or hasattr(L, "asdr_stream_pipeline_launches"):
        L.asdr_stream_pipeline_launches.argtypes = [vp]; L.asdr_stream_pipeline_launches.restype = C.c_long
    if path is None or hasattr(L, "asdr_set_stream_pipeline"):
        L.asdr_stream_pipeline_recoveries.argtypes = [vp]; L.asdr_stream_pipeline_recoveries.restype = C.c_long
        L.asdr_stream_pipeline_max_groups.argtypes = [vp]; L.asdr_stream_pipeline_max_groups.restype = _i
        if hasattr(L, "asdr_set_stream_fir_helpers"):
            L.asdr_set_stream_fir_helpers.argtypes = [vp, _i]; L.asdr_set_stream_fir_helpers.restype = _i
            L.asdr_stream_pipeline_h3_calls.argtypes = [vp]; L.asdr_stream_pipeline_h3_calls.restype = C.c_long
        L.asdr_set_stream_pipeline.argtypes = [vp, _i]; L.asdr_set_stream_pipeline.restype = _i
        L.asdr_set_sam_launch_form.argtypes = [vp, _i, _i]; L.asdr_set_sam_launch_form.restype = _i
        if hasattr(L, "asdr_set_als_launch_form"):
            L.asdr_set_als_launch_form.argtypes = [vp, _i]; L.asdr_set_als_launch_form.restype = _i
        L.asdr_debug_set_stream_spin_limit.argtypes = [vp, _u]; L.asdr_debug_set_stream_spin_limit.restype = _i
        L.asdr_debug_set_stream_max_groups.argtypes = [vp, _i]; L.asdr_debug_set_stream_max_groups.restype = _i
    if path is None or hasattr(L, "asdr_set_exact_unknown_mode"):
        L.asdr_set_exact_unknown_mode.argtypes = [vp, _i]; L.asdr_set_exact_unknown_mode.restype = _i
        L.asdr_get_exact_unknown_mode.argtypes = [vp]; L.asdr_get_exact_unknown_mode.restype = _i
    if path is None or hasattr(L, "asdr_schedule_layout"):
        L.asdr_schedule_layout.argtypes = [vp, C.POINTER(C.c_int)]; L.asdr_schedule_layout.restype = _i
    if path is None or hasattr(L, "asdr_region_timing_begin"):
        L.asdr_set_launch_timing.argtypes = [vp, _i]; L.asdr_set_launch_timing.restype = _i
        L.asdr_region_timing_begin.argtypes = [vp, vp]; L.asdr_region_timing_begin.restype = _i
        L.asdr_region_timing_end.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_long)]; L.asdr_region_timing_end.restype = _i
    L.asdr_kernel_timing_end.argtypes = [vp, fp, _i]; L.asdr_kernel_timing_end.restype = _i
    if path is None or hasattr(L, "asdr_create_sharded"):   # (older builds timed by tools/ablate.py lack these)
        L.asdr_create_sharded.argtypes = [_i, _i, C.POINTER(C.c_int)]; L.asdr_create_sharded.restype = vp
        L.asdr_n_shards.argtypes = [vp]; L.asdr_n_shards.restype = _i
        L.asdr_shard.argtypes = [vp, _i]; L.asdr_shard.restype = vp
        L.asdr_shard_first_channel.argtypes = [vp, _i]; L.asdr_shard_first_channel.restype = _i
        L.asdr_shard_device.argtypes = [vp, _i]; L.asdr_shard_device.restype = _i
        L.asdr_host_alloc.argtypes = [C.c_size_t]; L.asdr_host_alloc.restype = vp
        L.asdr_host_free.argtypes = [vp]; L.asdr_host_free.restype = None
        L.asdr_host_register.argtypes = [vp, C.c_size_t]; L.asdr_host_register.restype = _i
        L.asdr_host_unregister.argtypes = [vp]; L.asdr_host_unregister.restype = _i
        L.asdr_set_host_chunks.argtypes = [vp, _i]; L.asdr_set_host_chunks.restype = _i
        L.asdr_host_path_info.argtypes = [vp, C.POINTER(C.c_int)]; L.asdr_host_path_info.restype = _i
        L.asdr_stream_pipeline_alloc_failures.argtypes = [vp]; L.asdr_stream_pipeline_alloc_failures.restype = C.c_long
        L.asdr_set_launch_split.argtypes = [vp, _i, _i]; L.asdr_set_launch_split.restype = _i
        L.asdr_order_after.argtypes = [vp, vp]; L.asdr_order_after.restype = _i
        L.asdr_order_before.argtypes = [vp, vp]; L.asdr_order_before.restype = _i
        L.asdr_lane_calls.argtypes = [vp]; L.asdr_lane_calls.restype = C.c_long
        L.asdr_sam_role_calls.argtypes = [vp]; L.asdr_sam_role_calls.restype = C.c_long
        L.asdr_als_role_calls.argtypes = [vp]; L.asdr_als_role_calls.restype = C.c_long
        L.asdr_sam_chunk_calls.argtypes = [vp]; L.asdr_sam_chunk_calls.restype = C.c_long
        L.asdr_set_lanes.argtypes = [vp, _i, _i]; L.asdr_set_lanes.restype = _i
    if path is None or hasattr(L, "asdr_lanes_overlap_probe"):   # round 5
        L.asdr_lanes_overlap_probe.argtypes = [vp]; L.asdr_lanes_overlap_probe.restype = _i
        L.asdr_lanes_enabled.argtypes = [vp]; L.asdr_lanes_enabled.restype = _i
        L.asdr_stream_pipeline_headroom_refusals.argtypes = [vp]; L.asdr_stream_pipeline_headroom_refusals.restype = C.c_long
    if path is None:
        _lib = L
    return L


class AudioSDRBatch:
    """N independent AudioSDR channels on one MI355X.  Method names follow the reference class."""

    def __init__(self, n_channels, device=0, devices=None, _handle=None, _owner=None):
        """`device`: one HIP device ordinal (asdr_create).  `devices`: a list of ordinals, one per shard -> a sharded batch
        (asdr_create_sharded): shard g owns channels [g*C/G, (g+1)*C/G) on devices[g]; every method keeps taking GLOBAL channel
        indices."""
        self._L = load_library()
        self._owner = _owner          # a shard view keeps its sharded batch alive and never destroys the handle
        if _handle is not None:
            self._h = _handle
        elif devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            self._h = self._L.asdr_create_sharded(int(n_channels), len(devices), devs)
        else:
            self._h = self._L.asdr_create(int(n_channels), int(device))
        if not self._h:
            raise AsdrError("asdr_create failed: " + self._L.asdr_last_error().decode())
        self.n_channels = int(n_channels)
        self.device = int(device) if devices is None else None

    def close(self):
        if getattr(self, "_h", None):
            if self._owner is None:
                self._L.asdr_destroy(self._h)
            self._h = None

    # ---- sharded batches ----
    @property
    def n_shards(self):
        return int(self._L.asdr_n_shards(self._h))

    def shard(self, g):
        """Shard g as a batch of its own (local channel indices, the shard's device): for device-resident rows on several GPUs."""
        h = self._L.asdr_shard(self._h, int(g))
        if not h:
            raise AsdrError("no such shard")
        lo, hi = self.shard_range(g)
        v = AudioSDRBatch(hi - lo, device=self._L.asdr_shard_device(self._h, int(g)), _handle=h, _owner=self)
        return v

    def shard_range(self, g):
        return int(self._L.asdr_shard_first_channel(self._h, int(g))), int(self._L.asdr_shard_first_channel(self._h, int(g) + 1))

    def order_after(self, stream):
        self._chk(self._L.asdr_order_after(self._h, C.c_void_p(stream)))

    def order_before(self, stream):
        self._chk(self._L.asdr_order_before(self._h, C.c_void_p(stream)))

    def lane_calls(self):
        return int(self._L.asdr_lane_calls(self._h))

    def sam_role_calls(self):
        """Multi-block calls that ran the SAM roles on three chained streams (include/asdr.h)."""
        return int(self._L.asdr_sam_role_calls(self._h))

    def sam_chunk_calls(self):
        """... of which ran the roles in chunks of 8 blocks per launch (include/asdr.h)."""
        return int(self._L.asdr_sam_chunk_calls(self._h))

    def als_role_calls(self):
        """Multi-block calls of a small short-ALS-filter bank that ran as chain | filter launches on two chained streams (include/asdr.h)."""
        return int(self._L.asdr_als_role_calls(self._h))

    def lanes_overlap_probe(self):
        """1 = the pool's streams run concurrently on this device, 0 = they share a hardware queue (lanes default off), -1 = not probed."""
        return int(self._L.asdr_lanes_overlap_probe(self._h))

    def lanes_enabled(self):
        return bool(self._L.asdr_lanes_enabled(self._h))

    def stream_pipeline_headroom_refusals(self):
        return int(self._L.asdr_stream_pipeline_headroom_refusals(self._h))

    def set_lanes(self, on=True, min_waves=0):
        self._chk(self._L.asdr_set_lanes(self._h, 1 if on else 0, int(min_waves)))

    def set_launch_split(self, pieces, min_waves=0):
        self._chk(self._L.asdr_set_launch_split(self._h, int(pieces), int(min_waves)))

    def set_host_chunks(self, chunks):
        self._chk(self._L.asdr_set_host_chunks(self._h, int(chunks)))

    def host_path_info(self):
        out = (C.c_int * 2)()
        self._chk(self._L.asdr_host_path_info(self._h, out))
        return {"chunks": int(out[0]), "pinned": bool(out[1])}

    def stream_pipeline_alloc_failures(self):
        return int(self._L.asdr_stream_pipeline_alloc_failures(self._h))

    def update_into(self, I, Q, out):
        """asdr_update on caller-owned arrays (e.g. views of pinned memory from host_alloc): no allocation, no copy here."""
        n_blocks = I.size // (self.n_channels * BLOCK)
        p = C.POINTER(C.c_int16)
        self._chk(self._L.asdr_update(self._h, I.ctypes.data_as(p), Q.ctypes.data_as(p), out.ctypes.data_as(p), n_blocks))
        return out

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AsdrError(self._L.asdr_last_error().decode())

    # ---- hot path ----
    def update(self, I, Q):
        """I, Q: int16 [channels][blocks][128] (host).  Returns int16 audio of the same shape."""
        I = np.ascontiguousarray(I, dtype=np.int16)
        Q = np.ascontiguousarray(Q, dtype=np.int16)
        assert I.shape == Q.shape and I.size % (self.n_channels * BLOCK) == 0
        n_blocks = I.size // (self.n_channels * BLOCK)
        out = np.empty((self.n_channels, n_blocks, BLOCK), dtype=np.int16)
        p = C.POINTER(C.c_int16)
        self._chk(self._L.asdr_update(self._h, I.ctypes.data_as(p), Q.ctypes.data_as(p), out.ctypes.data_as(p), n_blocks))
        return out

    def update_device(self, dI, dQ, dOut, n_blocks, stream=0):
        """Raw device pointers (ints), asynchronous on `stream` (a hipStream_t handle as int)."""
        self._chk(self._L.asdr_update_device(self._h, C.c_void_p(dI), C.c_void_p(dQ), C.c_void_p(dOut), int(n_blocks),
                                             C.c_void_p(stream)))

    def update_device_strided(self, dI, dQ, dOut, n_blocks, in_stride_blocks, out_stride_blocks, stream=0):
        """As update_device, with explicit row strides (in blocks) of the I/Q rows and of the out rows."""
        self._chk(self._L.asdr_update_device_strided(self._h, C.c_void_p(dI), C.c_void_p(dQ), C.c_void_p(dOut), int(n_blocks),
                                                     int(in_stride_blocks), int(out_stride_blocks), C.c_void_p(stream)))

    def synchronize(self):
        self._chk(self._L.asdr_synchronize(self._h))

    # ---- capture sink: one contiguous audio row per channel in HBM (include/asdr.h) ----
    def capture_open(self, capacity_blocks):
        self._chk(self._L.asdr_capture_open(self._h, int(capacity_blocks)))

    def capture_close(self):
        self._chk(self._L.asdr_capture_close(self._h))

    def capture_rewind(self):
        self._chk(self._L.asdr_capture_rewind(self._h))

    @property
    def capture_position(self):
        return int(self._L.asdr_capture_position(self._h))

    @property
    def capture_capacity(self):
        return int(self._L.asdr_capture_capacity(self._h))

    def capture_device_ptr(self):
        return int(self._L.asdr_capture_device_ptr(self._h) or 0)

    def capture_update_device(self, dI, dQ, n_blocks, in_stride_blocks=None, stream=0):
        self._chk(self._L.asdr_capture_update_device(self._h, C.c_void_p(dI), C.c_void_p(dQ), int(n_blocks),
                                                     int(n_blocks if in_stride_blocks is None else in_stride_blocks), C.c_void_p(stream)))

    def capture_read(self, ch, first_block=0, n_blocks=None):
        """int16 [n_blocks*128] of channel `ch`, starting at block `first_block` of its capture row."""
        if n_blocks is None:
            n_blocks = self.capture_position - first_block
        out = np.empty(int(n_blocks) * BLOCK, dtype=np.int16)
        self._chk(self._L.asdr_capture_read(self._h, int(ch), int(first_block), int(n_blocks), out.ctypes.data_as(C.POINTER(C.c_int16))))
        return out

    def chain_constants(self, ch=0):
        """(hang count of `ch`, the 12 derived float constants of include/asdr.h asdr_get_chain_constants)."""
        out = (C.c_float * 12)()
        hang = int(self._L.asdr_get_chain_constants(self._h, int(ch), out))
        return hang, np.array(list(out), dtype=np.float32)

    def control_plane_flush(self):
        """Host half of the pre-launch flush on a control-plane-only batch: what the next update() would refill / rebuild."""
        st = (C.c_longlong * 4)()
        self._chk(self._L.asdr_control_plane_flush(self._h, st))
        return {"rows_refilled": int(st[0]), "schedule_rebuilt": bool(st[1]), "waves_plain": int(st[2]) & 0x1FFFFF,
                "waves_sam": (int(st[2]) >> 21) & 0x1FFFFF, "waves_als": (int(st[2]) >> 42) & 0x1FFFFF, "agc_tables_alive": int(st[3])}

    def last_kernel_ms(self):
        return float(self._L.asdr_last_kernel_ms(self._h))

    def set_exact_unknown_mode(self, on=True):
        """Keep every block's post-ALS audio row so that unknown mode values re-process it as the reference does (default on)."""
        self._chk(self._L.asdr_set_exact_unknown_mode(self._h, 1 if on else 0))

    def stream_pipeline_launches(self):
        return int(self._L.asdr_stream_pipeline_launches(self._h))

    def stream_pipeline_recoveries(self):
        """Pipeline calls whose bounded wait ran out and that were re-run on the in-kernel block loop from the snapshot (asdr.h)."""
        return int(self._L.asdr_stream_pipeline_recoveries(self._h))

    def stream_pipeline_max_groups(self):
        return int(self._L.asdr_stream_pipeline_max_groups(self._h))

    def set_stream_fir_helpers(self, mode):
        """-1: three FIR helper waves per role-2 workgroup of the block pipeline while every workgroup has a compute unit to itself (default);
        0: always one; 1: three wherever they are resident (asdr.h)."""
        self._chk(self._L.asdr_set_stream_fir_helpers(self._h, int(mode)))

    def stream_pipeline_h3_calls(self):
        return int(self._L.asdr_stream_pipeline_h3_calls(self._h))

    def set_stream_pipeline(self, on=True):
        self._chk(self._L.asdr_set_stream_pipeline(self._h, 1 if on else 0))

    def set_sam_launch_form(self, fused=False, split_min_channels=0):
        self._chk(self._L.asdr_set_sam_launch_form(self._h, 1 if fused else 0, int(split_min_channels)))

    def set_als_launch_form(self, split_min_channels=0):
        self._chk(self._L.asdr_set_als_launch_form(self._h, int(split_min_channels)))

    def debug_set_stream_max_groups(self, groups):
        self._chk(self._L.asdr_debug_set_stream_max_groups(self._h, int(groups)))

    def debug_set_stream_spin_limit(self, polls):
        self._chk(self._L.asdr_debug_set_stream_spin_limit(self._h, int(polls)))

    def schedule_layout(self):
        """Slots (8 per wave) per kernel kind, the remainders' sub-range and its kernel kind, the SAM launch form (asdr.h)."""
        out = (C.c_int * 8)()
        self._chk(self._L.asdr_schedule_layout(self._h, out))
        v = list(out)
        return {"plain": v[0], "sam": v[1], "als_long": v[2], "als_compact": v[3], "sam_als": v[4], "remainders": v[5],
                "remainder_kind": v[6], "sam_three_launches": bool(v[7] & 1), "als_two_launches": bool(v[7] & 2)}

    def set_launch_timing(self, on):
        self._chk(self._L.asdr_set_launch_timing(self._h, 1 if on else 0))

    def region_timing_begin(self, stream=None):
        self._chk(self._L.asdr_region_timing_begin(self._h, stream))

    def region_timing_end(self):
        """(elapsed milliseconds, update calls) since region_timing_begin."""
        ms, n = C.c_float(0.0), C.c_long(0)
        self._chk(self._L.asdr_region_timing_end(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def kernel_timing_begin(self, max_launches):
        self._chk(self._L.asdr_kernel_timing_begin(self._h, int(max_launches)))

    def kernel_timing_end(self, cap):
        ms = np.zeros(int(cap), dtype=np.float32)
        n = self._L.asdr_kernel_timing_end(self._h, ms.ctypes.data_as(C.POINTER(C.c_float)), int(cap))
        if n < 0:
            raise AsdrError(self._L.asdr_last_error().decode())
        return ms[:n]

    # ---- batch-only helpers ----
    def read_status(self):
        n = self.n_channels
        a = [np.zeros(n, dtype=np.int32) for _ in range(3)]
        f = [np.zeros(n, dtype=np.float32) for _ in range(2)]
        ip, fp = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        self._chk(self._L.asdr_read_status(self._h, a[0].ctypes.data_as(ip), a[1].ctypes.data_as(ip), a[2].ctypes.data_as(ip),
                                           f[0].ctypes.data_as(fp), f[1].ctypes.data_as(fp)))
        return {"agc_active": a[0], "nb_detected": a[1], "sam_locked": a[2], "sam_frequency": f[0], "am_carrier": f[1]}

    def enable_taps(self, on=True):
        self._chk(self._L.asdr_enable_taps(self._h, 1 if on else 0))

    def read_taps(self):
        t = np.zeros((len(TAPS), self.n_channels, BLOCK), dtype=np.float32)
        self._chk(self._L.asdr_read_taps(self._h, t.ctypes.data_as(C.POINTER(C.c_float))))
        return {name: t[i] for i, name in enumerate(TAPS)}

    # ---- reference control surface ----
    def setDemodMode(self, mode, ch=ALL):
        return float(self._L.asdr_setDemodMode(self._h, ch, int(mode)))

    def getDemodMode(self, ch=0):
        return int(self._L.asdr_getDemodMode(self._h, ch))

    def setALSfilterParams(self, m, Lambda, Delay, ch=ALL):
        self._L.asdr_setALSfilterParams(self._h, ch, int(m), float(Lambda), float(Delay))

    def getAGClookup(self, i, ch=0):
        return float(self._L.asdr_getAGClookup(self._h, ch, int(i)))


def host_alloc(shape, dtype=np.int16):
    """A numpy array in page-locked host memory (asdr_host_alloc): asdr_update DMA-copies from / to it without staging.  Free it
    with host_free(arr) when done (the array must not be used afterwards)."""
    L = load_library()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = L.asdr_host_alloc(n)
    if not p:
        raise AsdrError("asdr_host_alloc failed: " + L.asdr_last_error().decode())
    buf = (C.c_char * n).from_address(p)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    _PINNED[arr.__array_interface__["data"][0]] = p
    return arr


_PINNED = {}


def host_free(arr):
    p = _PINNED.pop(arr.__array_interface__["data"][0], None)
    if p:
        load_library().asdr_host_free(p)


def _add_methods():
    def mk_void(n):
        def f(self, ch=ALL):
            getattr(self._L, "asdr_" + n)(self._h, ch)
        return f

    def mk_set(n, conv):
        def f(self, v, ch=ALL):
            getattr(self._L, "asdr_" + n)(self._h, ch, conv(v))
        return f

    def mk_get(n, conv):
        def f(self, ch=0):
            return conv(getattr(self._L, "asdr_" + n)(self._h, ch))
        return f

    for n in _SETTERS_VOID:
        setattr(AudioSDRBatch, n, mk_void(n))
    for n in _SETTERS_F:
        setattr(AudioSDRBatch, n, mk_set(n, float))
    for n in _SETTERS_I:
        setattr(AudioSDRBatch, n, mk_set(n, int))
    for n in _GETTERS_F:
        setattr(AudioSDRBatch, n, mk_get(n, float))
    for n in _GETTERS_I:
        setattr(AudioSDRBatch, n, mk_get(n, int))


_add_methods()
