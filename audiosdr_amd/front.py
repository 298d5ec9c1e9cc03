"""ctypes mirror of include/asdr_front.h: the batched AudioStream blocks around the hot path, with the reference's
class and method names (AudioSDRpreProcessor.h:49-70, AudioIQgenerator.h:48-59, AudioGrabberComplex256.h:44-52).
Same rules as binding.py: the work happens in libasdr_hip.so on the GPU; there is no CPU fallback."""
import ctypes as C

import numpy as np

from .binding import ALL, BLOCK, AsdrError, load_library

NO_DEVICE = -1

FRONT_EXPORTS = ["asdr_pre_create", "asdr_pre_destroy", "asdr_pre_n_channels", "asdr_pre_update", "asdr_pre_update_device",
                 "asdr_pre_synchronize", "asdr_pre_startAutoI2SerrorDetection", "asdr_pre_stopAutoI2SerrorDetection",
                 "asdr_pre_getAutoI2SerrorDetectionStatus", "asdr_pre_setI2SerrorCompensation",
                 "asdr_pre_getI2SerrorCompensation", "asdr_pre_swapIQ", "asdr_pre_read_state", "asdr_pre_last_kernel_ms",
                 "asdr_iqgen_create", "asdr_iqgen_destroy", "asdr_iqgen_n_channels", "asdr_iqgen_update",
                 "asdr_iqgen_update_device", "asdr_iqgen_synchronize", "asdr_iqgen_setGainBalance", "asdr_iqgen_last_kernel_ms",
                 "asdr_grab_create", "asdr_grab_destroy", "asdr_grab_n_channels", "asdr_grab_update", "asdr_grab_update_device",
                 "asdr_grab_newDataAvailable", "asdr_grab_grab", "asdr_grab_grab_all", "asdr_grab_device_ptr",
                 "asdr_grab_synchronize", "asdr_grab_power_spectrum", "asdr_grab_power_spectrum_device"]


class PreState(C.Structure):
    """asdr_pre_state_t"""
    _fields_ = [("correction", C.c_int16), ("saved_sample", C.c_int16), ("failure_count", C.c_int16), ("success_count", C.c_int16),
                ("auto_detect", C.c_int32), ("swap", C.c_int32), ("max_line", C.c_int32), ("strong", C.c_int32),
                ("max_power", C.c_float), ("avg_power", C.c_float), ("ratio", C.c_float)]


PRE_STATE_DTYPE = np.dtype([("correction", "<i2"), ("saved_sample", "<i2"), ("failure_count", "<i2"), ("success_count", "<i2"),
                            ("auto_detect", "<i4"), ("swap", "<i4"), ("max_line", "<i4"), ("strong", "<i4"),
                            ("max_power", "<f4"), ("avg_power", "<f4"), ("ratio", "<f4")])
assert PRE_STATE_DTYPE.itemsize == C.sizeof(PreState) == 36

_typed = False


def _lib():
    global _typed
    L = load_library()
    if _typed:
        return L
    vp, i, lg, f, i16p = C.c_void_p, C.c_int, C.c_long, C.c_float, C.POINTER(C.c_int16)
    for pre in ("pre", "iqgen", "grab"):
        getattr(L, "asdr_%s_create" % pre).argtypes = [i, i]; getattr(L, "asdr_%s_create" % pre).restype = vp
        getattr(L, "asdr_%s_destroy" % pre).argtypes = [vp]; getattr(L, "asdr_%s_destroy" % pre).restype = None
        getattr(L, "asdr_%s_n_channels" % pre).argtypes = [vp]; getattr(L, "asdr_%s_n_channels" % pre).restype = i
        getattr(L, "asdr_%s_synchronize" % pre).argtypes = [vp]; getattr(L, "asdr_%s_synchronize" % pre).restype = i
    L.asdr_pre_update.argtypes = [vp, i16p, i16p, i]; L.asdr_pre_update.restype = i
    L.asdr_pre_update_device.argtypes = [vp, vp, vp, vp, vp, i, lg, lg, vp]; L.asdr_pre_update_device.restype = i
    for n in ("startAutoI2SerrorDetection", "stopAutoI2SerrorDetection"):
        getattr(L, "asdr_pre_" + n).argtypes = [vp, i]; getattr(L, "asdr_pre_" + n).restype = None
    L.asdr_pre_getAutoI2SerrorDetectionStatus.argtypes = [vp, i]; L.asdr_pre_getAutoI2SerrorDetectionStatus.restype = i
    L.asdr_pre_setI2SerrorCompensation.argtypes = [vp, i, i]; L.asdr_pre_setI2SerrorCompensation.restype = None
    L.asdr_pre_getI2SerrorCompensation.argtypes = [vp, i]; L.asdr_pre_getI2SerrorCompensation.restype = C.c_int16
    L.asdr_pre_swapIQ.argtypes = [vp, i, i]; L.asdr_pre_swapIQ.restype = None
    L.asdr_pre_read_state.argtypes = [vp, vp]; L.asdr_pre_read_state.restype = i
    L.asdr_pre_last_kernel_ms.argtypes = [vp]; L.asdr_pre_last_kernel_ms.restype = f
    L.asdr_iqgen_update.argtypes = [vp, i16p, i16p, i16p, i]; L.asdr_iqgen_update.restype = i
    L.asdr_iqgen_update_device.argtypes = [vp, vp, vp, vp, i, lg, lg, vp]; L.asdr_iqgen_update_device.restype = i
    L.asdr_iqgen_setGainBalance.argtypes = [vp, i, f]; L.asdr_iqgen_setGainBalance.restype = None
    L.asdr_iqgen_last_kernel_ms.argtypes = [vp]; L.asdr_iqgen_last_kernel_ms.restype = f
    L.asdr_grab_update.argtypes = [vp, i16p, i16p, i]; L.asdr_grab_update.restype = i
    L.asdr_grab_update_device.argtypes = [vp, vp, vp, i, lg, vp]; L.asdr_grab_update_device.restype = i
    L.asdr_grab_newDataAvailable.argtypes = [vp, i]; L.asdr_grab_newDataAvailable.restype = i
    L.asdr_grab_grab.argtypes = [vp, i, i16p]; L.asdr_grab_grab.restype = i
    L.asdr_grab_grab_all.argtypes = [vp, i16p]; L.asdr_grab_grab_all.restype = i
    L.asdr_grab_device_ptr.argtypes = [vp]; L.asdr_grab_device_ptr.restype = vp
    L.asdr_grab_power_spectrum.argtypes = [vp, C.POINTER(C.c_float)]; L.asdr_grab_power_spectrum.restype = i
    L.asdr_grab_power_spectrum_device.argtypes = [vp, vp, vp]; L.asdr_grab_power_spectrum_device.restype = i
    _typed = True
    return L


def _p16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


class _Batch:
    _prefix = ""

    def __init__(self, n_channels, device=0):
        self._L = _lib()
        self._h = getattr(self._L, "asdr_%s_create" % self._prefix)(int(n_channels), int(device))
        if not self._h:
            raise AsdrError("asdr_%s_create failed: %s" % (self._prefix, self._L.asdr_last_error().decode()))
        self.n_channels = int(n_channels)

    def close(self):
        if getattr(self, "_h", None):
            getattr(self._L, "asdr_%s_destroy" % self._prefix)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise AsdrError(self._L.asdr_last_error().decode())
        return rc

    def synchronize(self):
        self._chk(getattr(self._L, "asdr_%s_synchronize" % self._prefix)(self._h))

    def _blocks(self, a):
        a = np.ascontiguousarray(a, dtype=np.int16)
        assert a.size % (self.n_channels * BLOCK) == 0
        return a.reshape(self.n_channels, -1, BLOCK)


class AudioSDRpreProcessorBatch(_Batch):
    """N independent AudioSDRpreProcessor instances (AudioSDRpreProcessor.h:49-70)."""
    _prefix = "pre"

    def update(self, I, Q):
        """I, Q: int16 [channels][blocks][128] (host).  Returns the conditioned (I, Q); inputs are not modified."""
        I, Q = self._blocks(I).copy(), self._blocks(Q).copy()
        assert I.shape == Q.shape
        self._chk(self._L.asdr_pre_update(self._h, _p16(I), _p16(Q), I.shape[1]))
        return I, Q

    def update_device(self, dI, dQ, dIout, dQout, n_blocks, in_stride_blocks=None, out_stride_blocks=None, stream=0):
        self._chk(self._L.asdr_pre_update_device(self._h, C.c_void_p(dI), C.c_void_p(dQ), C.c_void_p(dIout), C.c_void_p(dQout),
                                                 int(n_blocks), int(in_stride_blocks or n_blocks), int(out_stride_blocks or n_blocks),
                                                 C.c_void_p(stream)))

    def last_kernel_ms(self):
        return float(self._L.asdr_pre_last_kernel_ms(self._h))

    def startAutoI2SerrorDetection(self, ch=ALL):
        self._L.asdr_pre_startAutoI2SerrorDetection(self._h, ch)

    def stopAutoI2SerrorDetection(self, ch=ALL):
        self._L.asdr_pre_stopAutoI2SerrorDetection(self._h, ch)

    def getAutoI2SerrorDetectionStatus(self, ch=0):
        return int(self._L.asdr_pre_getAutoI2SerrorDetectionStatus(self._h, ch))

    def setI2SerrorCompensation(self, correction, ch=ALL):
        self._L.asdr_pre_setI2SerrorCompensation(self._h, ch, int(correction))

    def getI2SerrorCompensation(self, ch=0):
        return int(self._L.asdr_pre_getI2SerrorCompensation(self._h, ch))

    def swapIQ(self, swap, ch=ALL):
        self._L.asdr_pre_swapIQ(self._h, ch, 1 if swap else 0)

    def read_state(self):
        """numpy structured array [n_channels] of asdr_pre_state_t."""
        st = np.zeros(self.n_channels, dtype=PRE_STATE_DTYPE)
        self._chk(self._L.asdr_pre_read_state(self._h, st.ctypes.data_as(C.c_void_p)))
        return st


class AudioIQgeneratorBatch(_Batch):
    """N independent AudioIQgenerator instances (AudioIQgenerator.h:48-59)."""
    _prefix = "iqgen"

    def update(self, x):
        """x: int16 [channels][blocks][128] real input (host).  Returns (I, Q) of the same shape."""
        x = self._blocks(x)
        I, Q = np.empty_like(x), np.empty_like(x)
        self._chk(self._L.asdr_iqgen_update(self._h, _p16(x), _p16(I), _p16(Q), x.shape[1]))
        return I, Q

    def update_device(self, dIn, dI, dQ, n_blocks, in_stride_blocks=None, out_stride_blocks=None, stream=0):
        self._chk(self._L.asdr_iqgen_update_device(self._h, C.c_void_p(dIn), C.c_void_p(dI), C.c_void_p(dQ), int(n_blocks),
                                                   int(in_stride_blocks or n_blocks), int(out_stride_blocks or n_blocks), C.c_void_p(stream)))

    def last_kernel_ms(self):
        return float(self._L.asdr_iqgen_last_kernel_ms(self._h))

    def setGainBalance(self, balance, ch=ALL):
        self._L.asdr_iqgen_setGainBalance(self._h, ch, float(balance))


class AudioGrabberComplex256Batch(_Batch):
    """N independent AudioGrabberComplex256 instances (AudioGrabberComplex256.h:44-52)."""
    _prefix = "grab"

    def update(self, I, Q):
        I, Q = self._blocks(I), self._blocks(Q)
        assert I.shape == Q.shape
        self._chk(self._L.asdr_grab_update(self._h, _p16(I), _p16(Q), I.shape[1]))

    def update_device(self, dI, dQ, n_blocks, in_stride_blocks=None, stream=0):
        self._chk(self._L.asdr_grab_update_device(self._h, C.c_void_p(dI), C.c_void_p(dQ), int(n_blocks),
                                                  int(in_stride_blocks or n_blocks), C.c_void_p(stream)))

    def newDataAvailable(self, ch=0):
        return int(self._L.asdr_grab_newDataAvailable(self._h, ch))

    def grab(self, ch=0, destination=None):
        """Returns (copied, destination[512]); destination is left untouched when no complete buffer exists yet."""
        d = np.zeros(512, dtype=np.int16) if destination is None else destination
        return self._chk(self._L.asdr_grab_grab(self._h, int(ch), _p16(d))), d

    def grab_all(self):
        d = np.zeros((self.n_channels, 512), dtype=np.int16)
        return self._chk(self._L.asdr_grab_grab_all(self._h, _p16(d))), d

    def device_ptr(self):
        return int(self._L.asdr_grab_device_ptr(self._h) or 0)

    def power_spectrum(self):
        """(valid, float32 [n_channels][256]): |FFT256|^2 of every channel's buffer, natural bin order (panadapter)."""
        d = np.zeros((self.n_channels, 256), dtype=np.float32)
        return self._chk(self._L.asdr_grab_power_spectrum(self._h, d.ctypes.data_as(C.POINTER(C.c_float)))), d

    def power_spectrum_device(self, dDst, stream=0):
        return self._chk(self._L.asdr_grab_power_spectrum_device(self._h, C.c_void_p(dDst), C.c_void_p(stream)))
