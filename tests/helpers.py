"""Shared helpers for the parity tests (setter scripts applied identically to the HIP batch and to the oracle)."""
import numpy as np


def S(method, *args, sel=None):
    return (method, args, sel)


def apply_setters(batch, oracles, setters):
    n = len(oracles)
    for meth, args, sel in setters:
        for c in range(n):
            if sel is None or sel(c):
                if batch is not None:
                    getattr(batch, meth)(*args, ch=c)
                getattr(oracles[c], meth)(*args)


def f32_bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def compare_status(A, batch, oracles):
    st = batch.read_status()
    for c, o in enumerate(oracles):
        assert int(st["agc_active"][c]) == o.AGCisActive(), "AGCisActive ch %d" % c
        assert int(st["nb_detected"][c]) == o.NoiseBlankerDetection(), "NoiseBlankerDetection ch %d" % c
        assert int(st["sam_locked"][c]) == o.getSAMphaseLockStatus(), "SAM lock ch %d" % c
        assert f32_bits(st["sam_frequency"][c]) == f32_bits(np.float32(o.getSAMfrequency())), "SAM frequency ch %d" % c
        assert f32_bits(st["am_carrier"][c]) == f32_bits(np.float32(o.getAMcarrierLevel())), "carrier ch %d" % c


class Hip:
    """Bare HIP runtime through ctypes (libamdhip64), as a C host application would use it: HBM buffers and streams
    owned by the caller, handed to the C ABI as raw pointers.  GPU tests only."""

    def __init__(self):
        import ctypes as C
        self.C = C
        h = self.h = C.CDLL("libamdhip64.so")
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        h.hipStreamSynchronize.argtypes = [C.c_void_p]
        h.hipStreamDestroy.argtypes = [C.c_void_p]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self._bufs = []

    def malloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.h.hipMalloc(self.C.byref(p), int(nbytes)) == 0
        self._bufs.append(p.value)
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        assert self.h.hipMemcpy(p, arr.ctypes.data_as(self.C.c_void_p), arr.nbytes, 1) == 0
        return p

    def copy_async(self, dst, src, nbytes, kind, stream):
        """hipMemcpyAsync (kind 1 = H2D, 2 = D2H); host side must be pinned for the copy to be truly asynchronous."""
        assert self.h.hipMemcpyAsync(dst, src, int(nbytes), kind, stream) == 0

    def fill(self, ptr, byte, nbytes):
        assert self.h.hipMemset(ptr, byte, nbytes) == 0

    def download(self, ptr, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        assert self.h.hipDeviceSynchronize() == 0
        assert self.h.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), ptr + offset_bytes, out.nbytes, 2) == 0
        return out

    def stream(self):
        s = self.C.c_void_p()
        assert self.h.hipStreamCreate(self.C.byref(s)) == 0
        return s.value

    def sync(self, stream=None):
        assert (self.h.hipStreamSynchronize(stream) if stream else self.h.hipDeviceSynchronize()) == 0

    def free_all(self):
        self.h.hipDeviceSynchronize()
        for p in self._bufs:
            self.h.hipFree(p)
        self._bufs = []
