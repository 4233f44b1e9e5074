"""Device buffers for the tests of the `_device` entry points, through the HIP runtime the library itself is
linked against (no second runtime in the process)."""
import numpy as np


class Hip:
    def __init__(self):
        import ctypes as C
        self.C = C
        # by soname: resolves to the copy already loaded in this process (the one liborbgpu.so is bound to)
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, nbytes, fill=0xA5):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(nbytes, 4)) == 0
        assert self.rt.hipMemset(p, fill, max(nbytes, 4)) == 0
        self.bufs.append(p)
        return p

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    def download(self, p, shape, dtype):
        a = np.zeros(shape, dtype)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(a.ctypes.data, p, a.nbytes, 2) == 0
        return a

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


