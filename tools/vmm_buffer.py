"""Measurement-build helper: an observation buffer built with HIP virtual memory management (mg_mem.h;
libmarlgrid_hip_ab.so exports mg_ab_vmm_*).  NOT product code — the product allocates with hipMalloc
(mg_obs_alloc); see profiles/r03/README.md section 2 for what these experiments found (and why a construction
that cannot be freed safely on ROCm 7.2 does not ship).

    from vmm_buffer import VmmBuffer, ab_lib      # sets MARLGRID_HIP_LIB to the measurement build on import
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MARLGRID_HIP_LIB", os.path.join(ROOT, "marlgrid_amd", "csrc", "libmarlgrid_hip_ab.so"))


def ab_lib():
    from marlgrid_amd import _native as N
    L = N.lib()
    vp, i32 = C.c_void_p, C.c_int32
    L.mg_ab_vmm_alloc.argtypes, L.mg_ab_vmm_alloc.restype = [C.c_uint64, i32, C.c_int64], vp
    L.mg_ab_vmm_ptr.argtypes, L.mg_ab_vmm_ptr.restype = [vp], vp
    L.mg_ab_vmm_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mg_ab_vmm_free.argtypes = [vp]
    L.mg_ab_vmm_rebase.argtypes, L.mg_ab_vmm_rebase.restype = [vp], vp
    L.mg_ab_vmm_select.argtypes = [vp, i32]
    L.mg_ab_vmm_trim.argtypes = [vp]
    L.mg_ab_obs_permute.argtypes = [vp, vp]
    L.mg_ab_obs_exchange.argtypes = [vp, vp, vp, i32]
    return L


class VmmBuffer(object):
    """chunk_bytes: 0 = hipMalloc; > 0 = one physical handle per chunk; < 0 = ONE handle for everything"""

    def __init__(self, lib, nbytes, device, chunk_bytes):
        self._lib, self.device, self.nbytes = lib, device, int(nbytes)
        self._h = lib.mg_ab_vmm_alloc(self.nbytes, device.index, int(chunk_bytes))
        self.ok = bool(self._h)
        if self.ok:
            self._moved()

    def _moved(self):
        self.ptr = self._lib.mg_ab_vmm_ptr(self._h)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def tensor(self, shape):
        import torch
        return torch.as_tensor(self, device=self.device).view(shape)

    def info(self):
        out = (C.c_uint64 * 4)()
        assert self._lib.mg_ab_vmm_info(self._h, out) == 0
        return dict(mapped=int(out[0]), chunk=int(out[1]), handles=int(out[2]), ranges=int(out[3]))

    # the same physical memory behind another virtual range; tensors made before a move point at an unmapped range
    def rebase(self):
        import torch
        torch.cuda.synchronize(self.device)
        ok = bool(self._lib.mg_ab_vmm_rebase(self._h))
        self._moved()
        return ok

    def select(self, i):
        import torch
        torch.cuda.synchronize(self.device)
        assert self._lib.mg_ab_vmm_select(self._h, int(i)) == 0
        self._moved()

    def trim(self):
        assert self._lib.mg_ab_vmm_trim(self._h) == 0

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                import torch
                torch.cuda.synchronize(self.device)
                self._lib.mg_ab_vmm_free(self._h)
            except Exception:
                pass
            self._h = None
