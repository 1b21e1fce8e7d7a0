"""Device memory for the Python harness, allocated through the SAME HIP runtime the product
libraries are bound to.

The PyTorch wheel bundles its own libamdhip64.so / librccl.so (another ROCm release) that carry
the same SONAMEs as /opt/rocm's (libamdhip64.so.7, librccl.so.1).  The dynamic loader therefore
binds libemf_hip.so / libemf_fusion.so to whichever copy entered the process first:
  * torch imported first (bench.py)  -> everything, torch included, runs on torch's copy;
  * torch never imported (tests)     -> everything runs on /opt/rocm's copy;
  * product libraries first, torch later -> torch loads its own copy by file name and the process
    holds TWO HIP runtimes, each blind to the other's streams (torch.cuda.synchronize() would not
    wait for our kernels, a torch stream handle would mean nothing to us).
This module never hands torch tensors to the kernels: it resolves "libamdhip64.so.7" through the
loader, i.e. the copy the product libraries use, and keeps every device pointer, stream, event
and synchronisation there.  torch is used by the harness for torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

_lib.load()  # pulls in the product's HIP runtime; the handle below resolves to the same library
_hip = C.CDLL("libamdhip64.so.7")

_H2D, _D2H, _D2D = 1, 2, 3


def _check(rc: int, what: str):
    if rc != 0:
        _hip.hipGetErrorString.restype = C.c_char_p
        raise RuntimeError(f"{what}: hipError {rc} ({_hip.hipGetErrorString(rc).decode()})")


for _name, _args in {
        "hipMalloc": [C.POINTER(C.c_void_p), C.c_size_t],
        "hipFree": [C.c_void_p],
        "hipMemcpy": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int],
        "hipMemcpy2D": [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                        C.c_int],
        "hipMemset": [C.c_void_p, C.c_int, C.c_size_t],
        "hipDeviceSynchronize": [],
        "hipSetDevice": [C.c_int],
        "hipGetDeviceCount": [C.POINTER(C.c_int)],
        "hipStreamCreate": [C.POINTER(C.c_void_p)],
        "hipStreamCreateWithFlags": [C.POINTER(C.c_void_p), C.c_uint],
        "hipStreamQuery": [C.c_void_p],
        "hipHostMalloc": [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint],
        "hipHostFree": [C.c_void_p],
        "hipStreamDestroy": [C.c_void_p],
        "hipStreamSynchronize": [C.c_void_p],
        "hipEventCreate": [C.POINTER(C.c_void_p)],
        "hipEventDestroy": [C.c_void_p],
        "hipEventRecord": [C.c_void_p, C.c_void_p],
        "hipEventSynchronize": [C.c_void_p],
        "hipEventElapsedTime": [C.POINTER(C.c_float), C.c_void_p, C.c_void_p],
        "hipMemGetInfo": [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)],
}.items():
    getattr(_hip, _name).argtypes = _args
    getattr(_hip, _name).restype = C.c_int


def device_count() -> int:
    n = C.c_int(0)
    rc = _hip.hipGetDeviceCount(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(i: int):
    _check(_hip.hipSetDevice(i), "hipSetDevice")


def synchronize():
    _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


def mem_info() -> Tuple[int, int]:
    free, total = C.c_size_t(), C.c_size_t()
    _check(_hip.hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value, total.value


class Stream:
    def __init__(self, non_blocking: bool = False):
        self.handle = C.c_void_p()
        if non_blocking:  # hipStreamNonBlocking: no implicit ordering with the null stream
            _check(_hip.hipStreamCreateWithFlags(C.byref(self.handle), 1), "hipStreamCreateWithFlags")
        else:
            _check(_hip.hipStreamCreate(C.byref(self.handle)), "hipStreamCreate")

    def synchronize(self):
        _check(_hip.hipStreamSynchronize(self.handle), "hipStreamSynchronize")

    def busy(self) -> bool:
        """hipStreamQuery: True while work enqueued on the stream has not completed."""
        rc = _hip.hipStreamQuery(self.handle)
        if rc == 600:  # hipErrorNotReady
            return True
        _check(rc, "hipStreamQuery")
        return False

    def __del__(self):
        if getattr(self, "handle", None):
            _hip.hipStreamDestroy(self.handle)


class HostWord:
    """One 32-bit word of pinned host memory the device can read (flags for probe kernels)."""

    def __init__(self, value: int = 0):
        self.ptr = C.c_void_p()
        # hipHostMallocCoherent | hipHostMallocMapped: fine-grained, so a spinning kernel sees the host's store
        _check(_hip.hipHostMalloc(C.byref(self.ptr), 64, 0x40000000 | 0x2), "hipHostMalloc")
        self._view = C.cast(self.ptr, C.POINTER(C.c_uint32))
        self._view[0] = value

    def set(self, value: int):
        self._view[0] = value

    def __del__(self):
        if getattr(self, "ptr", None):
            _hip.hipHostFree(self.ptr)
            self.ptr = None


class Event:
    def __init__(self):
        self.handle = C.c_void_p()
        _check(_hip.hipEventCreate(C.byref(self.handle)), "hipEventCreate")

    def record(self, stream: Optional[int] = None):
        _check(_hip.hipEventRecord(self.handle, C.c_void_p(stream or 0)), "hipEventRecord")

    def synchronize(self):
        _check(_hip.hipEventSynchronize(self.handle), "hipEventSynchronize")

    def elapsed_ms(self, later: "Event") -> float:
        ms = C.c_float()
        _check(_hip.hipEventElapsedTime(C.byref(ms), self.handle, later.handle),
               "hipEventElapsedTime")
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None):
            _hip.hipEventDestroy(self.handle)


class DeviceArray:
    """A dense array in HBM.  Images (H, W[, C]) may have padded rows (pitch > W*C*itemsize);
    volumes are always contiguous."""

    def __init__(self, shape, dtype, pad_cols: int = 0):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        inner = int(np.prod(self.shape[2:])) if len(self.shape) > 2 else 1
        if pad_cols:
            assert len(self.shape) >= 2
            self.row_bytes = self.shape[1] * inner * self.dtype.itemsize
            self.pitch = (self.shape[1] + pad_cols) * inner * self.dtype.itemsize
            self.nbytes = self.pitch * self.shape[0]
        else:
            self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
            rows = self.shape[0] if self.shape else 1
            self.pitch = self.row_bytes = self.nbytes // max(rows, 1)
        p = C.c_void_p()
        _check(_hip.hipMalloc(C.byref(p), max(self.nbytes, 1)), "hipMalloc")
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None):
            _hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    @property
    def padded(self) -> bool:
        return self.pitch != self.row_bytes

    @classmethod
    def zeros(cls, shape, dtype=np.float32, pad_cols: int = 0) -> "DeviceArray":
        a = cls(shape, dtype, pad_cols)
        _check(_hip.hipMemset(C.c_void_p(a.ptr), 0, a.nbytes), "hipMemset")
        return a

    @classmethod
    def full(cls, shape, value, dtype=np.float32, pad_cols: int = 0) -> "DeviceArray":
        return cls.from_numpy(np.full(shape, value, dtype), pad_cols)

    @classmethod
    def from_numpy(cls, a: np.ndarray, pad_cols: int = 0) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype, pad_cols)
        if d.padded:
            _check(_hip.hipMemset(C.c_void_p(d.ptr), 0xA5, d.nbytes), "hipMemset")  # poison
            _check(_hip.hipMemcpy2D(C.c_void_p(d.ptr), d.pitch, a.ctypes.data, d.row_bytes,
                                    d.row_bytes, d.shape[0], _H2D), "hipMemcpy2D H2D")
        elif d.nbytes:
            _check(_hip.hipMemcpy(C.c_void_p(d.ptr), a.ctypes.data, d.nbytes, _H2D),
                   "hipMemcpy H2D")
        return d

    def copy_from(self, a: np.ndarray):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.shape == self.shape and not self.padded
        _check(_hip.hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data, self.nbytes, _H2D),
               "hipMemcpy H2D")

    def numpy(self) -> np.ndarray:
        """Wait for all device work, then copy to the host."""
        synchronize()
        out = np.empty(self.shape, self.dtype)
        if self.padded:
            _check(_hip.hipMemcpy2D(out.ctypes.data, self.row_bytes, C.c_void_p(self.ptr),
                                    self.pitch, self.row_bytes, self.shape[0], _D2H),
                   "hipMemcpy2D D2H")
        elif self.nbytes:
            _check(_hip.hipMemcpy(out.ctypes.data, C.c_void_p(self.ptr), self.nbytes, _D2H),
                   "hipMemcpy D2H")
        return out

    def numpy_nosync(self) -> np.ndarray:
        """Copy to the host WITHOUT waiting for the device: the caller has synchronised the stream that
        produced the data (threads sharing a GPU must not wait for each other's streams)."""
        assert not self.padded
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            _check(_hip.hipMemcpy(out.ctypes.data, C.c_void_p(self.ptr), self.nbytes, _D2H), "hipMemcpy D2H")
        return out

    def zero_(self):
        _check(_hip.hipMemset(C.c_void_p(self.ptr), 0, self.nbytes), "hipMemset")
        return self


def memcpy_d2h(dst: np.ndarray, src_ptr: int):
    synchronize()
    _check(_hip.hipMemcpy(dst.ctypes.data, C.c_void_p(src_ptr), dst.nbytes, _D2H), "hipMemcpy D2H")


class DeviceView(DeviceArray):
    """Non-owning DeviceArray over memory that belongs to someone else (e.g. a volume held by the
    C++ host classes)."""

    def __init__(self, ptr: int, shape, dtype):  # noqa: D401 - deliberately skips allocation
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        rows = self.shape[0] if self.shape else 1
        self.pitch = self.row_bytes = self.nbytes // max(rows, 1)
        self.ptr = ptr

    def __del__(self):
        self.ptr = None
