"""TensorView: host mirror of lele::tensor::TensorView (/root/reference/src/tensor.rs:5-85).

The Rust type is a borrowed-or-owned flat row-major buffer plus a shape.  Here `data` is either a numpy
array (host) or a device-resident `_lib.DevTensor` (the result of a kernel call).  Device results are
materialised on the host lazily by `.numpy()` / `.data` -- the analogue of the Rust shim's Deref<[T]> with
lazy D2H -- so chains of kernel calls never leave HBM.
"""
import numpy as np

from . import _lib


class TensorView:
    def __init__(self, data, shape=None):
        if isinstance(data, _lib.DevTensor):
            if shape is not None and data.is_view and tuple(int(d) for d in shape) != tuple(data.shape):
                raise _lib.LeleError("reshape of a channel view (offset %d, pitch %d): copy it out first" % (data.offset, data.pitch))
            self._dev = data if shape is None or data.is_view else _lib.DevTensor(data.buf, shape, data.dtype)
            self._host = None
            self.shape = tuple(self._dev.shape)
        else:
            a = np.asarray(data)
            if a.dtype not in (np.float32, np.int64, np.int32, np.uint8, np.int8):
                a = a.astype(np.float32)
            if shape is not None:
                shape = tuple(int(s) for s in shape)
                assert a.size == int(np.prod(shape, dtype=np.int64)), "Data length mismatch"  # tensor.rs:29
                a = a.reshape(shape)
            lshape = tuple(a.shape)  # np.ascontiguousarray promotes 0-d to 1-d: keep the logical shape ourselves
            self._host = np.ascontiguousarray(a)
            self._dev = None
            self.shape = lshape

    # constructors named as in tensor.rs:27-71
    @classmethod
    def new(cls, data, shape):
        return cls(data, shape)

    from_owned = new
    from_slice = new

    @classmethod
    def empty(cls):
        t = cls(np.zeros((0,), np.float32))
        t.shape = ()
        return t

    def to_owned(self):
        return TensorView(self.numpy().copy())

    def dim(self):
        return len(self.shape)

    def size(self, dim):
        return self.shape[dim]

    @property
    def dtype(self):
        return self._dev.dtype if self._dev is not None else self._host.dtype

    @property
    def is_device(self):
        return self._dev is not None

    @property
    def pitch(self):
        """elements from one image to the next when this is a channel view of a wider NCHW tensor (0: dense)"""
        return self._dev.pitch if self._dev is not None else 0

    @property
    def is_view(self):
        return self._dev is not None and self._dev.is_view

    def channels(self, c0, c1):
        """the channel window [c0, c1) of a device tensor [N, C, ...] as a view (no copy): a Split result / Concat operand"""
        d = self._dev
        if d is None or len(d.shape) < 2:
            raise _lib.LeleError("channels(): a device tensor of rank >= 2 is required")
        inner = int(np.prod(d.shape[2:], dtype=np.int64))
        pitch = d.pitch or d.shape[1] * inner
        return TensorView(_lib.DevTensor(d.buf, (d.shape[0], c1 - c0) + tuple(d.shape[2:]), d.dtype, d.offset + c0 * inner, pitch))

    def numpy(self):
        if self._host is None:
            self._host = self._dev.numpy()
        return self._host.reshape(self.shape)

    @property
    def data(self):
        """flat row-major values, like TensorView.data in Rust"""
        return self.numpy().reshape(-1)

    def raw(self):
        """what to hand to the C ABI: the device tensor if there is one, else the numpy array"""
        if self._dev is not None:
            return self._dev
        return self._host.reshape(self.shape) if self.shape != () else self._host.reshape(())

    def __repr__(self):
        return "TensorView(shape=%s, dtype=%s, %s)" % (self.shape, self.dtype, "device" if self.is_device else "host")


def unwrap(x):
    """TensorView | numpy | None -> object accepted by _lib.as_tensor"""
    if x is None:
        return None
    if isinstance(x, TensorView):
        return x.raw()
    return x
