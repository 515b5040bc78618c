"""ctypes binding of liblele_hip.so (the C ABI declared in include/lele_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is no CPU
fallback anywhere in lele_amd (the CPU restatement under oracle/ is test infrastructure and is never
imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LELE_HIP_LAB=1: the developer's build (python -m lele_amd.build under the same variable), see lele_amd/build.py
# LELE_HIP_LIBRARY=<file name in this directory>: another build of the same library (a saved copy: A/B runs inside one gpurun call)
LIB_PATH = os.path.join(_HERE, os.environ.get("LELE_HIP_LIBRARY") or
                        ("liblele_hip_lab.so" if os.environ.get("LELE_HIP_LAB", "0") not in ("", "0") else "liblele_hip.so"))

MAX_RANK = 8
F32, I64, I32, U8, I8 = 0, 1, 2, 3, 4
MEM_HOST, MEM_DEVICE, MEM_WEIGHT = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2

_NP2DT = {np.dtype(np.float32): F32, np.dtype(np.int64): I64, np.dtype(np.int32): I32, np.dtype(np.uint8): U8,
          np.dtype(np.int8): I8}
_DT2NP = {v: k for k, v in _NP2DT.items()}


class LeleTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("shape", C.POINTER(C.c_int64)), ("rank", C.c_int32), ("dtype", C.c_int32),
                ("mem", C.c_int32)]


class LelePitch(C.Structure):
    """channel views (include/lele_hip.h): pitches / offset in elements, 0 = dense"""
    _fields_ = [("x_pitch", C.c_int64), ("y_pitch", C.c_int64), ("out_offset", C.c_int64), ("out_pitch", C.c_int64)]


class LeleFeatureConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_int64), ("n_mels", C.c_int64), ("frame_length_ms", C.c_float),
                ("frame_shift_ms", C.c_float), ("lfr_m", C.c_int64), ("lfr_n", C.c_int64)]


class LeleError(RuntimeError):
    """Raised where lele's Rust kernels would panic! (shape/attribute violations) or on a HIP failure."""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "lele_amd: %s is missing. Build it with `python -m lele_amd.build` (needs hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.lele_hip_last_error.restype = C.c_char_p
        _lib.lele_hip_buf_data.restype = C.c_void_p
        _lib.lele_hip_buf_bytes.restype = C.c_size_t
        _lib.lele_hip_ctx_stream.restype = C.c_void_p
    return _lib


def check(rc):
    if rc != 0:
        raise LeleError(lib().lele_hip_last_error().decode("utf-8", "replace"))


def exported_symbols():
    """Names declared in include/lele_hip.h (parsed from the header), for the symbol-coverage test."""
    import re
    hdr = os.path.join(_HERE, "..", "include", "lele_hip.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lele_hip_\w+)\s*\(", txt)))


_live_ctxs = []  # weak references; closed at interpreter exit, BEFORE the HIP runtime's own teardown (destroying graphs / events
                 # after that throws inside the runtime)


def _close_all_ctxs():
    for r in list(_live_ctxs):
        c = r()
        if c is not None:
            try:
                c.close()
            except Exception:
                pass


class Ctx:
    """LeleCtx: one HIP stream + staging arena + weight cache on one device."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().lele_hip_ctx_create(C.c_int(device), C.byref(self._h)))
        import atexit
        import weakref
        if not _live_ctxs:
            atexit.register(_close_all_ctxs)
        _live_ctxs.append(weakref.ref(self))
        self.device = device
        self._bufs = []
        self._graphs = []  # weak references: ctx_destroy destroys the graphs recorded on it, their wrappers must not do it again
        self._comms = []   # likewise the communicators bound to this ctx's stream

    def close(self):
        if self._h:
            for r in self._graphs + self._comms:
                g = r()
                if g is not None:
                    g.close()
            self._graphs, self._comms = [], []
            for b in self._bufs:
                b.close()
            lib().lele_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().lele_hip_sync(self._h))

    def timer_start(self):
        check(lib().lele_hip_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        check(lib().lele_hip_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def graph_begin(self):
        """start recording every op issued on this ctx into a hipGraph (see include/lele_hip.h)"""
        check(lib().lele_hip_graph_begin(self._h))

    def graph_end(self):
        h = C.c_void_p()
        rc = lib().lele_hip_graph_end(self._h, C.byref(h))
        if rc != 0:
            self.graph_abort()
        check(rc)
        g = Graph(self, h)
        import weakref
        self._graphs.append(weakref.ref(g))
        return g

    # lanes (include/lele_hip.h, lele_hip_lane_*): extra streams of this context for plans run as a DAG (lele_amd/lanes.py)
    cur_lane = 0
    _next_event = 0

    def lane_set(self, lane):
        if lane != self.cur_lane:
            check(lib().lele_hip_lane_set(self._h, C.c_int(int(lane))))
            self.cur_lane = int(lane)

    def lane_record(self, event):
        check(lib().lele_hip_lane_record(self._h, C.c_int(int(event))))

    def lane_wait(self, event):
        check(lib().lele_hip_lane_wait(self._h, C.c_int(int(event))))

    LANE_EVENT_IDS = 1 << 16   # lele_hip_lane_record's id space (csrc/context.hip)

    def lane_events(self, n):
        """reserve `n` consecutive event ids for one plan; returns the first.  Ranges come back through lane_events_release (a Runner
        does that when it is closed or collected) and are re-used first-fit, so a process that re-plans for ever does not run out."""
        n = int(n)
        free = self.__dict__.setdefault("_free_events", [])
        for i, (base, size) in enumerate(free):
            if size >= n:
                if size == n:
                    free.pop(i)
                else:
                    free[i] = (base + n, size - n)
                return base
        base = self._next_event
        if base + n > self.LANE_EVENT_IDS:
            raise LeleError("lane events exhausted: %d ids in use, %d more asked for (of %d); close the Runners of plans no longer run"
                            % (base, n, self.LANE_EVENT_IDS))
        self._next_event += n
        return base

    def lane_events_release(self, base, n):
        if n > 0:
            free = self.__dict__.setdefault("_free_events", [])
            free.append((int(base), int(n)))
            free.sort()
            merged = []
            for b, sz in free:   # coalesce neighbours
                if merged and merged[-1][0] + merged[-1][1] == b:
                    merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                else:
                    merged.append((b, sz))
            self._free_events = merged

    def quant_set_profiling(self, on):
        """per-stage stopwatch of fused_quantized_linear (eager calls only), see include/lele_hip.h"""
        check(lib().lele_hip_quant_set_profiling(self._h, C.c_int(int(on))))

    def quant_profile_read(self):
        """-> (range_ms, quantise_ms, gemm_ms, calls): average per call since the last read"""
        a, b, c, n = C.c_float(), C.c_float(), C.c_float(), C.c_int64()
        check(lib().lele_hip_quant_profile_read(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def graph_abort(self):
        lib().lele_hip_graph_abort(self._h)
        self.cur_lane = 0   # the library is back on lane 0 whatever lane the failed statement ran on

    def buf(self):
        b = Buf(self)
        self._bufs.append(b)
        return b


class Buf:
    """LeleBuf: growable device buffer = the `out: &mut Vec<f32>` of a lele kernel call."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._h = C.c_void_p()
        check(lib().lele_hip_buf_create(ctx._h, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().lele_hip_buf_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def ptr(self):
        return lib().lele_hip_buf_data(self._h)

    @property
    def nbytes(self):
        return lib().lele_hip_buf_bytes(self._h)

    def mark_dirty(self):
        """the contents were written behind the library's back (see lele_hip_buf_mark_dirty)"""
        check(lib().lele_hip_buf_mark_dirty(self._h))

    def reserve(self, nbytes):
        """make room for `nbytes` (lele_hip_buf_reserve): what the owner of an enclosing tensor does before its windows are written"""
        check(lib().lele_hip_buf_reserve(self._h, C.c_size_t(int(nbytes))))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        check(lib().lele_hip_buf_from_host(self._h, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        return DevTensor(self, arr.shape, arr.dtype)

    def to_numpy(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        check(lib().lele_hip_buf_to_host(self._h, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out


class Comm:
    """LeleComm: this process's membership of an RCCL communicator (one process per GPU), bound to a ctx stream.
    `Comm.from_file(ctx, path, rank, world)` is the torch-free rendezvous (rank 0 writes the unique id to `path`)."""

    def __init__(self, ctx, h, rank, world):
        import weakref
        self.ctx, self._h, self.rank, self.world = ctx, h, rank, world
        ctx._comms.append(weakref.ref(self))   # closed with the ctx (and so before the HIP runtime goes away at interpreter exit)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().lele_hip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_id(cls, ctx, uid, rank, world):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(lib().lele_hip_comm_init(ctx._h, buf, C.c_int(rank), C.c_int(world), C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def from_file(cls, ctx, path, rank, world, timeout_ms=120000):
        h = C.c_void_p()
        check(lib().lele_hip_comm_init_file(ctx._h, path.encode(), C.c_int(rank), C.c_int(world), C.c_int(timeout_ms), C.byref(h)))
        return cls(ctx, h, rank, world)

    def allgather_i32(self, send, out=None):
        """send: device-resident int32 tensor (same size on every rank) -> DevTensor [world, count]"""
        keep = []
        out = out or self.ctx.buf()
        sh = OutShape()
        check(lib().lele_hip_comm_allgather_i32(self._h, as_tensor(send, keep), out._h, sh.shape, C.byref(sh.rank)))
        return DevTensor(out, sh.get(), np.int32)

    def allgather(self, send, out=None):
        """send: device-resident tensor of any element type (same shape on every rank) -> DevTensor [world, ...shape]"""
        keep = []
        out = out or self.ctx.buf()
        sh = OutShape()
        t = as_tensor(send, keep)
        check(lib().lele_hip_comm_allgather(self._h, t, out._h, sh.shape, C.byref(sh.rank)))
        return DevTensor(out, sh.get(), send.dtype)

    def allreduce_max(self, value):
        v = C.c_int64(int(value))
        check(lib().lele_hip_comm_allreduce_max_i64(self._h, C.byref(v)))
        return v.value

    def barrier(self):
        check(lib().lele_hip_comm_barrier(self._h))

    def close(self):
        if self._h:
            lib().lele_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """A captured op sequence; launch() replays it on the ctx stream with one hipGraphLaunch."""

    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    def launch(self):
        check(lib().lele_hip_graph_launch(self._h))

    def close(self):
        if self._h:
            lib().lele_hip_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevTensor:
    """A device-resident tensor: (LeleBuf, shape, dtype) -- or a CHANNEL VIEW of one (include/lele_hip.h, LelePitch): the tensor
    starts `offset` elements into the buffer and image n (index along axis 0) starts n * pitch elements after image 0; inside an
    image it is dense.  pitch == 0: dense.  Only the *_pitched entry points accept views (as_tensor refuses them elsewhere)."""

    def __init__(self, buf, shape, dtype=np.float32, offset=0, pitch=0):
        self.buf = buf
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.offset, self.pitch = int(offset), int(pitch)
        if self.pitch and len(self.shape) and int(np.prod(self.shape[1:], dtype=np.int64)) == self.pitch and not self.offset:
            self.pitch = 0   # the whole tensor: dense

    @property
    def is_view(self):
        return bool(self.pitch or self.offset)

    def numpy(self):
        if int(np.prod(self.shape)) == 0:
            return np.zeros(self.shape, self.dtype)
        if not self.is_view:
            return self.buf.to_numpy(self.shape, self.dtype)
        n = self.shape[0]
        per = int(np.prod(self.shape[1:], dtype=np.int64))
        pitch = self.pitch or per
        flat = self.buf.to_numpy((self.offset + (n - 1) * pitch + per,), self.dtype)
        return np.stack([flat[self.offset + i * pitch:self.offset + i * pitch + per] for i in range(n)]).reshape(self.shape)


class Weight:
    """A host array that is immutable for the life of the ctx (a weights.bin slice): passed as LELE_MEM_WEIGHT, so
    the library uploads / pre-packs it once per ctx and caches the device copy by (pointer, bytes)."""

    _alive = []  # the contract is "immutable AND alive for the life of the ctx": a collected array's address could be
                 # handed to a new array of the same size, which the (pointer, bytes) cache would mistake for the old one

    def __init__(self, arr):
        a = np.asarray(arr)
        self.arr = np.ascontiguousarray(a if a.dtype in _NP2DT else a.astype(np.float32))
        self.shape = self.arr.shape
        Weight._alive.append(self.arr)


def as_tensor(x, keep, mem=None, views=False):
    """Build a LeleTensor for x (numpy array -> host memory, DevTensor -> device memory, Weight -> cached weight).
    `keep` collects the objects that must stay alive for the duration of the call.  views=True: the caller is a *_pitched entry
    point and hands the pitch over separately; everywhere else a channel view is an error, not a silently dense read."""
    if x is None:
        return None
    if isinstance(x, Weight):
        x, mem = x.arr, MEM_WEIGHT
    if isinstance(x, DevTensor):
        if x.is_view and not views:
            raise LeleError("this operator needs a dense tensor, got a channel view (offset %d, pitch %d)" % (x.offset, x.pitch))
        shape = (C.c_int64 * max(1, len(x.shape)))(*x.shape)
        keep.append(shape)
        t = LeleTensor(C.c_void_p(x.buf.ptr + x.offset * x.dtype.itemsize), shape, len(x.shape), _NP2DT[x.dtype], MEM_DEVICE)
        keep.append(t)
        return C.byref(t)
    a = np.asarray(x)
    if a.dtype not in _NP2DT:
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    keep.append(a)
    shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
    keep.append(shape)
    t = LeleTensor(a.ctypes.data_as(C.c_void_p), shape, a.ndim, _NP2DT[a.dtype], MEM_HOST if mem is None else mem)
    keep.append(t)
    return C.byref(t)


def i64_array(vals, keep):
    arr = (C.c_int64 * max(1, len(vals)))(*[int(v) for v in vals])
    keep.append(arr)
    return arr, C.c_size_t(len(vals))


class OutShape:
    def __init__(self):
        self.shape = (C.c_int64 * MAX_RANK)()
        self.rank = C.c_int32(0)

    def get(self):
        return tuple(int(self.shape[i]) for i in range(self.rank.value))
