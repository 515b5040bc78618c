"""Minimal ONNX protobuf reader/writer (wire format only; no `onnx` / generated classes).

lele parses ONNX with prost-generated types (src/model/onnx_proto, used by src/compiler/mod.rs:311-373); this module
decodes the same messages by field number -- ModelProto, GraphProto, NodeProto, AttributeProto, TensorProto,
ValueInfoProto -- keeping exactly the fields the compiler reads.  The writer exists so that tools and tests can build
models programmatically (no ONNX file ships with the reference).  Field numbers are those of onnx.proto (IR version 3+).
"""
import struct

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
NP_OF = {FLOAT: np.float32, UINT8: np.uint8, INT8: np.int8, UINT16: np.uint16, INT16: np.int16, INT32: np.int32,
         INT64: np.int64, BOOL: np.bool_, FLOAT16: np.float16, DOUBLE: np.float64}
ONNX_OF = {np.dtype(v): k for k, v in NP_OF.items()}


# ----------------------------------------------------------------------------------------------- wire format
def _read_varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def fields(buf):
    """yield (field number, wire type, value) -- value is an int (varint / fixed) or a memoryview (length-delimited)"""
    buf = memoryview(buf)
    i, n = 0, len(buf)
    while i < n:
        key, i = _read_varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8])
            i += 8
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = bytes(buf[i:i + 4])
            i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d (field %d)" % (wt, fno))
        yield fno, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [_signed(v)]
    out, i, n = [], 0, len(v)
    while i < n:
        x, i = _read_varint(v, i)
        out.append(_signed(x))
    return out


def _packed_fixed(v, wt, fmt, size):
    if wt != 2:
        return [struct.unpack("<" + fmt, v)[0]]
    return list(struct.unpack("<%d%s" % (len(v) // size, fmt), bytes(v)))


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(fno, wt):
    return _varint((fno << 3) | wt)


def _ld(fno, payload):
    return _key(fno, 2) + _varint(len(payload)) + bytes(payload)


# ----------------------------------------------------------------------------------------------- messages
class Tensor:
    def __init__(self, name="", array=None, dims=None, data_type=None):
        self.name = name
        self.array = None if array is None else np.asarray(array)
        self.dims = list(self.array.shape) if dims is None and self.array is not None else list(dims or [])
        self.data_type = data_type if data_type is not None else (ONNX_OF[self.array.dtype] if self.array is not None else FLOAT)

    @staticmethod
    def parse(buf):
        t = Tensor()
        raw = None
        f32, i32, i64, f64 = [], [], [], []
        for fno, wt, v in fields(buf):
            if fno == 1:
                t.dims += _packed_varints(v, wt)
            elif fno == 2:
                t.data_type = v
            elif fno == 4:
                f32 += _packed_fixed(v, wt, "f", 4)
            elif fno == 5:
                i32 += _packed_varints(v, wt)
            elif fno == 7:
                i64 += _packed_varints(v, wt)
            elif fno == 8:
                t.name = bytes(v).decode()
            elif fno == 9:
                raw = bytes(v)
            elif fno == 10:
                f64 += _packed_fixed(v, wt, "d", 8)
            elif fno == 14 and v != 0:
                raise ValueError("tensor %r: external data is not supported" % t.name)
        if t.data_type not in NP_OF:
            raise ValueError("tensor %r: unsupported ONNX data type %d" % (t.name, t.data_type))
        dt = np.dtype(NP_OF[t.data_type])
        if raw is not None:
            a = np.frombuffer(raw, dt.newbyteorder("<")).astype(dt)
        elif t.data_type == FLOAT:
            a = np.array(f32, np.float32)
        elif t.data_type == DOUBLE:
            a = np.array(f64, np.float64)
        elif t.data_type == INT64:
            a = np.array(i64, np.int64)
        elif t.data_type == FLOAT16:  # int32_data carries the bit patterns
            a = np.array(i32, np.int32).astype(np.uint16).view(np.float16)
        else:
            a = np.array(i32, np.int64).astype(dt)
        t.array = a.reshape(t.dims) if t.dims else a.reshape(())
        return t

    def serialize(self):
        a = np.ascontiguousarray(self.array)
        out = b"".join(_key(1, 0) + _varint(int(d)) for d in self.dims)
        out += _key(2, 0) + _varint(self.data_type)
        if self.name:
            out += _ld(8, self.name.encode())
        return out + _ld(9, a.astype(a.dtype.newbyteorder("<")).tobytes())


class Attribute:
    def __init__(self, name, value=None):
        self.name, self.f, self.i, self.s, self.t, self.g = name, None, None, None, None, None
        self.floats, self.ints, self.strings = [], [], []
        if isinstance(value, bool):
            self.i = int(value)
        elif isinstance(value, int):
            self.i = value
        elif isinstance(value, float):
            self.f = value
        elif isinstance(value, (str, bytes)):
            self.s = value.encode() if isinstance(value, str) else value
        elif isinstance(value, Tensor):
            self.t = value
        elif isinstance(value, np.ndarray):
            self.t = Tensor("", value)
        elif isinstance(value, Graph):
            self.g = value
        elif isinstance(value, (list, tuple)):
            if value and isinstance(value[0], float):
                self.floats = list(value)
            else:
                self.ints = [int(v) for v in value]
        elif value is not None:
            raise TypeError("attribute %s: unsupported value %r" % (name, value))

    @staticmethod
    def parse(buf):
        a = Attribute("")
        for fno, wt, v in fields(buf):
            if fno == 1:
                a.name = bytes(v).decode()
            elif fno == 2:
                a.f = struct.unpack("<f", v)[0]
            elif fno == 3:
                a.i = _signed(v)
            elif fno == 4:
                a.s = bytes(v)
            elif fno == 5:
                a.t = Tensor.parse(v)
            elif fno == 6:
                a.g = Graph.parse(v)
            elif fno == 7:
                a.floats += _packed_fixed(v, wt, "f", 4)
            elif fno == 8:
                a.ints += _packed_varints(v, wt)
            elif fno == 9:
                a.strings.append(bytes(v))
        return a

    def serialize(self):
        out = _ld(1, self.name.encode())
        if self.f is not None:
            out += _key(2, 5) + struct.pack("<f", self.f) + _key(20, 0) + _varint(1)
        if self.i is not None:
            out += _key(3, 0) + _varint(self.i) + _key(20, 0) + _varint(2)
        if self.s is not None:
            out += _ld(4, self.s) + _key(20, 0) + _varint(3)
        if self.t is not None:
            out += _ld(5, self.t.serialize()) + _key(20, 0) + _varint(4)
        if self.g is not None:
            out += _ld(6, self.g.serialize()) + _key(20, 0) + _varint(5)
        if self.floats:
            out += _ld(7, struct.pack("<%df" % len(self.floats), *self.floats)) + _key(20, 0) + _varint(6)
        if self.ints:
            out += _ld(8, b"".join(_varint(v) for v in self.ints)) + _key(20, 0) + _varint(7)
        return out


class Node:
    def __init__(self, op_type="", inputs=(), outputs=(), name="", **attrs):
        self.op_type, self.input, self.output, self.name = op_type, list(inputs), list(outputs), name
        self.attribute = [Attribute(k, v) for k, v in attrs.items()]

    def attr(self, name, default=None):
        for a in self.attribute:
            if a.name == name:
                return a
        return default

    @staticmethod
    def parse(buf):
        n = Node()
        for fno, _wt, v in fields(buf):
            if fno == 1:
                n.input.append(bytes(v).decode())
            elif fno == 2:
                n.output.append(bytes(v).decode())
            elif fno == 3:
                n.name = bytes(v).decode()
            elif fno == 4:
                n.op_type = bytes(v).decode()
            elif fno == 5:
                n.attribute.append(Attribute.parse(v))
        return n

    def serialize(self):
        out = b"".join(_ld(1, s.encode()) for s in self.input) + b"".join(_ld(2, s.encode()) for s in self.output)
        if self.name:
            out += _ld(3, self.name.encode())
        out += _ld(4, self.op_type.encode())
        return out + b"".join(_ld(5, a.serialize()) for a in self.attribute)


class ValueInfo:
    def __init__(self, name="", elem_type=FLOAT, shape=None):
        self.name, self.elem_type, self.shape = name, elem_type, shape  # shape: list of int | str (symbolic) | None

    @staticmethod
    def parse(buf):
        vi = ValueInfo()
        for fno, _wt, v in fields(buf):
            if fno == 1:
                vi.name = bytes(v).decode()
            elif fno == 2:
                for f2, _w2, v2 in fields(v):           # TypeProto
                    if f2 != 1:
                        continue
                    for f3, _w3, v3 in fields(v2):      # TypeProto.Tensor
                        if f3 == 1:
                            vi.elem_type = v3
                        elif f3 == 2:
                            vi.shape = []
                            for f4, _w4, v4 in fields(v3):  # TensorShapeProto.dim
                                if f4 != 1:
                                    continue
                                d = None
                                for f5, _w5, v5 in fields(v4):
                                    if f5 == 1:
                                        d = _signed(v5)
                                    elif f5 == 2:
                                        d = bytes(v5).decode()
                                vi.shape.append(d)
        return vi

    def serialize(self):
        tt = _key(1, 0) + _varint(self.elem_type)
        if self.shape is not None:
            dims = b""
            for d in self.shape:
                dims += _ld(1, _ld(2, d.encode()) if isinstance(d, str) else _key(1, 0) + _varint(int(d)))
            tt += _ld(2, dims)
        return _ld(1, self.name.encode()) + _ld(2, _ld(1, tt))


class Graph:
    def __init__(self, nodes=(), inputs=(), outputs=(), initializers=(), name="graph"):
        self.node, self.input, self.output, self.initializer, self.name = list(nodes), list(inputs), list(outputs), list(initializers), name

    @staticmethod
    def parse(buf):
        g = Graph()
        for fno, _wt, v in fields(buf):
            if fno == 1:
                g.node.append(Node.parse(v))
            elif fno == 2:
                g.name = bytes(v).decode()
            elif fno == 5:
                g.initializer.append(Tensor.parse(v))
            elif fno == 11:
                g.input.append(ValueInfo.parse(v))
            elif fno == 12:
                g.output.append(ValueInfo.parse(v))
        return g

    def serialize(self):
        return (b"".join(_ld(1, n.serialize()) for n in self.node) + _ld(2, self.name.encode())
                + b"".join(_ld(5, t.serialize()) for t in self.initializer)
                + b"".join(_ld(11, v.serialize()) for v in self.input) + b"".join(_ld(12, v.serialize()) for v in self.output))


class Model:
    def __init__(self, graph=None, ir_version=8, opset=17, producer="lele_amd"):
        self.graph, self.ir_version, self.opset, self.producer = graph, ir_version, opset, producer

    @staticmethod
    def parse(buf):
        m = Model()
        for fno, _wt, v in fields(buf):
            if fno == 1:
                m.ir_version = v
            elif fno == 2:
                m.producer = bytes(v).decode()
            elif fno == 7:
                m.graph = Graph.parse(v)
            elif fno == 8:
                for f2, _w2, v2 in fields(v):
                    if f2 == 2:
                        m.opset = _signed(v2)
        if m.graph is None:
            raise ValueError("not an ONNX model: no graph")
        return m

    def serialize(self):
        return (_key(1, 0) + _varint(self.ir_version) + _ld(2, self.producer.encode()) + _ld(7, self.graph.serialize())
                + _ld(8, _key(2, 0) + _varint(self.opset)))


def load(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    return Model.parse(data)
