"""Torch7 binary serialization (`torch.save` / `torch.load`, ".t7" files), host side.

The reference persists two things in this format:
  * training snapshots -- ``{it=, opt=, epoch=, params={[1]=embed, [2]=rnn, [3]=mult}}``
    written at experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:1188-1197 and
    read back by experiments/Ours_ResNet/Eval.lua:113-114, 344-347;
  * per-image feature maps -- one ``FloatTensor [D, W, H]`` per ``COCO_*.t7``
    (utils/vqa_prepro_loader.lua:549-552, utils/qa_utils.lua:15).

Torch7 itself is absent from the build image and un-vendored in the reference, so this
is a restatement of its published on-disk format (torch7 ``File.lua`` /
``Tensor.lua`` / ``Storage``'s ``write``/``read`` in binary little-endian mode,
``long`` = 8 bytes):

    object  := int32 tag, payload
    tag     :  0 nil | 1 number(float64) | 2 string(int32 n, n bytes) | 3 table
               | 4 torch object | 5 boolean(int32)
    table   := int32 ref-index, [first time only:] int32 npairs, npairs x (object, object)
    torch   := int32 ref-index, [first time only:] string "V 1", string class-name, body
    Tensor  body := int32 ndim, ndim x int64 size, ndim x int64 stride,
                    int64 storage-offset (1-based), object storage
    Storage body := int64 n, n raw elements

Objects are memoised by ref-index, so shared storages / tables round-trip as shared.
Nothing in a file is executed: tags 6-8 (Lua functions) are refused.

PARITY UNPINNED: no file written by a real Torch7 is available offline (the
reference's download_trained_model.sh needs the network), so the reader is pinned
only by byte fixtures assembled by hand from the format description
(tests/test_t7.py) and by write->read round trips.
"""
from __future__ import annotations

import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_STORAGE_DTYPES = {
    "torch.DoubleStorage": np.float64, "torch.FloatStorage": np.float32,
    "torch.LongStorage": np.int64, "torch.IntStorage": np.int32,
    "torch.ShortStorage": np.int16, "torch.CharStorage": np.int8,
    "torch.ByteStorage": np.uint8,
    "torch.CudaStorage": np.float32,          # cutorch writes CudaStorage as floats
}
_TENSOR_TO_STORAGE = {k.replace("Storage", "Tensor"): k for k in _STORAGE_DTYPES}


class T7Error(ValueError):
    pass


class Tensor:
    """A deserialised torch.*Tensor: its class name and a numpy view with the saved strides."""

    def __init__(self, array: np.ndarray, type_name: str = "torch.FloatTensor"):
        if type_name not in _TENSOR_TO_STORAGE:
            raise T7Error(f"unsupported tensor class {type_name}")
        self.array = np.asarray(array, _STORAGE_DTYPES[_TENSOR_TO_STORAGE[type_name]])
        self.type_name = type_name

    def __repr__(self):
        return f"t7.Tensor({self.type_name}, shape={self.array.shape})"


class TorchObject:
    """Any other torch class (e.g. an nn module): class name + its serialised table."""

    def __init__(self, type_name, fields):
        self.type_name, self.fields = type_name, fields


# ------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, data: bytes):
        self.b, self.o, self.memo = memoryview(data), 0, {}

    def _take(self, n):
        if self.o + n > len(self.b):
            raise T7Error("truncated .t7 file")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def i32(self):
        return struct.unpack("<i", self._take(4))[0]

    def i64(self):
        return struct.unpack("<q", self._take(8))[0]

    def string(self):
        n = self.i32()
        if n < 0:
            raise T7Error("negative string length")
        return bytes(self._take(n)).decode("latin-1")

    def obj(self):
        tag = self.i32()
        if tag == TYPE_NIL:
            return None
        if tag == TYPE_NUMBER:
            v = struct.unpack("<d", self._take(8))[0]
            return int(v) if v == int(v) and abs(v) < 2 ** 53 else v
        if tag == TYPE_STRING:
            return self.string()
        if tag == TYPE_BOOLEAN:
            return self.i32() != 0
        if tag == TYPE_TABLE:
            idx = self.i32()
            if idx in self.memo:
                return self.memo[idx]
            t = {}
            self.memo[idx] = t
            for _ in range(self.i32()):
                k = self.obj()
                t[k] = self.obj()
            return t
        if tag == TYPE_TORCH:
            idx = self.i32()
            if idx in self.memo:
                return self.memo[idx]
            first = self.string()
            cls = self.string() if first.startswith("V ") else first   # pre-versioning files
            o = self._torch_body(cls)
            self.memo[idx] = o
            return o
        raise T7Error(f"refusing type tag {tag} (Lua functions are never deserialised)")

    def _torch_body(self, cls):
        if cls in _STORAGE_DTYPES:
            n = self.i64()
            dt = np.dtype(_STORAGE_DTYPES[cls]).newbyteorder("<")
            return np.frombuffer(self._take(n * dt.itemsize), dt, n)
        if cls in _TENSOR_TO_STORAGE:
            nd = self.i32()
            size = [self.i64() for _ in range(nd)]
            stride = [self.i64() for _ in range(nd)]
            off = self.i64() - 1
            storage = self.obj()
            dt = _STORAGE_DTYPES[_TENSOR_TO_STORAGE[cls]]
            if storage is None or nd == 0:
                return Tensor(np.zeros([0] * max(nd, 1), dt), cls)
            need = off + sum((s - 1) * st for s, st in zip(size, stride)) + 1
            if off < 0 or any(s < 0 for s in size) or (min(size) > 0 and need > storage.size):
                raise T7Error("tensor view exceeds its storage")
            view = np.lib.stride_tricks.as_strided(
                storage[off:], shape=size, strides=[st * storage.itemsize for st in stride],
                writeable=False)
            return Tensor(view, cls)
        return TorchObject(cls, self.obj())    # generic torch class: one serialised table


def loads(data: bytes):
    return _Reader(data).obj()


def load(path):
    with open(path, "rb") as f:
        return loads(f.read())


# ------------------------------------------------------------------ writer
class _Writer:
    def __init__(self):
        self.out, self.memo, self.next = [], {}, 1

    def i32(self, v):
        self.out.append(struct.pack("<i", v))

    def i64(self, v):
        self.out.append(struct.pack("<q", v))

    def string(self, s):
        b = s.encode("latin-1")
        self.i32(len(b))
        self.out.append(b)

    def _ref(self, o):
        """(index, first_time) for the object identity of o."""
        k = id(o)
        if k in self.memo:
            return self.memo[k][0], False
        self.memo[k] = (self.next, o)      # keep o alive so ids stay unique
        self.next += 1
        return self.memo[k][0], True

    def obj(self, o):
        if o is None:
            self.i32(TYPE_NIL)
        elif isinstance(o, (bool, np.bool_)):
            self.i32(TYPE_BOOLEAN)
            self.i32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.i32(TYPE_NUMBER)
            self.out.append(struct.pack("<d", float(o)))
        elif isinstance(o, str):
            self.i32(TYPE_STRING)
            self.string(o)
        elif isinstance(o, dict):
            self.i32(TYPE_TABLE)
            idx, first = self._ref(o)
            self.i32(idx)
            if first:
                self.i32(len(o))
                for k, v in o.items():
                    self.obj(k)
                    self.obj(v)
        elif isinstance(o, (list, tuple)):          # Lua array part: keys 1..n
            self.obj({i + 1: v for i, v in enumerate(o)})
        elif isinstance(o, np.ndarray):
            self.obj(Tensor(o, {np.dtype(np.float64): "torch.DoubleTensor",
                                np.dtype(np.float32): "torch.FloatTensor",
                                np.dtype(np.int64): "torch.LongTensor",
                                np.dtype(np.int32): "torch.IntTensor",
                                np.dtype(np.uint8): "torch.ByteTensor"}[o.dtype]))
        elif isinstance(o, Tensor):
            self.i32(TYPE_TORCH)
            idx, first = self._ref(o)
            self.i32(idx)
            if first:
                a = np.ascontiguousarray(o.array)
                self.string("V 1")
                self.string(o.type_name)
                self.i32(a.ndim)
                for s in a.shape:
                    self.i64(s)
                st = 1
                strides = []
                for s in reversed(a.shape):
                    strides.append(st)
                    st *= s
                for s in reversed(strides):
                    self.i64(s)
                self.i64(1)                          # storage offset, 1-based
                self.i32(TYPE_TORCH)                 # the storage object
                self.i32(self.next)
                self.next += 1
                self.string("V 1")
                self.string(_TENSOR_TO_STORAGE[o.type_name])
                self.i64(a.size)
                self.out.append(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        else:
            raise T7Error(f"cannot serialise {type(o).__name__}")


def dumps(obj) -> bytes:
    w = _Writer()
    w.obj(obj)
    return b"".join(w.out)


def save(path, obj):
    with open(path, "wb") as f:
        f.write(dumps(obj))


# ------------------------------------------------------------------ the reference's two uses
GROUPS = ("embed", "rnn", "mult")     # checkpoint.params[1..3], SS:1196


def save_snapshot(path, params, it, epoch, opt, cuda=True):
    """`torch.save(savefile, {it=, opt=, epoch=, params=})` of SS:1188-1197.

    params: {"embed","rnn","mult"} -> flat float32 vectors in THIS library's layout
    (rau_layout_entry).  The reference saves CudaTensors (its flat vectors live on the GPU);
    cuda=False writes FloatTensors, loadable by a Torch7 without cutorch.
    """
    cls = "torch.CudaTensor" if cuda else "torch.FloatTensor"
    ckpt = {"it": it, "opt": dict(opt), "epoch": epoch,
            "params": {i + 1: Tensor(np.asarray(params[g], np.float32).ravel(), cls)
                       for i, g in enumerate(GROUPS)}}
    save(path, ckpt)


def load_snapshot(path):
    """-> (it, epoch, opt, {"embed","rnn","mult"} flat float32 arrays); Eval.lua:113-114,344-347."""
    snap = load(path)
    if not isinstance(snap, dict) or "params" not in snap:
        raise T7Error("not a training snapshot (no `params` field)")
    p = snap["params"]
    out = {}
    for i, g in enumerate(GROUPS):
        t = p.get(i + 1)
        if not isinstance(t, Tensor):
            raise T7Error(f"snapshot params[{i + 1}] is missing or not a tensor")
        out[g] = np.ascontiguousarray(t.array, np.float32).ravel()
    return snap.get("it"), snap.get("epoch"), snap.get("opt", {}), out


def load_feature(path, D, W, H):
    """One image's feature map as the loader hands it on (vqa_prepro_loader.lua:549-552):
    FloatTensor [D, W, H] -> float32 [D, W*H] (a row of rau_set_batch's feats)."""
    t = load(path)
    if not isinstance(t, Tensor):
        raise T7Error("feature file does not hold a tensor")
    a = np.ascontiguousarray(t.array, np.float32)
    if a.shape != (D, W, H):
        raise T7Error(f"feature shape {a.shape} != ({D}, {W}, {H})")   # the loader's asserts
    return a.reshape(D, W * H)


def remap_flat(flat, src_layout, dst_layout):
    """Re-order a flat parameter vector between two layouts of the same named tensors.

    layouts: [(name, offset, rows, cols)] as returned by RAU.layout(group).  Needed when a
    snapshot's flat vector follows another module order than rau_layout_entry -- the
    reference's order is nngraph's topological order, which cannot be verified offline.
    """
    src = {n: (o, r * c) for n, o, r, c in src_layout}
    out = np.empty(sum(r * c for _, _, r, c in dst_layout), np.float32)
    for n, o, r, c in dst_layout:
        if n not in src or src[n][1] != r * c:
            raise T7Error(f"layouts disagree on {n}")
        out[o:o + r * c] = flat[src[n][0]:src[n][0] + r * c]
    return out
