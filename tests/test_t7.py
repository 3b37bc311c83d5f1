"""Torch7 .t7 serialisation (rau_vqa_amd/t7.py): reader pinned by byte fixtures assembled
here by hand from the published format (independent of the writer), then writer->reader
round trips of the reference's two uses (snapshots SS:1188-1197, feature maps
vqa_prepro_loader.lua:549-552).  PARITY UNPINNED against a real Torch7 (none available)."""
import struct

import numpy as np
import pytest

from rau_vqa_amd import t7


def i32(v):
    return struct.pack("<i", v)


def i64(v):
    return struct.pack("<q", v)


def s(txt):
    return i32(len(txt)) + txt.encode()


def num(v):
    return i32(1) + struct.pack("<d", v)


def float_tensor_bytes(idx, sizes, strides, offset, data, cls="Float"):
    b = i32(4) + i32(idx) + s("V 1") + s(f"torch.{cls}Tensor") + i32(len(sizes))
    b += b"".join(i64(x) for x in sizes) + b"".join(i64(x) for x in strides) + i64(offset)
    b += i32(4) + i32(idx + 1) + s("V 1") + s(f"torch.{cls}Storage") + i64(len(data))
    b += np.asarray(data, "<f4").tobytes()
    return b


def test_reader_on_hand_assembled_tensor():
    raw = float_tensor_bytes(1, [2, 3], [3, 1], 1, [1, 2, 3, 4, 5, 6])
    t = t7.loads(raw)
    assert t.type_name == "torch.FloatTensor"
    np.testing.assert_array_equal(t.array, [[1, 2, 3], [4, 5, 6]])


def test_reader_honours_strides_and_offset():
    # a transposed 3x2 view starting at the second element of an 8-element storage
    raw = float_tensor_bytes(1, [3, 2], [1, 3], 2, [9, 1, 2, 3, 4, 5, 6, 9])
    np.testing.assert_array_equal(t7.loads(raw).array, [[1, 4], [2, 5], [3, 6]])


def test_reader_on_hand_assembled_table_with_shared_reference():
    st = lambda txt: i32(2) + s(txt)                                # a string OBJECT (tag + body)
    inner = i32(3) + i32(2) + i32(1) + st("k") + num(7.0)         # table #2 {k=7}, first sight
    pairs = (st("it") + num(1500.0) + st("name") + st("SS") + st("flag") + i32(5) + i32(1)
             + num(1.0) + inner)
    t = t7.loads(i32(3) + i32(1) + i32(4) + pairs)
    assert t == {"it": 1500, "name": "SS", "flag": True, 1: {"k": 7}}
    again = num(2.0) + i32(3) + i32(2)                              # [2] = back-reference to #2
    t2 = t7.loads(i32(3) + i32(1) + i32(5) + pairs + again)
    assert t2[1] is t2[2] and t2[2] == {"k": 7}


def test_reader_accepts_pre_versioning_class_header_and_cuda_tensors():
    b = i32(4) + i32(1) + s("torch.CudaTensor") + i32(1) + i64(3) + i64(1) + i64(1)
    b += i32(4) + i32(2) + s("torch.CudaStorage") + i64(3) + np.asarray([1, 2, 3], "<f4").tobytes()
    t = t7.loads(b)
    assert t.type_name == "torch.CudaTensor" and t.array.tolist() == [1.0, 2.0, 3.0]


def test_reader_refuses_functions_and_truncation():
    with pytest.raises(t7.T7Error, match="functions"):
        t7.loads(i32(6) + i32(1))
    good = float_tensor_bytes(1, [4], [1], 1, [1, 2, 3, 4])
    with pytest.raises(t7.T7Error, match="truncated"):
        t7.loads(good[:-3])
    bad = float_tensor_bytes(1, [5], [1], 1, [1, 2, 3, 4])        # view larger than its storage
    with pytest.raises(t7.T7Error, match="exceeds"):
        t7.loads(bad)


def test_snapshot_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    params = {"embed": rng.standard_normal(40).astype(np.float32),
              "rnn": rng.standard_normal(33).astype(np.float32),
              "mult": rng.standard_normal(57).astype(np.float32)}
    opt = {"nhop": 8, "alg_name": "Ours_SS", "batch_size": 100, "learning_rate": 3e-3,
           "use_cudnn": True, "init_from": ""}
    p = tmp_path / "snapshot_iter001500_epoch0.62.t7"
    t7.save_snapshot(p, params, it=1500, epoch=0.62, opt=opt)
    raw = p.read_bytes()
    assert b"torch.CudaTensor" in raw and b"torch.CudaStorage" in raw   # what SS:1196 saves
    it, epoch, opt2, got = t7.load_snapshot(p)
    assert (it, epoch) == (1500, 0.62)
    assert opt2["nhop"] == 8 and opt2["alg_name"] == "Ours_SS" and opt2["use_cudnn"] is True
    assert abs(opt2["learning_rate"] - 3e-3) < 1e-12
    for g in params:
        np.testing.assert_array_equal(got[g], params[g])
    t7.save_snapshot(p, params, 1, 0.0, opt, cuda=False)
    assert b"torch.FloatTensor" in p.read_bytes()
    with pytest.raises(t7.T7Error, match="snapshot"):
        t7.save(p, {"it": 1})
        t7.load_snapshot(p)


def test_feature_file_and_generic_round_trip(tmp_path):
    f = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    p = tmp_path / "COCO_train2014_000000357413.t7"
    t7.save(p, f)
    np.testing.assert_array_equal(t7.load_feature(p, 2, 3, 4), f.reshape(2, 12))
    with pytest.raises(t7.T7Error, match="shape"):
        t7.load_feature(p, 2, 4, 3)
    shared = {"a": 1}
    obj = {"x": [1.5, "two", None, False], "s1": shared, "s2": shared,
           "d": np.arange(5, dtype=np.float64), "l": np.arange(3, dtype=np.int64)}
    back = t7.loads(t7.dumps(obj))
    assert back["x"] == {1: 1.5, 2: "two", 3: None, 4: False}
    assert back["s1"] is back["s2"]
    assert back["d"].type_name == "torch.DoubleTensor" and back["l"].array.tolist() == [0, 1, 2]


def test_remap_flat_between_layouts():
    src = [("a.weight", 0, 2, 3), ("a.bias", 6, 2, 1), ("b.weight", 8, 1, 4)]
    dst = [("b.weight", 0, 1, 4), ("a.weight", 4, 2, 3), ("a.bias", 10, 2, 1)]
    flat = np.arange(12, dtype=np.float32)
    out = t7.remap_flat(flat, src, dst)
    assert out.tolist() == [8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6, 7]
    with pytest.raises(t7.T7Error):
        t7.remap_flat(flat, src, [("c.weight", 0, 1, 1)])
