"""The SequenceFile / MapFile codec (csrc/fy_seqfile.cpp) on the CPU: round trips, sync markers, byte-level header checks
against Hadoop 1.2.1's published record format, part-file directories, error behaviour.
Parity unpinned at the byte level: the reference holds no binary fixture of these files (its tests write them at run time
with the Hadoop classes, M/util/DataInitialization.java:155-222); what is pinned here is the published layout."""
import os
import struct

import numpy as np
import pytest

from util import pkg


@pytest.fixture(scope="module")
def sf():
    P = pkg()
    P.build()
    import importlib
    return importlib.import_module("filmyou-core_amd.seqfile")


def test_int_int_round_trip_and_header_bytes(sf, tmp_path):
    f = str(tmp_path / "clustering" / "data")          # the writer creates the parent directory (createIntIntFileParent)
    keys = np.arange(1, 31, dtype=np.int32)
    vals = (keys * 7 % 5).astype(np.int32)
    sf.write_int_int(f, keys, vals)
    raw = open(f, "rb").read()
    name = b"org.apache.hadoop.io.IntWritable"
    assert raw[:4] == b"SEQ\x06"
    assert raw[4] == len(name) and raw[5:5 + len(name)] == name                      # Text = vint length + UTF-8
    off = 5 + len(name)
    assert raw[off] == len(name) and raw[off + 1:off + 1 + len(name)] == name
    off += 1 + len(name)
    assert raw[off:off + 2] == b"\x00\x00" and raw[off + 2:off + 6] == b"\x00\x00\x00\x00"   # not compressed, no metadata
    sync = raw[off + 6:off + 22]
    rec = raw[off + 22:off + 22 + 16]
    assert struct.unpack(">iiii", rec) == (8, 4, 1, int(vals[0]))                       # record length, key length, key, value
    assert len(raw) == off + 22 + 16 * 30                                                   # 480 bytes of records: no sync escape yet
    k, v = sf.read_int_int(f)
    np.testing.assert_array_equal(k, keys)
    np.testing.assert_array_equal(v, vals)
    k, v = sf.read_int_int(str(tmp_path / "clustering"))                                    # the directory, like a job input path
    np.testing.assert_array_equal(k, keys)
    assert len(sync) == 16


def test_sync_markers_every_2000_bytes(sf, tmp_path):
    f = str(tmp_path / "big")
    n = 5000
    keys = np.arange(n, dtype=np.int32) - 17
    vals = np.linspace(-1e300, 1e300, n)
    sf.write_int_double(f, keys, vals)
    raw = open(f, "rb").read()
    header = raw.index(b"DoubleWritable") + len(b"DoubleWritable") + 6 + 16
    sync = raw[header - 16:header]
    n_sync = (len(raw) - header - 20 * n) // 20                                           # record = 8 + 4 + 8 bytes
    assert n_sync >= (20 * n) // 2020 and raw.count(b"\xff\xff\xff\xff" + sync) == n_sync
    k, v = sf.read_int_double(f)
    np.testing.assert_array_equal(k, keys)
    np.testing.assert_array_equal(v, vals)                                                  # bit-exact doubles


def test_intpair_float_and_part_files(sf, tmp_path):
    out = tmp_path / "recommendations"
    rng = np.random.default_rng(0)
    parts = []
    for r in range(3):
        u = rng.integers(1, 1000, 700).astype(np.int32)
        i = rng.integers(1, 1000, 700).astype(np.int32)
        s = rng.normal(-300, 50, 700).astype(np.float32)
        s[0] = -np.inf                                                                      # a one-user cluster's scores (quirk Q7)
        sf.write_intpair_float(str(out / ("part-r-%05d" % r)), u, i, s)
        parts.append((u, i, s))
    (out / "_SUCCESS").write_bytes(b"")
    (out / ".part-r-00000.crc").write_bytes(b"junk")
    u, i, s = sf.read_intpair_float(str(out))
    np.testing.assert_array_equal(u, np.concatenate([p[0] for p in parts]))
    np.testing.assert_array_equal(i, np.concatenate([p[1] for p in parts]))
    np.testing.assert_array_equal(s, np.concatenate([p[2] for p in parts]))
    raw = open(out / "part-r-00000", "rb").read()
    assert b"org.apache.mahout.common.IntPairWritable" in raw and b"org.apache.hadoop.io.FloatWritable" in raw
    first = raw.index(b"FloatWritable") + len(b"FloatWritable") + 6 + 16
    rl, kl, a, b = struct.unpack(">iiii", raw[first:first + 16])
    assert (rl, kl, a, b) == (12, 8, int(parts[0][0][0]), int(parts[0][1][0]))              # two big-endian int32: layout unpinned


def test_mapfile_data_and_index(sf, tmp_path):
    d = str(tmp_path / "itemColl" / "part-r-00000")
    keys = np.arange(1, 1001, dtype=np.int32) * 3
    vals = 1.0 / keys
    sf.write_mapfile_int_double(d, keys, vals)
    assert sorted(os.listdir(d)) == ["data", "index"]
    k, v = sf.read_int_double(d)                                                            # a MapFile directory reads its data file
    np.testing.assert_array_equal(k, keys)
    np.testing.assert_array_equal(v, vals)
    k, v = sf.read_int_double(str(tmp_path / "itemColl"))                                   # the job output directory above it
    np.testing.assert_array_equal(k, keys)
    raw = open(os.path.join(d, "index"), "rb").read()
    assert b"org.apache.hadoop.io.LongWritable" in raw
    first = raw.index(b"LongWritable") + len(b"LongWritable") + 6 + 16
    recs = [struct.unpack(">iiiq", raw[first + 20 * t:first + 20 * t + 20]) for t in range(8)]
    assert [r[2] for r in recs] == [int(keys[128 * t]) for t in range(8)]                   # every 128th key ...
    data = open(os.path.join(d, "data"), "rb").read()
    for _, _, key, pos in recs:                                                             # ... with the position of its record
        p = pos + 20 if data[pos:pos + 4] == b"\xff\xff\xff\xff" else pos
        assert struct.unpack(">iii", data[p:p + 12]) == (12, 4, key)
    with pytest.raises(IOError):
        sf.write_mapfile_int_double(str(tmp_path / "bad"), [3, 2], [0.1, 0.2])             # MapFile.Writer: keys must ascend


def test_errors(sf, tmp_path):
    with pytest.raises(IOError, match="no such file"):
        sf.read_int_int(str(tmp_path / "missing"))
    f = tmp_path / "notseq"
    f.write_bytes(b"hello world, not a sequence file")
    with pytest.raises(IOError, match="not a SequenceFile"):
        sf.read_int_int(str(f))
    g = str(tmp_path / "ii")
    sf.write_int_int(g, [1, 2], [3, 4])
    with pytest.raises(IOError, match="expected"):
        sf.read_int_double(g)                                                               # wrong value class
    raw = bytearray(open(g, "rb").read())
    comp = raw.index(b"IntWritable", raw.index(b"IntWritable") + 1) + len(b"IntWritable")
    raw[comp] = 1                                                                           # the "compressed" flag
    (tmp_path / "comp").write_bytes(bytes(raw))
    with pytest.raises(IOError, match="compressed"):
        sf.read_int_int(str(tmp_path / "comp"))
    (tmp_path / "trunc").write_bytes(bytes(open(g, "rb").read()[:-3]))
    with pytest.raises(IOError):
        sf.read_int_int(str(tmp_path / "trunc"))
    e = str(tmp_path / "empty")
    sf.write_int_int(e, [], [])
    k, v = sf.read_int_int(e)
    assert len(k) == 0 and len(v) == 0
