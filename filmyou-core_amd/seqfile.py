"""The Hadoop files on either side of the RM2 job, through the library's C++ codec (csrc/fy_seqfile.cpp, fy_seqfile_* of
include/filmyou.h): SequenceFile<IntWritable,IntWritable> (clustering, clusteringCount), <IntWritable,DoubleWritable>
(rm2/userSum, and rm2/itemColl as a MapFile), <IntPairWritable,FloatWritable> (ratings, recommendations) -- what
M/util/DataInitialization.java:155-222, M/rm/RM2Job.java:110-205 and M/rm/RM2HDFSReducer.java:44-50 write.
Byte layout: Hadoop 1.2.1's published format; PARITY UNPINNED at the byte level (no binary fixture in the reference)."""
import ctypes as C
import os

import numpy as np

from . import _native


class SeqFileError(IOError):
    pass


def _lib():
    return _native.load()


def _check(rc):
    if rc != 0:
        raise SeqFileError(_lib().fy_last_error().decode("utf-8", "replace"))


def _take(ptr, n, dtype):
    """copy a malloc'ed array into numpy and release it"""
    try:
        if n == 0 or not ptr.value:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).copy()
    finally:
        if ptr.value:
            _lib().fy_buffer_free(ptr)


def read_int_int(path):
    n, k, v = C.c_int64(), C.c_void_p(), C.c_void_p()
    _check(_lib().fy_seqfile_read_int_int(os.fsencode(path), C.byref(n), C.byref(k), C.byref(v)))
    return _take(k, n.value, np.int32), _take(v, n.value, np.int32)


def read_int_double(path):
    n, k, v = C.c_int64(), C.c_void_p(), C.c_void_p()
    _check(_lib().fy_seqfile_read_int_double(os.fsencode(path), C.byref(n), C.byref(k), C.byref(v)))
    return _take(k, n.value, np.int32), _take(v, n.value, np.float64)


def read_intpair_float(path):
    n, a, b, v = C.c_int64(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    _check(_lib().fy_seqfile_read_intpair_float(os.fsencode(path), C.byref(n), C.byref(a), C.byref(b), C.byref(v)))
    return _take(a, n.value, np.int32), _take(b, n.value, np.int32), _take(v, n.value, np.float32)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def write_int_int(file, key, value):
    k, v = _arr(key, np.int32), _arr(value, np.int32)
    assert len(k) == len(v)
    _check(_lib().fy_seqfile_write_int_int(os.fsencode(file), len(k), k.ctypes.data, v.ctypes.data))


def write_int_double(file, key, value):
    k, v = _arr(key, np.int32), _arr(value, np.float64)
    assert len(k) == len(v)
    _check(_lib().fy_seqfile_write_int_double(os.fsencode(file), len(k), k.ctypes.data, v.ctypes.data))


def write_intpair_float(file, first, second, value):
    a, b, v = _arr(first, np.int32), _arr(second, np.int32), _arr(value, np.float32)
    assert len(a) == len(b) == len(v)
    _check(_lib().fy_seqfile_write_intpair_float(os.fsencode(file), len(a), a.ctypes.data, b.ctypes.data, v.ctypes.data))


def write_mapfile_int_double(directory, key, value):
    k, v = _arr(key, np.int32), _arr(value, np.float64)
    assert len(k) == len(v)
    _check(_lib().fy_mapfile_write_int_double(os.fsencode(directory), len(k), k.ctypes.data, v.ctypes.data))
