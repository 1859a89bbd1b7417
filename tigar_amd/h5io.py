"""Minimal HDF5 reader / writer over ``libhdf5`` through ctypes (no h5py in this image): groups, n-d datasets of
float64 / int64 / uint64, string and numeric attributes -- what ``extraction-data.h5`` (tIGAr/common.py:460-467:
``HDF5File.write(mesh, "/mesh")`` and ``HDF5File.write(cpFuncs[i], "/control<i>")``) consists of.

The library is looked up at ``TIGAR_HDF5_LIB``, then by name, then under ``/opt/conda/lib``; ``available()`` tells
whether it could be loaded.  Host-side file I/O only: nothing here touches the device."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy

_lib = None
_err = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
hid_t = C.c_int64
hsize_t = C.c_uint64


def _load():
    global _lib, _err
    if _lib is not None or _err is not None:
        return _lib
    names = []
    if os.environ.get("TIGAR_HDF5_LIB"):
        names.append(os.environ["TIGAR_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        names.append(found)
    names += ["libhdf5.so"] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/*/libhdf5*.so*"))
    for nm in names:
        try:
            lib = C.CDLL(nm)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                continue
            _lib = lib
            break
        except OSError as e:           # noqa: PERF203
            _err = e
    if _lib is None:
        _err = _err or OSError("libhdf5 not found")
        return None
    L = _lib
    for f, res, args in [
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ("H5Fclose", C.c_int, [hid_t]), ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            ("H5Gclose", C.c_int, [hid_t]), ("H5Oopen", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Oclose", C.c_int, [hid_t]),
            ("H5Lexists", C.c_int, [hid_t, C.c_char_p, hid_t]),
            ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), ("H5Screate", hid_t, [C.c_int]),
            ("H5Sclose", C.c_int, [hid_t]), ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", C.c_int, [hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]),
            ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tclose", C.c_int, [hid_t]),
            ("H5Tget_class", C.c_int, [hid_t]), ("H5Tget_size", C.c_size_t, [hid_t]), ("H5Tget_sign", C.c_int, [hid_t]),
            ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]),
            ("H5Aexists", C.c_int, [hid_t, C.c_char_p]), ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]),
            ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]), ("H5Aget_type", hid_t, [hid_t]), ("H5Aget_space", hid_t, [hid_t]),
            ("H5Aclose", C.c_int, [hid_t]), ("H5Eset_auto2", C.c_int, [hid_t, C.c_void_p, C.c_void_p])]:
        fn = getattr(L, f)
        fn.restype = res
        fn.argtypes = args
    L.H5Eset_auto2(0, None, None)              # errors are reported through return values (checked below), not on stderr
    return _lib


def available():
    return _load() is not None


def _need():
    lib = _load()
    if lib is None:
        raise IOError("HDF5 library not available (%s); set TIGAR_HDF5_LIB to a libhdf5.so" % (_err,))
    return lib


def _tid(name):
    return hid_t.in_dll(_need(), name).value


def _mem_type(dtype):
    dtype = numpy.dtype(dtype)
    if dtype == numpy.float64:
        return _tid("H5T_NATIVE_DOUBLE_g"), _tid("H5T_IEEE_F64LE_g")
    if dtype == numpy.int64:
        return _tid("H5T_NATIVE_INT64_g"), _tid("H5T_STD_I64LE_g")
    if dtype == numpy.uint64:
        return _tid("H5T_NATIVE_UINT64_g"), _tid("H5T_STD_U64LE_g")
    raise TypeError("h5io: dtype %s is not supported (float64, int64, uint64)" % dtype)


def _chk(v, what):
    if v < 0:
        raise IOError("HDF5: %s failed" % what)
    return v


class H5File(object):
    """``with H5File(path, "w") as f: f.create_group("/mesh"); f.write_dataset("/mesh/coordinates", X, attrs={...})``"""

    def __init__(self, path, mode="r"):
        L = _need()
        self.L = L
        p = os.fsencode(path)
        if mode == "w":
            self.fid = _chk(L.H5Fcreate(p, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate(%s)" % path)
        elif mode == "r":
            self.fid = _chk(L.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fopen(%s)" % path)
        else:
            raise ValueError("mode must be 'r' or 'w'")

    def close(self):
        if self.fid is not None:
            self.L.H5Fclose(self.fid)
            self.fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- structure
    def exists(self, path):
        """every link along ``path`` exists"""
        parts = [q for q in path.split("/") if q]
        cur = ""
        for q in parts:
            cur += "/" + q
            if self.L.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def create_group(self, path):
        parts = [q for q in path.split("/") if q]
        cur = ""
        for q in parts:
            cur += "/" + q
            if self.L.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) > 0:
                continue
            g = _chk(self.L.H5Gcreate2(self.fid, cur.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2(%s)" % cur)
            self.L.H5Gclose(g)

    # ---- datasets
    def write_dataset(self, path, array, attrs=None):
        L = self.L
        a = numpy.ascontiguousarray(array)
        mem, filet = _mem_type(a.dtype)
        parent = path.rsplit("/", 1)[0]
        if parent:
            self.create_group(parent)
        dims = (hsize_t * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
        sp = _chk(L.H5Screate_simple(max(a.ndim, 1), dims, None), "H5Screate_simple")
        ds = _chk(L.H5Dcreate2(self.fid, path.encode(), filet, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2(%s)" % path)
        if a.size:
            _chk(L.H5Dwrite(ds, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), "H5Dwrite(%s)" % path)
        for k, v in (attrs or {}).items():
            self._write_attr(ds, k, v)
        L.H5Dclose(ds)
        L.H5Sclose(sp)

    def read_dataset(self, path):
        L = self.L
        ds = _chk(L.H5Dopen2(self.fid, path.encode(), H5P_DEFAULT), "H5Dopen2(%s)" % path)
        try:
            sp = L.H5Dget_space(ds)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[i]) for i in range(nd))
            L.H5Sclose(sp)
            t = L.H5Dget_type(ds)
            dtype = self._numpy_type(t, path)
            L.H5Tclose(t)
            out = numpy.empty(shape, dtype=dtype)
            if out.size:
                _chk(L.H5Dread(ds, _mem_type(dtype)[0], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)),
                     "H5Dread(%s)" % path)
            return out
        finally:
            L.H5Dclose(ds)

    def _numpy_type(self, t, what):
        cls, size = self.L.H5Tget_class(t), self.L.H5Tget_size(t)
        if cls == H5T_FLOAT and size == 8:
            return numpy.float64
        if cls == H5T_INTEGER and size == 8:
            return numpy.int64 if self.L.H5Tget_sign(t) != 0 else numpy.uint64
        raise TypeError("h5io: %s has a type this reader does not handle (class %d, %d bytes)" % (what, cls, size))

    # ---- attributes (on a group or a dataset)
    def _write_attr(self, obj, name, value):
        L = self.L
        if isinstance(value, str):
            raw = value.encode()
            t = L.H5Tcopy(_tid("H5T_C_S1_g"))
            L.H5Tset_size(t, max(len(raw), 1))
            sp = L.H5Screate(H5S_SCALAR)
            at = _chk(L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2(%s)" % name)
            buf = C.create_string_buffer(raw, max(len(raw), 1))
            _chk(L.H5Awrite(at, t, buf), "H5Awrite(%s)" % name)
            L.H5Aclose(at)
            L.H5Sclose(sp)
            L.H5Tclose(t)
            return
        a = numpy.asarray(value)
        if a.ndim:
            a = numpy.ascontiguousarray(a)
        mem, filet = _mem_type(a.dtype)
        if a.ndim == 0:
            sp = L.H5Screate(H5S_SCALAR)
        else:
            dims = (hsize_t * a.ndim)(*a.shape)
            sp = L.H5Screate_simple(a.ndim, dims, None)
        at = _chk(L.H5Acreate2(obj, name.encode(), filet, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2(%s)" % name)
        _chk(L.H5Awrite(at, mem, a.ctypes.data_as(C.c_void_p)), "H5Awrite(%s)" % name)
        L.H5Aclose(at)
        L.H5Sclose(sp)

    def write_attr(self, path, name, value):
        obj = _chk(self.L.H5Oopen(self.fid, path.encode(), H5P_DEFAULT), "H5Oopen(%s)" % path)
        try:
            self._write_attr(obj, name, value)
        finally:
            self.L.H5Oclose(obj)

    def has_attr(self, path, name):
        obj = _chk(self.L.H5Oopen(self.fid, path.encode(), H5P_DEFAULT), "H5Oopen(%s)" % path)
        try:
            return self.L.H5Aexists(obj, name.encode()) > 0
        finally:
            self.L.H5Oclose(obj)

    def read_attr(self, path, name):
        L = self.L
        obj = _chk(L.H5Oopen(self.fid, path.encode(), H5P_DEFAULT), "H5Oopen(%s)" % path)
        try:
            at = _chk(L.H5Aopen(obj, name.encode(), H5P_DEFAULT), "H5Aopen(%s of %s)" % (name, path))
            t = L.H5Aget_type(at)
            try:
                if L.H5Tget_class(t) == H5T_STRING:
                    n = L.H5Tget_size(t)
                    buf = C.create_string_buffer(n + 1)
                    _chk(L.H5Aread(at, t, buf), "H5Aread(%s)" % name)
                    return buf.raw[:n].split(b"\0", 1)[0].decode()
                dtype = self._numpy_type(t, "attribute %s of %s" % (name, path))
                sp = L.H5Aget_space(at)
                nd = L.H5Sget_simple_extent_ndims(sp)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    L.H5Sget_simple_extent_dims(sp, dims, None)
                L.H5Sclose(sp)
                out = numpy.empty(tuple(int(dims[i]) for i in range(nd)), dtype=dtype)
                _chk(L.H5Aread(at, _mem_type(dtype)[0], out.ctypes.data_as(C.c_void_p)), "H5Aread(%s)" % name)
                return out if nd else out[()]
            finally:
                L.H5Tclose(t)
                L.H5Aclose(at)
        finally:
            L.H5Oclose(obj)
