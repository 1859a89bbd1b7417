"""The HDF5 half of the on-disk extraction data (tIGAr/common.py:460-467): the ctypes binding to
libhdf5 (tigar_amd/h5io.py) checked by the HDF5 distribution's own ``h5dump`` where the image has it, and the arrays
``writeExtraction`` puts under ``/mesh`` and ``/control<i>`` checked against a brute-force element walk.  The layout
follows dolfin's HDF5File (2019); dolfin is absent, so a dolfin-written file cannot be compared: parity unpinned."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tigar_amd import h5io

pytestmark = pytest.mark.skipif(not h5io.available(), reason="libhdf5 not found")


def _h5dump():
    for c in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if c and os.path.exists(c):
            return c
    return None


def test_write_read_and_h5dump(tmp_path):
    p = str(tmp_path / "t.h5")
    X = np.arange(12.0).reshape(6, 2) / 7.0
    topo = np.array([[0, 1, 3, 4], [1, 2, 4, 5]], dtype=np.int64)
    with h5io.H5File(p, "w") as f:
        f.create_group("/mesh")
        f.write_dataset("/mesh/coordinates", X)
        f.write_dataset("/mesh/topology", topo, attrs={"celltype": "quadrilateral", "partition": np.zeros(1, dtype=np.uint64)})
        f.write_dataset("/control0/vector_0", np.array([0.5, -1.0, 2.0]))
        f.write_attr("/control0", "signature", "FiniteElement('Q', quadrilateral, 2)")
        f.write_attr("/control0", "count", np.int64(3))
        f.write_dataset("/empty", np.zeros(0))
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    with h5io.H5File(p, "r") as f:
        assert np.array_equal(f.read_dataset("/mesh/coordinates"), X)
        t2 = f.read_dataset("/mesh/topology")
        assert t2.dtype == np.int64 and np.array_equal(t2, topo)
        assert f.read_attr("/mesh/topology", "celltype") == "quadrilateral"
        assert f.read_attr("/mesh/topology", "partition").dtype == np.uint64
        assert f.read_attr("/control0", "signature") == "FiniteElement('Q', quadrilateral, 2)"
        assert int(f.read_attr("/control0", "count")) == 3
        assert f.exists("/control0/vector_0") and not f.exists("/control1") and not f.exists("/mesh/nothing")
        assert f.has_attr("/control0", "signature") and not f.has_attr("/control0", "other")
        assert f.read_dataset("/empty").shape == (0,)
        with pytest.raises(IOError):
            f.read_dataset("/mesh/nothing")
    with pytest.raises(IOError):
        h5io.H5File(str(tmp_path / "absent.h5"), "r")
    dump = _h5dump()
    if dump:                                   # an independent reader: the HDF5 distribution's own tool
        out = subprocess.run([dump, p], capture_output=True, text=True, check=True).stdout
        assert 'GROUP "mesh"' in out and 'DATASET "topology"' in out and 'ATTRIBUTE "celltype"' in out
        assert '"quadrilateral"' in out and "H5T_IEEE_F64LE" in out and "H5T_STD_I64LE" in out
        assert "DATASPACE  SIMPLE { ( 6, 2 ) / ( 6, 2 ) }" in out
        val = subprocess.run([dump, "-d", "/control0/vector_0", p], capture_output=True, text=True, check=True).stdout
        assert "0.5, -1, 2" in val


def test_knot_mesh_and_cell_dofs_of_a_tensor_node_grid():
    from tigar_amd import common as tc
    for dg in (False, True):
        p, nel = 2, [3, 2, 2]
        verts = [np.linspace(0, 1 + k, n + 1) for k, n in enumerate(nel)]
        per_el = [[np.linspace(verts[k][e], verts[k][e + 1], p + 1) for e in range(nel[k])] for k in range(3)]
        axes = [np.concatenate(q) if dg else np.unique(np.concatenate(q)) for q in per_el]
        g = tc.TensorNodeGrid(axes, verts, p, dg)
        X, topo = tc._knot_mesh_arrays(g)
        cd = tc._cell_dofs_arrays(g)
        XN = g.coordinates()
        assert topo.shape == (12, 8) and cd.shape == (12, 27) and X.shape == (4 * 3 * 3, 3)
        assert sorted(set(cd.ravel().tolist())) == list(range(g.num_nodes()))
        for c in range(12):
            e = [c % 3, (c // 3) % 2, c // 6]                      # cells with direction 0 fastest
            lo = np.array([verts[k][e[k]] for k in range(3)])
            hi = np.array([verts[k][e[k] + 1] for k in range(3)])
            V = X[topo[c]]
            for v in range(8):                                     # corners in lexicographic order, direction 0 fastest
                assert np.allclose(V[v], [hi[k] if (v >> k) & 1 else lo[k] for k in range(3)])
            N = XN[cd[c]]
            for a in range(27):
                loc = [a % 3, (a // 3) % 3, a // 9]
                assert np.allclose(N[a], [lo[k] + (hi[k] - lo[k]) * loc[k] / p for k in range(3)])
