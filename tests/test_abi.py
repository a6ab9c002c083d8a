"""The C-ABI library loads and exports exactly what include/lbm.h declares (no compute calls: this
runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lbm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lbm):
    lib = lbm.load_library()
    syms = header_symbols()
    assert len(syms) >= 19
    assert sorted(lbm.ABI_SYMBOLS) == syms          # the binding knows the whole header
    for s in syms:
        assert hasattr(lib, s), s                   # and the .so exports it


def test_params_struct_matches_reference_t_param(lbm):
    # t_param: 4 floats + 4 ints = 32 bytes, 32-byte aligned in the reference (d2q9-bgk.c:81-92)
    assert ctypes.sizeof(lbm.Params) == 32
    names = [f[0] for f in lbm.Params._fields_]
    assert names == ["nx", "ny", "max_iters", "reynolds_dim", "density", "accel", "omega", "free_cells_inv"]


def test_version_and_error_channel(lbm):
    lib = lbm.load_library()
    assert b"gfx950" in lib.lbm_version()
    assert lib.lbm_comm_id_size() == 128            # sizeof(ncclUniqueId)
    assert lib.lbm_steps_done(None) == -1
    # argument errors are reported without touching a device
    assert lib.lbm_run(None, 1) == 1                # LBM_ERR_ARG
    assert b"NULL" in lib.lbm_last_error()
    p = lbm.make_params(2, 2, 1)
    ctx = ctypes.c_void_p()
    rc = lib.lbm_create(ctypes.byref(ctx), ctypes.byref(p), None, 1, None)
    assert rc != 0 and not ctx.value


def test_no_cpu_fallback(lbm):
    """without a GPU the product path must fail loudly, never compute on the host"""
    import numpy as np
    lib = lbm.load_library()
    n = ctypes.c_int()
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    p = lbm.make_params(16, 16, 4)
    with pytest.raises(lbm.LBMError) as e:
        lbm.LBM(p, np.zeros((16, 16), dtype=np.int32))
    assert "HIP" in str(e.value) or "device" in str(e.value)


def test_read_inputs_counts_duplicate_obstacles_once(lbm):
    # every shipped obstacle file lists the corner cells twice (d2q9-bgk.c:583-585)
    from conftest import input_files
    expect = {"128x128": 15876, "128x256": 32130, "256x256": 64516, "1024x1024": 1043462}
    for size, free in expect.items():
        p, ob = lbm.read_inputs(*input_files(size))
        assert int(ob.size - ob.sum()) == free
        assert round(1.0 / p.free_cells_inv) == free or abs(1.0 / p.free_cells_inv - free) / free < 1e-6
