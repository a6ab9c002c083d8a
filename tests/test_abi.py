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


def test_every_launched_deep_kernel_instance_is_in_the_library(lbm):
    """the deep window kernels live in a translation unit of their own (csrc/lbm_deep.cpp instantiates the list of
    csrc/deep_instances.h, lbm_hip.cpp only declares them `extern template`): every instance lbm_hip.cpp launches must be on
    that list — the link runs with -z defs, and the library must carry no undefined symbol of the lbm namespace"""
    import subprocess
    so = os.path.join(ROOT, "opencl-lattice-boltzmann_amd", "liblbm_hip.so")
    undefined = subprocess.run(["nm", "-uC", so], capture_output=True, text=True, check=True).stdout
    assert "lbm::" not in undefined, [ln for ln in undefined.splitlines() if "lbm::" in ln][:5]
    src = open(os.path.join(ROOT, "opencl-lattice-boltzmann_amd", "csrc", "lbm_hip.cpp")).read()
    listed = open(os.path.join(ROOT, "opencl-lattice-boltzmann_amd", "csrc", "deep_instances.h")).read()
    launched = set(re.findall(r"hipLaunchKernelGGL\(\((d2q9_deep(?:_twin)?<[^>]*>)\)", src))
    assert len(launched) == 37
    consts = {"kDeepSteps": "8", "kDeepTwinSteps": "8", "kDeepTwinDefault": "5", "D5": "5"}
    n_listed = len(re.findall(r"X\(d2q9_deep", listed))
    assert n_listed == len(launched), (n_listed, len(launched))
    for inst in launched:      # same kernel family and depth at least (defaulted template arguments are spelled out in the list)
        name, args = inst.split("<", 1)
        first = consts.get(args.split(",")[0].strip(), args.split(",")[0].strip())
        assert re.search(r"X\(%s<%s, " % (name, first), listed), inst


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
