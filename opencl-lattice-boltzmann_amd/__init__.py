"""opencl-lattice-boltzmann_amd — MI355X-native D2Q9-BGK lattice-Boltzmann timestep.

Host-side mirror of the C ABI in include/lbm.h (liblbm_hip.so, hand-written HIP for gfx950).
The product's host is the C program `d2q9-bgk` (host/d2q9-bgk.c); this module is the ctypes
binding used by bench.py, __graft_entry__.py and the tests.  There is NO CPU fallback: loading
fails loudly when the HIP library is missing, and every entry point raises LBMError on a non-zero
return code (the reference's checkError() prints and exits, d2q9-bgk.c:858-866).

The directory name contains '-' and is therefore loaded through `lbm_amd.py` at the repo root
(`import lbm_amd`).
"""
import ctypes
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, os.environ.get("LBM_LIB", "liblbm_hip.so"))  # LBM_LIB: A/B of two builds (tools/ab_two_libs.py)

# every symbol include/lbm.h declares
ABI_SYMBOLS = [
    "lbm_create", "lbm_create_rank", "lbm_comm_id_size", "lbm_comm_get_id", "lbm_upload", "lbm_run",
    "lbm_run_timed", "lbm_sync", "lbm_download", "lbm_steps_done", "lbm_row_range", "lbm_final_state",
    "lbm_reynolds", "lbm_set_option", "lbm_get_option", "lbm_copy_bandwidth", "lbm_valu_rate", "lbm_destroy",
    "lbm_last_error", "lbm_version", "lbm_set_default", "lbm_peer_info_size", "lbm_peer_info", "lbm_connect_peers",
    "lbm_run_profiled", "lbm_upload_obstacles", "lbm_disconnect_peers", "lbm_host_alloc", "lbm_host_free",
]

TRANSPORTS = {"auto": 0, "rccl": 1, "copy": 2, "peer": 3}


class LBMError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """lbm_params == the reference's t_param (d2q9-bgk.c:81-92)."""
    _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("max_iters", ctypes.c_int),
                ("reynolds_dim", ctypes.c_int), ("density", ctypes.c_float), ("accel", ctypes.c_float),
                ("omega", ctypes.c_float), ("free_cells_inv", ctypes.c_float)]


def build_library(verbose=False):
    """Compile liblbm_hip.so and the d2q9-bgk host for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", ROOT, "-j4", "all"], check=True, stdout=out)


_lib = None


def load_library():
    """dlopen liblbm_hip.so and declare the prototypes of include/lbm.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LBMError("HIP library %s is missing: run `make` (or __graft_entry__.build()); "
                       "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
    L.lbm_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(Params), vp, ci, ctypes.POINTER(ci)]
    L.lbm_create_rank.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(Params), vp, ci, ci, ci, vp]
    L.lbm_comm_id_size.restype = ctypes.c_size_t
    L.lbm_comm_get_id.argtypes = [vp]
    L.lbm_upload.argtypes = [vp, vp]
    L.lbm_upload_obstacles.argtypes = [vp, vp]
    L.lbm_run.argtypes = [vp, ci]
    L.lbm_run_timed.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    L.lbm_run_profiled.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    L.lbm_sync.argtypes = [vp]
    L.lbm_download.argtypes = [vp, vp, vp]
    L.lbm_steps_done.argtypes = [vp]
    L.lbm_row_range.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.lbm_final_state.argtypes = [vp, vp, vp, vp, vp]
    L.lbm_reynolds.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.lbm_set_option.argtypes = [vp, cp, ctypes.c_long]
    L.lbm_get_option.argtypes = [vp, cp, ctypes.POINTER(ctypes.c_long)]
    L.lbm_copy_bandwidth.argtypes = [ctypes.c_size_t, ci, ctypes.POINTER(ctypes.c_double)]
    L.lbm_valu_rate.argtypes = [ci, ctypes.POINTER(ctypes.c_double)]
    L.lbm_set_default.argtypes = [cp, ctypes.c_long]
    L.lbm_peer_info_size.restype = ctypes.c_size_t
    L.lbm_peer_info.argtypes = [vp, vp]
    L.lbm_connect_peers.argtypes = [vp, vp, vp]
    L.lbm_disconnect_peers.argtypes = [vp]
    L.lbm_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    L.lbm_host_free.argtypes = [vp]
    L.lbm_destroy.argtypes = [vp]
    L.lbm_destroy.restype = None
    L.lbm_last_error.restype = cp
    L.lbm_version.restype = cp
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise LBMError("%s failed (code %d): %s" % (what, rc, load_library().lbm_last_error().decode()))


def make_params(nx, ny, max_iters, reynolds_dim=10, density=0.1, accel=0.005, omega=1.85, obstacles=None):
    """Run constants; free_cells_inv from the mask as in d2q9-bgk.c:583-591."""
    p = Params()
    p.nx, p.ny, p.max_iters, p.reynolds_dim = nx, ny, max_iters, reynolds_dim
    p.density, p.accel, p.omega = density, accel, omega
    free_cells = nx * ny if obstacles is None else int(obstacles.size - np.count_nonzero(obstacles))
    p.free_cells_inv = np.float32(1.0) / np.float32(free_cells)
    return p


def read_inputs(paramfile, obstaclefile):
    """Parse the reference's two input files (d2q9-bgk.c:466-492, 553-591) into (Params, mask)."""
    with open(paramfile) as f:
        tok = f.read().split()
    if len(tok) < 7:
        raise ValueError("could not read param file: %s" %
                         ["nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega"][len(tok)])
    nx, ny, max_iters, reynolds_dim = (int(t) for t in tok[:4])
    density, accel, omega = (float(t) for t in tok[4:7])
    obstacles = np.zeros((ny, nx), dtype=np.int32)
    with open(obstaclefile) as f:
        vals = f.read().split()
    if len(vals) % 3:
        raise ValueError("expected 3 values per line in obstacle file")
    if vals:
        tri = np.array(vals, dtype=np.int64).reshape(-1, 3)
        if np.any(tri[:, 0] < 0) or np.any(tri[:, 0] > nx - 1):
            raise ValueError("obstacle x-coord out of range")
        if np.any(tri[:, 1] < 0) or np.any(tri[:, 1] > ny - 1):
            raise ValueError("obstacle y-coord out of range")
        if np.any(tri[:, 2] != 1):
            raise ValueError("obstacle blocked value should be 1")
        obstacles[tri[:, 1], tri[:, 0]] = 1
    return make_params(nx, ny, max_iters, reynolds_dim, density, accel, omega, obstacles), obstacles


# ---- row partition: the host-side geometry of csrc/lbm_hip.cpp (split_rows, build_slab, exchange_halos) ----

# distributions that cross a slab edge in the pull scheme: a cell reads f2,f5,f6 from the row below and
# f4,f7,f8 from the row above (kernels.cl:104-112), so a slab sends its top row's 2,5,6 north and its
# bottom row's 4,7,8 south
HALO_PLANES = {"to_north": [2, 5, 6], "to_south": [4, 7, 8]}


def slab_rows(ny, nslabs, index):
    """(first global row, row count) of slab `index` of `nslabs`: contiguous rows, sizes differ by <= 1."""
    base, rem = divmod(ny, nslabs)
    return index * base + min(index, rem), base + (1 if index < rem else 0)


def ring_neighbours(nslabs, index):
    """(south, north) neighbours on the periodic ring of slabs (the grid wraps in y, kernels.cl:91-93)."""
    return (index + nslabs - 1) % nslabs, (index + 1) % nslabs


def accel_row_local(ny, y0, rows):
    """local index of the accelerated global row ny-2 (kernels.cl:18) inside a slab, or -1."""
    ar = ny - 2
    return ar - y0 if y0 <= ar < y0 + rows else -1


def valu_rate_tera(launches=40):
    """issue rate of packed fp32 FMAs in 1e12 lane-instructions per second (the roofline of the issue-bound kernels)"""
    g = ctypes.c_double(0.0)
    _check(load_library().lbm_valu_rate(launches, ctypes.byref(g)), "lbm_valu_rate")
    return g.value


def copy_bandwidth_gbps(nbytes=1 << 30, iters=20):
    """Measured float4 streaming-copy rate (read + write bytes per second) — the roofline denominator."""
    g = ctypes.c_double()
    _check(load_library().lbm_copy_bandwidth(nbytes, iters, ctypes.byref(g)), "lbm_copy_bandwidth")
    return g.value


def set_default(key, value):
    """Process-wide default for contexts created afterwards (lbm_set_default): force_halo, halo_depth, transport
    (number or one of TRANSPORTS), lanes_out."""
    if key == "transport" and isinstance(value, str):
        value = TRANSPORTS[value]
    _check(load_library().lbm_set_default(key.encode(), int(value)), "lbm_set_default(%s)" % key)


class HostBuffer:
    """Page-locked host memory from lbm_host_alloc, seen as a numpy array (`.array`): a read-back target that device -> host
    copies fill at the PCIe rate.  Free with close() (or leave it to the garbage collector)."""

    def __init__(self, shape, dtype=np.float32):
        self.lib = load_library()
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = ctypes.c_void_p()
        _check(self.lib.lbm_host_alloc(ctypes.byref(self.ptr), nbytes), "lbm_host_alloc")
        raw = (ctypes.c_char * nbytes).from_address(self.ptr.value)
        self.array = np.frombuffer(raw, dtype=dtype).reshape(shape)

    def close(self):
        if self.ptr:
            self.array = None
            self.lib.lbm_host_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_id():
    """RCCL unique id blob for lbm_create_rank (produce on one rank, broadcast to the others)."""
    L = load_library()
    buf = ctypes.create_string_buffer(L.lbm_comm_id_size())
    _check(L.lbm_comm_get_id(buf), "lbm_comm_get_id")
    return buf.raw


class LBM:
    """A simulation context (lbm_ctx).  Mirrors the call sequence of the reference's main()
    (d2q9-bgk.c:194-277): create -> upload -> run -> sync -> download -> destroy."""

    def __init__(self, params, obstacles, devices=None, rank=None, nranks=None, device=0, comm=None):
        self.lib = load_library()
        self.params = params
        self.nx, self.ny = params.nx, params.ny
        obst = np.ascontiguousarray(obstacles, dtype=np.int32)
        assert obst.shape == (params.ny, params.nx)
        self.obstacles = obst
        self.ctx = ctypes.c_void_p()
        if rank is not None:
            cid = ctypes.create_string_buffer(comm, len(comm)) if comm is not None else None
            _check(self.lib.lbm_create_rank(ctypes.byref(self.ctx), ctypes.byref(params), obst.ctypes.data,
                                            rank, nranks, device, cid), "lbm_create_rank")
        elif devices is None:
            _check(self.lib.lbm_create(ctypes.byref(self.ctx), ctypes.byref(params), obst.ctypes.data, 1, None),
                   "lbm_create")
        else:
            arr = (ctypes.c_int * len(devices))(*devices)
            _check(self.lib.lbm_create(ctypes.byref(self.ctx), ctypes.byref(params), obst.ctypes.data,
                                       len(devices), arr), "lbm_create")

    def upload(self, cells=None):
        if cells is None:
            _check(self.lib.lbm_upload(self.ctx, None), "lbm_upload")
        else:
            c = np.ascontiguousarray(cells, dtype=np.float32)
            assert c.shape == (9, self.ny, self.nx)
            _check(self.lib.lbm_upload(self.ctx, c.ctypes.data), "lbm_upload")

    def upload_obstacles(self, obstacles):
        """the obstacle map once more (d2q9-bgk.c:205-209); same shape as at creation"""
        ob = np.ascontiguousarray(obstacles, dtype=np.int32)
        assert ob.shape == (self.ny, self.nx)
        _check(self.lib.lbm_upload_obstacles(self.ctx, ob.ctypes.data), "lbm_upload_obstacles")
        self.obstacles = ob

    def run(self, nsteps):
        _check(self.lib.lbm_run(self.ctx, nsteps), "lbm_run")

    def run_timed(self, nsteps):
        """Runs nsteps and returns the HIP-event time of the step loop in milliseconds."""
        ms = ctypes.c_double()
        _check(self.lib.lbm_run_timed(self.ctx, nsteps, ctypes.byref(ms)), "lbm_run_timed")
        return ms.value

    def run_profiled(self, nsteps):
        """Runs nsteps with timing events around every launch of the first slab; dict of mean microseconds."""
        st = (ctypes.c_double * 8)()
        _check(self.lib.lbm_run_profiled(self.ctx, nsteps, st), "lbm_run_profiled")
        return {"sets": int(st[0]), "steps_per_set": st[1], "edge_us": st[2], "exchange_us": st[3], "interior_us": st[4],
                "set_period_us": st[5], "interior_start_lag_us": st[6],
                "transport": {0: "none", 1: "rccl", 2: "copy", 3: "peer"}.get(int(st[7]), "?")}

    def sync(self):
        _check(self.lib.lbm_sync(self.ctx), "lbm_sync")

    @property
    def steps_done(self):
        return self.lib.lbm_steps_done(self.ctx)

    def row_range(self):
        y0, y1 = ctypes.c_int(), ctypes.c_int()
        _check(self.lib.lbm_row_range(self.ctx, ctypes.byref(y0), ctypes.byref(y1)), "lbm_row_range")
        return y0.value, y1.value

    def download(self, cells=True, av_vels=True):
        """Returns (cells float32[9,ny,nx] or None, av_vels float32[steps_done] or None)."""
        c = np.zeros((9, self.ny, self.nx), dtype=np.float32) if cells else None
        a = np.zeros(max(self.steps_done, 1), dtype=np.float32) if av_vels else None
        _check(self.lib.lbm_download(self.ctx, c.ctypes.data if cells else None,
                                     a.ctypes.data if av_vels else None), "lbm_download")
        return c, (a[:self.steps_done] if av_vels else None)

    def final_state(self, out=None):
        """(u_x, u_y, u, pressure), each float32[ny,nx] — the columns of final_state.dat.  `out`: a float32[4,ny,nx] array
        to fill instead of fresh ones (e.g. HostBuffer((4, ny, nx)).array: page-locked)."""
        if out is not None:
            assert out.shape == (4, self.ny, self.nx) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
            outs = [out[i] for i in range(4)]
        else:
            outs = [np.zeros((self.ny, self.nx), dtype=np.float32) for _ in range(4)]
        _check(self.lib.lbm_final_state(self.ctx, *[o.ctypes.data for o in outs]), "lbm_final_state")
        return outs

    def reynolds(self):
        r = ctypes.c_float()
        _check(self.lib.lbm_reynolds(self.ctx, ctypes.byref(r)), "lbm_reynolds")
        return r.value

    def peer_info(self):
        """This rank's peer descriptor (bytes) for lbm_connect_peers on its ring neighbours."""
        buf = ctypes.create_string_buffer(self.lib.lbm_peer_info_size())
        _check(self.lib.lbm_peer_info(self.ctx, buf), "lbm_peer_info")
        return buf.raw

    def connect_peers(self, south_info, north_info):
        """Descriptors of ranks (rank-1) and (rank+1) mod nranks: switches the halo transport to peer stores."""
        so = ctypes.create_string_buffer(south_info, len(south_info))
        no = ctypes.create_string_buffer(north_info, len(north_info))
        _check(self.lib.lbm_connect_peers(self.ctx, so, no), "lbm_connect_peers")

    def disconnect_peers(self):
        """Unmap the neighbours' grids.  Between processes: every rank calls this, then a barrier, then close() — memory that
        another process still has mapped must not be freed."""
        _check(self.lib.lbm_disconnect_peers(self.ctx), "lbm_disconnect_peers")

    def set_option(self, key, value):
        _check(self.lib.lbm_set_option(self.ctx, key.encode(), int(value)), "lbm_set_option(%s)" % key)

    def get_option(self, key):
        v = ctypes.c_long()
        _check(self.lib.lbm_get_option(self.ctx, key.encode(), ctypes.byref(v)), "lbm_get_option(%s)" % key)
        return v.value

    def write_values(self, final_state_path="final_state.dat", av_vels_path="av_vels.dat"):
        """The two output files in the reference's format (d2q9-bgk.c:835,848-851)."""
        ux, uy, u, pr = self.final_state()
        _, av = self.download(cells=False)
        yy, xx = np.mgrid[0:self.ny, 0:self.nx]
        cols = np.stack([xx.ravel(), yy.ravel(), ux.ravel(), uy.ravel(), u.ravel(), pr.ravel(),
                         self.obstacles.ravel()], axis=1)
        np.savetxt(final_state_path, cols, fmt=["%d", "%d", "%.12E", "%.12E", "%.12E", "%.12E", "%d"])
        with open(av_vels_path, "w") as f:
            for i, v in enumerate(av):
                f.write("%d:\t%.12E\n" % (i, v))

    def close(self):
        if self.ctx:
            self.lib.lbm_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
