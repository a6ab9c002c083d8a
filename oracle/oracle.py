"""ctypes wrapper around the CPU oracle (oracle/d2q9_oracle.c).

ORACLE — TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product package.  The oracle restates the
reference's timestep (kernels.cl:9-231, d2q9-bgk.c:396-856) in plain C; parity is pinned
against the reference's golden files in tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def build(quiet=True):
    """Compile the oracle libraries and serial drivers with gcc (oracle/Makefile)."""
    subprocess.run(["make", "-C", HERE, "-j4"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Oracle:
    """One build of the oracle: precision 'f32' or 'f64', optionally the OpenMP variant."""

    def __init__(self, precision="f32", omp=False):
        assert precision in ("f32", "f64")
        name = "liboracle_%s%s.so" % (precision, "_omp" if omp else "")
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            build()
        if omp and "OMP_NUM_THREADS" not in os.environ:
            # a container may expose far more hardware threads than its CPU share
            os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, len(os.sched_getaffinity(0)))))
        self.lib = ctypes.CDLL(path)
        self.real = np.float32 if precision == "f32" else np.float64
        creal = ctypes.c_float if precision == "f32" else ctypes.c_double
        self.creal = creal

        class Params(ctypes.Structure):
            _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("max_iters", ctypes.c_int),
                        ("reynolds_dim", ctypes.c_int), ("density", creal), ("accel", creal),
                        ("omega", creal), ("free_cells_inv", creal)]

        self.Params = Params
        L = self.lib
        assert L.oracle_real_size() == np.dtype(self.real).itemsize
        P = ctypes.POINTER(Params)
        rp = np.ctypeslib.ndpointer(dtype=self.real, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.oracle_load_params.argtypes = [ctypes.c_char_p, P, ctypes.c_char_p]
        L.oracle_load_obstacles.argtypes = [ctypes.c_char_p, P, ip, ctypes.c_char_p]
        L.oracle_init_cells.argtypes = [P, rp]
        L.oracle_accelerate_flow.argtypes = [P, rp, ip]
        L.oracle_timestep.argtypes = [P, rp, rp, ip]
        L.oracle_timestep.restype = creal
        L.oracle_accelerate_row.argtypes = [P, rp, ip, ctypes.c_int]
        L.oracle_timestep_rows.argtypes = [P, rp, rp, ip, ctypes.c_int, ctypes.c_int]
        L.oracle_timestep_rows.restype = ctypes.c_double
        L.oracle_run.argtypes = [P, rp, rp, ip, ctypes.c_void_p, ctypes.c_int]
        L.oracle_av_velocity.argtypes = [P, rp, ip]
        L.oracle_av_velocity.restype = creal
        L.oracle_calc_reynolds.argtypes = [P, rp, ip]
        L.oracle_calc_reynolds.restype = creal
        L.oracle_total_density.argtypes = [P, rp]
        L.oracle_total_density.restype = creal
        L.oracle_final_fields.argtypes = [P, rp, ip, rp, rp, rp, rp]
        L.oracle_write_values.argtypes = [P, rp, ip, rp, ctypes.c_char_p, ctypes.c_char_p]

    # ---- inputs -------------------------------------------------------------------------
    def make_params(self, nx, ny, max_iters, reynolds_dim, density, accel, omega, free_cells=None):
        p = self.Params()
        p.nx, p.ny, p.max_iters, p.reynolds_dim = nx, ny, max_iters, reynolds_dim
        # the reference holds the three reals in fp32 (d2q9-bgk.c:83-85)
        if self.real is np.float32:
            p.density, p.accel, p.omega = np.float32(density), np.float32(accel), np.float32(omega)
        else:
            p.density, p.accel, p.omega = density, accel, omega
        free_cells = nx * ny if free_cells is None else free_cells
        p.free_cells_inv = self.real(1.0) / self.real(free_cells)
        return p

    def load(self, paramfile, obstaclefile):
        """Parse the reference's two input files; returns (params, obstacles int32[ny,nx])."""
        p = self.Params()
        err = ctypes.create_string_buffer(256)
        if self.lib.oracle_load_params(paramfile.encode(), ctypes.byref(p), err):
            raise ValueError(err.value.decode())
        obstacles = np.zeros((p.ny, p.nx), dtype=np.int32)
        if self.lib.oracle_load_obstacles(obstaclefile.encode(), ctypes.byref(p), obstacles, err):
            raise ValueError(err.value.decode())
        return p, obstacles

    def set_obstacles(self, p, obstacles):
        """free_cells_inv from a mask (d2q9-bgk.c:583-591)."""
        free_cells = int(obstacles.size - np.count_nonzero(obstacles))
        p.free_cells_inv = self.real(1.0) / self.real(free_cells)
        return p

    def init_cells(self, p):
        cells = np.empty((9, p.ny, p.nx), dtype=self.real)
        self.lib.oracle_init_cells(ctypes.byref(p), cells)
        return cells

    # ---- the path -----------------------------------------------------------------------
    def accelerate_flow(self, p, cells, obstacles):
        self.lib.oracle_accelerate_flow(ctypes.byref(p), cells, obstacles)

    def timestep(self, p, src, dst, obstacles):
        return float(self.lib.oracle_timestep(ctypes.byref(p), src, dst, obstacles))

    def accelerate_row(self, p, cells, obstacles, row):
        self.lib.oracle_accelerate_row(ctypes.byref(p), cells, obstacles, row)

    def timestep_rows(self, p, src, dst, obstacles, y0, y1):
        """rows [y0, y1) only; returns the raw sum of |j|/rho over their fluid cells"""
        return float(self.lib.oracle_timestep_rows(ctypes.byref(p), src, dst, obstacles, y0, y1))

    def run(self, p, cells, obstacles, nsteps):
        """nsteps of accelerate+timestep in place on `cells`; returns av_vels[nsteps]."""
        tmp = np.empty_like(cells)
        av = np.zeros(max(nsteps, 1), dtype=self.real)
        self.lib.oracle_run(ctypes.byref(p), cells, tmp, obstacles, av.ctypes.data, nsteps)
        return av[:nsteps]

    # ---- outputs ------------------------------------------------------------------------
    def av_velocity(self, p, cells, obstacles):
        return float(self.lib.oracle_av_velocity(ctypes.byref(p), cells, obstacles))

    def reynolds(self, p, cells, obstacles):
        return float(self.lib.oracle_calc_reynolds(ctypes.byref(p), cells, obstacles))

    def total_density(self, p, cells):
        return float(self.lib.oracle_total_density(ctypes.byref(p), cells))

    def final_fields(self, p, cells, obstacles):
        """(u_x, u_y, u, pressure), each [ny, nx] — the columns of final_state.dat."""
        outs = [np.empty((p.ny, p.nx), dtype=self.real) for _ in range(4)]
        self.lib.oracle_final_fields(ctypes.byref(p), cells, obstacles, *outs)
        return outs

    def write_values(self, p, cells, obstacles, av_vels, final_state_path, av_vels_path):
        av = np.ascontiguousarray(av_vels, dtype=self.real)
        assert av.size >= p.max_iters
        if self.lib.oracle_write_values(ctypes.byref(p), cells, obstacles, av,
                                        final_state_path.encode(), av_vels_path.encode()):
            raise OSError("could not open file output file")
