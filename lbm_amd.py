"""Import alias for the package directory `opencl-lattice-boltzmann_amd/` (its name is not a
valid Python identifier): `import lbm_amd` gives that package."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "opencl-lattice-boltzmann_amd")
_spec = importlib.util.spec_from_file_location("lbm_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lbm_amd"] = _mod
_spec.loader.exec_module(_mod)
