"""Import pygraphblas_amd/rmat.py WITHOUT importing the package (which loads the HIP library): the pure-CPU tests
only need the generator."""
import importlib.util
import os

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pygraphblas_amd", "rmat.py")
_spec = importlib.util.spec_from_file_location("_rmat_standalone", _p)
rmat = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(rmat)
