"""`.grb` binary reader/writer of SuiteSparse (LAGraph layout) is out of scope for the MI355X backend."""


def binread(filename, opener=open):
    raise NotImplementedError("suitesparse_graphblas.io.binary.binread is not provided by the MI355X shim")


def binwrite(A, filename, comments=None, opener=open):
    raise NotImplementedError("suitesparse_graphblas.io.binary.binwrite is not provided by the MI355X shim")
