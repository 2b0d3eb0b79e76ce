"""TEST HARNESS ONLY — a stand-in for the `graphviz` package, which is not installed in this image.

The reference's doctests (`pygraphblas.run_doctests`) draw many of their example matrices with `pygraphblas.gviz`; with no
`graphviz` module the `from pygraphblas import ... gviz` line of a docstring raises and every arithmetic example after it
fails with NameError.  This stub records the calls and renders nothing, so those examples run; it is put on PYTHONPATH by
tools/ref_doctests_gpu.sh alone and is not part of the shim or the library."""
from .dot import Digraph, Graph, Source  # noqa: F401
