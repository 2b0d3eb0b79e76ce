"""Offline stand-in for `ssgetpy` (test harness: lets the reference's notebooks run their own cells on a machine without a network).

The reference's `Matrix.ssget` (pygraphblas/matrix.py:597-625) does  `ssgetpy.search(name)[0].download(extract=True)`  and reads every `*.mtx` of the
directory it gets back.  The only matrix the north star's notebooks ask for by name is Zachary's karate club (`Matrix.ssget('Newman/karate')`,
demo/Triangle-Counting.ipynb:21, demo/TriangleCentrality.ipynb) — 34 vertices, 78 edges, 45 triangles — which networkx carries:
`networkx.karate_club_graph()` is the same graph as SuiteSparse's Newman/karate (the collection's file lists the 78 edges of the lower triangle as a
`pattern symmetric` coordinate matrix; the reference's docstring shows 156 stored entries of type BOOL after loading).  Round 6: `Matrix.ssget('Gleich/wikipedia-20070206')` (demo/PageRank.ipynb cell 2, the input of its PageRank run) is answered with a SYNTHETIC stand-in — a
directed R-MAT-13 pattern matrix from this repo's generator, written as `wikipedia-20070206.mtx` — so that the notebook's hot-path cells (`pagerank(W, d, 0.85,
100)`: `A.plus_second(w, out=r, accum=FP32.plus, desc=T0)` per iteration) execute through the shim.  It is NOT the Wikipedia graph and no expected value is attached
to it; the file's comment line says so.  Anything else raises, as a search without a network would."""
import os
import tempfile


class _Karate:
    name = "karate"; group = "Newman"; id = 2399; rows = cols = 34; nnz = 156

    def download(self, format="MM", destpath=None, extract=False):
        import networkx as nx
        G = nx.karate_club_graph()
        d = os.path.join(destpath or tempfile.mkdtemp(prefix="ssgetpy_stub_"), "karate")
        os.makedirs(d, exist_ok=True)
        edges = sorted((max(u, v) + 1, min(u, v) + 1) for u, v in G.edges())
        with open(os.path.join(d, "karate.mtx"), "w") as f:
            f.write("%%MatrixMarket matrix coordinate pattern symmetric\n% Newman/karate (offline stand-in: networkx.karate_club_graph)\n")
            f.write(f"{G.number_of_nodes()} {G.number_of_nodes()} {len(edges)}\n")
            for i, j in edges:
                f.write(f"{i} {j}\n")
        return d, None


class _WikipediaStandIn:
    name = "wikipedia-20070206"; group = "Gleich"; id = 0; rows = cols = 1 << 13; nnz = 0

    def download(self, format="MM", destpath=None, extract=False):
        import sys
        import numpy as np
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        sys.path.insert(0, root)
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("_grb_rmat_for_stub", os.path.join(root, "pygraphblas_amd", "rmat.py"))
            rm = importlib.util.module_from_spec(spec); spec.loader.exec_module(rm)      # (the generator alone: numpy only, no library handle)
        finally:
            sys.path.pop(0)
        scale = 13; n = 1 << scale
        s, d = rm.edges_numpy(scale, seed=42, edgefactor=8)
        key = np.unique((s[s != d] << np.uint64(32)) | d[s != d])
        i, j = (key >> np.uint64(32)).astype(np.int64) + 1, (key & np.uint64(0xFFFFFFFF)).astype(np.int64) + 1
        dd = os.path.join(destpath or tempfile.mkdtemp(prefix="ssgetpy_stub_"), "wikipedia-20070206")
        os.makedirs(dd, exist_ok=True)
        with open(os.path.join(dd, "wikipedia-20070206.mtx"), "w") as f:
            f.write("%%MatrixMarket matrix coordinate pattern general\n% SYNTHETIC STAND-IN (offline harness): directed R-MAT-13, edge factor 8 - NOT Gleich/wikipedia-20070206\n")
            f.write(f"{n} {n} {len(i)}\n")
            f.write("\n".join(f"{a} {b}" for a, b in zip(i.tolist(), j.tolist())) + "\n")
        return dd, None


def search(name_or_id=None, **kwargs):
    if name_or_id in ("Newman/karate", "karate", 2399):
        return [_Karate()]
    if name_or_id in ("Gleich/wikipedia-20070206", "wikipedia-20070206"):
        return [_WikipediaStandIn()]
    raise RuntimeError(f"ssgetpy stand-in: no network; only Newman/karate is served offline (asked for {name_or_id!r})")
