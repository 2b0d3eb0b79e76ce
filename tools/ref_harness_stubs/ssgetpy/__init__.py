"""Offline stand-in for `ssgetpy` (test harness: lets the reference's notebooks run their own cells on a machine without a network).

The reference's `Matrix.ssget` (pygraphblas/matrix.py:597-625) does  `ssgetpy.search(name)[0].download(extract=True)`  and reads every `*.mtx` of the
directory it gets back.  The only matrix the north star's notebooks ask for by name is Zachary's karate club (`Matrix.ssget('Newman/karate')`,
demo/Triangle-Counting.ipynb:21, demo/TriangleCentrality.ipynb) — 34 vertices, 78 edges, 45 triangles — which networkx carries:
`networkx.karate_club_graph()` is the same graph as SuiteSparse's Newman/karate (the collection's file lists the 78 edges of the lower triangle as a
`pattern symmetric` coordinate matrix; the reference's docstring shows 156 stored entries of type BOOL after loading).  Anything else raises, as a search
without a network would."""
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


def search(name_or_id=None, **kwargs):
    if name_or_id in ("Newman/karate", "karate", 2399):
        return [_Karate()]
    raise RuntimeError(f"ssgetpy stand-in: no network; only Newman/karate is served offline (asked for {name_or_id!r})")
