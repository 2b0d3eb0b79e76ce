"""Independent second opinions on the CPU oracle: scipy.sparse for PLUS_TIMES, brute-force numpy for
MIN_PLUS / LOR_LAND / masks / accum, networkx for triangle counts, scipy.csgraph for BFS levels, and the typed
`fast_*` baseline loops against the generic restatement."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as O
from helpers_cpu import rand_matrix, rand_vector
from pygraphblas_amd_rmat import rmat


def dense(t, fill=0):
    d = np.full((t.nrows, t.ncols), fill, dtype=np.float64 if t.typ.startswith("FP") else np.int64)
    p = np.zeros((t.nrows, t.ncols), bool)
    d[t.I.astype(int), t.J.astype(int)] = t.X
    p[t.I.astype(int), t.J.astype(int)] = True
    return d, p


@pytest.mark.parametrize("typ", ["FP64", "INT64", "INT32", "FP32"])
def test_plus_times_matches_scipy(typ):
    rng = np.random.default_rng(1)
    A, B = rand_matrix(rng, typ, 40, 50, 0.1), rand_matrix(rng, typ, 50, 30, 0.1)
    r = O.mxm(O.Tuples(typ, 40, 30), A, B, "PLUS", "TIMES", typ).sorted()
    sa = sp.csr_matrix((A.X.astype(np.float64), (A.I.astype(int), A.J.astype(int))), shape=(40, 50))
    sb = sp.csr_matrix((B.X.astype(np.float64), (B.I.astype(int), B.J.astype(int))), shape=(50, 30))
    ref = (sa @ sb).toarray()
    d, p = dense(r)
    # pattern: entry exists iff some k has both operands stored (explicit zeros included)
    pat = ((sa != 0).astype(int) if False else sp.csr_matrix((np.ones(A.nvals), (A.I.astype(int), A.J.astype(int))), shape=(40, 50))) @ \
        sp.csr_matrix((np.ones(B.nvals), (B.I.astype(int), B.J.astype(int))), shape=(50, 30))
    assert np.array_equal(p, pat.toarray() > 0)
    assert np.allclose(d[p], ref[p])


def brute(A, B, add, mul, ident):
    da, pa = dense(A); db, pb = dense(B)
    out = {}
    for i in range(A.nrows):
        for j in range(B.ncols):
            ks = np.flatnonzero(pa[i] & pb[:, j])
            if len(ks):
                acc = None
                for k in ks:
                    m = mul(da[i, k], db[k, j])
                    acc = m if acc is None else add(acc, m)
                out[(i, j)] = acc
    return out


@pytest.mark.parametrize("sr", ["MIN_PLUS", "MAX_TIMES", "PLUS_PAIR", "PLUS_SECOND", "PLUS_FIRST", "MIN_MAX"])
def test_other_semirings_match_bruteforce(sr):
    rng = np.random.default_rng(2)
    A, B = rand_matrix(rng, "INT64", 12, 15, 0.3), rand_matrix(rng, "INT64", 15, 9, 0.3)
    add, mul = sr.split("_")
    f = {"MIN": min, "MAX": max, "PLUS": lambda x, y: x + y, "TIMES": lambda x, y: x * y, "PAIR": lambda x, y: 1,
         "SECOND": lambda x, y: y, "FIRST": lambda x, y: x}
    exp = brute(A, B, f[add], f[mul], None)
    got = O.mxm(O.Tuples("INT64", 12, 9), A, B, add, mul, "INT64").to_dict()
    assert got == {k: int(v) for k, v in exp.items()}


def test_mask_accum_replace_semantics_bruteforce():
    """SURVEY.md App. A items 3-4, every combination on a small case."""
    rng = np.random.default_rng(3)
    A, B, Cm = rand_matrix(rng, "INT64", 8, 8, 0.4), rand_matrix(rng, "INT64", 8, 8, 0.4), rand_matrix(rng, "INT64", 8, 8, 0.4)
    M = rand_matrix(rng, "INT64", 8, 8, 0.5)            # valued: some stored zeros
    T = brute(A, B, lambda x, y: x + y, lambda x, y: x * y, 0)
    Cd = Cm.to_dict(); Md = M.to_dict()
    for accum in (None, "PLUS"):
        for comp in (False, True):
            for struct in (False, True):
                for repl in (False, True):
                    exp = {}
                    for i in range(8):
                        for j in range(8):
                            if accum:
                                z = (Cd[(i, j)] + T[(i, j)]) if (i, j) in Cd and (i, j) in T else T.get((i, j), Cd.get((i, j)))
                            else:
                                z = T.get((i, j))
                            m = ((i, j) in Md) and (struct or Md[(i, j)] != 0)
                            if comp:
                                m = not m
                            if m:
                                if z is not None:
                                    exp[(i, j)] = int(z)
                            elif not repl and (i, j) in Cd:
                                exp[(i, j)] = int(Cd[(i, j)])
                    got = O.mxm(Cm, A, B, "PLUS", "TIMES", "INT64", mask=M, accum=accum, replace=repl, mask_comp=comp, mask_struct=struct).to_dict()
                    assert got == exp, (accum, comp, struct, repl)


def test_typecast_rules():
    # float -> int saturates, NaN -> 0, anything -> BOOL is x != 0
    A = O.Tuples("FP64", 1, 4, [0, 0, 0, 0], [0, 1, 2, 3], [1e30, -1e30, float("nan"), 2.7])
    B = O.Tuples("FP64", 4, 4, [0, 1, 2, 3], [0, 1, 2, 3], [1.0, 1.0, 1.0, 1.0])
    r = O.mxm(O.Tuples("INT8", 1, 4), A, B, "PLUS", "TIMES", "FP64")
    assert r.X.tolist() == [127, -128, 0, 2]
    r = O.mxm(O.Tuples("BOOL", 1, 4), A, B, "PLUS", "TIMES", "FP64")
    assert r.X.tolist() == [True, True, True, True]
    # UINT8 wraparound (reference tests/test_matrix.py:858-864)
    A8 = O.Tuples("UINT8", 1, 2, [0, 0], [0, 1], [200, 100]); B8 = O.Tuples("UINT8", 2, 1, [0, 1], [0, 0], [2, 3])
    assert O.mxm(O.Tuples("UINT8", 1, 1), A8, B8, "PLUS", "TIMES", "UINT8").X.tolist() == [(400 + 300) % 256]


def test_triangles_networkx_and_fast_path():
    nx = pytest.importorskip("networkx")
    G = nx.karate_club_graph()
    e = np.array([(max(u, v), min(u, v)) for u, v in G.edges()], dtype=np.uint64)
    L = O.Tuples("INT64", 34, 34, e[:, 0], e[:, 1], np.ones(len(e), np.int64))
    Cm = O.mxm(O.Tuples("INT64", 34, 34), L, L, "PLUS", "PAIR", "INT64", mask=L)
    assert int(Cm.X.sum()) == 45 == sum(nx.triangles(G).values()) // 3          # demo/Triangle-Counting.ipynb:33,56
    rp, col = rmat.csr_numpy(9, symmetric=True, drop_self_loops=True, lower=True)
    n = 1 << 9
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    Lr = O.Tuples("INT64", n, n, rows, col, np.ones(len(col), np.int64))
    generic = int(O.mxm(O.Tuples("INT64", n, n), Lr, Lr, "PLUS", "PAIR", "INT64", mask=Lr).X.sum())
    g = nx.Graph(); g.add_nodes_from(range(n)); g.add_edges_from(zip(rows.tolist(), col.tolist()))
    assert generic == O.fast_tricount(rp, col) == sum(nx.triangles(g).values()) // 3


def test_bfs_fast_path_matches_scipy_and_generic():
    from scipy.sparse.csgraph import shortest_path
    rp, col = rmat.csr_numpy(8, symmetric=True, drop_self_loops=True)
    n = 1 << 8
    src = int(np.argmax(np.diff(rp.astype(np.int64))))
    lev, depth = O.fast_bfs(rp, col, src)
    A = sp.csr_matrix((np.ones(len(col)), col.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    d = shortest_path(A, unweighted=True, indices=src)
    exp = np.where(np.isinf(d), 0, d + 1).astype(np.uint8)
    assert np.array_equal(lev, exp) and depth == exp.max()
    # the reference loop, step by step, through the generic oracle: q<!v,replace> = v lor.land A
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    At = O.Tuples("BOOL", n, n, rows, col, np.ones(len(col), bool))
    v = O.row_vector("UINT8", n); q = O.row_vector("BOOL", n, [src], [True]); level = 1
    while q.nvals and q.X.any():
        # v.assign_scalar(level, mask=q)
        vd = dict(zip(v.J.tolist(), v.X.tolist())); vd.update({j: level for j, x in zip(q.J.tolist(), q.X.tolist()) if x})
        ks = sorted(vd); v = O.row_vector("UINT8", n, ks, [vd[k] for k in ks])
        q = O.vxm(q, v, At, "LOR", "LAND", "BOOL", mask=v, replace=True, mask_comp=True)
        level += 1
    out = np.zeros(n, np.uint8); out[v.J.astype(int)] = v.X
    assert np.array_equal(out, lev)


def test_fast_spmv_matches_generic():
    rng = np.random.default_rng(5)
    rp, col = rmat.csr_numpy(10)
    n = 1 << 10
    val = rng.random(len(col)); x = rng.random(n)
    y, pres = O.fast_spmv(rp, col, val, x)
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    r = O.mxv(O.col_vector("FP64", n), O.Tuples("FP64", n, n, rows, col, val), O.col_vector("FP64", n, np.arange(n), x), "PLUS", "TIMES", "FP64")
    assert np.array_equal(np.flatnonzero(pres), r.I.astype(int)) and np.array_equal(y[pres != 0], r.X)
    A = sp.csr_matrix((val, col.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    assert np.allclose(y, A @ x)
    y32, _ = O.fast_spmv(rp, col, None, x.astype(np.float32), "PLUS_SECOND")
    assert np.allclose(y32, sp.csr_matrix((np.ones(len(col), np.float32), col.astype(np.int64), rp.astype(np.int64)), shape=(n, n)) @ x.astype(np.float32), rtol=1e-5)


@pytest.mark.parametrize("wtype", ["INT64", "FP64"])
def test_sssp_fast_path_matches_generic_and_scipy(wtype):
    """`fast_sssp` = the reference's MIN_PLUS loop (demo/Intro-Prez.ipynb:1034-1045; vector.py:883-885): against the same loop
    through the generic restatement (sweep count included) and against scipy.sparse.csgraph's Bellman-Ford / Dijkstra."""
    from scipy.sparse.csgraph import bellman_ford, dijkstra
    rng = np.random.default_rng(9)
    rp, col = rmat.csr_numpy(9, drop_self_loops=True)
    n = 1 << 9
    nnz = len(col)
    val = rng.integers(1, 256, nnz).astype(np.int64) if wtype == "INT64" else (rng.random(nnz) + 1e-3)
    src = int(np.argmax(np.diff(rp.astype(np.int64))))
    dist, pres, sweeps = O.fast_sssp(rp, col, val, src)
    G = sp.csr_matrix((val.astype(np.float64), col.astype(np.int64), rp.astype(np.int64)), shape=(n, n))
    for algo in (bellman_ford, dijkstra):
        d = algo(G, directed=True, indices=src)
        assert np.array_equal(np.isfinite(d), pres != 0)
        if wtype == "INT64":
            assert np.array_equal(d[pres != 0].astype(np.int64), dist[pres != 0])
        else:
            assert np.allclose(d[pres != 0], dist[pres != 0], rtol=1e-12, atol=0.0)
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    At = O.Tuples(wtype, n, n, rows, col, val)
    v = O.row_vector(wtype, n, [src], [0]); k = 0
    while True:
        w = v
        v = O.vxm(v, v, At, "MIN", "PLUS", wtype, accum="MIN")
        k += 1
        if np.array_equal(w.J, v.J) and np.array_equal(w.X, v.X):
            break
    assert k == sweeps
    assert np.array_equal(np.flatnonzero(pres), v.J.astype(int)) and np.array_equal(dist[pres != 0], v.X)


def test_per_entry_masked_product_fast_path_matches_generic():
    """fast_masked_mxm (the at-scale checker of configs[3]'s C, entry by entry) against the generic restatement: PLUS_PAIR counts
    exactly, PLUS_TIMES on doubles to 1e-12, same pattern (mask entries no product meets are no entries)."""
    rp, col = rmat.csr_numpy(9, symmetric=True, drop_self_loops=True, lower=True)
    n = 1 << 9
    rows = np.repeat(np.arange(n, dtype=np.uint64), np.diff(rp.astype(np.int64)))
    rng = np.random.default_rng(3)
    vals = rng.random(len(col))
    Li = O.Tuples("INT64", n, n, rows, col, np.ones(len(col), np.int64))
    Ci = O.mxm(O.Tuples("INT64", n, n), Li, Li, "PLUS", "PAIR", "INT64", mask=Li)
    out, has = O.fast_masked_mxm(rp, col)
    assert 0 < has.sum() < len(col)
    assert np.array_equal(rows[has != 0], Ci.I) and np.array_equal(col[has != 0], Ci.J) and np.array_equal(out[has != 0].astype(np.int64), Ci.X)
    Lf = O.Tuples("FP64", n, n, rows, col, vals)
    Cf = O.mxm(O.Tuples("FP64", n, n), Lf, Lf, "PLUS", "TIMES", "FP64", mask=Lf)
    out, has2 = O.fast_masked_mxm(rp, col, vals)
    assert np.array_equal(has, has2) and np.array_equal(col[has2 != 0], Cf.J) and np.allclose(out[has2 != 0], Cf.X, rtol=1e-12, atol=0)


def test_batched_bc_fast_path_matches_networkx():
    """fast_bc (the at-scale checker of the gap/bcmark.py algorithm) against networkx's Brandes on the directed R-MAT-9: every vertex
    that is not a source holds the sum over the sources of its dependency."""
    nx = pytest.importorskip("networkx")
    scale, ns = 9, 4
    n = 1 << scale
    rp, col = rmat.csr_numpy(scale, drop_self_loops=True)
    rpt, colt = rmat.csr_numpy(scale, drop_self_loops=True, transpose=True)
    rows = np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))
    deg = np.diff(rp.astype(np.int64))
    sources = [int(x) for x in np.argsort(-deg, kind="stable")[:ns]]
    cent, depth, sizes = O.fast_bc(rp, col, rpt, colt, sources)
    G = nx.DiGraph(); G.add_nodes_from(range(n)); G.add_edges_from(zip(rows.tolist(), col.astype(np.int64).tolist()))
    want = nx.betweenness_centrality_subset(G, sources=sources, targets=list(range(n)), normalized=False)
    others = np.array([v for v in range(n) if v not in sources])
    w = np.array([want[v] for v in others])
    assert depth >= 3 and len(sizes) == depth and all(x > 0 for x in sizes) and w.max() > 1.0
    assert np.allclose(cent[others], w, rtol=1e-9, atol=1e-9), np.abs(cent[others] - w).max()


def test_sampled_rows_of_the_unmasked_product_fast_path_matches_generic_and_scipy():
    """fast_mxm_rows (the checker of sampled rows of A @ A at R-MAT-18 and bench.py's `aa` CPU baseline) against the generic restatement
    and scipy on a small R-MAT: same rows, same pattern (stored zeros included: values of +-1 cancel), values exact on integers."""
    rp, col = rmat.csr_numpy(9, seed=3, symmetric=True, drop_self_loops=True)
    n = len(rp) - 1
    rng = np.random.default_rng(11)
    val = rng.choice([-1.0, 1.0, 2.0], len(col))
    rows = np.unique(rng.integers(0, n, 150)).astype(np.uint32)
    off, oc, ov, prods = O.fast_mxm_rows(rp, col, val, rows)
    I = np.repeat(np.arange(n), np.diff(rp.astype(np.int64)))
    A = O.Tuples("FP64", n, n, I, col, val)
    full = O.mxm(O.Tuples("FP64", n, n), A, A, "PLUS", "TIMES", "FP64").sorted()
    crp = np.zeros(n + 1, np.int64); np.cumsum(np.bincount(full.I.astype(np.int64), minlength=n), out=crp[1:])
    sa = sp.csr_matrix((val, col.astype(np.int64), rp.astype(np.int64)), shape=(n, n)); ss = (sa @ sa).toarray()
    deg = np.diff(rp.astype(np.int64))
    cancelled = 0
    for s, i in enumerate(rows.tolist()):
        b, e = crp[i], crp[i + 1]
        assert np.array_equal(oc[off[s]:off[s + 1]], full.J[b:e].astype(np.uint32)), i
        assert np.array_equal(ov[off[s]:off[s + 1]], full.X[b:e]), i
        assert np.array_equal(ss[i, oc[off[s]:off[s + 1]].astype(np.int64)], ov[off[s]:off[s + 1]])
        assert prods[s] == deg[col[rp[i]:rp[i + 1]].astype(np.int64)].sum()
        cancelled += int((ov[off[s]:off[s + 1]] == 0).sum())
    assert cancelled > 0            # an entry whose products cancel is still an entry
