"""Non-blocking execution (pygraphblas_amd/csrc/grb_lazy.cpp): the reference initialises the library GrB_NONBLOCKING
(pygraphblas/__init__.py:251-256), so vector operations may be deferred and fused — results must stay as-if sequential
(SURVEY.md App. A item 8).  Checked here:

  * the PageRank iteration of gap/prmark.py:17-29 runs in <= 5 kernels' worth of deferred work (one element-wise chain for
    `w = t / d`, the fill `r[:] = teleport` folded into the product's store, one chain for `t -= r; abs; reduce_float`) and gives
    the values of the blocking run;
  * random programs of element-wise operations, fills, products, element writes, frees and reductions over a pool of
    aliasing vectors against a numpy model of the GraphBLAS rules — chains of every length, flushed by every kind of access;
  * the traps: an operand overwritten or freed while queued work still reads it, an output overwritten before it was ever
    computed, a fill consumed by a product that cannot fold it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(gb):
    a = [C.c_uint64(0) for _ in range(4)]
    assert gb.lib.GrBX_lazy_stats(*[C.byref(x) for x in a]) == 0
    return dict(zip(("chains", "nodes", "fills_folded", "reduces_fused"), [x.value for x in a]))


# ---- numpy model of a bitmap vector -------------------------------------------------------------------------------------------
class M:
    def __init__(self, val, pres):
        self.val, self.pres = val.copy(), pres.copy()

    def copy(self):
        return M(self.val, self.pres)


BIN = {"PLUS": lambda a, b: a + b, "MINUS": lambda a, b: a - b, "TIMES": lambda a, b: a * b, "MIN": np.minimum, "MAX": np.maximum,
       "FIRST": lambda a, b: a + 0 * b, "SECOND": lambda a, b: 0 * a + b}
UN = {"ABS": np.abs, "AINV": lambda a: -a, "IDENTITY": lambda a: a}


def m_ewise(u, v, op, union):
    both = u.pres & v.pres
    val = np.where(both, BIN[op](u.val, v.val), np.where(u.pres, u.val, v.val))
    pres = (u.pres | v.pres) if union else both
    return M(np.where(pres, val, 0), pres)


def to_model(v, dt):
    x, p = v.to_dense_arrays()
    return M(np.where(p != 0, x, 0).astype(dt), p != 0)


def same(v, m):
    x, p = v.to_dense_arrays()
    return np.array_equal(p != 0, m.pres) and np.array_equal(x[m.pres], m.val[m.pres])


@pytest.mark.parametrize("tname", ["FP64", "FP32", "INT32"])
def test_random_programs_against_a_model(gb, gpu, tname):
    typ = getattr(gb, tname)
    dt = typ._np
    rng = np.random.default_rng(1234)
    n = 3000
    # a small matrix (row-block kernel: cannot fold a fill) for products in the middle of the programs
    nz = 9000
    flat = np.sort(rng.choice(n * n, nz, replace=False)).astype(np.uint64)
    I, J = np.divmod(flat, np.uint64(n))
    AX = rng.integers(1, 4, nz).astype(dt)
    A = gb.Matrix.from_arrays(I, J, AX, n, n, typ)
    import scipy.sparse as sp
    As = sp.csr_matrix((AX.astype(np.float64), (I.astype(np.int64), J.astype(np.int64))), shape=(n, n))
    Ap = sp.csr_matrix((np.ones(nz), (I.astype(np.int64), J.astype(np.int64))), shape=(n, n))

    def rand_vec(density):
        k = int(n * density)
        idx = np.sort(rng.choice(n, k, replace=False)).astype(np.uint64)
        x = rng.integers(-6, 7, k).astype(dt)
        return gb.Vector.from_arrays(idx, x, n, typ)

    before = stats(gb)
    for prog in range(60):
        pool = [rand_vec(d) for d in (1.0, 0.6, 0.3, 1.0, 0.05)]
        model = [to_model(v, dt) for v in pool]
        for step in range(int(rng.integers(1, 12))):
            kind = rng.choice(["eadd", "emult", "apply", "bind", "fill", "mxv", "set", "reduce", "dup", "free", "nvals", "masked"],
                              p=[0.2, 0.15, 0.15, 0.08, 0.08, 0.06, 0.05, 0.08, 0.05, 0.04, 0.03, 0.03])
            a, b, c = (int(x) for x in rng.integers(0, len(pool), 3))
            if kind in ("eadd", "emult"):
                op = str(rng.choice(list(BIN)))
                (pool[a].eadd if kind == "eadd" else pool[a].emult)(pool[b], getattr(typ, op), out=pool[c])
                model[c] = m_ewise(model[a], model[b], op, kind == "eadd")
            elif kind == "apply":
                op = str(rng.choice(list(UN)))
                pool[a].apply(getattr(typ, op), out=pool[c])
                model[c] = M(np.where(model[a].pres, UN[op](model[a].val), 0).astype(dt), model[a].pres)
            elif kind == "bind":
                s = int(rng.integers(-3, 4))
                pool[a].apply_second(typ.TIMES, s, out=pool[c])
                model[c] = M(np.where(model[a].pres, model[a].val * dt(s), 0).astype(dt), model[a].pres)
            elif kind == "fill":
                s = int(rng.integers(-3, 4))
                pool[c][:] = s
                model[c] = M(np.full(n, s, dt), np.ones(n, bool))
            elif kind == "mxv":
                # w += A (+).(x) u with the monoid's operator: exercises the pending-fill and the in-place epilogue decisions
                A.mxv(pool[a], out=pool[c], accum=typ.PLUS, semiring=typ.PLUS_TIMES)
                ua = model[a]
                y = As @ np.where(ua.pres, ua.val, 0).astype(np.float64)
                has = (Ap @ ua.pres.astype(np.float64)) > 0
                old = model[c].copy() if c != a else ua.copy()
                val = np.where(has & old.pres, old.val + y.astype(dt), np.where(has, y.astype(dt), old.val))
                model[c] = M(np.where(has | old.pres, val, 0).astype(dt), has | old.pres)
            elif kind == "set":
                i = int(rng.integers(0, n)); s = int(rng.integers(-3, 4))
                pool[c][i] = s
                model[c].val[i] = s; model[c].pres[i] = True
            elif kind == "reduce":
                got = pool[a].reduce_int() if tname == "INT32" else pool[a].reduce_float()
                assert got == model[a].val[model[a].pres].astype(np.float64).sum(), (prog, step)
            elif kind == "dup":
                pool[c] = pool[a].dup()
                model[c] = model[a].copy()
            elif kind == "free":
                pool[c] = rand_vec(0.5)                       # the old object dies while queued work may still involve it
                model[c] = to_model(pool[c], dt)
            elif kind == "nvals":
                assert pool[a].nvals == int(model[a].pres.sum()), (prog, step)
            elif kind == "masked":
                s = int(rng.integers(-3, 4))
                pool[c].assign_scalar(s, mask=pool[b])        # not deferrable: must see every queued result it depends on
                allow = model[b].pres & (model[b].val != 0)
                model[c] = M(np.where(allow, dt(s), model[c].val), model[c].pres | allow)
            if kind in ("eadd", "emult", "apply", "bind", "mxv") and np.abs(model[c].val.astype(np.float64)).max() > 1e5:
                pool[c][:] = 1                                # keep every intermediate exactly representable in FP32 / INT32
                model[c] = M(np.ones(n, dt), np.ones(n, bool))
        for k in range(len(pool)):
            assert same(pool[k], model[k]), (prog, k)
    after = stats(gb)
    assert after["chains"] > before["chains"] and after["nodes"] - before["nodes"] > after["chains"] - before["chains"]      # chains of several operations did run


def test_pagerank_iteration_is_fused_and_matches_the_blocking_run(gb, gpu):
    """gap/prmark.py:17-29 on R-MAT-22 (kernel X): per iteration two chain kernels (w = t / d ; t -= r, abs, reduce), the fill
    folded into the product's store — and the ranks equal those of a GRB_MI355X_BLOCKING=1 process to 1e-6 (same iterations)."""
    code = r"""
import sys, json, ctypes as C
sys.path.insert(0, %r)
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops
S = 22; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
ones = torch.ones(nnz, dtype=torch.float32, device=dev)
A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32); pres = (deg > 0).to(torch.uint8)
d = gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)
a = [C.c_uint64(0) for _ in range(4)]
r, its, rdiff = loops.pagerank(A, d)
gb.lib.GrBX_lazy_stats(*[C.byref(x) for x in a])
x, p = r.to_dense_arrays()
np.save(sys.argv[1], x)
print(json.dumps({"its": its, "rdiff": rdiff, "full": bool(p.all()), "stats": [v.value for v in a], "plan": gb.last_kernel_plan()}))
""" % ROOT
    import json, tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("0", "1"):
            f = os.path.join(td, f"r{mode}.npy")
            r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=400, env=dict(os.environ, GRB_MI355X_BLOCKING=mode))
            assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
            out[mode] = (json.loads(r.stdout.strip().splitlines()[-1]), np.load(f))
    lazy, blocking = out["0"], out["1"]
    assert lazy[0]["full"] and blocking[0]["full"]
    assert lazy[0]["its"] == blocking[0]["its"]
    assert np.allclose(lazy[1], blocking[1], rtol=1e-6, atol=0.0)
    chains, nodes, fills, reduces = lazy[0]["stats"]
    its = lazy[0]["its"]
    assert blocking[0]["stats"] == [0, 0, 0, 0]
    assert fills == its                      # every `r[:] = teleport` folded into the product's store (FP32: kernel X from the first product on)
    assert reduces == its                    # every `t.reduce_float()` produced by the chain kernel of `t -= r; abs`
    assert chains == 2 * its and nodes == 3 * its


def test_traps(gb, gpu):
    n = 5000
    F = gb.FP64
    idx = np.arange(n, dtype=np.uint64)
    a = gb.Vector.from_arrays(idx, np.arange(n, dtype=np.float64), n, F)
    b = gb.Vector.from_arrays(idx[::2], np.ones(n // 2), n, F)
    # an operand is overwritten (by a fill, by an element write, by a product) while queued work still reads it
    c = a + b
    a[:] = 7.0
    x, p = c.to_dense_arrays()
    assert p.all() and np.array_equal(x, np.arange(n) + (np.arange(n) % 2 == 0))
    c = a * b                                 # a is a pending fill: written when the queue first needs it
    a[3] = -1.0
    x, p = c.to_dense_arrays()
    assert np.array_equal(np.flatnonzero(p), np.arange(0, n, 2)) and np.all(x[p != 0] == 7.0)
    assert a[3] == -1.0 and a[4] == 7.0 and a.nvals == n
    # an output overwritten before it was ever computed; an output freed before it was computed
    c = a + b
    a.emult(b, F.TIMES, out=c)
    d = a + c
    del d
    x, p = c.to_dense_arrays()
    assert np.array_equal(np.flatnonzero(p), np.arange(0, n, 2))
    # a long chain through one vector, with its own old value as an operand of every step
    t = gb.Vector.from_arrays(idx, np.ones(n), n, F)
    for k in range(11):
        t.eadd(b, F.PLUS, out=t)
    x, p = t.to_dense_arrays()
    assert p.all() and np.array_equal(x, 1.0 + 11.0 * (np.arange(n) % 2 == 0))
    # a fill read by a reduction, by dup, by a mask
    f = gb.Vector.sparse(F, n)
    f[:] = 0.5
    assert f.reduce_float() == 0.5 * n
    f[:] = 2.0
    g = f.dup()
    f[:] = 3.0
    assert g.reduce_float() == 2.0 * n and f.reduce_float() == 3.0 * n
    m = gb.Vector.sparse(gb.BOOL, n)
    m[:] = True
    h = gb.Vector.sparse(F, n)
    h.assign_scalar(4.0, mask=m)
    assert h.nvals == n and h.reduce_float() == 4.0 * n
    # mixed types in a row: each chain ends where the type changes
    i32 = gb.Vector.from_arrays(idx, np.arange(n, dtype=np.int32), n, gb.INT32)
    j32 = i32 + i32
    e = a + a
    k32 = j32 * i32
    assert k32.reduce_int() == int((2 * np.arange(n, dtype=np.int64) ** 2).sum())
    assert e.reduce_float() == float(2 * (7.0 * (n - 1) - 1.0))


def test_a_reduced_chain_is_not_stored_until_somebody_looks(gb, gpu):
    """`t -= r; t = abs(t); t.reduce_float()` (gap/prmark.py:24-26): the reduction runs the queue without storing t and keeps it queued
    (grb_lazy.cpp run_queue `keep`); a later whole-vector assignment of t drops the steps unrun, any look at t runs them — and an
    operand overwritten in between is still read first."""
    n = 70000
    F = gb.FP32
    idx = np.arange(n, dtype=np.uint64)
    tv = (np.arange(n) % 17).astype(np.float32); rv = (np.arange(n) % 5).astype(np.float32)
    want = np.abs(tv - rv)

    def fresh():
        return gb.Vector.from_arrays(idx, tv, n, F), gb.Vector.from_arrays(idx[::3], rv[::3], n, F)
    want_u = np.where(np.arange(n) % 3 == 0, want, tv)                # eWiseAdd: where r has no entry t keeps its value (abs of a non-negative)
    # 1. reduce, reduce again, then look
    t, r = fresh()
    t -= r; t.apply(F.ABS, out=t)
    s0 = stats(gb)
    assert t.reduce_float() == float(want_u.astype(np.float64).sum())
    s1 = stats(gb)
    assert s1["chains"] == s0["chains"] + 1 and s1["reduces_fused"] == s0["reduces_fused"] + 1
    assert t.reduce_float() == float(want_u.astype(np.float64).sum())
    x, p = t.to_dense_arrays()
    assert p.all() and np.array_equal(x, want_u)
    # 2. reduce, then the vector is assigned as a whole: the two steps never run again, and never stored
    t, r = fresh()
    t -= r; t.apply(F.ABS, out=t)
    assert t.reduce_float() == float(want_u.astype(np.float64).sum())
    s2 = stats(gb)
    t[:] = 4.0
    w = r * r                                                        # (a new step behind the dropped ones: only it runs)
    x, p = w.to_dense_arrays()
    s3 = stats(gb)
    assert s3["chains"] == s2["chains"] + 1 and s3["nodes"] == s2["nodes"] + 1
    assert np.array_equal(np.flatnonzero(p), np.arange(0, n, 3)) and np.array_equal(x[::3], rv[::3] ** 2)
    assert t.reduce_float() == 4.0 * n and r.nvals == len(idx[::3])
    # 3. reduce, then an operand of the kept steps is overwritten: they run (and store) first
    t, r = fresh()
    t -= r; t.apply(F.ABS, out=t)
    assert t.reduce_float() == float(want_u.astype(np.float64).sum())
    r[:] = 100.0
    x, p = t.to_dense_arrays()
    assert p.all() and np.array_equal(x, want_u)
    assert r.reduce_float() == 100.0 * n
    # 4. reduce, then a step that reads the kept result through the queue, then the kept vector dies
    t, r = fresh()
    t -= r; t.apply(F.ABS, out=t)
    assert t.reduce_float() == float(want_u.astype(np.float64).sum())
    u = t * t
    del t
    x, p = u.to_dense_arrays()
    assert p.all() and np.array_equal(x, want_u * want_u)
    # 5. a MIN reduction (its identity needs the presence bytes when nothing is present: stored at once, as before)
    t, r = fresh()
    t -= r; t.apply(F.ABS, out=t)
    assert t.reduce_float(F.MIN_MONOID) == float(want_u.min())
    x, p = t.to_dense_arrays()
    assert np.array_equal(x, want_u)


def jit_stats(gb):
    a = [C.c_uint64(0), C.c_uint64(0)]
    assert gb.lib.GrBX_chain_jit_stats(C.byref(a[0]), C.byref(a[1])) == 0
    return a[0].value, a[1].value


@pytest.mark.parametrize("tname", ["FP32", "FP64", "INT32", "INT64"])      # (the model draws negative scalars: the unsigned types are covered by the test below)
def test_random_programs_through_the_compiled_chains(gb, gpu, tname, monkeypatch, tmp_path):
    """The random programs of test_random_programs_against_a_model with every chain compiled by hipRTC at first sight
    (GRB_MI355X_CHAIN_JIT=2, grb_chain_jit.cpp) instead of run by the interpreter kernel: same numpy model, exact values.  Round 6: the 4- and 8-byte
    integer types too (wrap-around and SuiteSparse's integer division as helper functions of the generated text)."""
    monkeypatch.setenv("GRB_MI355X_CHAIN_JIT", "2")
    monkeypatch.setenv("GRB_MI355X_CACHE_DIR", str(tmp_path))          # (every kernel really compiled here: an empty cache)
    c0, l0 = jit_stats(gb)
    test_random_programs_against_a_model(gb, gpu, tname)
    c1, l1 = jit_stats(gb)
    assert c1 > c0 and l1 - l0 >= 20, (c0, c1, l0, l1)


def test_integer_division_and_wrap_around_rules_survive_the_compiler(gb, gpu, monkeypatch, tmp_path):
    """The rules of grb_ops.hpp that are not plain C, through compiled integer chains against the interpreter (GRB_MI355X_CHAIN_JIT=0) and numpy: x / 0 saturates
    by sign, 0 / 0 = 0, INT_MIN / -1 wraps instead of trapping, sums and products wrap modulo 2^bits, ABS(INT_MIN) = INT_MIN, AINV and MINV."""
    monkeypatch.setenv("GRB_MI355X_CACHE_DIR", str(tmp_path))
    n = 4096
    for tname, dt in (("INT32", np.int32), ("INT64", np.int64), ("UINT32", np.uint32), ("UINT64", np.uint64)):
        T = getattr(gb, tname); info = np.iinfo(dt)
        rng = np.random.default_rng(11)
        special = np.array([0, 1, info.max, info.min, info.max - 1, 2, 3] + ([-1, -2, info.min + 1] if info.min < 0 else [info.max // 2]), dtype=dt)
        xs = rng.choice(special, n).astype(dt); ys = rng.choice(special, n).astype(dt)
        res = {}
        for mode in ("0", "2"):
            monkeypatch.setenv("GRB_MI355X_CHAIN_JIT", mode)
            x = gb.Vector.from_dense_array(xs.copy(), T); y = gb.Vector.from_dense_array(ys.copy(), T)
            out = []
            for opn in ("DIV", "PLUS", "TIMES", "MINUS", "MIN", "MAX", "RDIV"):
                t = x.emult(y, getattr(T, opn)); t = t.eadd(x, T.PLUS); out.append(t.to_dense_arrays()[0].copy())      # two steps: a chain, not a single kernel
            for opn in ("ABS", "AINV", "MINV"):
                t = x.apply(getattr(T, opn)); t = t.emult(y, T.FIRST); out.append(t.to_dense_arrays()[0].copy())
            t = x.emult(y, T.TIMES); out.append(np.array([t.reduce_int()]))
            res[mode] = out
        for k, (a0, a2) in enumerate(zip(res["0"], res["2"])):
            assert np.array_equal(a0, a2), (tname, k)
        with np.errstate(over="ignore"):
            assert np.array_equal(res["2"][1], (xs + ys) + xs) and np.array_equal(res["2"][2], (xs * ys) + xs)                 # numpy integers wrap the same way
        d0 = res["2"][0] - xs                                                                                                    # the DIV step alone (the PLUS x undone, modulo 2^bits)
        zero = ys == 0
        assert np.array_equal(d0[zero & (xs == 0)], np.zeros((zero & (xs == 0)).sum(), dt))
        assert (d0[zero & (xs > 0)] == info.max).all()
        if info.min < 0:
            assert (d0[zero & (xs < 0)] == info.min).all() and (d0[(ys == -1) & (xs == info.min)] == info.min).all()


def test_pagerank_loop_gives_the_same_bits_with_and_without_the_chain_compiler(gb, gpu, tmp_path):
    """gap/prmark.py:17-29 on R-MAT-20 in three processes: GRB_MI355X_CHAIN_JIT=0 (interpreter and ahead-of-time shapes only), the default with an empty code-object
    cache (its two chains are compiled by hipRTC at their second appearance), and the default again on the cache the second left behind: the rank vectors are the
    same bits, the third process compiles NOTHING (its kernels come from GRB_MI355X_CACHE_DIR) and still launches through them."""
    code = r"""
import sys, json, ctypes as C
sys.path.insert(0, %r)
import numpy as np, torch
import pygraphblas_amd as gb
from pygraphblas_amd import rmat, loops
S = 20; n = 1 << S; dev = torch.device("cuda", 0)
rowptr, col = rmat.csr_torch(S, dev, seed=42); nnz = int(col.numel())
ones = torch.ones(nnz, dtype=torch.float32, device=dev)
A = gb.Matrix.from_csr(gb.FP32, n, n, rowptr.data_ptr(), col.data_ptr(), (ones.data_ptr(), nnz), device=True)
deg = (rowptr[1:] - rowptr[:-1]).to(torch.float32); pres = (deg > 0).to(torch.uint8)
d = gb.Vector.from_dense_array((deg.data_ptr(), n), gb.FP32, present=pres.data_ptr(), device=True)
r, its, rdiff = loops.pagerank(A, d, fixed_iterations=12)
a = [C.c_uint64(0) for _ in range(3)]
gb.lib.GrBX_chain_jit_stats2(*[C.byref(x) for x in a])
np.save(sys.argv[1], r.to_dense_arrays()[0])
print(json.dumps({"rdiff": rdiff, "jit": [v.value for v in a]}))
""" % ROOT
    import json
    cache = tmp_path / "cache"; cache.mkdir()
    runs = []
    for k, jit in enumerate(("0", "1", "1")):
        f = str(tmp_path / f"r{k}.npy")
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=400, env=dict(os.environ, GRB_MI355X_CHAIN_JIT=jit, GRB_MI355X_CACHE_DIR=str(cache)))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        runs.append((json.loads(r.stdout.strip().splitlines()[-1]), np.load(f)))
    (j0, r0), (j1, r1), (j2, r2) = runs
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32)) and np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    assert j0["rdiff"] == j1["rdiff"] == j2["rdiff"]
    assert j0["jit"] == [0, 0, 0]
    assert j1["jit"][0] == 2 and j1["jit"][2] == 0 and j1["jit"][1] >= 20, j1            # two chains compiled, none found on disk, the loop launched through them
    assert j2["jit"][0] == 0 and j2["jit"][2] == 2 and j2["jit"][1] >= 20, j2            # the second start compiles nothing
    assert len(list(cache.glob("chain-*.co"))) == 2


def test_a_chain_outside_the_pagerank_loop_runs_as_fast_as_the_compiled_shapes(gb, gpu, monkeypatch, capsys, tmp_path):
    """`reduce(+, abs(x * y - z))` over 2^25 FP32 positions — a chain gap/prmark.py does not contain: four streams of 4 bytes.  The second time the library sees
    it, hipRTC compiles its steps (default mode: GRB_MI355X_CHAIN_JIT=1), and from then on it moves its bytes within 1.3 x of the rate of the ahead-of-time
    shape `reduce(+, abs(x - y))` (k_vec_chain<..., SPEC = 1>: two streams), where the interpreter kernel is bound by instruction issue.  Values against numpy."""
    monkeypatch.delenv("GRB_MI355X_CHAIN_JIT", raising=False)
    monkeypatch.setenv("GRB_MI355X_CACHE_DIR", str(tmp_path))          # (an empty cache: on a box whose ~/.cache/grb_mi355x holds this chain from an earlier run it would be loaded, not compiled)
    n = 1 << 25
    rng = np.random.default_rng(4)
    xs, ys, zs = (rng.random(n, dtype=np.float32) for _ in range(3))
    x, y, z = (gb.Vector.from_dense_array(a, gb.FP32) for a in (xs, ys, zs))

    def generic():
        t = x.emult(y, gb.FP32.TIMES); t = t.eadd(z, gb.FP32.MINUS); t = t.apply(gb.FP32.ABS)
        return t.reduce_float()

    def spec():
        t = x.eadd(y, gb.FP32.MINUS); t = t.apply(gb.FP32.ABS)
        return t.reduce_float()

    def timed(fn, reps=20):
        fn(); fn()
        best = 1e9
        for _ in range(3):
            gb.lib.GrBX_timer_start()
            for _ in range(reps):
                r = fn()
            ms = C.c_float(0); gb.lib.GrBX_timer_stop(C.byref(ms)); best = min(best, ms.value / reps)
        return r, best
    c0, _ = jit_stats(gb)
    got, ms_gen = timed(generic)
    c1, l1 = jit_stats(gb)
    assert c1 == c0 + 1 and l1 > 0                                       # compiled once, at its second appearance
    want = float(np.abs(xs.astype(np.float64) * ys - zs).sum())
    assert abs(got - want) <= 1e-5 * want, (got, want)
    monkeypatch.setenv("GRB_MI355X_CHAIN_JIT", "3")                    # (the ahead-of-time kernel itself: by default its shape, too, is compiled at its second appearance)
    _, ms_spec = timed(spec)
    monkeypatch.setenv("GRB_MI355X_CHAIN_JIT", "0")
    got_i, ms_int = timed(generic)
    assert abs(got_i - want) <= 1e-5 * want
    rate_gen, rate_spec, rate_int = 3 * 4 * n / ms_gen, 2 * 4 * n / ms_spec, 3 * 4 * n / ms_int           # bytes per ms
    with capsys.disabled():
        print(f"\n[chain abs(x*y - z) reduced, 2^25 FP32: compiled {ms_gen * 1e3:.1f} us = {rate_gen / 1e9:.2f} TB/s; interpreter {ms_int * 1e3:.1f} us = {rate_int / 1e9:.2f} TB/s; "
              f"ahead-of-time shape abs(x - y) reduced {ms_spec * 1e3:.1f} us = {rate_spec / 1e9:.2f} TB/s]")
    assert rate_gen * 1.3 >= rate_spec, (rate_gen, rate_spec)
    assert ms_gen < ms_int
