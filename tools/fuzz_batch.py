#!/usr/bin/env python3
"""Differential fuzzing of the BATCH paths (grb_mxm_rows.cpp, round 6: matrices of <= 64 very long rows as bitmaps) — test infrastructure, run on the GPU box.
  * mxm: random row counts / widths / types / semirings / masks / accumulators / replace / transposed B on batch shapes against the CPU ORACLE
    (tests/test_mxm_gpu.py::run_case), with the batch path forced, and the one-pass product (k_spb_blocks) forced on, forced off and left to the library;
  * eWiseAdd / eWiseMult / apply chains on batch shapes: the bitmap kernels against the generic CSR kernels (GRB_MI355X_BATCH=0), which the small-shape
    fuzzer (tools/fuzz_parity.py) holds against the oracle.
Stops at the first mismatch (the assertion prints the case)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_mxv_vxm_gpu as TV
import test_mxm_gpu as TM
from helpers import TYPE, rand_matrix, to_matrix, matrix_tuples
import pygraphblas_amd as gb
from pygraphblas_amd import descriptor as D

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
TYPES = TV.ALL
MASKS = [None, None, {"typ": "BOOL"}, {"typ": "BOOL", "comp": True}, {"typ": "INT32", "struct": True}, {"typ": "FP64", "comp": True, "struct": True}, {"typ": "FP32", "comp": True},
         {"typ": "UINT8", "dens": 0.0, "comp": True}, {"typ": "INT8", "dens": 0.9}]
ACC = {"BOOL": ["LOR", "LAND", "LXOR"], "INT": ["PLUS", "MIN", "MAX", "TIMES"], "FP": ["PLUS", "MIN", "MAX"]}
EW = {"BOOL": ["LOR", "LAND", "LXOR", "FIRST", "SECOND"], "INT": ["PLUS", "MIN", "MAX", "TIMES", "FIRST", "SECOND", "MINUS"], "FP": ["PLUS", "MIN", "MAX", "TIMES", "DIV", "MINUS", "SECOND"]}
ROWS = [1, 2, 3, 4, 7, 8, 16, 33, 64]
WIDTHS = [65536, 65536, 65600, 98304, 131072]
ENVK = ("GRB_MI355X_BATCH", "GRB_MI355X_SPMM", "GRB_MI355X_EWISE_ROWS", "GRB_MI355X_MXM_ROWS")


def setenv(**kw):
    for k in ENVK: os.environ.pop(k, None)
    os.environ.update(kw)


def same(a, b, what):
    assert np.array_equal(a.I, b.I) and np.array_equal(a.J, b.J), ("pattern", what)
    assert np.allclose(a.X.astype(np.float64), b.X.astype(np.float64), rtol=1e-6, atol=0.0, equal_nan=True), ("values", what)


t0 = time.time(); cnt = {"mxm": 0, "ewise": 0}
while time.time() - t0 < args.seconds:
    typ = TYPES[rng.integers(len(TYPES))]; fam = TV.family(typ)
    mask = MASKS[rng.integers(len(MASKS))]
    accum = None if rng.random() < 0.6 else ACC[fam][rng.integers(len(ACC[fam]))]
    replace = bool(rng.random() < 0.5)
    m = int(ROWS[rng.integers(len(ROWS))])
    if rng.random() < 0.55:
        sr = TV.SEMIRINGS[fam][rng.integers(len(TV.SEMIRINGS[fam]))]
        if "DIV" in sr or "MINUS" in sr or "ISGT" in sr: sr = "LOR_LAND" if fam == "BOOL" else "PLUS_TIMES"
        k, n = int(WIDTHS[rng.integers(len(WIDTHS))]), int(WIDTHS[rng.integers(len(WIDTHS))])
        da = float(rng.choice([2e-5, 1e-3, 0.02])); db = float(rng.choice([2e-5, 6e-5, 1.5e-4]))
        tb = bool(rng.random() < 0.25)
        if mask is not None: mask = dict(mask); mask.setdefault("dens", float(rng.choice([0.01, 0.4])))
        cd = float(rng.choice([0.0, 0.0, 3e-4]))
        state = rng.bit_generator.state
        for env in ({"GRB_MI355X_BATCH": "1"}, {"GRB_MI355X_BATCH": "1", "GRB_MI355X_SPMM": "1"}, {"GRB_MI355X_BATCH": "1", "GRB_MI355X_SPMM": "0"}):
            setenv(**env)
            rng.bit_generator.state = state                           # the same matrices under every setting
            TM.run_case(rng, typ, sr, m, k, n, da, db, mask=mask, accum=accum, replace=replace, tb=tb, c_dens=cd)
            assert "mxm_batch" in gb.last_kernel_plan(), (gb.last_kernel_plan(), env)
            cnt["mxm"] += 1
    else:
        n = int(WIDTHS[rng.integers(len(WIDTHS))])
        opn = EW[fam][rng.integers(len(EW[fam]))]
        union = bool(rng.random() < 0.5)
        At, Bt, Ct = (rand_matrix(rng, typ, m, n, float(rng.choice([0.0, 0.01, 0.3, 0.7]))) for _ in range(3))
        Mt = rand_matrix(rng, mask["typ"], m, n, mask.get("dens", float(rng.choice([0.05, 0.5])))) if mask else None
        flags = ("R" if replace else "") + ("S" if mask and mask.get("struct") else "") + ("C" if mask and mask.get("comp") else "")
        alias = int(rng.integers(0, 4))                               # 0: separate output, 1: out is A, 2: out is B, 3: the mask is A (same type only)
        then_apply = bool(rng.random() < 0.5)

        def one():
            A, B, Cm = to_matrix(At), to_matrix(Bt), to_matrix(Ct)
            M = to_matrix(Mt) if Mt is not None else None
            if alias == 3 and Mt is None: M = A
            out = A if alias == 1 else B if alias == 2 else Cm
            (A.eadd if union else A.emult)(B, getattr(TYPE[typ], opn), out=out, mask=M, accum=getattr(TYPE[typ], accum) if accum else None, desc=getattr(D, flags) if flags else None)
            res = [matrix_tuples(out)]
            if then_apply:                                              # a second batch operation on the result while it lives as a bitmap, then a leave through a CSR consumer
                R = gb.Matrix.sparse(TYPE[typ], m, n)
                out.apply(TYPE[typ].IDENTITY if fam != "BOOL" else TYPE[typ].LNOT, out=R)
                res += [matrix_tuples(R), matrix_tuples(R.transpose())]
            return res, out.nvals
        setenv(GRB_MI355X_BATCH="0", GRB_MI355X_EWISE_ROWS="0"); (ra, na) = one()
        setenv(GRB_MI355X_BATCH="1"); (rb, nb) = one()
        what = f"{typ}.{opn} union={union} mask={mask} accum={accum} flags={flags} alias={alias} m={m} n={n}"
        assert na == nb, (na, nb, what)
        for x, y in zip(ra, rb): same(x, y, what)
        cnt["ewise"] += 1
setenv()
print(f"batch fuzz ok: {cnt['mxm']} mxm cases against the oracle, {cnt['ewise']} element-wise chains against the generic kernels in {time.time() - t0:.0f} s, seed {args.seed}")
