#!/usr/bin/env python3
"""Differential fuzzing of the hot path against the CPU oracle (test infrastructure; run on the GPU box):
random shapes / densities / types / semirings / masks / accumulators / descriptors for GrB_mxv, GrB_vxm and GrB_mxm in the
library's automatic kernel selection, plus the row-wise and entry-parallel paths forced on small shapes.  Stops at the first
mismatch (the failing case is printed by the assertion)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_mxv_vxm_gpu as TV
import test_mxm_gpu as TM

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
TYPES = TV.ALL
MASKS = [None, {"typ": "BOOL"}, {"typ": "BOOL", "comp": True}, {"typ": "INT32", "struct": True}, {"typ": "FP64", "comp": True, "struct": True}, {"typ": "UINT8", "dens": 0.0, "comp": True},
         {"typ": "INT8", "dens": 1.0}]
ACC = {"BOOL": ["LOR", "LAND", "LXOR"], "INT": ["PLUS", "MIN", "MAX", "TIMES", "SECOND"], "FP": ["PLUS", "MIN", "MAX", "SECOND"]}
t0 = time.time(); n = {"mxv": 0, "mxm": 0}
while time.time() - t0 < args.seconds:
    typ = TYPES[rng.integers(len(TYPES))]; fam = TV.family(typ)
    sr = TV.SEMIRINGS[fam][rng.integers(len(TV.SEMIRINGS[fam]))]
    mask = MASKS[rng.integers(len(MASKS))]
    accum = None if rng.random() < 0.5 else ACC[fam][rng.integers(len(ACC[fam]))]
    replace = bool(rng.random() < 0.4)
    if rng.random() < 0.6:
        nr, nc = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        TV.run_case(rng, typ, sr, nr, nc, float(rng.choice([0.01, 0.05, 0.2, 0.6])), float(rng.choice([0.0, 0.02, 0.3, 1.0])), vxm=bool(rng.random() < 0.5),
                    tran=bool(rng.random() < 0.3), mask=mask, accum=accum, replace=replace)
        n["mxv"] += 1
    else:
        m, k, nn = int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.integers(1, 70))
        for env in ({}, {"GRB_MI355X_MXM_ROWS": "1"}, {"GRB_MI355X_SPGEMM": "esc"}, {"GRB_MI355X_EWISE_ROWS": "1"}):
            for kk in ("GRB_MI355X_MXM_ROWS", "GRB_MI355X_SPGEMM", "GRB_MI355X_EWISE_ROWS"): os.environ.pop(kk, None)
            os.environ.update(env)
            sr2 = sr if ("DIV" not in sr and "MINUS" not in sr and "ISGT" not in sr) else ("LOR_LAND" if fam == "BOOL" else "PLUS_TIMES")
            TM.run_case(rng, typ, sr2, m, k, nn, float(rng.choice([0.05, 0.2, 0.5])), float(rng.choice([0.05, 0.2, 0.5])), mask=mask, accum=accum if accum != "SECOND" else None,
                        replace=replace, ta=bool(rng.random() < 0.3), tb=bool(rng.random() < 0.3), c_dens=float(rng.choice([0.0, 0.3])))
            n["mxm"] += 1
        for kk in ("GRB_MI355X_MXM_ROWS", "GRB_MI355X_SPGEMM", "GRB_MI355X_EWISE_ROWS"): os.environ.pop(kk, None)
print(f"fuzz ok: {n['mxv']} mxv/vxm cases, {n['mxm']} mxm cases in {time.time() - t0:.0f} s, seed {args.seed}")
