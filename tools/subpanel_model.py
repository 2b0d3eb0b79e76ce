"""CPU model of kernel X's plan with column sub-panels (round 4): how many sub-rows, how many cold gathers and how many bytes of `u`
a sub-panel's cold gathers touch, for S sub-panels per XCD.  Pure numpy; no GPU.  python tools/subpanel_model.py SCALE TSIZE [S ...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pygraphblas_amd import rmat

def build(scale):
    src, dst = rmat.edges_numpy(scale)
    key = np.unique((src << np.uint64(32)) | dst)          # dedup, row-major
    return (key >> np.uint64(32)).astype(np.uint32), (key & np.uint64(0xFFFFFFFF)).astype(np.uint32)

def model(row, col, n, tsize, S, own_tables=False):
    H = {8: 19454, 4: 39932}[tsize]
    line = 128 // tsize
    nlines = (n + line - 1) // line
    cnt = np.bincount(col, minlength=n).astype(np.int64)
    lw = np.add.reduceat(np.concatenate([cnt, np.zeros(nlines * line - n, np.int64)]), np.arange(0, nlines * line, line))
    order = np.argsort(-lw, kind="stable")
    NP = 8 * S
    m = np.arange(nlines) % (2 * NP)
    vp_of_rank = np.where(m < NP, m, 2 * NP - 1 - m)
    vp_line = np.empty(nlines, np.int32); vp_line[order] = vp_of_rank
    # virtual panel vp = s * 8 + k
    vp_col = vp_line[np.arange(n) // line]
    k_col = vp_col & 7
    grp = vp_col if own_tables else k_col
    # hot: top-H columns per group by count
    hot = np.zeros(n, bool)
    o = np.lexsort((-cnt, grp))
    g_sorted = grp[o]
    starts = np.searchsorted(g_sorted, np.arange(grp.max() + 1))
    rank = np.arange(n) - starts[g_sorted]
    hot[o] = (rank < H) & (cnt[o] > 0)
    e_hot = hot[col]; e_vp = vp_col[col].astype(np.int64); e_k = e_vp & 7
    nnz = len(col)
    ncold = int((~e_hot).sum())
    if own_tables:
        sub = len(np.unique(row.astype(np.int64) * NP + e_vp))
    else:
        pairs_all = np.unique(row.astype(np.int64) * 8 + e_k)
        cold_keys = np.unique(row[~e_hot].astype(np.int64) * NP + e_vp[~e_hot])        # distinct (row, vp) among cold entries
        pairs_cold = np.unique((cold_keys // NP) * 8 + ((cold_keys % NP) & 7))         # (row, k) having a cold entry
        sub = len(cold_keys) + (len(pairs_all) - len(pairs_cold))
    # footprint: lines with >= 1 cold entry per vp
    cold_lines = np.unique(col[~e_hot] // line)
    fp = np.bincount(vp_line[cold_lines], minlength=NP) * 128
    return dict(S=S, own=own_tables, nnz=nnz, cold_frac=ncold / nnz, subrows=sub, sub_per_entry=sub / nnz, footprint_MB_max=fp.max() / 2**20, footprint_MB_mean=fp.mean() / 2**20)

if __name__ == "__main__":
    scale, tsize = int(sys.argv[1]), int(sys.argv[2])
    Ss = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
    t = time.time(); row, col = build(scale); print("built", len(col), "entries in", round(time.time() - t, 1), "s", flush=True)
    for S in Ss:
        for own in (False, True):
            if S == 1 and own: continue
            t = time.time(); print(model(row, col, 1 << scale, tsize, S, own), round(time.time() - t, 1), "s", flush=True)
