"""Rows or a 2 x 4 grid at P = 8?  (settle it by measurement.)

What differs between the two layouts is the SHAPE of the launches a rank issues per panel and what sits on the cross-rank cycle;
the arithmetic, the tile kernel and the total flops per rank are the same.  This script issues, for BASELINE config 4
(N = 32768, nb = 1024, fp64, M = 1024 right-hand-side rows), every MFMA launch of one rank of either layout -- through the
C ABI, on one stream, HIP events around every launch -- and adds them up:

  rows  (the product: row blocks dealt by g3h_deal)   per panel k:  the rank's rows below block k as ONE K-triangular product
        against V_k (its panel solve), the column updates k+1 and k+2 (rows x nb x nb) and the staircase update of everything
        right of them against the gathered panel (g3_gemm_nt_stair);
  grid  (2 x 4 block-cyclic)               per panel k:  the ranks of process column k mod 4 solve n_k / 2 tiles
        (4x what a row rank solves, on the step's critical path: nothing can be updated before the panel has been solved AND
        broadcast along rows and columns), every rank updates its tiles (I, J), I >= J > k, from the row panel and the column
        panel it received: one staircase launch over its 16 x 8 local tiles.

Printed per layout: the busiest rank's bulk / column / solve time (compute side, the other ranks infinitely fast: what the
replay transport measures for rows with all three streams running), and the part of it that sits on the CROSS-RANK cycle of a
step -- rows: the head solve (nb rows) of the next block's owner; grid: the whole panel solve of a process column -- summed over
the sweep.  usage: python scripts/r5_grid_vs_rows.py  -> gpurun_out/r5_grid_vs_rows.txt
"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.distributed import deal_blocks

N, nb, P, RHS = 32768, 1024, 8, 1152
nblk = N // nb
dt, code, es = np.float64, _lib.G3_F64, 8
dev = g3.Device(0)
st = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(st)
dev.set_stream(st.cuda_stream)
lib = dev.lib
rng = torch.Generator(device='cuda'); rng.manual_seed(1)


def rnd(rows, cols):
    return torch.randn((rows, cols), dtype=torch.float64, device='cuda', generator=rng) * 0.01


def timed(fn, reps=2):
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def stair(Cm, A, B, seg_rows, seg_cols, diag):
    if not seg_rows or max(seg_cols) <= 0:
        return 0.0
    return timed(lambda: dev.gemm_nt_stair(Cm.data_ptr(), Cm.stride(0), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), nb,
                                           seg_rows, seg_cols, dt, alpha=-1.0, beta=1.0, seg_diag=diag))


def solve(V, X, Xo, m):
    if m <= 0:
        return 0.0
    def run():
        rc = lib.g3_trsm_full(dev.ctx, V.data_ptr(), nb, nb, X.data_ptr(), m, X.stride(0), Xo.data_ptr(), nb, code)
        assert rc == 0, rc
    return timed(run)


V = torch.tril(rnd(nb, nb))
out = []
# ------------------------------------------------------------------ rows
owner = deal_blocks(P, nblk)
rhs_of = [RHS // 128 // P * 128 + (128 if r < (RHS // 128) % P else 0) for r in range(P)]
res_rows = []
for r in range(P):
    mine = [I for I in range(nblk) if owner[I] == r]
    rows_mat = len(mine) * nb
    A = rnd(rows_mat + rhs_of[r], N)                    # the rank's local rows, full width
    G = rnd(N, nb)                                      # a gathered panel (block order: no table needed for timing)
    Xo = rnd(rows_mat + rhs_of[r], nb)
    bulk = col = sol = head = 0.0
    for k in range(nblk - 1):
        lo = sum(1 for I in mine if I <= k) * nb
        m = rows_mat - lo + rhs_of[r]
        sol += solve(V, A[lo:, k * nb:], Xo, m)
        if k + 1 < nblk and owner[k + 1] == r:
            head += solve(V, A[lo:, k * nb:], Xo, nb)
        for c in (k + 1, k + 2):                        # the two column updates (chain stream: k+1; bulk stream: k+2)
            if c < nblk:
                blocks = [I for I in mine if I >= c]
                lo2 = (len(mine) - len(blocks)) * nb
                sr = [nb] * len(blocks) + ([rhs_of[r]] if rhs_of[r] else [])
                col += stair(A[lo2:, c * nb:], A[lo2:, k * nb:], G, sr, [nb] * len(sr), [1 if I == c else 0 for I in blocks] + ([0] if rhs_of[r] else []))
        blocks = [I for I in mine if I >= k + 3]
        if blocks or (rhs_of[r] and nblk - k - 3 > 0):
            lo2 = (len(mine) - len(blocks)) * nb
            sr = [nb] * len(blocks) + ([rhs_of[r]] if rhs_of[r] and nblk - k - 3 > 0 else [])
            sc = [(I - k - 2) * nb for I in blocks] + ([(nblk - k - 3) * nb] if rhs_of[r] and nblk - k - 3 > 0 else [])
            bulk += stair(A[lo2:, (k + 3) * nb:], A[lo2:, k * nb:], G, sr, sc, [1] * len(blocks) + ([0] if len(sr) > len(blocks) else []))
    res_rows.append((bulk, col, sol, head))
    del A, G, Xo
    torch.cuda.empty_cache()
    out.append('rows rank %d (%d blocks): bulk staircase %.2f ms, column updates %.2f ms, panel solves %.2f ms (head solves on the cycle: %.2f ms)'
               % (r, len(mine), bulk, col, sol, head))
    print(out[-1], flush=True)
# ------------------------------------------------------------------ 2 x 4 grid
Pr, Pc = 2, 4
res_grid = []
for pr in range(Pr):
    for pc in range(Pc):
        rows_I = [I for I in range(nblk) if I % Pr == pr]
        cols_J = [J for J in range(nblk) if J % Pc == pc]
        rhs = RHS // 128 // Pr * 128 + (128 if pr < (RHS // 128) % Pr else 0)
        Cl = rnd(len(rows_I) * nb + rhs, len(cols_J) * nb)                 # the rank's tiles, compact
        Rp = rnd(len(rows_I) * nb + rhs, nb)                               # row panel (my rows of panel k)
        Cp = rnd(len(cols_J) * nb, nb)                                     # column panel (rows J of panel k, my columns J)
        Xo = rnd(len(rows_I) * nb + rhs, nb)
        bulk = sol = 0.0
        for k in range(nblk - 1):
            rI = [I for I in rows_I if I > k]
            cJ = [J for J in cols_J if J > k]
            if pc == k % Pc:                                               # my process column solves panel k
                m = len(rI) * nb + rhs
                sol += solve(V, Cl[(len(rows_I) - len(rI)) * nb:, :nb], Xo, m)
            if not cJ:
                continue
            lo_r, lo_c = (len(rows_I) - len(rI)) * nb, (len(cols_J) - len(cJ)) * nb
            sr, sc, sd = [], [], []
            for I in rI:
                w = sum(1 for J in cJ if J <= I)
                sr.append(nb); sc.append(w * nb); sd.append(1 if (I % Pc == pc and w > 0) else 0)
            if rhs:
                sr.append(rhs); sc.append(len(cJ) * nb); sd.append(0)
            keep = [i for i in range(len(sr)) if sc[i] > 0]
            first = keep[0] if keep else 0
            skip_rows = sum(sr[:first])
            bulk += stair(Cl[lo_r + skip_rows:, lo_c:], Rp[lo_r + skip_rows:], Cp[lo_c:], sr[first:], sc[first:], sd[first:])
        res_grid.append((bulk, sol))
        del Cl, Rp, Cp, Xo
        torch.cuda.empty_cache()
        out.append('grid rank (%d, %d): trailing update (one staircase launch per panel, K = %d) %.2f ms, panel solves %.2f ms (all of them on the cycle of their step)'
                   % (pr, pc, nb, bulk, sol))
        print(out[-1], flush=True)
br = max(range(P), key=lambda r: sum(res_rows[r][:3]))
bg = max(range(Pr * Pc), key=lambda i: sum(res_grid[i]))
out.append('')
out.append('busiest rank, MFMA launches issued one after the other on one stream (no overlap between a rank\'s streams, the other ranks infinitely fast):')
out.append('  rows: %.2f ms  (bulk %.2f + column updates %.2f + panel solves %.2f)   on the cross-rank cycle of the sweep: head solves %.2f ms'
           % ((sum(res_rows[br][:3]),) + res_rows[br][:3] + (sum(x[3] for x in res_rows),)))
out.append('  grid: %.2f ms  (trailing update %.2f + panel solves %.2f)               on the cross-rank cycle of the sweep: panel solves %.2f ms (process column k mod 4 at step k) + two panel broadcasts per step'
           % (sum(res_grid[bg]), res_grid[bg][0], res_grid[bg][1], sum(max(res_grid[pr_ * Pc + k % Pc][1] for pr_ in range(Pr)) for k in range(Pc)) ))
print('\n'.join(out[-3:]), flush=True)
os.makedirs(os.path.join(R, 'gpurun_out'), exist_ok=True)
open(os.path.join(R, 'gpurun_out', 'r5_grid_vs_rows.txt'), 'w').write('\n'.join(out) + '\n')
