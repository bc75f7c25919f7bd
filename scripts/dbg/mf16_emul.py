"""Lane-level numpy emulation of the 16-wide MFMA LDL^T of be_solve (csrc/be_solve.hip: ldlt_mf16 / bs_mf16): every register is a 64-vector, every
cross-lane move is the one the kernel uses (LDS gather, row_newbcast, v_mfma_f64_16x16x4_f64 lane maps of cdna_hip_programming.md).  It pins the
algebra (transposed tiles as MFMA operands, Gauss-Jordan W = L_kk^-1 by rank-4 MFMAs, right-hand side riding as row n, block back substitution)
against numpy before the HIP version is debugged on the GPU.  Not product code, not a test."""
import numpy as np

L64 = np.arange(64)
ROW0, COL = L64 >> 4, L64 & 15          # C layout: reg r of lane l = M[(l>>4) + 4r][l&15]


def to_c(M):            # 16x16 -> [4][64]
    return np.stack([M[ROW0 + 4 * r, COL] for r in range(4)])


def from_c(F):
    M = np.zeros((16, 16))
    for r in range(4):
        M[ROW0 + 4 * r, COL] = F[r]
    return M


def mfma(a, b, c4):
    """v_mfma_f64_16x16x4_f64: A[row = l&15][k = l>>4] = a[l]; B[k = l>>4][col = l&15] = b[l]; C/D layout above"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[COL, ROW0] = a; B[ROW0, COL] = b
    return c4 + to_c(A @ B)


def bcast(v, lane_in_row):        # row_newbcast:n
    return v[(L64 & 48) + lane_in_row]


def diag_factor(T, nvalid):
    """T: diag tile (C layout, symmetric).  Returns W = L^-1 (C layout), d[16], Lhat columns per lane"""
    Wt = to_c(np.eye(16))
    d = np.zeros(16); Lcols = np.zeros((16, 16))
    for q in range(4):
        p = 4 * q
        gat = T[q].copy()                                   # LDS gather buffer: gat[l] = row (l>>4)+4q, col l&15
        g = [gat[16 * m + COL] for m in range(4)]           # g[m][lane] = A[p+m][c]
        a00 = bcast(g[0], p); a10 = bcast(g[1], p); a20 = bcast(g[2], p); a30 = bcast(g[3], p)
        a11 = bcast(g[1], p + 1); a21 = bcast(g[2], p + 1); a31 = bcast(g[3], p + 1)
        a22 = bcast(g[2], p + 2); a32 = bcast(g[3], p + 2); a33 = bcast(g[3], p + 3)
        d0 = a00; i0 = 1 / d0
        l10, l20, l30 = a10 * i0, a20 * i0, a30 * i0
        d1 = a11 - l10 * a10; i1 = 1 / d1
        t21 = a21 - l20 * a10; t31 = a31 - l30 * a10
        l21, l31 = t21 * i1, t31 * i1
        d2 = a22 - l20 * a20 - l21 * t21; i2 = 1 / d2
        t32 = a32 - l30 * a20 - l31 * t21
        l32 = t32 * i2
        d3 = a33 - l30 * a30 - l31 * t31 - l32 * t32; i3 = 1 / d3
        d[p:p + 4] = [d0[0], d1[0], d2[0], d3[0]]
        p0 = g[0]; p1 = g[1] - p0 * l10; p2 = g[2] - p0 * l20 - p1 * l21; p3 = g[3] - p0 * l30 - p1 * l31 - p2 * l32
        pv = [p0, p1, p2, p3]; iv = [i0, i1, i2, i3]
        Lh = [np.where(COL > p + m, pv[m] * iv[m], 0.0) for m in range(4)]
        Ph = [np.where(COL > p + m, pv[m], 0.0) for m in range(4)]
        for m in range(4):
            Lcols[:, p + m] = Lh[m][:16]
        sel = lambda arr: np.choose(ROW0, arr)              # lane l picks entry k = l>>4
        T = mfma(-sel(Lh), sel(Ph), T)
        w10 = -l10; w21 = -l21; w32 = -l32
        w20 = -l20 + l21 * l10; w31 = -l31 + l32 * l21
        w30 = -l30 + l31 * l10 + l32 * l20 - l32 * l21 * l10
        Z = [-(Lh[0] + Lh[1] * w10 + Lh[2] * w20 + Lh[3] * w30), -(Lh[1] + Lh[2] * w21 + Lh[3] * w31), -(Lh[2] + Lh[3] * w32), -Lh[3]]
        Wt = mfma(sel(Z), Wt[q].copy(), Wt)
    return Wt, d, Lcols


def solve(A, rhs):
    n = len(rhs)
    NB = (n + 1 + 15) >> 4
    N = 16 * NB
    Ap = np.eye(N)
    Ap[:n, :n] = A
    Ap[n, :n] = rhs; Ap[:n, n] = rhs; Ap[n, n] = 1e300
    U = {(I, J): to_c(Ap[16 * I:16 * I + 16, 16 * J:16 * J + 16].T) for J in range(NB) for I in range(J, NB)}      # transposed tiles, C layout
    Wd, V, dv = {}, {}, np.zeros(N)
    yv = np.zeros(N)
    IB, c0 = n >> 4, n & 15
    Wd[0], dv[0:16], Lc = diag_factor(U[(0, 0)], 16)
    if IB == 0:
        yv[:c0] = Lc[c0, :c0]
    for k in range(NB):
        Wk = from_c(Wd[k])
        wfrag = np.stack([Wk[COL, ROW0 + 4 * s] for s in range(4)])          # A-layout fragment of W (transposed read)
        ivk = 1.0 / dv[16 * k:16 * k + 16]
        for I in range(k + 1, NB):
            Y = np.zeros((4, 64))
            for s in range(4):
                Y = mfma(wfrag[s], U[(I, k)][s], Y)
            V[(I, k)] = np.stack([Y[r] * ivk[ROW0 + 4 * r] for r in range(4)])
            if I == IB:
                for r in range(4):
                    m = COL == c0
                    yv[16 * k + ROW0[m] + 4 * r] = V[(I, k)][r][m]
        if k == NB - 1:
            break
        dk = dv[16 * k:16 * k + 16]
        for J in range(k + 1, NB):
            for I in range(J, NB):
                for s in range(4):
                    U[(I, J)] = mfma(V[(J, k)][s], -dk[ROW0 + 4 * s] * V[(I, k)][s], U[(I, J)])
        Wd[k + 1], dv[16 * k + 16:16 * k + 32], Lc = diag_factor(U[(k + 1, k + 1)], 16)
        if k + 1 == IB:
            yv[16 * IB:16 * IB + c0] = Lc[c0, :c0]
    # back substitution
    yv[n:] = 0.0
    x = np.zeros(N)
    for i in range(NB - 1, -1, -1):
        W = from_c(Wd[i])
        xi = W.T @ yv[16 * i:16 * i + 16]
        if i == IB:
            xi[c0:] = 0.0
        x[16 * i:16 * i + 16] = xi
        for k in range(i):
            yv[16 * k:16 * k + 16] -= from_c(V[(i, k)]) @ xi           # V = L_ik^T
    return x[:n], dv[:n]


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n in (165, 79, 66, 15, 16, 31):
        M = rng.normal(size=(n, n + 40)); A = M @ M.T + 0.5 * np.eye(n); r = rng.normal(size=n)
        s = 1 / np.sqrt(np.diag(A)); A = A * s[:, None] * s[None, :]
        x, d = solve(A, r)
        ref = np.linalg.solve(A, r)
        print(n, "rel err", np.abs(x - ref).max() / np.abs(ref).max(), "min pivot", d.min())
