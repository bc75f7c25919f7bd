// be_mf16.h — the 16-wide LDL^T on the f64 matrix cores (v_mfma_f64_16x16x4_f64): tile types, the diagonal tile, the tile plan's slots and a plain factorisation loop.
// Shared by the window solve (be_solve.hip: ldlt_mf16, which adds the scaling of the prefetched tiles, the side work beside the first tile and the phase stamps) and
// by the marginalization (be_marg.hip: elimination of the dropped block and the prior's constant c0 in ONE factorisation, rank revealing).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "wave_dpp.h"

// Barrier for exchanges that live in LDS only: s_waitcnt lgkmcnt(0) + s_barrier without the workgroup fence of __syncthreads(), which also drains vmcnt —
// global loads requested ahead of the barrier stay in flight across it (that is the point: operands of a later phase are fetched behind the reductions of
// this one).  NOT a substitute where other threads' global stores must become visible.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double fast_rcp(double d) {      // v_rcp_f64 + two Newton steps (full precision, half the latency of a division)
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// ---------------------------------------------------------------------------------------------------------------------------
// MF16 — the same LDL^T re-blocked 16 wide on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
// Every experiment of rounds 1-2 ended at the same per-step cost (~1.2 us per barrier-separated 4-pivot step, 42 steps): so the steps become 11 sixteen-pivot
// steps.  The (n+1) x (n+1) system — row n carries the right-hand side, so the forward substitution is part of the factorisation and row n of L is
// y = D^-1 L^-1 rhs — is cut into 16 x 16 tiles; tile (I, J), I >= J, lives TRANSPOSED in the registers of one wave in the MFMA result layout
// (lane l, register r: U[(l >> 4) + 4 r][l & 15] = A[16 I + (l & 15)][16 J + (l >> 4) + 4 r]).  In that layout a tile is, without any data movement,
//   * the B operand for a product that sums over its row index, and
//   * the A operand of its transpose,
// which is exactly what the two tile operations need:
//   panel    V_I = X_I^T = D^-1 W_k U_Ik                 A = W_k = L_kk^-1 (fragment read transposed from LDS), B = the tile's registers
//   update   U_IJ -= X_J D X_I^T = V_J^T (D V_I)          A = the stored fragment of V_J, B = the stored fragment of V_I scaled by d  (4 MFMAs per tile)
// The fragments V_I stay in LDS as the factor (L_Ik^T); the diagonal slots hold W_k.  Diagonal tile (one wave, ~1 us): four 4-pivot sub-steps; the four pivot
// rows are gathered through LDS, the 4x4 pivot block reaches every lane by row_newbcast, every lane runs the pivot chain and, by symmetry, the forward
// substitution of ITS column (lane c holds A[p..p+3][c] = A[c][p..p+3]); ONE MFMA applies the rank-4 update to the tile and a second one the same elementary
// block transformation to W (Gauss-Jordan: W <- L_q^-1 W), so that W_k = L_kk^-1 comes out of the factorisation.  Tile ownership (be_mf16_plan): wave j owns
// diagonal tile j and only off-diagonal tiles of columns < j, so the wave that factors diagonal k+1 has no other tile to update in step k.
// Two workgroup barriers per 16 pivots.  Same solution as ldlt_wavecol to rounding (different summation order): parity by tolerance (iteration sequences, 1e-6).
#define MF_SLOTS 5
#define MF_MAXNB 11
#define MF_RHO 1e300
typedef double mf_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mf_d4 mf_mfma(double a, double b, mf_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mf_tix(int I, int J, int NB) { return J * NB - (J * (J - 1)) / 2 + (I - J); }
template <int LANE> __device__ __forceinline__ double mf_rowbc(double v) { return dpp_f64<0x150 + LANE, 0xf>(v); }      // row_newbcast:LANE (gfx90a+): lane LANE of every 16-lane row to the whole row
__device__ __forceinline__ double mf_sel4(int rho, double v0, double v1, double v2, double v3) {      // v[rho] as three selects (a nested ?: chain is compiled into exec-mask branches)
    const bool b0 = (rho & 1) != 0, b1 = (rho & 2) != 0;
    const double t0 = b0 ? v1 : v0, t1 = b0 ? v3 : v2;
    return b1 ? t1 : t0;
}
// (Round 5: hardware wave h runs on SIMD h & 3 (measured, scripts/dbg/diag_bench.hip: trailing-update MFMAs on waves 4, 8, 12 stretch a diagonal tile on wave 0 from
//  1.52 to 2.11 us, the same load on the other twelve waves does nothing).  Dealing the plan's logical waves to the SIMDs in blocks of four — the chain wave's
//  SIMD-mates are then its index neighbours, idle or nearly so — changed the kernel by nothing measurable (61.7 against 61.6 us, diagonal tiles 1.85 against 1.90 us):
//  the mates that remain still own the tiles of the next columns.  Left at the identity.)
__device__ __forceinline__ int mf_wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
struct MfLds { double* Tl; double* gat; double* dv; double* iv; double* yv; };      // tiles | per-wave gather buffers (64 each) | D | 1 / D | y (then the running right-hand side of the back substitution)

// one 4-pivot sub-step of the diagonal tile.  c0: local index of the right-hand-side row (or -1); yk: this block's 16 entries of y.
// The instruction stream of this one wave IS the critical path of the factorisation (~10 cycles per instruction in this mix of dependent fp64, LDS and MFMA
// operations), so everything that is not arithmetic is kept out of it: the pivot block comes from the gather buffer by same-address (broadcast) LDS reads next
// to the column reads, selections are flat v_cndmask pairs, D / 1/D / y leave through single masked 32-byte stores, the pivot check runs once per tile.
// RR (rank revealing, the marginalization): a pivot <= 1e-8 is SKIPPED — its reciprocal is taken as 0, so its column of L vanishes, it contributes nothing to any update
// and its entry of y is 0: the LDL^T counterpart of the reference's eigenvalue-clamped pseudo-inverse (marginalization_factor.cpp:286-289, eps = 1e-8); D keeps the pivot
// itself (the caller reads the smallest one and the clamp flag from it).
template <int Q, bool RR>
__device__ __forceinline__ void mf_diag_substep(mf_d4& T, mf_d4& Wt, double* gat, double* dv16, double* iv16, int c0, double* yk) {
    constexpr int P = 4 * Q;
    const int lane = threadIdx.x & 63, c = lane & 15, rho = lane >> 4;
    gat[lane] = T[Q];                                   // rows P .. P+3 sit in register Q of the four 16-lane rows
    wave_lds_sync();
    const double g0 = gat[c], g1 = gat[16 + c], g2 = gat[32 + c], g3 = gat[48 + c];      // g_m = A[P + m][c] (= A[c][P + m])
    const double d0 = gat[P], a10 = gat[16 + P], a11 = gat[16 + P + 1], a20 = gat[32 + P], a21 = gat[32 + P + 1], a22 = gat[32 + P + 2];
    const double a30 = gat[48 + P], a31 = gat[48 + P + 1], a32 = gat[48 + P + 2], a33 = gat[48 + P + 3];
    wave_lds_sync();
    const double i0 = RR ? (d0 > 1e-8 ? fast_rcp(d0) : 0.0) : fast_rcp(d0);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
    const double d1 = __builtin_fma(-l10, a10, a11), i1 = RR ? (d1 > 1e-8 ? fast_rcp(d1) : 0.0) : fast_rcp(d1);
    const double t21 = __builtin_fma(-l20, a10, a21), t31 = __builtin_fma(-l30, a10, a31);
    const double l21 = t21 * i1, l31 = t31 * i1;
    const double d2 = __builtin_fma(-l21, t21, __builtin_fma(-l20, a20, a22)), i2 = RR ? (d2 > 1e-8 ? fast_rcp(d2) : 0.0) : fast_rcp(d2);
    const double t32 = __builtin_fma(-l31, t21, __builtin_fma(-l30, a20, a32));
    const double l32 = t32 * i2;
    const double d3 = __builtin_fma(-l32, t32, __builtin_fma(-l31, t31, __builtin_fma(-l30, a30, a33))), i3 = RR ? (d3 > 1e-8 ? fast_rcp(d3) : 0.0) : fast_rcp(d3);
    // this lane's column, read as row c of the panel: p = (L D)[c][P..P+3], x = L[c][P..P+3]; only the strictly lower part of L counts
    const double p0 = g0, p1 = __builtin_fma(-p0, l10, g1);
    const double p2 = __builtin_fma(-p1, l21, __builtin_fma(-p0, l20, g2));
    const double p3 = __builtin_fma(-p2, l32, __builtin_fma(-p1, l31, __builtin_fma(-p0, l30, g3)));
    const double lh0 = c > P ? p0 * i0 : 0.0, lh1 = c > P + 1 ? p1 * i1 : 0.0, lh2 = c > P + 2 ? p2 * i2 : 0.0, lh3 = c > P + 3 ? p3 * i3 : 0.0;
    // T -= Lhat (D Lhat)^T: rank 4, the whole tile in one instruction.  Only the A operand is masked: the unmasked columns of B reach finished columns of the
    // tile only, which are never read again.
    T = mf_mfma(-mf_sel4(rho, lh0, lh1, lh2, lh3), mf_sel4(rho, p0, p1, p2, p3), T);
    // W <- L_q^-1 W with L_q^-1 = I - Lhat W_qq embedded in columns P .. P+3 (W_qq = inverse of the unit lower 4x4 pivot block)
    const double w20 = __builtin_fma(l21, l10, -l20), w31 = __builtin_fma(l32, l21, -l31);
    const double w30 = __builtin_fma(-l32, w20, __builtin_fma(l31, l10, -l30));      // -l30 + l31 l10 + l32 l20 - l32 l21 l10
    const double z0 = -__builtin_fma(lh3, w30, __builtin_fma(lh2, w20, __builtin_fma(-lh1, l10, lh0)));
    const double z1 = -__builtin_fma(lh3, w31, __builtin_fma(-lh2, l21, lh1));
    const double z2 = -__builtin_fma(-lh3, l32, lh2);
    const double wq = Wt[Q];
    Wt = mf_mfma(mf_sel4(rho, z0, z1, z2, -lh3), wq, Wt);
    if (lane == 0) {
        mf_d4 dd = { d0, d1, d2, d3 }, ii = { i0, i1, i2, i3 };
        *reinterpret_cast<mf_d4*>(dv16 + P) = dd; *reinterpret_cast<mf_d4*>(iv16 + P) = ii;
    }
    if (lane == c0) { mf_d4 yy = { lh0, lh1, lh2, lh3 }; *reinterpret_cast<mf_d4*>(yk + P) = yy; }      // row c0 of L_kk = this block's share of y (zero from column c0 on: lh is masked)
}
template <bool RR = false>
__device__ __forceinline__ void mf_diag_factor(mf_d4& T, const MfLds& m, int k, int NB, int n, int* s_fail) {
    const int lane = threadIdx.x & 63, c = lane & 15, rho = lane >> 4, wave = mf_wave();
    __builtin_amdgcn_s_setprio(3);
    mf_d4 Wt;
#pragma unroll
    for (int r = 0; r < 4; ++r) Wt[r] = (rho + 4 * r == c) ? 1.0 : 0.0;
    double* gat = m.gat + wave * 64;
    const int nv = n - 16 * k, c0 = (nv >= 0 && nv < 16) ? nv : -1;      // nv: rows of this tile that belong to the n x n system
    // The tile that holds the right-hand-side row (the last one: c0 >= 0) ends with that row and identity padding: a sub-step whose four pivots all lie behind the
    // row (4 Q > c0) touches nothing that is read again — W stays the identity there, y is zero-initialised, the pivot check below looks at c < nv only — and is
    // skipped (n = 165: two of the last tile's four sub-steps, ~1.1 us per solve).
    mf_diag_substep<0, RR>(T, Wt, gat, m.dv + 16 * k, m.iv + 16 * k, c0, m.yv + 16 * k);
    if (c0 < 0 || c0 >= 4) mf_diag_substep<1, RR>(T, Wt, gat, m.dv + 16 * k, m.iv + 16 * k, c0, m.yv + 16 * k);
    if (c0 < 0 || c0 >= 8) mf_diag_substep<2, RR>(T, Wt, gat, m.dv + 16 * k, m.iv + 16 * k, c0, m.yv + 16 * k);
    if (c0 < 0 || c0 >= 12) mf_diag_substep<3, RR>(T, Wt, gat, m.dv + 16 * k, m.iv + 16 * k, c0, m.yv + 16 * k);
    *reinterpret_cast<mf_d4*>(m.Tl + (size_t)mf_tix(k, k, NB) * 256 + lane * 4) = Wt;      // W_k = L_kk^-1, result layout (the workgroup barrier behind this tile publishes it)
    // ... and a second image in the PANEL's operand order (lane (rho, c) of a panel wave wants W[c][rho + 4 q], q = 0..3: here four consecutive doubles at lane * 4): the
    // transposing read of the result-layout image is a 4-way bank conflict, and all sixteen waves issue it in the same instant behind barrier A — the panel phase was
    // bound by the LDS pipe (0.8 us per step, phase stamps), not by its four MFMAs.  One wave pays the scattered WRITE once instead.  The image lives in the gather
    // buffers of waves 12-15, idle during the factorisation (diagonal tiles belong to waves <= 10); one image at a time suffices: W_k is dead behind barrier B of step k.
    {
        double* wt = m.gat + 768;
#pragma unroll
        for (int r = 0; r < 4; ++r) wt[((((c & 3) << 4) + rho + 4 * r) << 2) + (c >> 2)] = Wt[r];
    }
    __builtin_amdgcn_s_setprio(0);      // (the pivots are checked once, behind the last tile: ldlt_mf16)
}
// Three other forms of this tile were built and measured in round 5 (scripts/dbg/diag_bench.hip keeps them: rank-1 MFMA Gauss-Jordan in place, the same with the next
// reciprocal formed in the MFMA's shadow, column per lane on v_fmac_f64 DPP row_newbcast): 1.65 / 1.62 / 1.76 us per tile against 1.52 us for the form above, alone on
// its SIMD — an MFMA result takes ~200 cycles to reach a VALU consumer, and DP-ALU DPP operations issue at half rate.  What the measurements DID show: trailing-update
// MFMAs on the chain wave's own SIMD stretch the tile to 2.11 us (the figure seen inside the kernel); mf_wave() below deals with that.
__device__ __forceinline__ void mf_diag(mf_d4& T, const MfLds& m, int k, int NB, int n, int* s_fail) { mf_diag_factor<false>(T, m, k, NB, n, s_fail); }
__device__ __forceinline__ void mf_slots(const uint8_t* plan, int wave, int NB, int (&sI)[5], int (&sJ)[5]) {
    sI[0] = sJ[0] = wave < NB ? wave : -1;
    // the wave's four plan bytes as ONE dword through the scalar path (the plan sits in the kernel arguments / the constant-space table, `wave` is an SGPR); byte
    // loads have no scalar form on gfx950 and came back as four dependent vector loads in front of everything that needs the slots (~2 us per use)
    const uint32_t w4 = reinterpret_cast<const uint32_t*>(plan)[wave];
#pragma unroll
    for (int s = 1; s < 5; ++s) { const int b = (int)((w4 >> (8 * (s - 1))) & 0xFFu); sI[s] = b == 0xFF ? -1 : (b >> 4); sJ[s] = b == 0xFF ? -1 : (b & 15); }
}

// The plain factorisation loop on tiles that are already final in the registers (U by the plan's slots, yv zeroed, a barrier behind both): schedule as ldlt_mf16
// (be_solve.hip).  after_update(k): called by EVERY wave once step k's updates of its tiles are in its registers — by the chain wave for its diagonal tile (slot 0)
// before it factors it — i.e. at the one moment the trailing tiles hold the Schur complement of the first 16 (k + 1) pivots (the marginalization writes A', b' there).
// On return: D, 1 / D in m.dv / m.iv, y = row n of L in m.yv (natural order), the factor's fragments in m.Tl; a workgroup barrier has been passed.
template <bool RR, class After>
__device__ __forceinline__ void mf16_factor_core(mf_d4 (&U)[MF_SLOTS], const uint8_t* plan, int n, const MfLds& m, int* s_fail, After after_update) {
    const int lane = threadIdx.x & 63, wave = mf_wave(), c = lane & 15, rho = lane >> 4;
    const int NB = (n + 16) >> 4, IB = n >> 4, c0 = n & 15;
    int sI[MF_SLOTS], sJ[MF_SLOTS];
    mf_slots(plan, wave, NB, sI, sJ);
    if (wave == 0) mf_diag_factor<RR>(U[0], m, 0, NB, n, s_fail);
    lds_barrier();                                       // A: W_0 published
    auto panel = [&](mf_d4& T, int I, int k) -> mf_d4 {      // returns W_k U (before the division by D)
        const mf_d4 wf = *reinterpret_cast<const mf_d4*>(m.gat + 768 + lane * 4);
        double ivr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ivr[q] = m.iv[16 * k + rho + 4 * q];
        mf_d4 Y = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int q = 0; q < 4; ++q) Y = mf_mfma(wf[q], T[q], Y);
        const mf_d4 Y0 = Y;
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[r] *= ivr[r];
        T = Y;
        *reinterpret_cast<mf_d4*>(m.Tl + (size_t)mf_tix(I, k, NB) * 256 + lane * 4) = Y;
        if (I == IB && c == c0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) m.yv[16 * k + rho + 4 * r] = Y[r];
        }
        return Y0;
    };
    for (int k = 0; k + 1 < NB; ++k) {
        if (wave == k + 1) {
            __builtin_amdgcn_s_setprio(3);
            const mf_d4 Y0 = panel(U[1], k + 1, k);
#pragma unroll
            for (int q = 0; q < 4; ++q) U[0] = mf_mfma(U[1][q], -Y0[q], U[0]);      // D V = W U: no read of D on the chain (a skipped pivot has V = 0 there: its product vanishes as before)
            after_update(k);
            if (k + 2 < NB) lds_barrier();               // B
            mf_diag_factor<RR>(U[0], m, k + 1, NB, n, s_fail);
            lds_barrier();                               // A
        } else {
#pragma unroll
            for (int s = 1; s < MF_SLOTS; ++s) if (sJ[s] == k) panel(U[s], sI[s], k);
            if (k + 2 < NB) lds_barrier();               // B
            double dk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dk[q] = -m.dv[16 * k + rho + 4 * q];
#pragma unroll
            for (int s = 0; s < MF_SLOTS; ++s) if (sJ[s] > k) {
                const mf_d4 a = *reinterpret_cast<const mf_d4*>(m.Tl + (size_t)mf_tix(sJ[s], k, NB) * 256 + lane * 4);
                const mf_d4 b = *reinterpret_cast<const mf_d4*>(m.Tl + (size_t)mf_tix(sI[s], k, NB) * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) U[s] = mf_mfma(a[q], dk[q] * b[q], U[s]);
            }
            after_update(k);
            lds_barrier();                               // A
        }
    }
}
