// ntt_core.h — negacyclic NTT building blocks for CDNA4, shared by every kernel.
//
// A size-N transform (N = 2^n, 1024 <= N <= 32768) is split as N = N1 x 1024:
//   * column pass : N1 <= 32 point transforms over stride-1024 columns; one lane owns one column in
//                   registers; its twiddles are the same for every column (wave-uniform -> SGPRs).
//   * row pass    : 1024-point transforms over contiguous rows; ONE 64-lane wave owns one row, 16
//                   elements per lane; the 10 stages run as 4 + 4 + 2 register-resident stages with two
//                   LDS exchanges between them (layouts A -> B -> C below, padded against bank conflicts).
// Forward (Cooley-Tukey, natural -> bit-reversed) = column pass then row pass; inverse (Gentleman-Sande)
// = row pass then column pass with N^-1 folded into the last stage.  The ordering of the NTT form
// (bit-reversed evaluations, twiddle table w[bitrev(i)] = psi^i) is the one SEAL uses, so device slabs are
// interchangeable with seal::Ciphertext::data().
//
// Lane programs are written as "phases" (code between two LDS hand-offs) so that tests/csim can execute
// the identical code lane by lane on the CPU.
#pragma once
#include "modarith.h"

namespace he355 {

constexpr int kRowLog = 10;
constexpr int kRowN = 1 << kRowLog; // 1024 elements per row
constexpr int kRowE = 16;           // elements per lane
constexpr int kLdsRow = kRowN + (kRowN >> 6) * 4; // 1088: +4 elements per 64 to spread banks

HE_HD int lds_pad(int e) { return e + ((e >> 6) << 2); }

// element index held in register r of lane `lane` under the three layouts
HE_HD int elemA(int lane, int r) { return (r << 6) | lane; }                                   // r = bits 9..6
HE_HD int elemB(int lane, int r) { return ((lane >> 2) << 6) | (r << 2) | (lane & 3); }       // r = bits 5..2
HE_HD int elemC(int lane, int r) { return ((r >> 2) << 8) | (lane << 2) | (r & 3); }          // r = bits 9,8,1,0

template <class T> HE_HD void lds_store_A(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemA(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_A(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemA(lane, r))];
}
template <class T> HE_HD void lds_store_B(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemB(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_B(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemB(lane, r))];
}
template <class T> HE_HD void lds_store_C(T *lds, int lane, const T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) lds[lds_pad(elemC(lane, r))] = x[r];
}
template <class T> HE_HD void lds_load_C(const T *lds, int lane, T x[kRowE])
{
#pragma unroll
    for (int r = 0; r < kRowE; ++r) x[r] = lds[lds_pad(elemC(lane, r))];
}

// ---------------------------------------------------------------------------------------------------
// Row pass, forward.  rowbase = N1 + a for row a: stage s' (0..9) uses tw[(rowbase << s') + (e >> (10-s'))]
// ---------------------------------------------------------------------------------------------------
// stages 0..3 on layout A (register bit 3-s' is the butterfly bit); twiddles are lane-uniform
template <class Ar> HE_HD void row_fwd_A(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *tw, u32 rowbase)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int bit = 3 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            const Tw16 w = tw[(rowbase << s) + (r >> (4 - s))];
            ar.bfly_fwd(x[r], x[r | (1 << bit)], w);
        }
    }
}
// stages 4..7 on layout B
template <class Ar> HE_HD void row_fwd_B(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *tw, u32 rowbase, int lane)
{
    const u32 hi4 = (u32)lane >> 2;
#pragma unroll
    for (int s = 4; s < 8; ++s) {
        const int bit = 7 - s; // register bit
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            // e >> (10-s) = (hi4 << (s-4)) | (r >> (8-s))
            const Tw16 w = tw[(rowbase << s) + ((hi4 << (s - 4)) | ((u32)r >> (8 - s)))];
            ar.bfly_fwd(x[r], x[r | (1 << bit)], w);
        }
    }
}
// stages 8,9 on layout C (register bits 1,0)
template <class Ar> HE_HD void row_fwd_C(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *tw, u32 rowbase, int lane)
{
#pragma unroll
    for (int s = 8; s < 10; ++s) {
        const int bit = 9 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            const u32 e = (u32)elemC(lane, r);
            const Tw16 w = tw[(rowbase << s) + (e >> (10 - s))];
            ar.bfly_fwd(x[r], x[r | (1 << bit)], w);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Row pass, inverse: stages 9,8 (layout C), 7..4 (layout B), 3..0 (layout A)
// ---------------------------------------------------------------------------------------------------
template <class Ar> HE_HD void row_inv_C(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *itw, u32 rowbase, int lane)
{
#pragma unroll
    for (int s = 9; s >= 8; --s) {
        const int bit = 9 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            const u32 e = (u32)elemC(lane, r);
            const Tw16 w = itw[(rowbase << s) + (e >> (10 - s))];
            ar.bfly_inv(x[r], x[r | (1 << bit)], w);
        }
    }
}
template <class Ar> HE_HD void row_inv_B(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *itw, u32 rowbase, int lane)
{
    const u32 hi4 = (u32)lane >> 2;
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.renorm(x[r]);
    }
#pragma unroll
    for (int s = 7; s >= 4; --s) {
        const int bit = 7 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            const Tw16 w = itw[(rowbase << s) + ((hi4 << (s - 4)) | ((u32)r >> (8 - s)))];
            ar.bfly_inv(x[r], x[r | (1 << bit)], w);
        }
    }
}
// LAST = this is the final stage of the whole transform (N1 == 1): fold N^-1 using itw_scaled for stage 0
template <class Ar, bool LAST> HE_HD void row_inv_A(const Ar &ar, typename Ar::T x[kRowE], const Tw16 *itw, u32 rowbase, const Tw16 &w0_scaled)
{
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int r = 0; r < kRowE; ++r) x[r] = ar.renorm(x[r]);
    }
#pragma unroll
    for (int s = 3; s >= 0; --s) {
        const int bit = 3 - s;
#pragma unroll
        for (int r = 0; r < kRowE; ++r) {
            if (r & (1 << bit)) continue;
            if (LAST && s == 0) {
                ar.bfly_inv_last(x[r], x[r | (1 << bit)], w0_scaled);
            } else {
                const Tw16 w = itw[(rowbase << s) + (r >> (4 - s))];
                ar.bfly_inv(x[r], x[r | (1 << bit)], w);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Column pass: LOGN1 stages over N1 = 2^LOGN1 registers (row index a = register index)
// ---------------------------------------------------------------------------------------------------
template <class Ar, int LOGN1> HE_HD void col_fwd(const Ar &ar, typename Ar::T x[1 << LOGN1], const Tw16 *tw)
{
    constexpr int N1 = 1 << LOGN1;
#pragma unroll
    for (int s = 0; s < LOGN1; ++s) {
        const int gap = N1 >> (s + 1);
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            if (a & gap) continue;
            const Tw16 w = tw[(1 << s) + (a / (2 * gap))];
            ar.bfly_fwd(x[a], x[a + gap], w);
        }
    }
}
// inverse column pass; the very last stage (s == 0) folds N^-1: w0_scaled = itw[1] * N^-1
template <class Ar, int LOGN1> HE_HD void col_inv(const Ar &ar, typename Ar::T x[1 << LOGN1], const Tw16 *itw, const Tw16 &w0_scaled)
{
    constexpr int N1 = 1 << LOGN1;
    if (Ar::kNeedsRenormInv) {
#pragma unroll
        for (int a = 0; a < N1; ++a) x[a] = ar.renorm(x[a]);
    }
#pragma unroll
    for (int s = LOGN1 - 1; s >= 0; --s) {
        const int gap = N1 >> (s + 1);
#pragma unroll
        for (int a = 0; a < N1; ++a) {
            if (a & gap) continue;
            if (s == 0) {
                ar.bfly_inv_last(x[a], x[a + gap], w0_scaled);
            } else {
                const Tw16 w = itw[(1 << s) + (a / (2 * gap))];
                ar.bfly_inv(x[a], x[a + gap], w);
            }
        }
    }
}

} // namespace he355
