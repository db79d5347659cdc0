// The search kernels' joiner sweep on the matrix pipe (k_greedy, k_decoder_table in greedy.hip; k_beam_loop in beam.hip).
#pragma once
#include "kernels.h"

namespace k2hip {
namespace {

constexpr int GF = 8;      // frames per round
constexpr int GT = 512;    // threads per workgroup (8 waves; 1024 spilled 65 VGPRs to scratch and was slower, 256 starves the sweep)
// (the vector sweep's 8 k slices per wave read actT rows kper*GF floats apart and were skewed by APAD floats against the 8-way bank
// conflict; on the matrix pipe a wave reads one row at a time: no skew)
constexpr int APAD = 0;
// weight rows in flight per lane in the joiner sweep
constexpr int GL = 8;
constexpr int kPsumFloats = (GT / 64) * GF * 256;   // 64 KB

// ---- the sweep on the matrix pipe ------------------------------------------------------------------
// c += act[GF frames][rows] . W[rows][the lane's 4 columns] for the `nrows` rows of one k slice, one wave.
// v_mfma_f32_4x4x1_16B_f32 is 16 independent 4 x 4 outer products: block = lane / 4, A[lane] = the row's activation of frame
// lane % 4 (the same in every block: al = the slice's activations + (lane & 3), GF floats per row), B[lane] = the lane's weight, and
// accumulator register i of a lane is frame i x the lane's column -- 8 instructions per weight row give 8 frames x 256 columns, none
// of the shape wasted, each an fma onto the slice's running sum exactly like the vector FMAs they replace
// (tools/probes/sweep_compute_probe.hip compares the two bit for bit).  The vector pipe needed 32 v_fma (~4 cycles each with two waves
// on the SIMD) for what these 8 (~9 cycles each) do: 7.3 -> 5.2 us per 0.5 MB slab in the probe, 8.9 us as the compiler packed the
// FMAs (v_pk_fma_f32 is slower than two v_fma on gfx950).
// wp: the lane's 4 columns of the slice's first row (16-byte aligned), ldw floats between rows.
typedef float f32x4 __attribute__((ext_vector_type(4)));
// NH = 1: only frames 0..3 (c[0]): half the MFMAs, for callers with at most 4 live rows (beam <= 4)
template <int NH = 2>
__device__ __forceinline__ void mfma_sweep_rows(const float* __restrict__ wp, long long ldw, const float* al, int nrows, f32x4 (&c)[2][4]) {
    if (nrows <= 0) return;
    // A ring of GL weight rows per lane: row k + GL is requested as soon as row k has been used, and the next row's activations are
    // read one row ahead; the scheduling barrier keeps the compiler from sinking the requests to just before their use (which left
    // 2-4 in flight: the whole L2 latency, ~0.5 us, per group of rows).
    float4 wv[GL];
#pragma unroll
    for (int i = 0; i < GL; i++) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)min(i, nrows - 1) * ldw);
    float anext[2] = {al[0], al[4]};
    auto row = [&](int k, int i, bool reload) {
        const float alo = anext[0], ahi = anext[1];
        const int kn = min(k + 1, nrows - 1);
        anext[0] = al[kn * GF];
        if (NH == 2) anext[1] = al[kn * GF + 4];
        const float4 wk = wv[i];
        if (reload) wv[i] = *reinterpret_cast<const float4*>(wp + (long long)min(k + GL, nrows - 1) * ldw);
        c[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.x, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.y, c[0][1], 0, 0, 0);
        c[0][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.z, c[0][2], 0, 0, 0);
        c[0][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, wk.w, c[0][3], 0, 0, 0);
        if (NH == 2) {
            c[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.x, c[1][0], 0, 0, 0);
            c[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.y, c[1][1], 0, 0, 0);
            c[1][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.z, c[1][2], 0, 0, 0);
            c[1][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(ahi, wk.w, c[1][3], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kb = 0;
    for (; kb + 2 * GL <= nrows; kb += GL) {
#pragma unroll
        for (int i = 0; i < GL; i++) row(kb + i, i, true);
    }
    for (; kb + GL <= nrows; kb += GL) {   // the last whole group: only the tail's rows are still to be requested
#pragma unroll
        for (int i = 0; i < GL; i++) row(kb + i, i, kb + GL + i < nrows);
    }
#pragma unroll
    for (int i = 0; i < GL; i++)
        if (kb + i < nrows) row(kb + i, i, false);   // (uniform)
}
// Two 256-column chunks at once for at most 4 live rows (the beam search at beam <= 4): c[ch][q] = chunk ch, column quad q, rows 0..3.
// The same fma chains as two calls of mfma_sweep_rows<1>; one ring of 2 x GL weight rows, one activation read per row.
__device__ __forceinline__ void mfma_sweep_rows_2c(const float* __restrict__ wp0, const float* __restrict__ wp1, long long ldw, const float* al,
                                                   int nrows, f32x4 (&c)[2][4]) {
    if (nrows <= 0) return;
    float4 wa[GL], wb[GL];
#pragma unroll
    for (int i = 0; i < GL; i++) {
        wa[i] = *reinterpret_cast<const float4*>(wp0 + (long long)min(i, nrows - 1) * ldw);
        wb[i] = *reinterpret_cast<const float4*>(wp1 + (long long)min(i, nrows - 1) * ldw);
    }
    float anext = al[0];
    auto row = [&](int k, int i, bool reload) {
        const float alo = anext;
        anext = al[min(k + 1, nrows - 1) * GF];
        const float4 ka = wa[i], kb4 = wb[i];
        if (reload) {
            wa[i] = *reinterpret_cast<const float4*>(wp0 + (long long)min(k + GL, nrows - 1) * ldw);
            wb[i] = *reinterpret_cast<const float4*>(wp1 + (long long)min(k + GL, nrows - 1) * ldw);
        }
        c[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, ka.x, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, ka.y, c[0][1], 0, 0, 0);
        c[0][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, ka.z, c[0][2], 0, 0, 0);
        c[0][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, ka.w, c[0][3], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, kb4.x, c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, kb4.y, c[1][1], 0, 0, 0);
        c[1][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, kb4.z, c[1][2], 0, 0, 0);
        c[1][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(alo, kb4.w, c[1][3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int kb = 0;
    for (; kb + 2 * GL <= nrows; kb += GL) {
#pragma unroll
        for (int i = 0; i < GL; i++) row(kb + i, i, true);
    }
    for (; kb + GL <= nrows; kb += GL) {
#pragma unroll
        for (int i = 0; i < GL; i++) row(kb + i, i, kb + GL + i < nrows);
    }
#pragma unroll
    for (int i = 0; i < GL; i++)
        if (kb + i < nrows) row(kb + i, i, false);   // (uniform)
}
// the k slices' partial sums of one pass go through LDS: psum[slice][frame][256 columns]
template <int NH = 2>
__device__ __forceinline__ void psum_store(float* psum, int wave, int lane, const f32x4 (&c)[2][4]) {
#pragma unroll
    for (int hh = 0; hh < NH; hh++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<float4*>(psum + ((wave * GF) + 4 * hh + i) * 256 + 4 * lane) = make_float4(c[hh][0][i], c[hh][1][i], c[hh][2][i], c[hh][3][i]);
}

}  // namespace
}  // namespace k2hip
