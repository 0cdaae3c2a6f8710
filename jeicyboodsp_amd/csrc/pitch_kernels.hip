// pitch_kernels.hip -- PitchEstimation_method1.cpp:69-116 (CalcPitch) on gfx950, one block per
// wavefront: frame = [previous block, block] (no window) -> forward transform -> |X|^2 ->
// inverse transform -> autocorrelation r[0..511] -> arg max over lags 511 .. 101.
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

template <int J>
__device__ __forceinline__ void power_presplit_j(const float2 *lds, float2 *zout, int lane, const float2 *wsp)
{
    const int m = 128 * J + 2 * lane;
    const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
    float2 zr0, zr1;
    load_mirror_pair(lds, m, zr0, zr1);
    float2 lo0, hi0, lo1, hi1;
    split_fwd<J>(make_float2(zz.x, zz.y), zr0, wsp[0], lo0, hi0);
    split_fwd<J>(make_float2(zz.z, zz.w), zr1, wsp[1], lo1, hi1);
    // :91-92  |X|^2 + 0j
    const float2 pl0 = make_float2(lo0.x * lo0.x + lo0.y * lo0.y, 0.f), ph0 = make_float2(hi0.x * hi0.x + hi0.y * hi0.y, 0.f);
    const float2 pl1 = make_float2(lo1.x * lo1.x + lo1.y * lo1.y, 0.f), ph1 = make_float2(hi1.x * hi1.x + hi1.y * hi1.y, 0.f);
    zout[2 * J] = presplit_inv<J>(pl0, ph0, wsp[0]);
    zout[2 * J + 1] = presplit_inv<J>(pl1, ph1, wsp[1]);
}

__global__ __launch_bounds__(64) void pitch_autocorr_kernel(const short *__restrict__ pcm, long n_blocks,
                                                            const short *__restrict__ prev_block,
                                                            const float2 *__restrict__ table, int *__restrict__ arg,
                                                            float *__restrict__ rmax, float *__restrict__ autocorr)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long b = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (b >= n_blocks) return;
    // raw[r]: the sample pair (2 lane + 128 r, +1) of the frame [keep buffer (:74,:79-81), block]: four
    // coalesced dword loads per half, already in the transform's layout
    const unsigned int *p0 = b > 0 ? reinterpret_cast<const unsigned int *>(pcm + (b - 1) * 512)
                                   : reinterpret_cast<const unsigned int *>(prev_block);
    const unsigned int *p1 = reinterpret_cast<const unsigned int *>(pcm + b * 512);
    unsigned int raw[8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        raw[r] = p0 ? p0[lane + 64 * r] : 0u;
        raw[r + 4] = p1[lane + 64 * r];
    }
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 s = unpack_i16x2(raw[r]);
        v[r] = make_float2(0.5f * s.x, 0.5f * s.y);                     // 0.5: the split's convention (frame_io.h)
    }
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
    float2 z[8];
    power_presplit_j<0>(lds, z, lane, wsp);
    power_presplit_j<1>(lds, z, lane, wsp);
    power_presplit_j<2>(lds, z, lane, wsp);
    power_presplit_j<3>(lds, z, lane, wsp);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 4; j++)
        *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y);
    wave_lds_fence();
    float2 y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, tw);
    // y[d] = (r[2 lane + 128 d], r[2 lane + 128 d + 1]) * 1024 ; lags 0..511 are d = 0..3 (:95-97)
    float best = -INFINITY;
    int at = 0x7fffffff;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int i0 = 2 * lane + 128 * d;
        const float a = y[d].x * (1.0f / 1024.0f), c = y[d].y * (1.0f / 1024.0f);
        if (autocorr) *reinterpret_cast<float2 *>(autocorr + b * 512 + i0) = make_float2(a, c);
        // :102-108 scans 511 -> 101 with >=: the largest value wins, ties go to the SMALLEST lag
        // (bitwise, not short-circuit, conditions: selects instead of divergent branches)
        const bool ta = (i0 > 100) & ((a > best) | ((a == best) & (i0 < at)));
        best = ta ? a : best;
        at = ta ? i0 : at;
        const bool tc = (i0 + 1 > 100) & ((c > best) | ((c == best) & (i0 + 1 < at)));
        best = tc ? c : best;
        at = tc ? i0 + 1 : at;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(at, o);
        const bool to = (ob > best) | ((ob == best) & (oa < at));
        best = to ? ob : best;
        at = to ? oa : at;
    }
    if (lane == 0) {
        arg[b] = at;
        rmax[b] = best;
    }
}

// The same blocks with the spectrum in registers (frame_io.h, pair-owned bins: five split items per lane, |X|^2 of the
// bins m and m + 512 for m = lane + 64 d, d < 5, the rest of the inverse transform's input from the symmetry of a power
// spectrum), persistent waves that walk `run` consecutive blocks (each block's samples loaded once and kept for the next
// frame, the following block requested before this block's arithmetic), and the arg max reduced with DPP moves instead
// of twelve ds_bpermute.
#ifndef JDSP_PITCH_RUN
#define JDSP_PITCH_RUN 1
#endif
#ifndef JDSP_PITCH_WAVES
#define JDSP_PITCH_WAVES 4
#endif
// one step of the wave-wide arg max: the candidate of the lane that `CTRL` names replaces this lane's when it is larger,
// or equal with the smaller lag (:102-108 scans 511 -> 101 with >=)
#define JDSP_ARGMAX_STEP(CTRL, ROWS)                                                                                   \
    {                                                                                                                  \
        const float ob = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(best), __float_as_int(best), CTRL, ROWS, 0xf, false)); \
        const int oa = __builtin_amdgcn_update_dpp(at, at, CTRL, ROWS, 0xf, false);                                     \
        const bool to = (ob > best) | ((ob == best) & (oa < at));                                                       \
        best = to ? ob : best;                                                                                          \
        at = to ? oa : at;                                                                                              \
    }

__global__ __launch_bounds__(64, JDSP_PITCH_WAVES) void pitch_run_kernel(const short *__restrict__ pcm, long n_blocks,
                                                                         const short *__restrict__ prev_block,
                                                                         const float2 *__restrict__ table, int *__restrict__ arg,
                                                                         float *__restrict__ rmax, float *__restrict__ autocorr,
                                                                         int run)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                // XCD-aware run order (speed only)
    const long b0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * run;
    if (b0 >= n_blocks) return;
    const long b1 = b0 + run < n_blocks ? b0 + run : n_blocks;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    PairTwiddles pw;
    load_pair_twiddles(pw, table, lane);
    // raw[r]: the sample pair (2 lane + 128 r, +1) of the frame [keep buffer (:74,:79-81), block]
    unsigned int raw[8], nxt[4];
    {
        const unsigned int *p0 = b0 > 0 ? reinterpret_cast<const unsigned int *>(pcm + (b0 - 1) * 512)
                                        : reinterpret_cast<const unsigned int *>(prev_block);
        const unsigned int *p1 = reinterpret_cast<const unsigned int *>(pcm + b0 * 512);
#pragma unroll
        for (int r = 0; r < 4; r++) { raw[r + 4] = p0 ? p0[lane + 64 * r] : 0u; nxt[r] = p1[lane + 64 * r]; }
    }
    for (long b = b0; b < b1; b++) {
#pragma unroll
        for (int r = 0; r < 4; r++) { raw[r] = raw[r + 4]; raw[r + 4] = nxt[r]; }
        if (b + 1 < b1) {
            const unsigned int *pn = reinterpret_cast<const unsigned int *>(pcm + (b + 1) * 512);
#pragma unroll
            for (int r = 0; r < 4; r++) nxt[r] = pn[lane + 64 * r];
        }
        float2 v[8], zr[5], y[8], ret[4];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 sp = unpack_i16x2(raw[r]);
            v[r] = make_float2(0.5f * sp.x, 0.5f * sp.y);                   // 0.5: the split's convention (frame_io.h)
        }
        wave_fft512<false>(v, lds, lane, tw);
        wave_lds_fence();
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
            const float2 t = cmul(pw.w[d], o);
            const float2 lo = cadd(e, t), hi = csub(e, t);
            // :91-92  |X|^2 + 0j for the bins m and m + 512; Z'[m] = S + j D conj(W^m), Z'[512 - m] = conj(S - j D conj(W^m))
            const float pl = lo.x * lo.x + lo.y * lo.y, ph = hi.x * hi.x + hi.y * hi.y;
            const float S = pl + ph, D = pl - ph;
            const float rx = D * pw.w[d].x, ry = -D * pw.w[d].y;
            y[d] = make_float2(S - ry, rx);
            if (d < 4) ret[d] = make_float2(S + ry, rx);
        }
        pair_return_lds(ret, lds, lane, y);
        wave_fft512<true>(y, lds, lane, tw);
        wave_lds_fence();
        // y[d] = (r[2 lane + 128 d], r[2 lane + 128 d + 1]) * 1024 ; lags 0..511 are d = 0..3 (:95-97)
        float best = -INFINITY;
        int at = 0x7fffffff;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int i0 = 2 * lane + 128 * d;
            const float a = y[d].x * (1.0f / 1024.0f), c = y[d].y * (1.0f / 1024.0f);
            if (autocorr) *reinterpret_cast<float2 *>(autocorr + b * 512 + i0) = make_float2(a, c);
            // the largest value wins, ties go to the SMALLEST lag (bitwise conditions: selects, not branches)
            const bool ta = (i0 > 100) & ((a > best) | ((a == best) & (i0 < at)));
            best = ta ? a : best;
            at = ta ? i0 : at;
            const bool tc = (i0 + 1 > 100) & ((c > best) | ((c == best) & (i0 + 1 < at)));
            best = tc ? c : best;
            at = tc ? i0 + 1 : at;
        }
        JDSP_ARGMAX_STEP(0xB1, 0xf)      // quad_perm [1,0,3,2]
        JDSP_ARGMAX_STEP(0x4E, 0xf)      // quad_perm [2,3,0,1]
        JDSP_ARGMAX_STEP(0x141, 0xf)     // row_half_mirror
        JDSP_ARGMAX_STEP(0x140, 0xf)     // row_mirror: every lane of a row holds the row's winner
        JDSP_ARGMAX_STEP(0x142, 0xa)     // row_bcast15 into rows 1 and 3
        JDSP_ARGMAX_STEP(0x143, 0xc)     // row_bcast31 into rows 2 and 3: lane 63 holds the wave's winner
        if (lane == 63) {
            arg[b] = at;
            rmax[b] = best;
        }
    }
}

// ONE block per wave, nothing kept between blocks (the form that paid for the 512-FFT MFCC kernel): tables loaded where
// they are used.  JDSP_PITCH_ONE selects it (A/B in profiles/r02_pitch_run.txt).
#ifndef JDSP_PITCH_ONE
#define JDSP_PITCH_ONE 1
#endif
#ifndef JDSP_PITCH_ONE_WAVES
#define JDSP_PITCH_ONE_WAVES 6
#endif
__global__ __launch_bounds__(64, JDSP_PITCH_ONE_WAVES) void pitch_one_kernel(const short *__restrict__ pcm, long n_blocks,
                                                       const short *__restrict__ prev_block,
                                                       const float2 *__restrict__ table, int *__restrict__ arg,
                                                       float *__restrict__ rmax, float *__restrict__ autocorr)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long b = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (b >= n_blocks) return;
    const unsigned int *p0 = b > 0 ? reinterpret_cast<const unsigned int *>(pcm + (b - 1) * 512)
                                   : reinterpret_cast<const unsigned int *>(prev_block);
    const unsigned int *p1 = reinterpret_cast<const unsigned int *>(pcm + b * 512);
    float2 v[8], y[8];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float2 a = unpack_i16x2(p0 ? p0[lane + 64 * r] : 0u), c = unpack_i16x2(p1[lane + 64 * r]);
        v[r] = make_float2(0.5f * a.x, 0.5f * a.y);
        v[r + 4] = make_float2(0.5f * c.x, 0.5f * c.y);
    }
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    wave_fft512<false>(v, lds, lane, tw);
    wave_lds_fence();
    {
        PairTwiddles pw;
        load_pair_twiddles(pw, table, lane);
        float2 zr[5], ret[4];
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
            const float2 t = cmul(pw.w[d], o);
            const float2 lo = cadd(e, t), hi = csub(e, t);
            const float pl = lo.x * lo.x + lo.y * lo.y, ph = hi.x * hi.x + hi.y * hi.y;
            const float S = pl + ph, D = pl - ph;
            const float rx = D * pw.w[d].x, ry = -D * pw.w[d].y;
            y[d] = make_float2(S - ry, rx);
            if (d < 4) ret[d] = make_float2(S + ry, rx);
        }
        pair_return_lds(ret, lds, lane, y);
    }
    wave_fft512<true>(y, lds, lane, tw);
    float best = -INFINITY;
    int at = 0x7fffffff;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int i0 = 2 * lane + 128 * d;
        const float a = y[d].x * (1.0f / 1024.0f), c = y[d].y * (1.0f / 1024.0f);
        if (autocorr) *reinterpret_cast<float2 *>(autocorr + b * 512 + i0) = make_float2(a, c);
        const bool ta = (i0 > 100) & ((a > best) | ((a == best) & (i0 < at)));
        best = ta ? a : best;
        at = ta ? i0 : at;
        const bool tc = (i0 + 1 > 100) & ((c > best) | ((c == best) & (i0 + 1 < at)));
        best = tc ? c : best;
        at = tc ? i0 + 1 : at;
    }
    JDSP_ARGMAX_STEP(0xB1, 0xf)
    JDSP_ARGMAX_STEP(0x4E, 0xf)
    JDSP_ARGMAX_STEP(0x141, 0xf)
    JDSP_ARGMAX_STEP(0x140, 0xf)
    JDSP_ARGMAX_STEP(0x142, 0xa)
    JDSP_ARGMAX_STEP(0x143, 0xc)
    if (lane == 63) {
        arg[b] = at;
        rmax[b] = best;
    }
}

int launch_pitch(hipStream_t s, const short *pcm, long n_blocks, const short *prev_block, const float2 *table, int *arg,
                 float *rmax, float *autocorr)
{
    if (n_blocks <= 0) return 0;
    if (JDSP_PITCH_ONE) {
        hipLaunchKernelGGL(pitch_one_kernel, dim3((unsigned)((n_blocks + 7) / 8 * 8)), dim3(64), 0, s, pcm, n_blocks, prev_block, table,
                           arg, rmax, autocorr);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (JDSP_PITCH_RUN) {
        // one round of resident waves of a 256-CU part; never fewer than 4 blocks per wave
        const long slots = 1024L * JDSP_PITCH_WAVES;
        long waves = (n_blocks + 3) / 4;
        if (waves > slots) waves = slots;
        const long run = (n_blocks + waves - 1) / waves;
        waves = (n_blocks + run - 1) / run;
        hipLaunchKernelGGL(pitch_run_kernel, dim3((unsigned)((waves + 7) / 8 * 8)), dim3(64), 0, s, pcm, n_blocks, prev_block,
                           table, arg, rmax, autocorr, (int)run);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const long grid = (n_blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(pitch_autocorr_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, prev_block, table,
                       arg, rmax, autocorr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
