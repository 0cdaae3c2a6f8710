// mvdrn_kernels.hip -- the MVDR beamformer generalised to n_mics <= 8 microphones with a per-bin
// covariance (BASELINE config 5 / SURVEY §8f rank 3).  The reference (BeamForming_MVDR_ver1.cpp)
// has 2 microphones and ONE real 2x2 matrix for all bins -- that exact algorithm is
// mvdr_kernels.hip; this file keeps its framing, VAD, run counter and weight formula
// (w = R^-1 c / (c^H R^-1 c), :170-171) and makes R a Hermitian n_mics x n_mics matrix per bin.
// No MFMA: 8x8 systems, one per bin, eight of them per wave (a matrix row per lane), eliminated in FP64.
//
//   vad_kernel / plan_kernel (denoise_kernels.hip)   as for the 2-microphone path
//   mvdrn_event_spectra_kernel    X_m[k], k = 0..512, of every estimation frame and microphone
//   mvdrn_chunk_sums / _prefix / mvdrn_update_kernel   R_k += X X^H / N per event, the weights after each
//   mvdrn_apply_kernel            one wave per block: n_mics transforms, y = IDFT(w^H X)
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

constexpr int kMvnBins = 513;

__device__ __forceinline__ u32x4 mvn_block(const short *__restrict__ chan, long n_blocks, const short *__restrict__ prev,
                                           long j, int lane)
{
    u32x4 zero = {0u, 0u, 0u, 0u};
    if (j >= 0 && j < n_blocks) return reinterpret_cast<const u32x4 *>(chan + j * 512)[lane];
    if (j == -1 && prev) return reinterpret_cast<const u32x4 *>(prev)[lane];
    return zero;
}

__device__ __forceinline__ void mvn_spectrum(float2 (&v)[8], float2 *lds, int lane, const WaveTwiddles &tw,
                                             const float2 *wsp, float2 (&lo)[8], float2 (&hi)[8])
{
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
#define JDSP_SPLIT(J)                                                                          \
    {                                                                                          \
        const int m = 128 * J + 2 * lane;                                                      \
        const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);                          \
        float2 zr0, zr1; load_mirror_pair(lds, m, zr0, zr1);                           \
        split_fwd<J>(make_float2(zz.x, zz.y), zr0, wsp[0], lo[2 * J], hi[2 * J]);             \
        split_fwd<J>(make_float2(zz.z, zz.w), zr1, wsp[1], lo[2 * J + 1], hi[2 * J + 1]);     \
    }
    JDSP_SPLIT(0) JDSP_SPLIT(1) JDSP_SPLIT(2) JDSP_SPLIT(3)
#undef JDSP_SPLIT
    wave_lds_fence();
}

// spec[(e * n_mics + m) * 513 + k] = X_m[k] of the frame [block j-1, block j], j = events[e]
__global__ __launch_bounds__(64) void mvdrn_event_spectra_kernel(const short *__restrict__ pcm, long chan_stride,
                                                                 int n_mics, long n_blocks,
                                                                 const short *__restrict__ prev_all,
                                                                 const int *__restrict__ events,
                                                                 const DenoisePlan *__restrict__ plan,
                                                                 const float2 *__restrict__ table,
                                                                 float2 *__restrict__ spec)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage[256];
    const int lane = threadIdx.x;
    const long total = (long)plan->n_events * n_mics;
    if ((long)blockIdx.x >= total) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const int e = (int)(w / n_mics), m = (int)(w % n_mics);
        const long j = events[e];
        const short *chan = pcm + (size_t)m * chan_stride;
        const short *prev = prev_all + (size_t)m * 512;
        unsigned int raw[8];
        relayout_half(stage, lane, mvn_block(chan, n_blocks, prev, j - 1, lane), raw);
        relayout_half(stage, lane, mvn_block(chan, n_blocks, prev, j, lane), raw + 4);
        float2 v[8], lo[8], hi[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const float2 s = unpack_i16x2(raw[r]); v[r] = make_float2(0.5f * s.x, 0.5f * s.y); }
        mvn_spectrum(v, lds, lane, tw, wsp, lo, hi);
        float2 *dst = spec + (size_t)w * kMvnBins;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            dst[128 * q + 2 * lane] = lo[2 * q];
            dst[128 * q + 2 * lane + 1] = lo[2 * q + 1];
        }
        if (lane == 0) dst[512] = hi[0];
    }
}

// ---- complex FP64 helpers on (re, im) pairs
struct cd { double x, y; };
__device__ __forceinline__ cd cd_mul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd cd_sub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }

// The covariance after every estimation frame and the weights that go with it.  R_k after event e is a prefix sum
// over the events and every (bin, version) solve is independent, so the event list is cut into kMvnChunks chunks:
//   mvdrn_chunk_sums_kernel    (8 bins, chunk): the chunk's sum of X X^H / N
//   mvdrn_chunk_prefix_kernel  (bin): the matrix entering every chunk, and the one carried out of the call
//   mvdrn_update_kernel        (8 bins, chunk): walks the chunk's events from its entering matrix, one solve per event
// (One wave per bin walking the whole list took 3.6 us per event: 5.9 ms at 10 % pauses in 16,384 blocks,
// profiles/r02_denoise_events.txt.)  FP64 sums: the grouping moves nothing above 1e-16.
//
// A wave holds EIGHT bins, one matrix ROW per lane (lane = 8 * bin-in-wave + row, the row's eight entries in
// registers).  With one ENTRY per lane (a bin per wave) every lane repeated the pivot's reciprocal and the right-hand
// side's update -- 17 of the 25 FP64 operations of an elimination step -- and a step was a round of exchanges per
// bin; a row per lane shares them between eight bins and lets step p touch only the columns right of p
// (profiles/r02_mvdr_pairs.txt).
struct MvnChunks { int per_chunk, n_chunks; };
__device__ __forceinline__ MvnChunks mvn_chunks(int n_events)
{
    MvnChunks g;
    g.per_chunk = n_events > kMvnChunks ? (n_events + kMvnChunks - 1) / kMvnChunks : 1;
    g.n_chunks = (n_events + g.per_chunk - 1) / g.per_chunk;
    return g;
}

__device__ __forceinline__ cd cd_of(double2 a) { return {a.x, a.y}; }
__device__ __forceinline__ double2 d2_of(cd a) { return make_double2(a.x, a.y); }
// 1 / a with one reciprocal: hardware estimate + two Newton steps (relative error ~1e-16: the weights are stored as FP32)
__device__ __forceinline__ cd cd_inv_fast(cd a)
{
    const double d = a.x * a.x + a.y * a.y;
    double y = __builtin_amdgcn_rcp(d);
    y = y * (2.0 - d * y);
    y = y * (2.0 - d * y);
    return {a.x * y, -a.y * y};
}

// this lane's spectrum value to its group, the group's eight back: row[c] += X_r conj(X_c) / N   (N a power of two: exact)
__device__ __forceinline__ void mvn_row_add(cd (&row)[8], float2 xr, float2 *xs, int r, double inv_n)
{
    wave_lds_fence();
    xs[r] = xr;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 two = *reinterpret_cast<const float4 *>(&xs[2 * q]);
        row[2 * q].x += ((double)xr.x * two.x + (double)xr.y * two.y) * inv_n;
        row[2 * q].y += ((double)xr.y * two.x - (double)xr.x * two.y) * inv_n;
        row[2 * q + 1].x += ((double)xr.x * two.z + (double)xr.y * two.w) * inv_n;
        row[2 * q + 1].y += ((double)xr.y * two.z - (double)xr.x * two.w) * inv_n;
    }
}

__global__ __launch_bounds__(64) void mvdrn_chunk_sums_kernel(const float2 *__restrict__ spec, int n_mics, int n_bins,
                                                              double inv_n, const DenoisePlan *__restrict__ plan,
                                                              double2 *__restrict__ chunk_sum)
{
    __shared__ __attribute__((aligned(16))) float2 sx[8][8];
    const int lane = threadIdx.x, grp = lane >> 3, r = lane & 7, chunk = blockIdx.y;
    const int n_events = plan->n_events;
    const MvnChunks g = mvn_chunks(n_events);
    if (chunk >= g.n_chunks) return;
    const int kb = blockIdx.x * 8 + grp;
    const int k = kb < n_bins ? kb : n_bins - 1;                 // a group past the last bin repeats it and stores nothing
    const int e0 = chunk * g.per_chunk, e1 = e0 + g.per_chunk < n_events ? e0 + g.per_chunk : n_events;
    cd S[8];
#pragma unroll
    for (int c = 0; c < 8; c++) S[c] = {0.0, 0.0};
    float2 xr = make_float2(0.f, 0.f);
    if (r < n_mics && e0 < e1) xr = spec[((size_t)e0 * n_mics + r) * n_bins + k];
    for (int e = e0; e < e1; e++) {
        const float2 x = xr;
        if (r < n_mics && e + 1 < e1) xr = spec[((size_t)(e + 1) * n_mics + r) * n_bins + k];
        mvn_row_add(S, x, sx[grp], r, inv_n);
    }
    if (kb < n_bins) {
        double2 *dst = chunk_sum + ((size_t)chunk * n_bins + k) * 64 + 8 * r;
#pragma unroll
        for (int c = 0; c < 8; c++) dst[c] = d2_of(S[c]);
    }
}

__global__ __launch_bounds__(64) void mvdrn_chunk_prefix_kernel(const double2 *__restrict__ chunk_sum, int n_mics, int n_bins,
                                                                const DenoisePlan *__restrict__ plan,
                                                                const double2 *__restrict__ cov_in, double2 *__restrict__ cov_out,
                                                                double2 *__restrict__ chunk_start)
{
    const int k = blockIdx.x, lane = threadIdx.x;
    const MvnChunks g = mvn_chunks(plan->n_events);
    const bool live = (lane >> 3) < n_mics && (lane & 7) < n_mics;
    const double2 rin = cov_in[(size_t)k * 64 + lane];
    double2 R = make_double2(live ? rin.x : 0.0, live ? rin.y : 0.0);
    chunk_start[(size_t)k * 64 + lane] = R;                       // chunk 0 exists even without events: version 0
    for (int c0 = 0; c0 < g.n_chunks; c0 += 8) {                  // eight sums in flight: the walk is latency, not arithmetic
        double2 s[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
            s[i] = c0 + i < g.n_chunks ? chunk_sum[((size_t)(c0 + i) * n_bins + k) * 64 + lane] : make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (c0 + i >= g.n_chunks) break;
            if (c0 + i > 0) chunk_start[((size_t)(c0 + i) * n_bins + k) * 64 + lane] = R;
            R.x += s[i].x;
            R.y += s[i].y;
        }
    }
    cov_out[(size_t)k * 64 + lane] = R;
}

// After every event the weights are recomputed: Gaussian elimination of [R' | c] to a diagonal (R' = R + loading *
// tr(R) / n * I, Hermitian positive definite once loaded, so no pivoting; pivot rows are left unnormalised and each
// unknown is divided by its pivot at the end), then w = x / (c^H x).  Version 0 = the matrix carried in (chunk 0).
// Step p: the lane holding row p puts its entries right of the diagonal and its right-hand side in LDS, the group's
// other rows subtract their multiple of it.
__global__ __launch_bounds__(64) void mvdrn_update_kernel(const float2 *__restrict__ spec, int n_mics, int n_bins,
                                                          double inv_n, const DenoisePlan *__restrict__ plan,
                                                          const double2 *__restrict__ chunk_start,
                                                          const double2 *__restrict__ steer, double loading,
                                                          float2 *__restrict__ weights)
{
    __shared__ __attribute__((aligned(16))) float2 sx[8][8];       // the event's spectra, [bin in wave][microphone]
    __shared__ __attribute__((aligned(16))) double2 srow[8][9];    // the pivot row and its right-hand side
    __shared__ __attribute__((aligned(16))) double2 sc[8][8];      // steering vectors
    __shared__ __attribute__((aligned(16))) double2 sv[8][8];      // diagonals, then the unknowns
    const int lane = threadIdx.x, grp = lane >> 3, r = lane & 7, chunk = blockIdx.y;
    const int n_events = plan->n_events;
    const MvnChunks g = mvn_chunks(n_events);
    if (chunk > 0 && chunk >= g.n_chunks) return;
    const int kb = blockIdx.x * 8 + grp;
    const int k = kb < n_bins ? kb : n_bins - 1;
    const bool row_live = r < n_mics;
    cd R[8];
    {
        const double2 *src = chunk_start + ((size_t)chunk * n_bins + k) * 64 + 8 * r;
#pragma unroll
        for (int c = 0; c < 8; c++) R[c] = cd_of(src[c]);
    }
    const double2 sr = steer[(size_t)k * 8 + r];
    const cd cr = {row_live ? sr.x : 0.0, row_live ? sr.y : 0.0};
    sc[grp][r] = d2_of(cr);
    const int e0 = chunk * g.per_chunk;
    const int e1 = e0 + g.per_chunk < n_events ? e0 + g.per_chunk : n_events;
    float2 xr = make_float2(0.f, 0.f);
    if (row_live && e0 < e1) xr = spec[((size_t)e0 * n_mics + r) * n_bins + k];     // event e0 is the first this chunk adds
    for (int v = chunk == 0 ? 0 : e0 + 1; v <= e1; v++) {
        if (v > e0) mvn_row_add(R, xr, sx[grp], r, inv_n);
        if (row_live && v < e1) xr = spec[((size_t)v * n_mics + r) * n_bins + k];   // event v is the one version v + 1 adds
        double dgx = R[0].x;                                   // the diagonal entry (real) of this lane's row
#pragma unroll
        for (int c = 1; c < 8; c++) dgx = r == c ? R[c].x : dgx;
        wave_lds_fence();
        sv[grp][r] = make_double2(dgx, 0.0);
        wave_lds_fence();
        double tr = 0.0;
        for (int d = 0; d < n_mics; d++) tr += sv[grp][d].x;
        const double load = loading * tr / n_mics;
        cd A[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            A[c].x = r == c ? (row_live ? R[c].x + load : 1.0) : R[c].x;
            A[c].y = r == c && !row_live ? 0.0 : R[c].y;
        }
        cd b = cr, my_inv = {1.0, 0.0};
#pragma unroll
        for (int p = 0; p < 8; p++) {
            if (p < n_mics) {
                wave_lds_fence();
                if (r == p) {
#pragma unroll
                    for (int c = p; c < 8; c++) srow[grp][c] = d2_of(A[c]);
                    srow[grp][8] = d2_of(b);
                }
                wave_lds_fence();
                const cd inv = cd_inv_fast(cd_of(srow[grp][p]));
                cd f = cd_mul(A[p], inv);                       // this row's multiple of the pivot row
                my_inv.x = r == p ? inv.x : my_inv.x;
                my_inv.y = r == p ? inv.y : my_inv.y;
                f.x = r == p ? 0.0 : f.x;
                f.y = r == p ? 0.0 : f.y;
#pragma unroll
                for (int c = p + 1; c < 8; c++) A[c] = cd_sub(A[c], cd_mul(f, cd_of(srow[grp][c])));
                b = cd_sub(b, cd_mul(f, cd_of(srow[grp][8])));
            }
        }
        const cd x = cd_mul(b, my_inv);
        wave_lds_fence();
        sv[grp][r] = d2_of(x);
        wave_lds_fence();
        cd den = {0.0, 0.0};                                   // c^H x
        for (int d = 0; d < n_mics; d++) {
            const cd xd = cd_of(sv[grp][d]);
            const cd cdv = cd_of(sc[grp][d]);
            den.x += cdv.x * xd.x + cdv.y * xd.y;
            den.y += cdv.x * xd.y - cdv.y * xd.x;
        }
        const cd w = cd_mul(x, cd_inv_fast(den));
        // [version][microphone][bin]: the apply kernel reads one microphone's weights for consecutive bins
        if (row_live && kb < n_bins) weights[((size_t)v * 8 + r) * n_bins + k] = make_float2((float)w.x, (float)w.y);
    }
}

__global__ __launch_bounds__(64) void mvdrn_apply_kernel(const short *__restrict__ pcm, long chan_stride, int n_mics,
                                                         long n_blocks, long calls_before,
                                                         const short *__restrict__ prev_in, short *__restrict__ prev_out,
                                                         const int *__restrict__ ver_base,
                                                         const unsigned long long *__restrict__ snap_mask,
                                                         const float2 *__restrict__ weights,
                                                         const float2 *__restrict__ table, short *__restrict__ out,
                                                         float *__restrict__ precast)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage32[520];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long j = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (j >= n_blocks) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    const bool have_prev = calls_before + j > 0;
    const float2 *W = weights + (size_t)version_of(ver_base, snap_mask, j) * kMvnBins * 8;

    float2 ylo[8], yhi[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { ylo[q] = make_float2(0.f, 0.f); yhi[q] = make_float2(0.f, 0.f); }
    for (int m = 0; m < n_mics; m++) {
        const short *chan = pcm + (size_t)m * chan_stride;
        const short *prev = prev_in + (size_t)m * 512;
        float2 v[8], lo[8], hi[8];
        mvdr_frame_pairs(stage32, lane, mvn_block(chan, n_blocks, prev, have_prev ? j - 1 : -2, lane),
                         mvn_block(chan, n_blocks, prev, j, lane), v, 0.5f);
        mvn_spectrum(v, lds, lane, tw, wsp, lo, hi);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int bin = 128 * (q >> 1) + 2 * lane + (q & 1);            // 0..511; its partner is bin + 512
            // Y[k] = sum_m conj(w_k[m]) X_m[k];  for k > 512, w_k = conj(w_{1024-k})
            const float2 *Wm = W + (size_t)m * kMvnBins;
            const float2 wl = Wm[bin];
            ylo[q].x += wl.x * lo[q].x + wl.y * lo[q].y;
            ylo[q].y += wl.x * lo[q].y - wl.y * lo[q].x;
            const float2 wh = Wm[512 - bin];                                // bin + 512 mirrors to 512 - bin
            if (bin == 0) {                                                 // k = 512 itself: not mirrored
                yhi[q].x += wh.x * hi[q].x + wh.y * hi[q].y;
                yhi[q].y += wh.x * hi[q].y - wh.y * hi[q].x;
            } else {
                yhi[q] = cadd(yhi[q], cmul(wh, hi[q]));
            }
        }
        if (j == n_blocks - 1)
            reinterpret_cast<u32x4 *>(prev_out + (size_t)m * 512)[lane] = reinterpret_cast<const u32x4 *>(chan + j * 512)[lane];
    }
    float2 z[8];
    z[0] = presplit_inv<0>(ylo[0], yhi[0], wsp[0]); z[1] = presplit_inv<0>(ylo[1], yhi[1], wsp[1]);
    z[2] = presplit_inv<1>(ylo[2], yhi[2], wsp[0]); z[3] = presplit_inv<1>(ylo[3], yhi[3], wsp[1]);
    z[4] = presplit_inv<2>(ylo[4], yhi[4], wsp[0]); z[5] = presplit_inv<2>(ylo[5], yhi[5], wsp[1]);
    z[6] = presplit_inv<3>(ylo[6], yhi[6], wsp[0]); z[7] = presplit_inv<3>(ylo[7], yhi[7], wsp[1]);
#pragma unroll
    for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4 *>(&lds[128 * q + 2 * lane]) = make_float4(z[2 * q].x, z[2 * q].y, z[2 * q + 1].x, z[2 * q + 1].y);
    wave_lds_fence();
    float2 y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, tw);
    const long first_emit = calls_before >= 1 ? 0 : 1;
    if (j >= first_emit) {
        short *o = out + (j - first_emit) * 512;
        float *pc = precast ? precast + (j - first_emit) * 512 : nullptr;
#pragma unroll
        for (int dd = 0; dd < 8; dd++) {
            const int i0 = 2 * lane + 128 * dd - 511;
            const float s0 = y[dd].x * (1.0f / 1024.0f), s1 = y[dd].y * (1.0f / 1024.0f);
            if (i0 >= 0 && i0 < 512) { o[i0] = (short)cast_i16_bits(s0); if (pc) pc[i0] = s0; }
            if (i0 + 1 >= 0 && i0 + 1 < 512) { o[i0 + 1] = (short)cast_i16_bits(s1); if (pc) pc[i0 + 1] = s1; }
        }
    }
}

// The same with pair-owned bins (frame_io.h): lane l works on the bins m, m + 512 of m = l + 64 d, d < 5.  Y[k] =
// sum_m conj(w_k[m]) X_m[k] with w_k = conj(w_{1024-k}) above 512 is Hermitian when the X_m are, so the 513 bins the
// five items cover are all of it: ten weights and ten products per lane and microphone instead of sixteen, and the
// inverse transform's mirrored inputs come from presplit_inv_pair.
#ifndef JDSP_MVN_APPLY_PAIRS
#define JDSP_MVN_APPLY_PAIRS 1
#endif
__global__ __launch_bounds__(64) void mvdrn_apply_pairs_kernel(const short *__restrict__ pcm, long chan_stride, int n_mics,
                                                               long n_blocks, long calls_before,
                                                               const short *__restrict__ prev_in, short *__restrict__ prev_out,
                                                               const int *__restrict__ ver_base,
                                                               const unsigned long long *__restrict__ snap_mask,
                                                               const float2 *__restrict__ weights,
                                                               const float2 *__restrict__ table, short *__restrict__ out,
                                                               float *__restrict__ precast)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage32[528];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long j = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (j >= n_blocks) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    PairTwiddles pw;
    load_pair_twiddles(pw, table, lane);
    const bool have_prev = calls_before + j > 0;
    const float2 *W = weights + (size_t)version_of(ver_base, snap_mask, j) * kMvnBins * 8;
    const float nyq = lane == 0 ? -1.f : 1.f;                      // k = 512 (lane 0, d = 0) is its own mirror: conj(w) there

    float2 ylo[5], yhi[5];
#pragma unroll
    for (int d = 0; d < 5; d++) { ylo[d] = make_float2(0.f, 0.f); yhi[d] = make_float2(0.f, 0.f); }
    // a microphone's two blocks are requested while the one before it is transformed (the LDS fences would hold the loads back)
    const long jp = have_prev ? j - 1 : -2;
    u32x4 nxt_prev = mvn_block(pcm, n_blocks, prev_in, jp, lane), nxt_cur = mvn_block(pcm, n_blocks, prev_in, j, lane);
    for (int m = 0; m < n_mics; m++) {
        const short *chan = pcm + (size_t)m * chan_stride;
        const float2 *Wm = W + (size_t)m * kMvnBins;
        float2 wl[5], wh[5];
#pragma unroll
        for (int d = 0; d < 5; d++) { wl[d] = Wm[lane + 64 * d]; wh[d] = Wm[512 - lane - 64 * d]; }
        const u32x4 img_prev = nxt_prev, img_cur = nxt_cur;
        if (m + 1 < n_mics) {
            nxt_prev = mvn_block(chan + chan_stride, n_blocks, prev_in + (size_t)(m + 1) * 512, jp, lane);
            nxt_cur = mvn_block(chan + chan_stride, n_blocks, prev_in + (size_t)(m + 1) * 512, j, lane);
        }
        float2 v[8], zr[5];
        mvdr_frame_pairs(stage32, lane, img_prev, img_cur, v, 0.5f);
        wave_fft512<false>(v, lds, lane, tw);
        wave_lds_fence();
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
            const float2 t = cmul(pw.w[d], o);
            const float2 lo = cadd(e, t), hi = csub(e, t);         // X[m], X[m + 512]
            ylo[d].x += wl[d].x * lo.x + wl[d].y * lo.y;           // conj(w_m) X[m]
            ylo[d].y += wl[d].x * lo.y - wl[d].y * lo.x;
            const float why = d == 0 ? nyq * wh[d].y : wh[d].y;
            yhi[d].x += wh[d].x * hi.x - why * hi.y;               // w_{512-m} X[m + 512]
            yhi[d].y += wh[d].x * hi.y + why * hi.x;
        }
        if (j == n_blocks - 1)
            reinterpret_cast<u32x4 *>(prev_out + (size_t)m * 512)[lane] = reinterpret_cast<const u32x4 *>(chan + j * 512)[lane];
    }
    float2 y[8], ret[4];
#pragma unroll
    for (int d = 0; d < 5; d++) {
        if (d < 4) presplit_inv_pair(ylo[d], yhi[d], pw.w[d], y[d], ret[d]);
        else y[d] = presplit_inv_reg(ylo[d], yhi[d], pw.w[d]);
    }
    pair_return_lds(ret, lds, lane, y);
    wave_fft512<true>(y, lds, lane, tw);
    const long first_emit = calls_before >= 1 ? 0 : 1;
    if (j >= first_emit) {
        short *o = out + (j - first_emit) * 512;
        float *pc = precast ? precast + (j - first_emit) * 512 : nullptr;
#pragma unroll
        for (int dd = 0; dd < 8; dd++) {
            const int i0 = 2 * lane + 128 * dd - 511;
            const float s0 = y[dd].x * (1.0f / 1024.0f), s1 = y[dd].y * (1.0f / 1024.0f);
            if (i0 >= 0 && i0 < 512) { o[i0] = (short)cast_i16_bits(s0); if (pc) pc[i0] = s0; }
            if (i0 + 1 >= 0 && i0 + 1 < 512) { o[i0 + 1] = (short)cast_i16_bits(s1); if (pc) pc[i0 + 1] = s1; }
        }
    }
}

// chunk_ws: 2 * chunk_cap * n_bins * 64 double2 (the chunk sums, then the matrices entering the chunks), [chunk][bin][64];
// chunk_cap = min(blocks the workspace is sized for, kMvnChunks) >= the chunks any call of that size can have
static void launch_mvdrn_update(hipStream_t s, const float2 *spec, int n_mics, int n_bins, double inv_n, const DenoisePlan *plan,
                                const double2 *cov_in, double2 *cov_out, double2 *chunk_ws, int chunk_cap, const double2 *steer,
                                double loading, float2 *weights)
{
    double2 *sums = chunk_ws, *start = chunk_ws + (size_t)chunk_cap * n_bins * 64;
    hipLaunchKernelGGL(mvdrn_chunk_sums_kernel, dim3((n_bins + 7) / 8, kMvnChunks), dim3(64), 0, s, spec, n_mics, n_bins, inv_n, plan, sums);
    hipLaunchKernelGGL(mvdrn_chunk_prefix_kernel, dim3(n_bins), dim3(64), 0, s, sums, n_mics, n_bins, plan, cov_in, cov_out, start);
    hipLaunchKernelGGL(mvdrn_update_kernel, dim3((n_bins + 7) / 8, kMvnChunks), dim3(64), 0, s, spec, n_mics, n_bins, inv_n, plan, start,
                       steer, loading, weights);
}

int launch_mvdrn(hipStream_t s, const short *pcm, long chan_stride, int n_mics, long n_blocks, long calls_before,
                 const short *prev_in, short *prev_out, const int *events, const DenoisePlan *plan, const int *ver_base,
                 const unsigned long long *snap_mask, float2 *spec, const double2 *cov_in, double2 *cov_out,
                 const double2 *steer, double loading, float2 *weights, const float2 *table, short *out, float *precast,
                 double2 *chunk_ws, int chunk_cap)
{
    if (n_blocks <= 0) return 0;
    const long g1 = n_blocks * n_mics < 4096 ? n_blocks * n_mics : 4096;
    hipLaunchKernelGGL(mvdrn_event_spectra_kernel, dim3((unsigned)g1), dim3(64), 0, s, pcm, chan_stride, n_mics, n_blocks,
                       prev_in, events, plan, table, spec);
    launch_mvdrn_update(s, spec, n_mics, kMvnBins, 1.0 / 1024.0, plan, cov_in, cov_out, chunk_ws, chunk_cap, steer, loading, weights);
    const long grid = (n_blocks + 7) / 8 * 8;
#if JDSP_MVN_APPLY_PAIRS
    hipLaunchKernelGGL(mvdrn_apply_pairs_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, chan_stride, n_mics, n_blocks,
                       calls_before, prev_in, prev_out, ver_base, snap_mask, weights, table, out, precast);
#else
    hipLaunchKernelGGL(mvdrn_apply_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, chan_stride, n_mics, n_blocks,
                       calls_before, prev_in, prev_out, ver_base, snap_mask, weights, table, out, precast);
#endif
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =======================================================================================
// FFT_PROCESSING_LEN 512 (BASELINE config 5 as worded: "8-mic array, 512-pt STFT"): blocks of 256 samples, KEEP_LEN
// 255, frames [first 255 samples of the previous block, block, 0], 257 bins, R_k += X X^H / 512, samples 255..510 out.
// A 512-sample real frame is half a wave transform, so frames ride it in PAIRS (z = a + j b, see denoise512_kernel):
// two MICROPHONES per forward transform, two consecutive BLOCKS per inverse transform.
constexpr int kMvn512Bins = 257;

// block j of one microphone, 256 samples: sample i; j == -1 is the previous call's last block, else silence
__device__ __forceinline__ float mvn512_sample(const short *__restrict__ chan, long n_blocks,
                                               const short *__restrict__ prev, long j, int i)
{
    if (j >= 0 && j < n_blocks) return (float)chan[j * 256 + i];
    if (j == -1 && prev) return (float)prev[i];
    return 0.f;
}

// Zh = FFT512(0.5 (a + j b)) as a natural-order image, then A[k], B[k] for this lane's bins k = lane + 64 q and 256
__device__ __forceinline__ void mvn512_pair_spectra(const float (&xa)[8], const float (&xb)[8], const WaveTwiddles &tw,
                                                    float2 *lds, int lane, float2 (&A)[5], float2 (&B)[5])
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = make_float2(0.5f * xa[r], 0.5f * xb[r]);
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int k = lane + 64 * q;
        const float2 zk = lds[k], zm = lds[512 - k];
        A[q] = cadd_conj(zk, zm);
        B[q] = csub_conj_mj(zk, zm);
    }
    const float2 z = lds[256];
    A[4] = make_float2(2.f * z.x, 0.f);
    B[4] = make_float2(2.f * z.y, 0.f);
    wave_lds_fence();
}

// spec[(e * n_mics + m) * 257 + k] = X_m[k] of the frame [block j-1, block j], j = events[e]; microphones 2p, 2p+1 per wave
__global__ __launch_bounds__(64) void mvdrn512_event_spectra_kernel(const short *__restrict__ pcm, long chan_stride,
                                                                    int n_mics, long n_blocks,
                                                                    const short *__restrict__ prev_all,
                                                                    const int *__restrict__ events,
                                                                    const DenoisePlan *__restrict__ plan,
                                                                    const float2 *__restrict__ table,
                                                                    float2 *__restrict__ spec)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const int n_pairs = (n_mics + 1) >> 1;
    const long total = (long)plan->n_events * n_pairs;
    if ((long)blockIdx.x >= total) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const int e = (int)(w / n_pairs), m0 = 2 * (int)(w % n_pairs), m1 = m0 + 1;
        const long j = events[e];
        float xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int pos = lane + 64 * r;                                   // 0..255: block j-1, 256..511: block j
            const long jb = pos < 256 ? j - 1 : j;
            const int i = pos & 255;
            xa[r] = mvn512_sample(pcm + (size_t)m0 * chan_stride, n_blocks, prev_all + (size_t)m0 * 512, jb, i);
            xb[r] = m1 < n_mics ? mvn512_sample(pcm + (size_t)m1 * chan_stride, n_blocks, prev_all + (size_t)m1 * 512, jb, i) : 0.f;
        }
        float2 A[5], B[5];
        mvn512_pair_spectra(xa, xb, tw, lds, lane, A, B);
        float2 *da = spec + ((size_t)e * n_mics + m0) * kMvn512Bins, *db = da + kMvn512Bins;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            da[lane + 64 * q] = A[q];
            if (m1 < n_mics) db[lane + 64 * q] = B[q];
        }
        if (lane == 0) {
            da[256] = A[4];
            if (m1 < n_mics) db[256] = B[4];
        }
    }
}

// One wave = two consecutive blocks (j, j+1): per block ceil(n_mics / 2) forward transforms, Y_t[k] = sum_m conj(w_k[m]) X_m[k]
// for k <= 256; then ONE inverse transform of Y_0 + j Y_1 (both Hermitian) gives the two output frames.
#ifndef JDSP_MVN512_EARLY_WEIGHTS
#define JDSP_MVN512_EARLY_WEIGHTS 1
#endif
#ifndef JDSP_MVN512_APPLY_WAVES
#define JDSP_MVN512_APPLY_WAVES 3
#endif
__global__ __launch_bounds__(64, JDSP_MVN512_APPLY_WAVES) void mvdrn512_apply_kernel(const short *__restrict__ pcm, long chan_stride, int n_mics,
                                                            long n_blocks, long calls_before,
                                                            const short *__restrict__ prev_in, short *__restrict__ prev_out,
                                                            const int *__restrict__ ver_base,
                                                            const unsigned long long *__restrict__ snap_mask,
                                                            const float2 *__restrict__ weights,
                                                            const float2 *__restrict__ table, short *__restrict__ out,
                                                            float *__restrict__ precast)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long j0 = 2 * ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3));
    if (j0 >= n_blocks) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float2 Y[2][5];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < 5; q++) Y[t][q] = make_float2(0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const long j = j0 + t;
        if (j >= n_blocks) break;
        const bool have_prev = calls_before + j > 0;
        const float2 *W = weights + (size_t)version_of(ver_base, snap_mask, j) * kMvn512Bins * 8;
        // The frame's samples of microphones m0, m0 + 1 as raw halfwords.  They are requested one microphone pair ahead
        // and converted only where they are used: converted next to the loads, the s_waitcnt sits there too, and the
        // guarded weight loads below each waited for themselves -- about six memory round trips in series per pair and
        // 48 per wave, which was most of this kernel's time (ISA before: 30 s_waitcnt vmcnt(0) in this loop).
        auto fetch = [&](int m0, int (&ra)[8], int (&rb)[8]) {
            const int m1 = m0 + 1 < n_mics ? m0 + 1 : m0;
            const short *ca = pcm + (size_t)m0 * chan_stride, *cb = pcm + (size_t)m1 * chan_stride;
            if (j >= 1) {
                // both blocks inside this call's buffer (wave-uniform): frame position pos is stream sample
                // (j - 1) 256 + pos, one further from 255 on (the previous block's last sample is not in the frame),
                // and 511 is the zero -- one load per sample, no per-lane branches
                const short *fa = ca + (j - 1) * 256 + lane, *fb = cb + (j - 1) * 256 + lane;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int pos = lane + 64 * r;
                    const int off = 64 * r + (pos >= 255 ? 1 : 0) - (pos == 511 ? 1 : 0);      // (keeps the last lane in bounds)
                    ra[r] = fa[off];
                    rb[r] = fb[off];
                }
            } else {
                const short *pa = prev_in + (size_t)m0 * 512, *pb = prev_in + (size_t)m1 * 512;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int pos = lane + 64 * r;        // frame position: < 255 previous block, 255..510 this block, 511 zero
                    float a = 0.f, b = 0.f;
                    if (pos < 255) {
                        if (have_prev) { a = mvn512_sample(ca, n_blocks, pa, j - 1, pos); b = mvn512_sample(cb, n_blocks, pb, j - 1, pos); }
                    } else if (pos < 511) {
                        a = (float)ca[j * 256 + pos - 255];
                        b = (float)cb[j * 256 + pos - 255];
                    }
                    ra[r] = (int)a;
                    rb[r] = (int)b;
                }
            }
        };
        int ra[8], rb[8];
        fetch(0, ra, rb);
        for (int m0 = 0; m0 < n_mics; m0 += 2) {
            const int m1 = m0 + 1;
            const bool two = m1 < n_mics;
            // this pair's weights, unguarded (a lone last microphone reads its own row twice: its B spectrum is zero)
            const float2 *Wa = W + (size_t)m0 * kMvn512Bins, *Wb = W + (size_t)(two ? m1 : m0) * kMvn512Bins;
            float2 wa[5], wb[5];
#if JDSP_MVN512_EARLY_WEIGHTS
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int k = q < 4 ? lane + 64 * q : 256;
                wa[q] = Wa[k];
                wb[q] = Wb[k];
            }
#endif
            float xa[8], xb[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int pos = lane + 64 * r;
                xa[r] = pos < 511 ? (float)ra[r] : 0.f;
                xb[r] = (pos < 511 && two) ? (float)rb[r] : 0.f;
            }
            if (m0 + 2 < n_mics) fetch(m0 + 2, ra, rb);
            float2 A[5], B[5];
            mvn512_pair_spectra(xa, xb, tw, lds, lane, A, B);
#if !JDSP_MVN512_EARLY_WEIGHTS
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const int k = q < 4 ? lane + 64 * q : 256;
                wa[q] = Wa[k];
                wb[q] = Wb[k];
            }
#endif
#pragma unroll
            for (int q = 0; q < 5; q++) {
                Y[t][q].x += wa[q].x * A[q].x + wa[q].y * A[q].y;           // conj(w) X
                Y[t][q].y += wa[q].x * A[q].y - wa[q].y * A[q].x;
                if (two) {                                                   // (wave-uniform)
                    Y[t][q].x += wb[q].x * B[q].x + wb[q].y * B[q].y;
                    Y[t][q].y += wb[q].x * B[q].y - wb[q].y * B[q].x;
                }
            }
            if (j == n_blocks - 1) {                                         // the previous block of the next call
                const short *ca = pcm + (size_t)m0 * chan_stride, *cb = pcm + (size_t)(two ? m1 : m0) * chan_stride;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    prev_out[(size_t)m0 * 512 + lane + 64 * d] = ca[j * 256 + lane + 64 * d];
                    if (two) prev_out[(size_t)m1 * 512 + lane + 64 * d] = cb[j * 256 + lane + 64 * d];
                }
            }
        }
    }
    // A block whose weights are not finite (R_k singular: before enough estimation frames, like the reference's output
    // before its first estimate) has a NaN spectrum and therefore an all-NaN frame -- but it must not poison the block
    // it shares the inverse transform with: its spectrum goes in as zero and its samples come out as NaN.
    bool bad[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 5; q++) acc += Y[t][q].x * 0.f + Y[t][q].y * 0.f;       // NaN iff any part is NaN or inf
        bad[t] = __ballot(acc != acc) != 0ull;
        if (bad[t]) {
#pragma unroll
            for (int q = 0; q < 5; q++) Y[t][q] = make_float2(0.f, 0.f);
        }
    }
    // packed inverse: bin k holds Y0 + j Y1, bin 512-k its Hermitian partner conj(Y0) + j conj(Y1) = conj(Y0 - j Y1)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int k = lane + 64 * q;
        const float2 m = cadd_mj(Y[0][q], Y[1][q]);
        lds[512 - k] = make_float2(m.x, -m.y);                               // k = 0 lands in the spare slot 512
        lds[k] = cadd_pj(Y[0][q], Y[1][q]);
    }
    if (lane == 0) lds[256] = cadd_pj(make_float2(Y[0][4].x, 0.f), make_float2(Y[1][4].x, 0.f));   // the Nyquist bin is real
    wave_lds_fence();
    float2 y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, tw);
    const long first_emit = calls_before >= 1 ? 0 : 1;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const long j = j0 + t;
        if (j >= n_blocks || j < first_emit) continue;
        short *o = out + (j - first_emit) * 256;
        float *pc = precast ? precast + (j - first_emit) * 256 : nullptr;
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const int i = lane + 64 * d - 255;                               // samples 255..510 of the frame
            const float sv = bad[t] ? __builtin_nanf("") : (t ? y[d].y : y[d].x) * (1.0f / 512.0f);
            if (i >= 0 && i < 256) { o[i] = (short)cast_i16_bits(sv); if (pc) pc[i] = sv; }
        }
    }
}

int launch_mvdrn512(hipStream_t s, const short *pcm, long chan_stride, int n_mics, long n_blocks, long calls_before,
                    const short *prev_in, short *prev_out, const int *events, const DenoisePlan *plan, const int *ver_base,
                    const unsigned long long *snap_mask, float2 *spec, const double2 *cov_in, double2 *cov_out,
                    const double2 *steer, double loading, float2 *weights, const float2 *table, short *out, float *precast,
                    double2 *chunk_ws, int chunk_cap)
{
    if (n_blocks <= 0) return 0;
    const long work = n_blocks * ((n_mics + 1) / 2);
    const long g1 = work < 4096 ? work : 4096;
    hipLaunchKernelGGL(mvdrn512_event_spectra_kernel, dim3((unsigned)g1), dim3(64), 0, s, pcm, chan_stride, n_mics, n_blocks,
                       prev_in, events, plan, table, spec);
    launch_mvdrn_update(s, spec, n_mics, kMvn512Bins, 1.0 / 512.0, plan, cov_in, cov_out, chunk_ws, chunk_cap, steer, loading, weights);
    const long grid = ((n_blocks + 1) / 2 + 7) / 8 * 8;
    hipLaunchKernelGGL(mvdrn512_apply_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, chan_stride, n_mics, n_blocks,
                       calls_before, prev_in, prev_out, ver_base, snap_mask, weights, table, out, precast);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
