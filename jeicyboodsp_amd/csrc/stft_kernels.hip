// stft_kernels.hip -- batched STFT analysis for 1024-point frames (gfx950).
//
// Work done per frame (reference: SpectralSubtraction_final.cpp:218-230,
// WienerFilter_final.cpp:181-193): int16 -> double, * Hamming, 1024-point
// forward c2c DFT of the real frame, all 1024 bins kept.
//
// Mapping: one wavefront owns K consecutive frames (K = 2 by default).  A frame
// is a 512-point complex FFT in the wave's registers/LDS (wave_fft512.h) plus a
// split step; the 8 KB spectrum leaves as eight fully coalesced 1 KB
// nontemporal wave stores.  HBM-bound: 1 KB in + 8 KB out per frame.
//
// What the measurements on MI355X said (profiles/r01_*; DESIGN.md "STFT kernel"):
//   * nontemporal STORES + plain LOADS: 92 us; plain stores 129 us; nontemporal
//     loads 126-155 us (the half-frame two neighbouring waves share stops hitting L2).
//   * short waves that issue their stores and end beat a persistent wave that
//     loops with a prefetch (124 us): vmcnt retires loads and stores in issue
//     order, so a looping wave that waits for its next input also waits for its
//     previous spectrum to be acknowledged by HBM.
//   * XCD-aware block -> chunk mapping (blocks b and b+8 share an XCD's L2):
//     92 -> 88 us, and PCM over-fetch disappears.
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

// ---- build-time variants (tools/tune_stft.py A/Bs them on the GPU) -------------------
#ifndef JDSP_STFT_K
#define JDSP_STFT_K 2             // consecutive frames per wavefront ("stft.frames_per_wave" = 0 picks this)
#endif
#ifndef JDSP_STFT_MINWAVES
#define JDSP_STFT_MINWAVES 4      // __launch_bounds__ 2nd arg: waves per SIMD the register budget must allow
#endif
#ifndef JDSP_STFT_XCD
#define JDSP_STFT_XCD 1           // 1: XCD-aware chunk -> block mapping
#endif
#ifndef JDSP_STFT_NT_LOAD
#define JDSP_STFT_NT_LOAD 0       // 1: nontemporal PCM loads (slower: see above)
#endif
#ifndef JDSP_STFT_NT
#define JDSP_STFT_NT 1            // 1: nontemporal spectrum stores
#endif
#ifndef JDSP_STFT_DIRECT
#define JDSP_STFT_DIRECT 1        // 1: four dword loads per lane per half-frame, already in the transform's layout
#endif                            //    (no LDS relayout; same bytes through the TA, about 1 us faster per launch);
                                  // 0: one dwordx4 per lane + a 1 KB LDS relayout
#ifndef JDSP_STFT_ABLATE
#define JDSP_STFT_ABLATE 0        // timing-only ablations (1: no stores, 2: no transform); never shipped
#endif

__device__ __forceinline__ void store_spec(float4 *p, float4 v)
{
#if JDSP_STFT_NT
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(p));
#else
    *p = v;
#endif
}

// Spectrum of one frame from the natural-order image of Zh in LDS.
template <int J>
__device__ __forceinline__ void split_store_j(const float2 *lds, int lane, const float2 *wsp, float4 *dst)
{
    const int m = 128 * J + 2 * lane;
    const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
    float2 zr0, zr1;
    load_mirror_pair(lds, m, zr0, zr1);
    float2 lo0, hi0, lo1, hi1;
    split_fwd<J>(make_float2(zz.x, zz.y), zr0, wsp[0], lo0, hi0);
    split_fwd<J>(make_float2(zz.z, zz.w), zr1, wsp[1], lo1, hi1);
    store_spec(dst + 64 * J, make_float4(lo0.x, lo0.y, lo1.x, lo1.y));
    store_spec(dst + 64 * J + 256, make_float4(hi0.x, hi0.y, hi1.x, hi1.y));
}

__device__ __forceinline__ void split_and_store(const float2 *lds, int lane, const float2 *wsp, float2 *out_frame)
{
    float4 *dst = reinterpret_cast<float4 *>(out_frame) + lane;
    split_store_j<0>(lds, lane, wsp, dst);
    split_store_j<1>(lds, lane, wsp, dst);
    split_store_j<2>(lds, lane, wsp, dst);
    split_store_j<3>(lds, lane, wsp, dst);
}

// hop == 512.  One wave owns K consecutive frames = K+1 half-frames of 512 samples; ALL
// loads are issued up front (four coalesced dword loads per lane per half-frame, each lane fetching the
// sample pairs 2 lane + 128 r it transforms), see the header comment.
template <int K>
__global__ __launch_bounds__(64, JDSP_STFT_MINWAVES) void stft1024_hop512_kernel(
    const short *__restrict__ pcm, float2 *__restrict__ spec, long n_frames, const float2 *__restrict__ table)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage[256];
    const int lane = threadIdx.x;
#if JDSP_STFT_XCD
    // Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
    // chunks so the half-frame two neighbouring waves both need is served by one L2.
    // Placement only changes speed, never results.
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long f0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * K;
#else
    const long f0 = (long)blockIdx.x * K;
#endif
    if (f0 >= n_frames) return;

    // half-frame h holds samples [512 h, 512 h + 512); the stream has n_frames + 1 of them
#if JDSP_STFT_DIRECT
    const unsigned int *pcm32 = reinterpret_cast<const unsigned int *>(pcm) + lane;   // sample pair per lane
    unsigned int half[K + 1][4];
#pragma unroll
    for (int h = 0; h <= K; h++) {
        const long hh = f0 + h <= n_frames ? f0 + h : n_frames;            // clamp: stays in bounds
#pragma unroll
        for (int q = 0; q < 4; q++) half[h][q] = pcm32[hh * 256 + 64 * q];
    }
#else
    const u32x4 *pcm128 = reinterpret_cast<const u32x4 *>(pcm) + lane;     // 8 samples per lane
    u32x4 half[K + 1];
#pragma unroll
    for (int h = 0; h <= K; h++) {
        const long hh = f0 + h <= n_frames ? f0 + h : n_frames;            // clamp: stays in bounds
#if JDSP_STFT_NT_LOAD
        half[h] = __builtin_nontemporal_load(pcm128 + hh * 64);
#else
        half[h] = pcm128[hh * 64];
#endif
    }
#endif

    FrameTables t;
    load_frame_tables(t, table, lane);

    // raw[r]: the int16 pair (2*lane + 128*r, +1) of the frame.  The second half of frame f
    // is the first half of f+1 and sits in the same lane, so raw[4..7] slide down to raw[0..3].
    unsigned int raw[8];
#if JDSP_STFT_DIRECT
#pragma unroll
    for (int q = 0; q < 4; q++) raw[4 + q] = half[0][q];
#else
    relayout_half(stage, lane, half[0], raw + 4);
#endif

#pragma unroll
    for (int i = 0; i < K; i++) {
        const long f = f0 + i;
        if (f >= n_frames) break;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) raw[r] = raw[r + 4];
#if JDSP_STFT_DIRECT
#pragma unroll
        for (int q = 0; q < 4; q++) raw[4 + q] = half[i + 1][q];
#else
        relayout_half(stage, lane, half[i + 1], raw + 4);
#endif
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 s = unpack_i16x2(raw[r]);
            v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
        }
#if JDSP_STFT_ABLATE == 2   /* timing-only build: no transform, stores only */
        {
            float4 *dst = reinterpret_cast<float4 *>(spec + f * 1024) + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) store_spec(dst + 64 * j, make_float4(v[j].x, v[j].y, v[j].x, v[j].y));
            continue;
        }
#endif
        wave_fft512<false>(v, lds, lane, t.tw);

        // natural-order image of Zh, then the split reads Zh[m] and Zh[512-m]
        store_natural_image(lds, lane, v);
        wave_lds_fence();
#if JDSP_STFT_ABLATE == 1   /* timing-only build: transform kept alive, stores never taken */
        if (n_frames < 0)
#endif
        split_and_store(lds, lane, t.wsp, spec + f * 1024);
        wave_lds_fence();
    }
}

// ---- half spectrum: bins 0..512 only (the other 511 are their conjugates) -------------------------
// [frame][513] complex64, 4,104 B per frame, so odd rows start 8 bytes off a 16-byte boundary.  Every
// store is still a full aligned 16 bytes: on even rows a lane writes the bins (m, m+1), m = 128 j + 2 lane,
// and bin 512 goes out alone; on odd rows it writes (m+1, m+2) -- the last pair is (511, 512) -- and
// bin 0 goes out alone.  (8-byte stores at a 16-byte stride turn every line into partial writes and cost
// the whole saving.)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int J, bool ODD>
__device__ __forceinline__ void split_store_half_j(const float2 *lds, int lane, const float2 *w2, float2 *row)
{
    const int m = 128 * J + 2 * lane + (ODD ? 1 : 0);            // first bin of this lane's pair
    float2 z0, z1;
    if (ODD) { z0 = lds[m]; z1 = lds[(m + 1) & 511]; }
    else { const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]); z0 = make_float2(zz.x, zz.y); z1 = make_float2(zz.z, zz.w); }
    float2 zr0, zr1;
    load_mirror_pair(lds, m, zr0, zr1);
    float2 lo0, hi0, lo1, hi1;
    split_fwd<J>(z0, zr0, w2[0], lo0, hi0);                      // W^m = w_8^J * W^(m - 128 J); m = 512 gives -1
    split_fwd<J>(z1, zr1, w2[1], lo1, hi1);
    f32x4 t = {lo0.x, lo0.y, lo1.x, lo1.y};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(row + m));
    if (J == 0 && lane == 0) {
        if (ODD) {                                               // bin 0: E + O of Z[0] with itself
            const float2 z = lds[0];
            float2 l, h;
            split_fwd<0>(z, z, make_float2(1.f, 0.f), l, h);
            f32x2 n = {l.x, l.y};
            __builtin_nontemporal_store(n, reinterpret_cast<f32x2 *>(row));
        } else {                                                 // bin 512 = X[0 + 512]
            f32x2 n = {hi0.x, hi0.y};
            __builtin_nontemporal_store(n, reinterpret_cast<f32x2 *>(row + 512));
        }
    }
}

template <bool ODD>
__device__ __forceinline__ void split_store_half(const float2 *lds, int lane, const float2 *w2, float2 *row)
{
    split_store_half_j<0, ODD>(lds, lane, w2, row);
    split_store_half_j<1, ODD>(lds, lane, w2, row);
    split_store_half_j<2, ODD>(lds, lane, w2, row);
    split_store_half_j<3, ODD>(lds, lane, w2, row);
}

template <int K>
__global__ __launch_bounds__(64, JDSP_STFT_MINWAVES) void stft1024_hop512_half_kernel(
    const short *__restrict__ pcm, float2 *__restrict__ spec, long n_frames, long pitch,
    const float2 *__restrict__ table)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long f0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * K;
    if (f0 >= n_frames) return;
    const unsigned int *pcm32 = reinterpret_cast<const unsigned int *>(pcm) + lane;   // sample pair per lane
    unsigned int half[K + 1][4];
#pragma unroll
    for (int h = 0; h <= K; h++) {
        const long hh = f0 + h <= n_frames ? f0 + h : n_frames;
#pragma unroll
        for (int q = 0; q < 4; q++) half[h][q] = pcm32[hh * 256 + 64 * q];
    }
    FrameTables t;
    load_frame_tables(t, table, lane);
    static_assert(K % 2 == 0, "row parity is taken from the in-wave frame index");
    const float2 wsp2 = table[kStftSplit + 2 * lane + 2];        // W^(2 lane + 2); lane 63: W^128
    unsigned int raw[8];
#pragma unroll
    for (int q = 0; q < 4; q++) raw[4 + q] = half[0][q];
#pragma unroll
    for (int i = 0; i < K; i++) {
        const long f = f0 + i;
        if (f >= n_frames) break;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) { raw[r] = raw[r + 4]; raw[r + 4] = half[i + 1][r]; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 s = unpack_i16x2(raw[r]);
            v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
        }
        wave_fft512<false>(v, lds, lane, t.tw);
        store_natural_image(lds, lane, v);
        wave_lds_fence();
        float2 *row = spec + f * pitch;
        if ((i & 1) && (pitch & 1)) {                            // f0 is even (K is): odd row of an odd pitch
            const float2 wodd[2] = {t.wsp[1], wsp2};
            split_store_half<true>(lds, lane, wodd, row);
        } else {
            split_store_half<false>(lds, lane, t.wsp, row);
        }
        wave_lds_fence();
    }
}

int launch_stft1024_half(hipStream_t stream, const short *pcm, long n_frames, float2 *spec, long pitch,
                         const float2 *table)
{
    if (n_frames <= 0) return 0;
    const long grid = ((n_frames + 1) / 2 + 7) / 8 * 8;
    hipLaunchKernelGGL(stft1024_hop512_half_kernel<2>, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames,
                       pitch, table);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// any hop: every frame fetches its own 1024 samples with 16-bit loads.
__global__ __launch_bounds__(64) void stft1024_anyhop_kernel(const short *__restrict__ pcm, float2 *__restrict__ spec,
                                                             long n_frames, int frames_per_wave, long hop,
                                                             const float2 *__restrict__ table)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * frames_per_wave;
    long f1 = f0 + frames_per_wave;
    if (f1 > n_frames) f1 = n_frames;
    if (f0 >= f1) return;
    FrameTables t;
    load_frame_tables(t, table, lane);
    for (long f = f0; f < f1; f++) {
        float2 v[8];
        const short *src = pcm + f * hop + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; r++)
            v[r] = make_float2((float)src[128 * r] * t.win[r].x, (float)src[128 * r + 1] * t.win[r].y);
        wave_fft512<false>(v, lds, lane, t.tw);
        store_natural_image(lds, lane, v);
        wave_lds_fence();
        split_and_store(lds, lane, t.wsp, spec + f * 1024);
        wave_lds_fence();
    }
}

// n_fft = 512 (BASELINE config 3 as written; the reference's SS/Wiener use 1024): the 512-point
// spectrum of a frame is exactly the even bins of the 1024-point spectrum of the frame
// zero-padded, so the same wave transform serves; only the even half of the split is formed.
template <int J>
__device__ __forceinline__ void split_store_even_j(const float2 *lds, int lane, float2 wsp0, float2 *dst)
{
    const int m = 128 * J + 2 * lane;
    const float2 zm = lds[m];
    const float2 zr = lds[512 - m];
    float2 lo, hi;
    split_fwd<J>(zm, zr, wsp0, lo, hi);
    __builtin_nontemporal_store(lo.x, &dst[64 * J + lane].x);
    __builtin_nontemporal_store(lo.y, &dst[64 * J + lane].y);
    __builtin_nontemporal_store(hi.x, &dst[256 + 64 * J + lane].x);
    __builtin_nontemporal_store(hi.y, &dst[256 + 64 * J + lane].y);
}

__global__ __launch_bounds__(64) void stft512_kernel(const short *__restrict__ pcm, float2 *__restrict__ spec,
                                                     long n_frames, int frames_per_wave, long hop,
                                                     const float2 *__restrict__ table,
                                                     const float2 *__restrict__ win512)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * frames_per_wave;
    long f1 = f0 + frames_per_wave;
    if (f1 > n_frames) f1 = n_frames;
    if (f0 >= f1) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp0 = table[kStftSplit + 2 * lane];
    float2 win[4];
#pragma unroll
    for (int r = 0; r < 4; r++) win[r] = win512[lane + 64 * r];
    for (long f = f0; f < f1; f++) {
        float2 v[8];
        const short *src = pcm + f * hop + 2 * lane;
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[r] = make_float2((float)src[128 * r] * win[r].x, (float)src[128 * r + 1] * win[r].y);
#pragma unroll
        for (int r = 4; r < 8; r++) v[r] = make_float2(0.f, 0.f);
        wave_fft512<false>(v, lds, lane, tw);
        store_natural_image(lds, lane, v);
        wave_lds_fence();
        float2 *dst = spec + f * 512;
        split_store_even_j<0>(lds, lane, wsp0, dst);
        split_store_even_j<1>(lds, lane, wsp0, dst);
        split_store_even_j<2>(lds, lane, wsp0, dst);
        split_store_even_j<3>(lds, lane, wsp0, dst);
        wave_lds_fence();
    }
}

int launch_stft512(hipStream_t stream, int n_cu, const short *pcm, long n_frames, long hop, float2 *spec,
                   const float2 *table, const float2 *win512)
{
    if (n_frames <= 0) return 0;
    const long fpw = n_frames >= 2L * n_cu * 16 ? 2 : 1;
    const long grid = (n_frames + fpw - 1) / fpw;
    hipLaunchKernelGGL(stft512_kernel, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames, (int)fpw, hop,
                       table, win512);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void fill_win512(float2 *w, int window_kind)
{
    const double wa = window_kind == 1 ? 0.5 : 0.54, wb = window_kind == 1 ? 0.5 : 0.46;
    for (int i = 0; i < 256; i++) {
        const double w0 = (wa - wb * cos(2 * 3.141592 * (2 * i) / (512 - 1)));
        const double w1 = (wa - wb * cos(2 * 3.141592 * (2 * i + 1) / (512 - 1)));
        w[i] = make_float2((float)(0.5 * w0), (float)(0.5 * w1));
    }
}

template <int K>
static void launch_hop512(hipStream_t stream, const short *pcm, long n_frames, float2 *spec, const float2 *table)
{
    long grid = (n_frames + K - 1) / K;
#if JDSP_STFT_XCD
    grid = (grid + 7) / 8 * 8;
#endif
    hipLaunchKernelGGL(stft1024_hop512_kernel<K>, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames,
                       table);
}

// ---- read pass: pulls a PCM range into the Infinity Cache ahead of the transform ---------------------------------
// The transform kernel above runs at the chip's store rate (88-90 us per 65,536 frames against 83 us for a
// hipMemset of the 512 MiB of spectra) -- when its PCM comes out of the 256 MiB Infinity Cache, which is what a
// benchmark that re-reads one buffer measures.  With the input in HBM, where a stream of audio is, the same
// launch takes 147 us (profiles/r02_cold_input_probe.txt): a wave's loads queue behind the write stream, every wave
// holds its slot for microseconds doing nothing, and HBM's 1 : 8 mix of reads and writes runs at 5.4-5.6 TB/s even
// for a plain copy of this shape (tools/membw.hip, "COLD in": 107-112 us).  What was tried instead of this pass:
// more frames per wave (K = 4: 131 us), more waves per SIMD (spills), touching the chunk D places ahead from inside
// the transform (D = 1024, K = 4: 117 us), a software-pipelined persistent wave that loads chunk c+1 before it
// stores chunk c, with hand-placed s_waitcnt (121-123 us cold, 99 us warm).  Separating the two streams in TIME wins:
// one read-only launch streams the slab's PCM through (64 MiB in 9 us = 7.5 TB/s; reads allocate in the Infinity
// Cache, the transform's nontemporal stores do not displace them), then the transform runs at its store rate:
// 98 us per 65,536 cold frames.  A warm input pays the 9 us for nothing; "stft.read_pass" = 0 turns the pass off.
constexpr long kReadPassSlabFrames = 65536;       // 64 MiB of PCM per slab
constexpr long kReadPassMinFrames = 16384;        // auto mode: smaller batches live in L2 / Infinity Cache anyway

__global__ __launch_bounds__(256) void pcm_touch_kernel(const u32x4 *__restrict__ p, long n16, unsigned int *__restrict__ sink)
{
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned int acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {               // four 16-byte loads per lane in flight
        const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a.x ^ b.x ^ c.x ^ d.x;
    }
    for (; i < n16; i += stride) acc ^= p[i].x;
    if (sink && acc == 0x9e3779b9u) *sink = acc;                   // keeps the loads alive; sink is NULL in every launch
}

static int launch_pcm_touch(hipStream_t stream, int n_cu, int wg_per_cu, const short *pcm, long n_samples)
{
    const long n16 = n_samples / 8;                                // whole 16-byte units (pcm is 16-byte aligned)
    if (n16 <= 0) return 0;
    long grid = (long)n_cu * (wg_per_cu > 0 ? wg_per_cu : 4);      // 4 x 256 threads x 4 loads of 16 B: 64 KB in flight per CU
    const long need = (n16 + 255) / 256;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(pcm_touch_kernel, dim3((unsigned)grid), dim3(256), 0, stream, reinterpret_cast<const u32x4 *>(pcm), n16,
                       (unsigned int *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the same pass for other transforms of hop-512 streams (the FP64 analysis, fft_c2c_kernels.hip)
int launch_pcm_read_pass(hipStream_t stream, int n_cu, int wg_per_cu, const short *pcm, long n_samples)
{
    return launch_pcm_touch(stream, n_cu, wg_per_cu, pcm, n_samples);
}

// One slab of the fast path: frames [0, n_frames) of `pcm` with K frames per wave.
static void launch_hop512_any(hipStream_t stream, int fpw_opt, const short *pcm, long n_frames, float2 *spec,
                              const float2 *table)
{
    switch (fpw_opt) {
    case 1: launch_hop512<1>(stream, pcm, n_frames, spec, table); break;
    case 2: launch_hop512<2>(stream, pcm, n_frames, spec, table); break;
    case 3: launch_hop512<3>(stream, pcm, n_frames, spec, table); break;
    case 4: launch_hop512<4>(stream, pcm, n_frames, spec, table); break;
    default: launch_hop512<JDSP_STFT_K>(stream, pcm, n_frames, spec, table); break;
    }
}

// read_pass: 0 never, 1 always, -1 auto (batches of kReadPassMinFrames frames or more).  touch_wg_per_cu: 0 = default.
int launch_stft1024(hipStream_t stream, int n_cu, int fpw_opt, const short *pcm, long n_frames, long hop,
                    float2 *spec, const float2 *table, int read_pass, int touch_wg_per_cu)
{
    if (n_frames <= 0) return 0;
    if (hop == 512 && ((uintptr_t)pcm & 15u) == 0) {
        const bool touch = read_pass > 0 || (read_pass < 0 && n_frames >= kReadPassMinFrames);
#ifdef JDSP_PROBE_READ_PASS_ALONE
        // timing-only build (tools/build_variant.sh -DJDSP_PROBE_READ_PASS_ALONE, tools/prefetch_probe.py): "stft.read_pass" = 1
        // issues the read pass and NO transform -- spectra are not written.  Never part of the shipped library.
        if (read_pass == 1) {
            for (long f0 = 0; f0 < n_frames; f0 += kReadPassSlabFrames) {
                const long nf = n_frames - f0 < kReadPassSlabFrames ? n_frames - f0 : kReadPassSlabFrames;
                if (launch_pcm_touch(stream, n_cu, touch_wg_per_cu, pcm + f0 * 512, 512 * (nf + 1))) return -1;
            }
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
#endif
        if (!touch) {
            launch_hop512_any(stream, fpw_opt, pcm, n_frames, spec, table);
        } else {
            // slab by slab, so that a slab's PCM (64 MiB) is a quarter of the Infinity Cache whatever the batch is
            for (long f0 = 0; f0 < n_frames; f0 += kReadPassSlabFrames) {
                const long nf = n_frames - f0 < kReadPassSlabFrames ? n_frames - f0 : kReadPassSlabFrames;
                if (launch_pcm_touch(stream, n_cu, touch_wg_per_cu, pcm + f0 * 512, 512 * (nf + 1))) return -1;
                // behind the read pass one frame per wave is the fastest (98.5 us per 65,536 cold frames against
                // 100-102 for two and 108-110 for three, profiles/r02_cold_input_probe.txt)
                launch_hop512_any(stream, fpw_opt > 0 ? fpw_opt : 1, pcm + f0 * 512, nf, spec + f0 * 1024, table);
            }
        }
    } else {
        // short-lived waves here too: a wave that loops reads its next frame after its stores
        // and so waits for them (see the header); two frames amortise the table loads
        const long fpw = n_frames >= 2L * n_cu * 16 ? 2 : 1;
        long grid = (n_frames + fpw - 1) / fpw;
        hipLaunchKernelGGL(stft1024_anyhop_kernel, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames,
                           (int)fpw, hop, table);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int stft1024_table_count() { return kStftTableCount; }

// Host-side table contents (double precision, rounded once to float).
void fill_stft1024_table(float2 *t, int window_kind)
{
    // window_kind 0: the reference's Hamming 0.54 - 0.46 cos; 1: Hann 0.5 - 0.5 cos (same argument);
    // 2: rectangular (the partitioned convolver's frames are not windowed)
    const double wa = window_kind == 2 ? 1.0 : (window_kind == 1 ? 0.5 : 0.54);
    const double wb = window_kind == 2 ? 0.0 : (window_kind == 1 ? 0.5 : 0.46);
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 1; k < 8; k++)
        for (int l = 0; l < 64; l++) {
            double a = -two_pi * (double)(l * k) / 512.0;
            t[kTwT1 + (k - 1) * 64 + l] = make_float2((float)cos(a), (float)sin(a));
        }
    for (int c = 1; c < 8; c++)
        for (int b = 0; b < 8; b++) {
            double a = -two_pi * (double)(b * c) / 64.0;
            t[kTwT2 + (c - 1) * 8 + b] = make_float2((float)cos(a), (float)sin(a));
        }
    // Hamming exactly as the reference writes it (PI 3.141592, SS:52,226); halved copy for the split
    for (int i = 0; i < 512; i++) {
        double w0 = (wa - wb * cos(2 * 3.141592 * (2 * i) / (1024 - 1)));
        double w1 = (wa - wb * cos(2 * 3.141592 * (2 * i + 1) / (1024 - 1)));
        t[kStftWin + i] = make_float2((float)(0.5 * w0), (float)(0.5 * w1));
        t[kStftWinFull + i] = make_float2((float)w0, (float)w1);
    }
    for (int m = 0; m < 512; m++) {
        double a = -two_pi * (double)m / 1024.0;
        t[kStftSplit + m] = make_float2((float)cos(a), (float)sin(a));
    }
}

}  // namespace jdsp
