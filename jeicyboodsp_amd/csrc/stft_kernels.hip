// stft_kernels.hip -- batched STFT analysis for 1024-point frames (gfx950).
//
// Work done per frame (reference: SpectralSubtraction_final.cpp:218-230,
// WienerFilter_final.cpp:181-193): int16 -> double, * Hamming, 1024-point
// forward c2c DFT of the real frame, all 1024 bins kept.
//
// Mapping: one wavefront owns `frames_per_wave` CONSECUTIVE frames.  With
// hop = 512 the second half of frame f is the first half of frame f+1 and,
// in the "lane + 64 r" layout, lands in the same lane -- so after the first
// frame a wave loads only the 512 new samples (4 dwords per lane) and every
// PCM sample is read from HBM exactly once.  The 8 KB spectrum leaves as eight
// fully coalesced 1 KB wave stores.  HBM-bound: 1 KB in + 8 KB out per frame.
#include "jdsp_internal.h"
#include "wave_fft512.h"

namespace jdsp {

// table layout (float2 units) appended after the wave twiddles
constexpr int kStftWin = kTwCount;            // [512] pairs: 0.5*w[2i], 0.5*w[2i+1]
constexpr int kStftSplit = kStftWin + 512;    // [512] W^m = exp(-2*pi*j*m/1024)
constexpr int kStftTableCount = kStftSplit + 512;

__device__ __forceinline__ float2 unpack_i16x2(unsigned int raw)
{
    return make_float2((float)(short)(raw & 0xffffu), (float)((int)raw >> 16));
}

// Split of the packed transform: Zh = FFT512(z)/2 (the 1/2 is folded into the
// window).  For m in [0,512):  E = Zh[m] + conj(Zh[512-m]),
// O = -j (Zh[m] - conj(Zh[512-m])),  X[m] = E + W^m O,  X[m+512] = E - W^m O.
__device__ __forceinline__ void split_pair(float2 zm, float2 zr, float2 w, float2 &lo, float2 &hi)
{
    const float2 e = make_float2(zm.x + zr.x, zm.y - zr.y);
    const float2 o = make_float2(zm.y + zr.y, zr.x - zm.x);
    const float2 t = cmul(w, o);
    lo = cadd(e, t);
    hi = csub(e, t);
}

// ---- build-time variants (tools/tune_stft.py A/Bs them on the GPU) -------------------
#ifndef JDSP_STFT_MINWAVES
#define JDSP_STFT_MINWAVES 4      // __launch_bounds__ 2nd arg: waves per SIMD the register budget must allow
#endif
#ifndef JDSP_STFT_K
#define JDSP_STFT_K 2             // consecutive frames per wavefront (default of "stft.frames_per_wave" = 0)
#endif
#ifndef JDSP_STFT_NT_LOAD
#define JDSP_STFT_NT_LOAD 0       // 1: nontemporal PCM loads
#endif
#ifndef JDSP_STFT_NT
#define JDSP_STFT_NT 1            // 1: nontemporal spectrum stores
#endif
#ifndef JDSP_STFT_WSP_SMALL
#define JDSP_STFT_WSP_SMALL 1     // 1: 2 split twiddles per lane + w_8 rotations instead of 8
#endif
#ifndef JDSP_STFT_ABLATE
#define JDSP_STFT_ABLATE 0        // timing-only ablations (1: no stores, 2: no transform); never shipped
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
template <bool SMALL>
__device__ __forceinline__ void split_and_store(const float2 *lds, int lane, const float2 *wsp, float2 *out_frame)
{
    float4 *dst = reinterpret_cast<float4 *>(out_frame) + lane;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = 128 * j + 2 * lane;
        const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
        const float2 zr0 = lds[(512 - m) & 511];
        const float2 zr1 = lds[511 - m];
        float2 lo0, hi0, lo1, hi1;
        if (SMALL) {
            // W^(128 j + 2 lane + e) = w_8^j * W^(2 lane + e): rotate the odd part instead of the twiddle
            const float2 e0 = make_float2(zz.x + zr0.x, zz.y - zr0.y), o0 = make_float2(zz.y + zr0.y, zr0.x - zz.x);
            const float2 e1 = make_float2(zz.z + zr1.x, zz.w - zr1.y), o1 = make_float2(zz.w + zr1.y, zr1.x - zz.z);
            float2 t0 = cmul(wsp[0], o0), t1 = cmul(wsp[1], o1);
            if (j == 1) { t0 = rot45<false>(t0); t1 = rot45<false>(t1); }
            if (j == 2) { t0 = rot90<false>(t0); t1 = rot90<false>(t1); }
            if (j == 3) { t0 = rot135<false>(t0); t1 = rot135<false>(t1); }
            lo0 = cadd(e0, t0); hi0 = csub(e0, t0);
            lo1 = cadd(e1, t1); hi1 = csub(e1, t1);
        } else {
            split_pair(make_float2(zz.x, zz.y), zr0, wsp[2 * j], lo0, hi0);
            split_pair(make_float2(zz.z, zz.w), zr1, wsp[2 * j + 1], lo1, hi1);
        }
        store_spec(dst + 64 * j, make_float4(lo0.x, lo0.y, lo1.x, lo1.y));
        store_spec(dst + 64 * j + 256, make_float4(hi0.x, hi0.y, hi1.x, hi1.y));
    }
}

constexpr int kNWsp = JDSP_STFT_WSP_SMALL ? 2 : 8;

__device__ __forceinline__ void load_split_twiddles(float2 *wsp, const float2 *__restrict__ table, int lane)
{
    if (JDSP_STFT_WSP_SMALL) {
        wsp[0] = table[kStftSplit + 2 * lane];
        wsp[1] = table[kStftSplit + 2 * lane + 1];
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wsp[2 * j] = table[kStftSplit + 128 * j + 2 * lane];
            wsp[2 * j + 1] = table[kStftSplit + 128 * j + 2 * lane + 1];
        }
    }
}

// hop == 512.  One wave owns K consecutive frames = K+1 half-frames of 512 samples; every
// PCM sample is fetched once (one dwordx4 per lane per half-frame) and ALL loads are issued
// up front.  That is deliberate: gfx950's vmcnt retires loads and stores in issue order, so a
// wave that waits for a load issued after a frame's stores also waits for those stores to be
// acknowledged by HBM (measured: a prefetching loop runs 25-50 % slower than this form, whose
// waves issue their last stores and simply end; see DESIGN.md "STFT kernel").
template <int K>
__global__ __launch_bounds__(64, JDSP_STFT_MINWAVES) void stft1024_hop512_kernel(
    const short *__restrict__ pcm, float2 *__restrict__ spec, long n_frames, const float2 *__restrict__ table)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage[256];
    const int lane = threadIdx.x;
    const long f0 = (long)blockIdx.x * K;
    if (f0 >= n_frames) return;

    // half-frame h holds samples [512 h, 512 h + 512); the stream has n_frames + 1 of them
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

    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float2 win[8], wsp[kNWsp];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = table[kStftWin + lane + 64 * r];
    load_split_twiddles(wsp, table, lane);

    // raw[r]: the int16 pair (2*lane + 128*r, +1) of the frame, re-laid out through LDS from
    // the 16-byte-per-lane load image.  The second half of frame f is the first half of f+1
    // and sits in the same lane, so raw[4..7] slide down to raw[0..3].
    unsigned int raw[8];
    reinterpret_cast<u32x4 *>(stage)[lane] = half[0];
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; r++) raw[r + 4] = stage[lane + 64 * r];
    wave_lds_fence();

#pragma unroll
    for (int i = 0; i < K; i++) {
        const long f = f0 + i;
        if (f >= n_frames) break;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) raw[r] = raw[r + 4];
        reinterpret_cast<u32x4 *>(stage)[lane] = half[i + 1];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; r++) raw[r + 4] = stage[lane + 64 * r];
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 s = unpack_i16x2(raw[r]);
            v[r] = make_float2(s.x * win[r].x, s.y * win[r].y);
        }
#if JDSP_STFT_ABLATE == 2   /* timing-only build: no transform, stores only */
        {
            float4 *dst = reinterpret_cast<float4 *>(spec + f * 1024) + lane;
#pragma unroll
            for (int j = 0; j < 8; j++) store_spec(dst + 64 * j, make_float4(v[j].x, v[j].y, v[j].x, v[j].y));
            continue;
        }
#endif
        wave_fft512<false>(v, lds, lane, tw);

        // natural-order image of Zh, then the split reads Zh[m] and Zh[512-m]
#pragma unroll
        for (int d = 0; d < 8; d++) lds[lane + 64 * d] = v[d];
        wave_lds_fence();
#if JDSP_STFT_ABLATE == 1   /* timing-only build: transform kept alive, stores never taken */
        if (n_frames < 0)
#endif
        split_and_store<JDSP_STFT_WSP_SMALL != 0>(lds, lane, wsp, spec + f * 1024);
        wave_lds_fence();
    }
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
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float2 win[8], wsp[kNWsp];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = table[kStftWin + lane + 64 * r];
    load_split_twiddles(wsp, table, lane);
    for (long f = f0; f < f1; f++) {
        float2 v[8];
        const short *src = pcm + f * hop + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; r++)
            v[r] = make_float2((float)src[128 * r] * win[r].x, (float)src[128 * r + 1] * win[r].y);
        wave_fft512<false>(v, lds, lane, tw);
#pragma unroll
        for (int d = 0; d < 8; d++) lds[lane + 64 * d] = v[d];
        wave_lds_fence();
        split_and_store<JDSP_STFT_WSP_SMALL != 0>(lds, lane, wsp, spec + f * 1024);
        wave_lds_fence();
    }
}

template <int K>
static void launch_hop512(hipStream_t stream, const short *pcm, long n_frames, float2 *spec, const float2 *table)
{
    const long grid = (n_frames + K - 1) / K;
    hipLaunchKernelGGL(stft1024_hop512_kernel<K>, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames,
                       table);
}

int launch_stft1024(hipStream_t stream, int n_cu, int fpw_opt, const short *pcm, long n_frames, long hop,
                    float2 *spec, const float2 *table)
{
    if (n_frames <= 0) return 0;
    if (hop == 512 && ((uintptr_t)pcm & 15u) == 0) {
        switch (fpw_opt) {
        case 1: launch_hop512<1>(stream, pcm, n_frames, spec, table); break;
        case 2: launch_hop512<2>(stream, pcm, n_frames, spec, table); break;
        case 3: launch_hop512<3>(stream, pcm, n_frames, spec, table); break;
        case 4: launch_hop512<4>(stream, pcm, n_frames, spec, table); break;
        case 6: launch_hop512<6>(stream, pcm, n_frames, spec, table); break;
        case 8: launch_hop512<8>(stream, pcm, n_frames, spec, table); break;
        default: launch_hop512<JDSP_STFT_K>(stream, pcm, n_frames, spec, table); break;
        }
    } else {
        long target_waves = (long)n_cu * 16;
        long fpw = (n_frames + target_waves - 1) / target_waves;
        if (fpw > 16) fpw = 16;
        if (fpw < 1) fpw = 1;
        long grid = (n_frames + fpw - 1) / fpw;
        hipLaunchKernelGGL(stft1024_anyhop_kernel, dim3((unsigned)grid), dim3(64), 0, stream, pcm, spec, n_frames,
                           (int)fpw, hop, table);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int stft1024_table_count() { return kStftTableCount; }

// Host-side table contents (double precision, rounded once to float).
void fill_stft1024_table(float2 *t)
{
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
    // Hamming exactly as the reference writes it (PI 3.141592, SS:52,226), halved for the split
    for (int i = 0; i < 512; i++) {
        double w0 = (0.54 - 0.46 * cos(2 * 3.141592 * (2 * i) / (1024 - 1)));
        double w1 = (0.54 - 0.46 * cos(2 * 3.141592 * (2 * i + 1) / (1024 - 1)));
        t[kStftWin + i] = make_float2((float)(0.5 * w0), (float)(0.5 * w1));
    }
    for (int m = 0; m < 512; m++) {
        double a = -two_pi * (double)m / 1024.0;
        t[kStftSplit + m] = make_float2((float)cos(a), (float)sin(a));
    }
}

}  // namespace jdsp
