// mfcc_kernels.hip -- MFCCFeatureExtraction_auto_version1.cpp:194-231 on gfx950, one frame
// per wavefront: pre-emphasis (:208-210) -> Hamming (:212-214) -> 1024-point forward transform
// (:216-217) -> |X| (:218-220) -> mel filterbank + ln (:154-174) -> DCT-II (:176-183) ->
// sinusoidal lifter (:185-192), n_cep doubles out per frame.
//
// n_fft = 512 (BASELINE config 4: 400-sample window, 512-FFT) runs on the same 1024-point
// machinery: the 512-point spectrum of a frame is exactly the even bins of the 1024-point
// spectrum of the same frame zero-padded.
#include <type_traits>
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

// redo (or NULL): {count, pair indices...} left by mfcc512_run_kernel -- the frames 2 q and 2 q + 1 of every listed pair
// are computed again here, each in a transform of its own, by waves that stride over the list.
__global__ __launch_bounds__(64) void mfcc_kernel(const short *__restrict__ pcm, const long long *__restrict__ starts,
                                                  long n_frames, MfccDev p, const float2 *__restrict__ table,
                                                  double *__restrict__ feats, const int *__restrict__ redo)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ float mag[512];
    __shared__ float logmel[64];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    long f = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    long it = blockIdx.x;
  for (;;) {
    if (redo) {
        if (it >= 2L * redo[0]) return;
        f = 2L * redo[1 + (it >> 1)] + (it & 1);
        it += gridDim.x;
        if (f >= n_frames) continue;
    } else if (f >= n_frames) return;
    const short *src = pcm + (starts ? starts[f] : (long long)p.hop * f);

    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp0 = table[kStftSplit + 2 * lane];
    const float2 wsp1 = table[kStftSplit + 2 * lane + 1];

    // x[i] = s[i] - preemph * s[i-1] for 1 <= i < win_len, x[0] = 0 (:208 starts at i = 1), zero beyond
    float2 v[8];
    if (p.win_len == 1024 && (((uintptr_t)src) & 3u) == 0) {
        // full-length window, 4-byte aligned frame: one dword per sample pair, plus the dword in
        // front of it for the sample the pre-emphasis reaches back to
        const unsigned int *s32 = reinterpret_cast<const unsigned int *>(src) + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 cur = unpack_i16x2(s32[64 * r]);
            const bool first = (lane == 0 && r == 0);
            const float sm = first ? 0.f : (float)((int)s32[first ? 0 : 64 * r - 1] >> 16);
            const float2 w = p.window[lane + 64 * r];
            const float x0 = first ? 0.f : cur.x - p.preemph * sm;
            const float x1 = cur.y - p.preemph * cur.x;
            v[r] = make_float2(x0 * w.x, x1 * w.y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i0 = 2 * lane + 128 * r;
            float sm = 0.f, s0 = 0.f, s1 = 0.f;
            if (i0 >= 1 && i0 - 1 < p.win_len) sm = (float)src[i0 - 1];
            if (i0 < p.win_len) s0 = (float)src[i0];
            if (i0 + 1 < p.win_len) s1 = (float)src[i0 + 1];
            const float2 w = p.window[lane + 64 * r];            // halved Hamming pair, zero beyond win_len
            float x0 = (i0 >= 1) ? s0 - p.preemph * sm : 0.f;
            float x1 = s1 - p.preemph * s0;
            v[r] = make_float2(x0 * w.x, x1 * w.y);
        }
    }
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
    // |X[m]| for m = 128 j + 2 lane + e < 512 (:218-220 computes all 1024, only these are used)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = 128 * j + 2 * lane;
        const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
        float2 zr0, zr1;
        load_mirror_pair(lds, m, zr0, zr1);
        float2 lo0, hi0, lo1, hi1;
        if (j == 0) { split_fwd<0>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<0>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 1) { split_fwd<1>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<1>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 2) { split_fwd<2>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<2>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 3) { split_fwd<3>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<3>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        // hardware square root (1 ulp; sqrtf() expands to ~10 instructions of scaling and fix-up per value)
        const float a0 = __builtin_amdgcn_sqrtf(lo0.x * lo0.x + lo0.y * lo0.y);
        const float a1 = __builtin_amdgcn_sqrtf(lo1.x * lo1.x + lo1.y * lo1.y);
        if (p.bin_stride == 1) *reinterpret_cast<float2 *>(&mag[m]) = make_float2(a0, a1);
        else mag[m >> 1] = a0;                               // 512-point bins = even 1024-point bins
    }
    wave_lds_fence();
    // mel filterbank (:157-168): every lane walks 8 CONSECUTIVE bins; consecutive bins feed the
    // same one or two channels, so a lane sums them in registers and adds to the channel
    // accumulators in LDS only when the channel index changes (2-4 LDS atomics per lane instead
    // of a 90-iteration loop on the 38 lanes that own a channel)
    logmel[lane] = 0.f;
    wave_lds_fence();
    {
        const int i0 = 8 * lane;
        if (i0 < p.n_bins) {
            const float4 m0 = *reinterpret_cast<const float4 *>(&mag[i0]), m1 = *reinterpret_cast<const float4 *>(&mag[i0 + 4]);
            const float4 f0 = *reinterpret_cast<const float4 *>(p.mel_fb + i0), f1 = *reinterpret_cast<const float4 *>(p.mel_fb + i0 + 4);
            const int4 k0 = *reinterpret_cast<const int4 *>(p.mel_k + i0), k1 = *reinterpret_cast<const int4 *>(p.mel_k + i0 + 4);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            const float ff[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            const int kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
            int cur = kk[0];
            float lo = 0.f, hi = 0.f;                      // partial sums for channels cur-1 and cur
#pragma unroll
            for (int t = 0; t < 8; t++) {
                if (kk[t] != cur) {
                    if (cur >= 1) atomicAdd(&logmel[cur - 1], lo);
                    if (cur < p.n_chan) atomicAdd(&logmel[cur], hi);
                    cur = kk[t];
                    lo = hi = 0.f;
                }
                if (cur == 0) hi += (1.f - ff[t]) * mm[t];                 // :161
                else {
                    lo += ff[t] * mm[t];                                   // :164
                    if (cur != p.n_chan) hi += (1.f - ff[t]) * mm[t];      // :165-166
                }
            }
            if (cur >= 1) atomicAdd(&logmel[cur - 1], lo);
            if (cur < p.n_chan) atomicAdd(&logmel[cur], hi);
        }
    }
    wave_lds_fence();
    if (lane < p.n_chan) logmel[lane] = logf(logmel[lane]);               // :171
    wave_lds_fence();
    // DCT-II (:178-182) and lifter (:189): cepstrum i on lanes i, i+16, i+32, i+48, each summing every
    // fourth channel (four short independent chains of table loads instead of one long one)
    {
        // up to 16 cepstra: four lane groups each sum every fourth channel; 17..32: two groups, every second
        const bool wide = p.n_cep > 16;
        const int i = wide ? (lane & 31) : (lane & 15), part = wide ? (lane >> 5) : (lane >> 4), step = wide ? 2 : 4;
        double acc = 0.0;
        if (i < p.n_cep) {
#pragma unroll 4
            for (int k = part; k < p.n_chan; k += step) acc += p.dct[k * 32 + i] * (double)logmel[k];
        }
        const double other = __shfl_xor(acc, 16);
        if (!wide) acc += other;
        acc += __shfl_xor(acc, 32);
        if (lane < p.n_cep) feats[f * p.n_cep + lane] = acc * p.lifter_w[lane];
    }
    if (!redo) return;
    wave_lds_fence();
  }
}


// ---- two frames per wavefront, in lock-step --------------------------------------------------------------
// The one-frame kernel spends 65 % of its wave cycles waiting (profiles/r01_chains_sq_counters.csv): a serial
// chain of short phases separated by LDS fences and table loads, at about five waves per SIMD.  Here a wave
// carries two frames through every phase together: the table loads are shared, every fence covers two
// frames, and two independent dependency chains interleave.  |X| goes through registers and reuses the
// transform scratch, which keeps the LDS footprint at 9.8 KB per wave (four waves per SIMD).
__device__ __forceinline__ void mfcc_load_frame(const short *__restrict__ src, const MfccDev &p, int lane, float2 (&v)[8])
{
    // x[i] = s[i] - preemph * s[i-1] for 1 <= i < win_len, x[0] = 0 (:208 starts at i = 1), zero beyond
    if (p.win_len == 1024 && (((uintptr_t)src) & 3u) == 0) {
        const unsigned int *s32 = reinterpret_cast<const unsigned int *>(src) + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 cur = unpack_i16x2(s32[64 * r]);
            const bool first = (lane == 0 && r == 0);
            const float sm = first ? 0.f : (float)((int)s32[first ? 0 : 64 * r - 1] >> 16);
            const float2 w = p.window[lane + 64 * r];
            const float x0 = first ? 0.f : cur.x - p.preemph * sm;
            const float x1 = cur.y - p.preemph * cur.x;
            v[r] = make_float2(x0 * w.x, x1 * w.y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i0 = 2 * lane + 128 * r;
            float sm = 0.f, s0 = 0.f, s1 = 0.f;
            if (i0 >= 1 && i0 - 1 < p.win_len) sm = (float)src[i0 - 1];
            if (i0 < p.win_len) s0 = (float)src[i0];
            if (i0 + 1 < p.win_len) s1 = (float)src[i0 + 1];
            const float2 w = p.window[lane + 64 * r];            // halved Hamming pair, zero beyond win_len
            const float x0 = (i0 >= 1) ? s0 - p.preemph * sm : 0.f;
            const float x1 = s1 - p.preemph * s0;
            v[r] = make_float2(x0 * w.x, x1 * w.y);
        }
    }
}

// The same in two halves, so that a kernel can have every load of a wave -- both frames' samples, window, twiddles,
// filterbank piece -- in flight before it waits for the first: mfcc_load_frame's `aligned?` branch sits between the
// two frames' loads and the tables', which made three memory round trips in series of what can be one (ISA: frame a's
// loads, s_waitcnt vmcnt(0), frame b's, vmcnt(0), tables), and its general path guards every halfword with a branch
// and a wait of its own.  ALIGNED: the dword holding samples 2 lane + 128 r, +1 (the sample before them is the previous lane's).  Otherwise
// three halfwords at positions clamped into [0, win_len): the window is zero wherever the clamp changes a position.
template <bool ALIGNED> struct MfccRaw;
template <> struct MfccRaw<true> { unsigned int cur[8]; };
template <> struct MfccRaw<false> { int sm[8], s0[8], s1[8]; };

__device__ __forceinline__ void mfcc_fetch(const short *__restrict__ src, const MfccDev &p, int lane, MfccRaw<true> &raw)
{
    const unsigned int *s32 = reinterpret_cast<const unsigned int *>(src) + lane;
#pragma unroll
    for (int r = 0; r < 8; r++) raw.cur[r] = s32[64 * r];
}
__device__ __forceinline__ void mfcc_fetch(const short *__restrict__ src, const MfccDev &p, int lane, MfccRaw<false> &raw)
{
    const int last = p.win_len - 1;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i0 = 2 * lane + 128 * r;
        raw.sm[r] = src[min(max(i0 - 1, 0), last)];
        raw.s0[r] = src[min(i0, last)];
        raw.s1[r] = src[min(i0 + 1, last)];
    }
}
// x[i] = s[i] - preemph * s[i-1] for 1 <= i < win_len, x[0] = 0 (:208 starts at i = 1), times the window (zero beyond)
__device__ __forceinline__ void mfcc_finish(const MfccRaw<true> &raw, const float2 (&w)[8], float preemph, int lane, float2 (&v)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 cur = unpack_i16x2(raw.cur[r]);
        // the dword before this one: lane - 1's (wave_shr:1), lane 0's is lane 63's of the row before
        const int up = __builtin_amdgcn_update_dpp(0, (int)raw.cur[r], 0x138, 0xf, 0xf, true);
        const int wrap = r > 0 ? __builtin_amdgcn_readlane((int)raw.cur[r > 0 ? r - 1 : 0], 63) : 0;
        const float sm = (float)((lane == 0 ? wrap : up) >> 16);
        const float x0 = (r == 0 && lane == 0) ? 0.f : cur.x - preemph * sm;
        const float x1 = cur.y - preemph * cur.x;
        v[r] = make_float2(x0 * w[r].x, x1 * w[r].y);
    }
}
__device__ __forceinline__ void mfcc_finish(const MfccRaw<false> &raw, const float2 (&w)[8], float preemph, int lane, float2 (&v)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float sm = (float)raw.sm[r], s0 = (float)raw.s0[r], s1 = (float)raw.s1[r];
        const float x0 = (r == 0 && lane == 0) ? 0.f : s0 - preemph * sm;
        const float x1 = s1 - preemph * s0;
        v[r] = make_float2(x0 * w[r].x, x1 * w[r].y);
    }
}

// |X[m]|, m = 128 j + 2 lane + e < 512, of the natural-order image `img` (:218-220)
__device__ __forceinline__ void mfcc_magnitudes(const float2 *img, int lane, float2 wsp0, float2 wsp1, float2 (&amp)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = 128 * j + 2 * lane;
        const float4 zz = *reinterpret_cast<const float4 *>(&img[m]);
        float2 zr0, zr1;
        load_mirror_pair(img, m, zr0, zr1);
        float2 lo0, hi0, lo1, hi1;
        if (j == 0) { split_fwd<0>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<0>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 1) { split_fwd<1>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<1>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 2) { split_fwd<2>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<2>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        if (j == 3) { split_fwd<3>(make_float2(zz.x, zz.y), zr0, wsp0, lo0, hi0); split_fwd<3>(make_float2(zz.z, zz.w), zr1, wsp1, lo1, hi1); }
        amp[j] = make_float2(__builtin_amdgcn_sqrtf(lo0.x * lo0.x + lo0.y * lo0.y),
                             __builtin_amdgcn_sqrtf(lo1.x * lo1.x + lo1.y * lo1.y));
    }
}

// mel filterbank (:157-168) of one frame: `mag` -> channel sums in `logmel` (LDS atomics, see mfcc_kernel)
__device__ __forceinline__ void mfcc_mel(const float *mag, float *logmel, const MfccDev &p, int lane, const float (&ff)[8],
                                         const int (&kk)[8])
{
    const int i0 = 8 * lane;
    if (i0 >= p.n_bins) return;
    const float4 m0 = *reinterpret_cast<const float4 *>(&mag[i0]), m1 = *reinterpret_cast<const float4 *>(&mag[i0 + 4]);
    const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    int cur = kk[0];
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        if (kk[t] != cur) {
            if (cur >= 1) atomicAdd(&logmel[cur - 1], lo);
            if (cur < p.n_chan) atomicAdd(&logmel[cur], hi);
            cur = kk[t];
            lo = hi = 0.f;
        }
        if (cur == 0) hi += (1.f - ff[t]) * mm[t];                 // :161
        else {
            lo += ff[t] * mm[t];                                   // :164
            if (cur != p.n_chan) hi += (1.f - ff[t]) * mm[t];      // :165-166
        }
    }
    if (cur >= 1) atomicAdd(&logmel[cur - 1], lo);
    if (cur < p.n_chan) atomicAdd(&logmel[cur], hi);
}

// a + a[lane ^ 16] / a + a[lane ^ 32] for a double: the swap leaves {even-row value, odd-row value} (resp. {lower-half,
// upper-half}) of the pair in the two results on both lanes of the pair, so their sum is the same bit pattern on both
__device__ __forceinline__ double sum_xor16_f64(double a)
{
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double sum_xor32_f64(double a)
{
    const int lo = __double2loint(a), hi = __double2hiint(a);
    const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
}

// Channel sums from the lanes' pieces.  Lane L holds (lo, hi) = its piece's contributions to the channels seg[L].z - 1 and
// seg[L].z; the pieces of one index are consecutive lanes, so channel ch is the sum of `hi` over lanes [src.x, +src.y)
// and of `lo` over lanes [src.z, +src.w) (chan_src[ch], at most four each when chan_ok).  Two plain stores per lane
// and frame and a few reads on the channel lanes instead of two LDS atomics per lane and frame: the atomics were 10 us
// of mfcc_x2_kernel's 95 (profiles/r02_mfcc512_run.txt), and their order -- hence the last bit of the sum -- was not
// reproducible.  pieces: 2 x 2 x 64 floats of LDS.
__device__ __forceinline__ void mel_channel_sums(float (*pieces)[2][64], float (*logmel)[64], const MfccDev &p, int lane,
                                                 float lo_a, float hi_a, float lo_b, float hi_b)
{
    pieces[0][0][lane] = lo_a; pieces[0][1][lane] = hi_a;
    pieces[1][0][lane] = lo_b; pieces[1][1][lane] = hi_b;
    wave_lds_fence();
    if (lane < p.n_chan) {
        const int4 src = p.chan_src[lane];
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int ih = src.x + (t < src.y ? t : 0), il = src.z + (t < src.w ? t : 0);
            const float ha = pieces[0][1][ih], hb = pieces[1][1][ih], la = pieces[0][0][il], lb = pieces[1][0][il];
            sa += (t < src.y ? ha : 0.f) + (t < src.w ? la : 0.f);
            sb += (t < src.y ? hb : 0.f) + (t < src.w ? lb : 0.f);
        }
        // :171.  Hardware log2 times ln 2 (1 ulp of the FP32 logarithm; logf() is ~25 instructions of range fix-up
        // per value for the last half ulp, of which nothing survives the 1e-5 bar); ln 0 = -inf as in the reference
        logmel[0][lane] = __logf(sa);
        logmel[1][lane] = __logf(sb);
    }
    wave_lds_fence();
}

#ifndef JDSP_MFCC_X2_PAIRS
#define JDSP_MFCC_X2_PAIRS 1
#endif
#ifndef JDSP_MFCC_ABLATE
#define JDSP_MFCC_ABLATE 0        // timing-only ablations of mfcc_x2_kernel's tail (tools/build_variant.sh): wrong results
#endif
// The tail both two-frame kernels share: |X| of two frames, bin i at mag[i + (i >> 4)] -> feats[fa], feats[fb].  sg/sw/cw:
// this lane's filterbank piece (MfccDev::seg, seg_wc), dc: its ten DCT coefficients (rows cpart + cstep t of column ci,
// unguarded loads from the zero-padded table), lw: its lifter weight -- all requested by the caller long before.
template <int PL> struct MfccLaneTables { int4 sg; float sw[PL], cw[PL]; };
template <int PL>
__device__ __forceinline__ void mfcc_load_lane_tables(MfccLaneTables<PL> &m, const MfccDev &p, int lane)
{
    m.sg = p.seg[lane];
#pragma unroll
    for (int q = 0; q < PL / 2; q++) {
        const float4 a = p.seg_wc[q * 64 + lane];
        m.sw[2 * q] = a.x; m.cw[2 * q] = a.y; m.sw[2 * q + 1] = a.z; m.cw[2 * q + 1] = a.w;
    }
}
__device__ __forceinline__ void mfcc_load_dct(double (&dc)[10], const MfccDev &p, int lane)
{
    const bool wide = p.n_cep > 16;                                  // see mfcc_kernel
    const int ci = wide ? (lane & 31) : (lane & 15), cpart = wide ? (lane >> 5) : (lane >> 4), cstep = wide ? 2 : 4;
#pragma unroll
    for (int t = 0; t < 10; t++) dc[t] = p.dct[(cpart + cstep * t) * 32 + ci];   // (a guarded load is a branch and a wait of its own)
}
template <int PL>
__device__ __forceinline__ void mfcc_tail_pre(const float *mag_a, const float *mag_b, float (*logmel)[64], float (*pieces)[2][64],
                                              const MfccDev &p, int lane, const MfccLaneTables<PL> &mt, const double (&dc)[10],
                                              double lw, long fa, long fb, bool two, double *__restrict__ feats)
{
    const bool wide = p.n_cep > 16;
    const int ci = wide ? (lane & 31) : (lane & 15), cpart = wide ? (lane >> 5) : (lane >> 4), cstep = wide ? 2 : 4;
    const int4 sg = mt.sg;
    const float (&sw)[PL] = mt.sw;
    const float (&cw)[PL] = mt.cw;
    const int last_bin = p.n_bins - 1;
    // mel filterbank (:157-168), one channel index per lane: bin i of index k adds f_i m_i to channel k - 1 and
    // (1 - f_i) m_i to channel k, so a lane whose bins all share k sums both in registers and issues exactly two
    // LDS atomics.  (Walking 8 consecutive bins per lane and flushing whenever the index changed took ~18
    // atomic instructions per frame, ~32 LDS cycles each: the LDS pipe was busy 72 % of the kernel.)
    {
        float lo_a = 0.f, hi_a = 0.f, lo_b = 0.f, hi_b = 0.f;
        // bins h .. h + N - 1 of the piece, 2 N reads in flight together (all 32 at once: 138 registers, three waves
        // per SIMD).  PL (8, 12 or 16) = the longest piece, the smallest for which the pieces fit 64 lanes
        // (mfcc_api.hip, MfccDev::piece_len): every lane walks PL bins, whatever its own piece's length.
        auto batch = [&](auto H, auto N) {
            constexpr int h = decltype(H)::value, n = decltype(N)::value;
            float ma[n], mb[n];
#pragma unroll
            for (int t = 0; t < n; t++) {
                const int bin = min(sg.x + h + t, last_bin);         // past the piece: any finite value, its weights are 0
                const int q = bin + (bin >> 4);
#if JDSP_MFCC_ABLATE & 1                                             /* timing-only: no filterbank reads */
                ma[t] = (float)q; mb[t] = (float)(q + 1);
#else
                ma[t] = mag_a[q];
                mb[t] = mag_b[q];
#endif
            }
#pragma unroll
            for (int t = 0; t < n; t++) {
                lo_a = fmaf(sw[h + t], ma[t], lo_a); hi_a = fmaf(cw[h + t], ma[t], hi_a);   // :164 / :161,:165-166
                lo_b = fmaf(sw[h + t], mb[t], lo_b); hi_b = fmaf(cw[h + t], mb[t], hi_b);
            }
        };
        batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        if constexpr (PL > 8) batch(std::integral_constant<int, 8>{}, std::integral_constant<int, PL - 8>{});
        if (p.chan_ok) {
            if (sg.y <= 0) { lo_a = hi_a = lo_b = hi_b = 0.f; }
            mel_channel_sums(pieces, logmel, p, lane, lo_a, hi_a, lo_b, hi_b);
        } else {
            if (sg.y > 0) {
                if (sg.z >= 1) { atomicAdd(&logmel[0][sg.z - 1], lo_a); atomicAdd(&logmel[1][sg.z - 1], lo_b); }
                if (sg.z < p.n_chan) { atomicAdd(&logmel[0][sg.z], hi_a); atomicAdd(&logmel[1][sg.z], hi_b); }
            }
            wave_lds_fence();
            if (lane < p.n_chan) {
                logmel[0][lane] = __logf(logmel[0][lane]);               // :171, as in mel_channel_sums
                logmel[1][lane] = __logf(logmel[1][lane]);
            }
            wave_lds_fence();
        }
    }
    // DCT-II (:178-182) and lifter (:189), both frames off one pass over the table
    {
        const int i = ci, part = cpart, step = cstep;
        double acc_a = 0.0, acc_b = 0.0;
        if (i < p.n_cep) {
#pragma unroll
            for (int t = 0; t < 10; t++) {
                const int k = min(part + step * t, p.n_chan - 1);        // past the last channel: dc[t] is 0 -- but ln can be -inf
                const float la = logmel[0][k], lb = logmel[1][k];
                if (part + step * t < p.n_chan) {
                    acc_a += dc[t] * (double)la;
                    acc_b += dc[t] * (double)lb;
                }
            }
#pragma unroll 4
            for (int k = part + 10 * step; k < p.n_chan; k += step) {
                const double c = p.dct[k * 32 + i];
                acc_a += c * (double)logmel[0][k];
                acc_b += c * (double)logmel[1][k];
            }
        }
        // lane ^ 16 and lane ^ 32 partners by v_permlane16/32_swap (one VALU instruction per dword; a __shfl_xor is
        // a ds_bpermute, 8.9 issue slots each on this chip: tools/valu_rate.hip)
        if (!wide) { acc_a = sum_xor16_f64(acc_a); acc_b = sum_xor16_f64(acc_b); }
        acc_a = sum_xor32_f64(acc_a); acc_b = sum_xor32_f64(acc_b);
        if (lane < p.n_cep) {
            feats[fa * p.n_cep + lane] = acc_a * lw;
            if (two) feats[fb * p.n_cep + lane] = acc_b * lw;
        }
    }
}

template <bool ALIGNED, int PL>
__device__ __forceinline__ void mfcc_x2_body(const short *__restrict__ src_a, const short *__restrict__ src_b, long fa, long fb, bool two,
                                             const MfccDev &p, const float2 *__restrict__ table, double *__restrict__ feats,
                                             float2 (*lds)[kWaveLdsComplex], int lane)
{
    float (*logmel)[64] = reinterpret_cast<float (*)[64]>(reinterpret_cast<float *>(lds[0]) + 640);       // [2][64]
    float (*pieces)[2][64] = reinterpret_cast<float (*)[2][64]>(reinterpret_cast<float *>(lds[1]) + 640); // [2][2][64]
    // every load the wave needs before its tail, requested in the order of use and none waited for yet (mfcc_fetch)
    MfccRaw<ALIGNED> raw_a, raw_b;
    mfcc_fetch(src_a, p, lane, raw_a);
    mfcc_fetch(src_b, p, lane, raw_b);
    float2 win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = p.window[lane + 64 * r];
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
#if !JDSP_MFCC_X2_PAIRS
    const float2 wsp0 = table[kStftSplit + 2 * lane], wsp1 = table[kStftSplit + 2 * lane + 1];
#endif
    float2 va[8], vb[8];
    mfcc_finish(raw_a, win, p.preemph, lane, va);
    mfcc_finish(raw_b, win, p.preemph, lane, vb);
    // (requested once the samples' registers are free: they are not needed before the transforms are done)
    // this lane's piece of the filterbank: bins [sg.x, sg.x + sg.y) of channel index sg.z, weights sw[t] towards
    // channel sg.z - 1 and cw[t] = 1 - sw[t] towards channel sg.z; both are ZERO past the piece's last bin, so the
    // sixteen steps below need neither a branch nor a select (the branchy form compiled to 94 exec-mask regions
    // with a dependent LDS read and an s_waitcnt in each: the kernel spent 65 % of its wave cycles waiting)
    MfccLaneTables<PL> mt;
    mfcc_load_lane_tables(mt, p, lane);
    const double lw = p.lifter_w[lane & 31];

    wave_fft512_x2<false>(va, vb, lds[0], lds[1], lane, tw);
#if JDSP_MFCC_X2_PAIRS
    // |X[m]| and |X[512 - m]| (= |X[m + 512]|) for m = lane + 64 d, d < 5 -- bins 0..319 and 193..511 -- from the
    // pair-owned split (frame_io.h): five mirror operands per frame through LDS instead of a natural-order image and
    // two reads of it (11 KB instead of 24 KB per pair of frames through the LDS pipe, which bounds this kernel)
    float *mag_a = reinterpret_cast<float *>(lds[0]), *mag_b = reinterpret_cast<float *>(lds[1]);
    {
        PairTwiddles pw;
        load_pair_twiddles(pw, table, lane);
        float2 za[5], zb[5];
        wave_lds_fence();
#pragma unroll
        for (int d = 3; d < 8; d++) { xchg_st(lds[0], lane + 64 * d, va[d]); xchg_st(lds[1], lane + 64 * d, vb[d]); }
        if (lane == 0) { lds[0][512] = va[0]; lds[1][512] = vb[0]; }
        wave_lds_fence();
#pragma unroll
        for (int d = 0; d < 5; d++) { za[d] = xchg_ld(lds[0], 512 - lane - 64 * d); zb[d] = xchg_ld(lds[1], 512 - lane - 64 * d); }
        wave_lds_fence();
        logmel[0][lane] = 0.f;                                       // (their home overlaps the image just read)
        logmel[1][lane] = 0.f;
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const int m = lane + 64 * d, mm = 512 - m;               // mm = 512 (m = 0): X[512], not a mel bin
            const bool pick = p.bin_stride == 1 || !(m & 1);         // 512-point bins = even 1024-point bins
            const int hm = p.bin_stride == 1 ? m : m >> 1, hmm = p.bin_stride == 1 ? mm : mm >> 1;
            const int qm = hm + (hm >> 4), qmm = hmm + (hmm >> 4);
            {
                const float2 e = cadd_conj(va[d], za[d]), o = csub_conj_mj(va[d], za[d]);
                const float2 t = cmul(pw.w[d], o);
                const float2 lo = cadd(e, t), hi = csub(e, t);
                if (pick) mag_a[qm] = __builtin_amdgcn_sqrtf(lo.x * lo.x + lo.y * lo.y);
                if (pick && mm < 512) mag_a[qmm] = __builtin_amdgcn_sqrtf(hi.x * hi.x + hi.y * hi.y);
            }
            {
                const float2 e = cadd_conj(vb[d], zb[d]), o = csub_conj_mj(vb[d], zb[d]);
                const float2 t = cmul(pw.w[d], o);
                const float2 lo = cadd(e, t), hi = csub(e, t);
                if (pick) mag_b[qm] = __builtin_amdgcn_sqrtf(lo.x * lo.x + lo.y * lo.y);
                if (pick && mm < 512) mag_b[qmm] = __builtin_amdgcn_sqrtf(hi.x * hi.x + hi.y * hi.y);
            }
        }
    }
#else
    store_natural_image(lds[0], lane, va);
    store_natural_image(lds[1], lane, vb);
    wave_lds_fence();
    float2 amp_a[4], amp_b[4];
    mfcc_magnitudes(lds[0], lane, wsp0, wsp1, amp_a);
    mfcc_magnitudes(lds[1], lane, wsp0, wsp1, amp_b);
    wave_lds_fence();                                                // every lane's split reads are done: overwrite
    logmel[0][lane] = 0.f;
    logmel[1][lane] = 0.f;
    float *mag_a = reinterpret_cast<float *>(lds[0]), *mag_b = reinterpret_cast<float *>(lds[1]);
    // |X| is stored PADDED, bin i at i + (i >> 4): the filterbank below reads it one piece per lane, and the pieces
    // of the wide upper channels start 16 bins apart -- unpadded, those lanes' addresses are 16 words apart and fall
    // on two banks (the counter pass showed the LDS pipe busy 72 % of the kernel, 18 % of it bank conflicts)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int m = 128 * j + 2 * lane;
        if (p.bin_stride == 1) {
            const int q = m + (m >> 4);                              // m even: m and m + 1 share a group of 16
            mag_a[q] = amp_a[j].x; mag_a[q + 1] = amp_a[j].y;
            mag_b[q] = amp_b[j].x; mag_b[q + 1] = amp_b[j].y;
        } else {                                                     // 512-point bins = even 1024-point bins
            const int h = m >> 1, q = h + (h >> 4);
            mag_a[q] = amp_a[j].x;
            mag_b[q] = amp_b[j].x;
        }
    }
#endif
    wave_lds_fence();
    // the DCT coefficients this lane will need (up to ten channels per quarter of the wave: 40 channels), requested
    // now -- the transforms' registers are free again -- so that they have arrived when the channel logarithms have
    double dc[10];
    mfcc_load_dct(dc, p, lane);
    mfcc_tail_pre(mag_a, mag_b, logmel, pieces, p, lane, mt, dc, lw, fa, fb, two, feats);
}

template <int PL>
__global__ __launch_bounds__(64) void mfcc_x2_kernel(const short *__restrict__ pcm, const long long *__restrict__ starts,
                                                     long n_frames, MfccDev p, const float2 *__restrict__ table,
                                                     double *__restrict__ feats)
{
    // 9,344 B per wave: the two transform scratches; |X| (544 floats each), the filterbank pieces and the channel
    // logarithms live in the scratches' second halves (seventeen waves per CU instead of fourteen)
    __shared__ __attribute__((aligned(16))) float2 lds[2][kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long fa = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * 2;
    if (fa >= n_frames) return;
    const bool two = fa + 1 < n_frames;
    const long fb = two ? fa + 1 : fa;                               // odd tail: the second slot repeats the first
    const short *src_a = pcm + (starts ? starts[fa] : (long long)p.hop * fa);
    const short *src_b = pcm + (starts ? starts[fb] : (long long)p.hop * fb);
    // full-length window and both frames 4-byte aligned: one dword per sample pair (wave-uniform)
    if (p.win_len == 1024 && ((((uintptr_t)src_a) | ((uintptr_t)src_b)) & 3u) == 0)
        mfcc_x2_body<true, PL>(src_a, src_b, fa, fb, two, p, table, feats, lds, lane);
    else
        mfcc_x2_body<false, PL>(src_a, src_b, fa, fb, two, p, table, feats, lds, lane);
}

// ---- persistent waves, spectrum in registers ------------------------------------------------------------------------
// (Round 2, when mfcc_x2_kernel still spent 44 % of its wave cycles in s_waitcnt:) the kernels below keep the tables in
// registers over a grid-stride loop of frame pairs, request the next pair's samples before this pair's arithmetic, take
// |X| from registers (frame_io.h: mirror operands by pair_fetch_lds, no natural-order image) and share a filterbank /
// ln / DCT / lifter tail.  The persistent form stays selectable (JDSP_MFCC512_ONE = 0); production is the pair kernel.
struct MelPiece { int4 sg; float sw[16], cw[16]; };

__device__ __forceinline__ void load_mel_piece(MelPiece &m, const MfccDev &p, int lane)
{
    m.sg = p.seg[lane];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float4 a = p.seg_wc[q * 64 + lane];
        m.sw[2 * q] = a.x; m.cw[2 * q] = a.y; m.sw[2 * q + 1] = a.z; m.cw[2 * q + 1] = a.w;
    }
}

// |X| of two frames, bin i at mag[i + (i >> 4)] (see mfcc_x2_kernel) -> feats[fa], feats[fb]
__device__ __forceinline__ void mfcc_tail_x2(const float *mag_a, const float *mag_b, float (*logmel)[64], float (*pieces)[2][64],
                                             const MfccDev &p, int lane, const MelPiece &mp, long fa, long fb, bool two,
                                             double *__restrict__ feats)
{
    {
        float lo_a = 0.f, hi_a = 0.f, lo_b = 0.f, hi_b = 0.f;
        float ma[16], mb[16];
        const int last = p.n_bins - 1;
#pragma unroll
        for (int t = 0; t < 16; t++) {                               // all thirty-two reads in flight together
            const int bin = min(mp.sg.x + t, last);                  // past the piece: a finite value, its weights are 0
            const int q = bin + (bin >> 4);
            ma[t] = mag_a[q];
            mb[t] = mag_b[q];
        }
#pragma unroll
        for (int t = 0; t < 16; t++) {
            lo_a = fmaf(mp.sw[t], ma[t], lo_a); hi_a = fmaf(mp.cw[t], ma[t], hi_a);   // :164 / :161,:165-166
            lo_b = fmaf(mp.sw[t], mb[t], lo_b); hi_b = fmaf(mp.cw[t], mb[t], hi_b);
        }
        if (p.chan_ok) {
            if (mp.sg.y <= 0) { lo_a = hi_a = lo_b = hi_b = 0.f; }
            mel_channel_sums(pieces, logmel, p, lane, lo_a, hi_a, lo_b, hi_b);
        } else {
            if (mp.sg.y > 0) {
                if (mp.sg.z >= 1) { atomicAdd(&logmel[0][mp.sg.z - 1], lo_a); atomicAdd(&logmel[1][mp.sg.z - 1], lo_b); }
                if (mp.sg.z < p.n_chan) { atomicAdd(&logmel[0][mp.sg.z], hi_a); atomicAdd(&logmel[1][mp.sg.z], hi_b); }
            }
            wave_lds_fence();
            if (lane < p.n_chan) {                                   // :171, hardware log2 (see mel_channel_sums)
                logmel[0][lane] = __logf(logmel[0][lane]);
                logmel[1][lane] = __logf(logmel[1][lane]);
            }
            wave_lds_fence();
        }
    }
    // (requesting the DCT coefficients ahead of the filterbank, as mfcc_x2_kernel does, costs this kernel 20 registers
    // it does not have: 73 spills at three waves per SIMD, 168 us)
    const bool wide = p.n_cep > 16;                                  // DCT-II (:178-182) and lifter (:189), see mfcc_kernel
    const int i = wide ? (lane & 31) : (lane & 15), part = wide ? (lane >> 5) : (lane >> 4), step = wide ? 2 : 4;
    double acc_a = 0.0, acc_b = 0.0;
    if (i < p.n_cep) {
#pragma unroll 4
        for (int k = part; k < p.n_chan; k += step) {
            const double c = p.dct[k * 32 + i];
            acc_a += c * (double)logmel[0][k];
            acc_b += c * (double)logmel[1][k];
        }
    }
    if (!wide) { acc_a = sum_xor16_f64(acc_a); acc_b = sum_xor16_f64(acc_b); }
    acc_a = sum_xor32_f64(acc_a); acc_b = sum_xor32_f64(acc_b);
    if (lane < p.n_cep) {
        const double lw = p.lifter_w[lane];
        feats[fa * p.n_cep + lane] = acc_a * lw;
        if (two) feats[fb * p.n_cep + lane] = acc_b * lw;
    }
    wave_lds_fence();                                                // logmel and the magnitudes are rewritten next
}

// sum over the wave, the same value in every lane's copy of lane 63 (five DPP adds and a v_readlane; a __shfl_xor tree
// is six ds_bpermute, 8.9 issue slots each)
__device__ __forceinline__ float wave_sum_f32(float v)
{
#define JDSP_DPP_ADD(CTRL, ROWS) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, true))
    JDSP_DPP_ADD(0xB1, 0xf);     // quad_perm [1,0,3,2]
    JDSP_DPP_ADD(0x4E, 0xf);     // quad_perm [2,3,0,1]
    JDSP_DPP_ADD(0x141, 0xf);    // row_half_mirror: sums of 8
    JDSP_DPP_ADD(0x140, 0xf);    // row_mirror: sums of 16
    JDSP_DPP_ADD(0x142, 0xa);    // row_bcast15 into rows 1 and 3
    JDSP_DPP_ADD(0x143, 0xc);    // row_bcast31 into rows 2 and 3
#undef JDSP_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// previous lane's value (wave_shr:1); lane 0 gets `lane0`
__device__ __forceinline__ float prev_lane(float v, float lane0, int lane)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true);
    return lane == 0 ? lane0 : __int_as_float(t);
}

// Two frames' samples (lane + 64 r) -> pre-emphasis (:208-210), window, one transform, |A[k]|, |B[k]| for k = lane + 64 d,
// d < 4.  Returns whether the two frames' energies are more than 36 dB apart (wave-uniform).
__device__ __forceinline__ bool mfcc512_pair_mags(const float (&sa)[8], const float (&sb)[8], float preemph,
                                                  const float (&win)[8], const WaveTwiddles &tw, float2 *lds, int lane,
                                                  float (&ma)[4], float (&mb)[4])
{
    // x[i] = s[i] - preemph * s[i-1] for 1 <= i < win_len, x[0] = 0 (:208 starts at i = 1); the window is zero beyond
    float2 v[8];
    float ea = 0.f, eb = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float pa = prev_lane(sa[r], r > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa[r > 0 ? r - 1 : 0]), 63)) : 0.f, lane);
        const float pb = prev_lane(sb[r], r > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sb[r > 0 ? r - 1 : 0]), 63)) : 0.f, lane);
        float xa = (sa[r] - preemph * pa) * win[r], xb = (sb[r] - preemph * pb) * win[r];
        if (r == 0 && lane == 0) { xa = 0.f; xb = 0.f; }
        ea = fmaf(xa, xa, ea);
        eb = fmaf(xb, xb, eb);
        v[r] = make_float2(xa, xb);
    }
    ea = wave_sum_f32(ea);
    eb = wave_sum_f32(eb);
    wave_fft512<false>(v, lds, lane, tw);
    wave_lds_fence();
    // bins k = lane + 64 d, d < 4 (0..255): mirrors Z[512 - k] are registers 4..7 of other lanes and Z[512] = Z[0]
#pragma unroll
    for (int d = 4; d < 8; d++) xchg_st(lds, lane + 64 * d, v[d]);
    if (lane == 0) lds[512] = v[0];
    wave_lds_fence();
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const float2 zm = xchg_ld(lds, 512 - lane - 64 * d);
        const float2 A = cadd_conj(v[d], zm), B = csub_conj_mj(v[d], zm);
        ma[d] = __builtin_amdgcn_sqrtf(A.x * A.x + A.y * A.y);   // hardware square root, 1 ulp (see mfcc_kernel)
        mb[d] = __builtin_amdgcn_sqrtf(B.x * B.x + B.y * B.y);
    }
    wave_lds_fence();
    return !(ea <= 4096.f * eb && eb <= 4096.f * ea);            // also when exactly one frame is all zeros
}

#ifndef JDSP_MFCC512_WAVES
#define JDSP_MFCC512_WAVES 3
#endif
// n_fft = 512 (BASELINE config 4: 400-sample window, 512-FFT): TWO frames per 512-point transform, z[n] = a[n] + j b[n],
// A[k] = (Z[k] + conj Z[512-k]) / 2, B[k] = -j (Z[k] - conj Z[512-k]) / 2 (the 1/2 is in the window table) -- half the
// transform work of the zero-padded 1024-point form mfcc_x2_kernel uses for this configuration.  Lane l holds sample
// l + 64 r of both frames (2-byte loads: any frame start), the sample before it comes from lane l - 1.
// The two spectra come apart exactly only in exact arithmetic: frame b's rounding (6e-8 of ITS magnitudes) lands in
// frame a's bins.  Between neighbours of similar level that is the FP32 noise floor the 1e-5 bar already allows for;
// when the two frames' energies differ by more than 36 dB (or one of them is all zeros, whose ln 0 = -inf, :171, must
// stay exact) the pair's index goes onto a list and mfcc_kernel computes both frames again, one per transform (a branch
// inside this kernel cost 15 % of its speed in registers: profiles/r02_mfcc512_run.txt).
__global__ __launch_bounds__(64, JDSP_MFCC512_WAVES) void mfcc512_run_kernel(const short *__restrict__ pcm, const long long *__restrict__ starts,
                                                            long n_frames, MfccDev p, const float2 *__restrict__ table,
                                                            double *__restrict__ feats, int *__restrict__ redo)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ float logmel[2][64];
    __shared__ float pieces[2][2][64];
    const int lane = threadIdx.x;
    const long n_pairs = (n_frames + 1) >> 1;
    if ((long)blockIdx.x >= n_pairs) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = reinterpret_cast<const float *>(p.window)[lane + 64 * r];   // zero from win_len on
    MelPiece mp;
    load_mel_piece(mp, p, lane);
    const int rows = (p.win_len + 63) >> 6;                          // register rows that hold samples (wave-uniform)
    float na[8], nb[8];                                              // the next pair's samples
    auto fetch = [&](long q) {
        const long fa = 2 * q, fb = fa + 1 < n_frames ? fa + 1 : fa;
        const short *sa = pcm + (starts ? starts[fa] : (long long)p.hop * fa) + lane;
        const short *sb = pcm + (starts ? starts[fb] : (long long)p.hop * fb) + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const bool in = r < rows - 1 || (r == rows - 1 && lane + 64 * r < p.win_len);
            na[r] = in ? (float)sa[64 * r] : 0.f;
            nb[r] = in ? (float)sb[64 * r] : 0.f;
        }
    };
    fetch(blockIdx.x);
    for (long q = blockIdx.x; q < n_pairs; q += gridDim.x) {
        const long fa = 2 * q;
        const bool two = fa + 1 < n_frames;
        const long fb = two ? fa + 1 : fa;                           // odd tail: the second slot repeats the first
        float sa[8], sb[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { sa[r] = na[r]; sb[r] = nb[r]; }
        if (q + gridDim.x < n_pairs) fetch(q + gridDim.x);
        float ma[4], mb[4];
        if (mfcc512_pair_mags(sa, sb, p.preemph, win, tw, lds, lane, ma, mb) && lane == 0)
            redo[1 + atomicAdd(redo, 1)] = (int)q;                   // rare: both frames again, apart (mfcc_kernel)
        float *mag_a = reinterpret_cast<float *>(lds), *mag_b = mag_a + 320;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int k = lane + 64 * d, qk = k + (k >> 4);
            mag_a[qk] = ma[d];
            mag_b[qk] = mb[d];
        }
        logmel[0][lane] = 0.f;
        logmel[1][lane] = 0.f;
        wave_lds_fence();
        mfcc_tail_x2(mag_a, mag_b, logmel, pieces, p, lane, mp, fa, fb, two, feats);
    }
}

// The same pair of frames, ONE pair per wave: nothing is kept between pairs, so the tables are loaded where they are
// used and die there -- 72 registers instead of 168, seven waves per SIMD instead of three.  Round 2: 53 us per
// 65,536 frames against the persistent kernel's 78 (profiles/r02_mfcc512_run.txt); round 3, with every load of the
// wave unguarded and in flight together and the pieces cut to the filterbank: 43 us, the 10,000-utterance batch 2.05 ms
// (profiles/r03_guarded_loads.txt).
// JDSP_MFCC512_ONE = 0 selects the persistent kernel.
#ifndef JDSP_MFCC512_ONE
#define JDSP_MFCC512_ONE 1
#endif
#ifndef JDSP_MFCC512_EARLY_TABLES
#define JDSP_MFCC512_EARLY_TABLES 0      // 1: 128 registers, four waves per SIMD, 50.5 us per 65,536 frames; 0: 82, five waves, 48.1 us
#endif
template <int PL>
__global__ __launch_bounds__(64) void mfcc512_pair_kernel(const short *__restrict__ pcm, const long long *__restrict__ starts,
                                                          long n_frames, MfccDev p, const float2 *__restrict__ table,
                                                          double *__restrict__ feats, int *__restrict__ redo)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    float (*logmel)[64] = reinterpret_cast<float (*)[64]>(reinterpret_cast<float *>(lds) + 768);          // [2][64]
    float (*pieces)[2][64] = reinterpret_cast<float (*)[2][64]>(reinterpret_cast<float *>(lds) + 896);    // [2][2][64]
    const int lane = threadIdx.x;
    const long n_pairs = (n_frames + 1) >> 1;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long q = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (q >= n_pairs) return;
    const long fa = 2 * q;
    const bool two = fa + 1 < n_frames;
    const long fb = two ? fa + 1 : fa;
    // Every load of the wave is requested before the first is waited for, in the order of use: samples, window,
    // twiddles, then (JDSP_MFCC512_EARLY_TABLES) the filterbank piece and the DCT column, which the tail needs only
    // after the transform.  The samples are read UNGUARDED at positions clamped into the window -- the window table is
    // zero from win_len on, so what a clamped position returns never counts.  (Guarded, each halfword was a branch
    // with an s_waitcnt vmcnt(0) of its own: sixteen memory round trips in series per wave, hidden only by running
    // seven waves per SIMD.)
    float sa[8], sb[8];
    {
        const short *pa = pcm + (starts ? starts[fa] : (long long)p.hop * fa);
        const short *pb = pcm + (starts ? starts[fb] : (long long)p.hop * fb);
        const int last = p.win_len - 1;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int i = min(lane + 64 * r, last);
            sa[r] = (float)pa[i];
            sb[r] = (float)pb[i];
        }
    }
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = reinterpret_cast<const float *>(p.window)[lane + 64 * r];
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    MfccLaneTables<PL> mt;
    double dc[10];
#if JDSP_MFCC512_EARLY_TABLES
    mfcc_load_lane_tables(mt, p, lane);
    mfcc_load_dct(dc, p, lane);
#endif
    const double lw = p.lifter_w[lane & 31];
    float ma[4], mb[4];
    if (mfcc512_pair_mags(sa, sb, p.preemph, win, tw, lds, lane, ma, mb) && lane == 0)
        redo[1 + atomicAdd(redo, 1)] = (int)q;
#if !JDSP_MFCC512_EARLY_TABLES
    mfcc_load_lane_tables(mt, p, lane);
    mfcc_load_dct(dc, p, lane);
#endif
    float *mag_a = reinterpret_cast<float *>(lds), *mag_b = mag_a + 320;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const int k = lane + 64 * d, qk = k + (k >> 4);
        mag_a[qk] = ma[d];
        mag_b[qk] = mb[d];
    }
    logmel[0][lane] = 0.f;                                           // (the atomics of the !chan_ok form add to them)
    logmel[1][lane] = 0.f;
    wave_lds_fence();
    mfcc_tail_pre(mag_a, mag_b, logmel, pieces, p, lane, mt, dc, lw, fa, fb, two, feats);
}

// (Tried for n_fft = 1024 too -- two frames per iteration through wave_fft512_x2, pair-owned |X|, tables in registers:
// 206 registers = two waves per SIMD, 102 us per 65,536 frames against mfcc_x2_kernel's 97 us; at three waves, with
// spills, 154 us.  And the other way round -- ONE frame per wave with nothing kept, as mfcc512_pair_kernel does: 85
// registers, five or six waves per SIMD, 119-134 us: twice the table loads per frame.  profiles/r02_mfcc512_run.txt.
// The 1024-point configurations stay with mfcc_x2_kernel.)

int launch_mfcc(hipStream_t s, const short *pcm, const long long *starts, long n_frames, const MfccDev &p,
                const float2 *table, double *feats, int *redo)
{
    if (n_frames <= 0) return 0;
#ifndef JDSP_MFCC_X2
#define JDSP_MFCC_X2 1             // 1: two frames per wavefront in lock-step (mfcc_x2_kernel); 0: one frame per wavefront
#endif
#ifndef JDSP_MFCC_RUN
#define JDSP_MFCC_RUN 1            // 1: persistent register-resident kernels where they apply
#endif
    if (JDSP_MFCC_RUN && p.seg_ok && p.bin_stride == 2 && p.win_len <= 512 && redo) {
        const long n_pairs = (n_frames + 1) / 2;
        if (hipMemsetAsync(redo, 0, sizeof(int), s) != hipSuccess) return -1;
        if (JDSP_MFCC512_ONE) {
            const dim3 grid((unsigned)((n_pairs + 7) / 8 * 8));
            if (p.piece_len <= 8) hipLaunchKernelGGL(mfcc512_pair_kernel<8>, grid, dim3(64), 0, s, pcm, starts, n_frames, p, table, feats, redo);
            else if (p.piece_len <= 12) hipLaunchKernelGGL(mfcc512_pair_kernel<12>, grid, dim3(64), 0, s, pcm, starts, n_frames, p, table, feats, redo);
            else hipLaunchKernelGGL(mfcc512_pair_kernel<16>, grid, dim3(64), 0, s, pcm, starts, n_frames, p, table, feats, redo);
        } else {
            const long slots = 1024L * JDSP_MFCC512_WAVES;          // resident waves of a 256-CU part
            const long grid = n_pairs < slots ? n_pairs : slots;
            hipLaunchKernelGGL(mfcc512_run_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats, redo);
        }
        hipLaunchKernelGGL(mfcc_kernel, dim3(1024), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats, (const int *)redo);
    } else if (JDSP_MFCC_X2 && p.seg_ok) {
        const long grid = ((n_frames + 1) / 2 + 7) / 8 * 8;
        if (p.piece_len <= 8) hipLaunchKernelGGL(mfcc_x2_kernel<8>, dim3((unsigned)grid), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats);
        else if (p.piece_len <= 12) hipLaunchKernelGGL(mfcc_x2_kernel<12>, dim3((unsigned)grid), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats);
        else hipLaunchKernelGGL(mfcc_x2_kernel<16>, dim3((unsigned)grid), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats);
    } else {                              // filterbanks that do not fit one piece per lane (many narrow channels)
        const long grid = (n_frames + 7) / 8 * 8;
        hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, starts, n_frames, p, table, feats,
                           (const int *)nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
