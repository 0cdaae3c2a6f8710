// fastconv_kernels.hip -- Fast_Convolution_Based_3DAudio_Impl.cpp:102-177 on gfx950:
// overlap-save fast convolution, y = IFFT(FFT(x_segment) * FFT(h)) / n_fft, last `block`
// samples of every segment kept (:156-158).
//
//   n_fft = 8192 (reference-native: 7169-tap room impulse response, 1024-sample blocks):
//       one 512-thread workgroup per output block, the real segment as a 4096-point complex
//       FFT = one radix-8 pass across the workgroup + eight 512-point wave FFTs, in LDS.
//   n_fft = 1024 (BASELINE config 2: 256-tap HRIR pair, 769-sample blocks):
//       one wavefront per output block, wave_fft512 forward once and inverse once per filter.
//
// The filter spectrum is computed ONCE per handle (the reference recomputes it for every
// block, :140,:143) and the unused atan2 pass (:145-147) is dropped.
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

__device__ __forceinline__ float conv_sample(const ConvStream &s, long pos)
{
    if (pos + s.global0 < s.valid_from) return 0.f;
    if (pos >= 0) return pos < s.n_samples ? (float)s.pcm[pos] : 0.f;
    const long h = pos + s.hist_len;
    return h >= 0 ? (float)s.hist[h] : 0.f;
}

__global__ void spectrum_to_f32_kernel(const double2 *__restrict__ in, float2 *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2((float)in[i].x, (float)in[i].y);
}

__global__ void conv_hist_update_kernel(ConvStream s, short *__restrict__ hist_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // new hist[i] = sample at pos n_samples - hist_len + i
    if (i >= s.hist_len) return;
    const long pos = s.n_samples - s.hist_len + i;
    short v = 0;
    if (pos >= 0) v = s.pcm[pos];
    else if (pos + s.hist_len >= 0) v = s.hist[pos + s.hist_len];
    hist_out[i] = v;
}

// ---------------------------------------------------------------------------------------
// n_fft = 1024: one wave per output block, NF filters sharing the forward transform.
template <int J>
__device__ __forceinline__ void conv_mul_presplit_j(const float2 *lds, float2 *zout, int lane, const float2 *wsp,
                                                    const float2 *__restrict__ H)
{
    const int m = 128 * J + 2 * lane;
    const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
    float2 zr0, zr1;
    load_mirror_pair(lds, m, zr0, zr1);
    float2 lo0, hi0, lo1, hi1;
    split_fwd<J>(make_float2(zz.x, zz.y), zr0, wsp[0], lo0, hi0);
    split_fwd<J>(make_float2(zz.z, zz.w), zr1, wsp[1], lo1, hi1);
    const float4 hl = *reinterpret_cast<const float4 *>(H + m);          // H[m], H[m+1]
    const float4 hh = *reinterpret_cast<const float4 *>(H + m + 512);
    lo0 = cmul(lo0, make_float2(hl.x, hl.y)); lo1 = cmul(lo1, make_float2(hl.z, hl.w));   // :150-151
    hi0 = cmul(hi0, make_float2(hh.x, hh.y)); hi1 = cmul(hi1, make_float2(hh.z, hh.w));
    zout[2 * J] = presplit_inv<J>(lo0, hi0, wsp[0]);
    zout[2 * J + 1] = presplit_inv<J>(lo1, hi1, wsp[1]);
}

__global__ __launch_bounds__(64) void fastconv1024_kernel(ConvStream s, long n_out_blocks, int first_block, int block,
                                                          int n_taps, int n_filters, const float2 *__restrict__ Hall,
                                                          const float2 *__restrict__ table, short *__restrict__ out,
                                                          float *__restrict__ precast, long plane)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) float2 spec[520];       // natural-order image + Z[512]
    const int lane = threadIdx.x;
    const long e = blockIdx.x;
    if (e >= n_out_blocks) return;
    const long end = (long)(first_block + e + 1) * block;       // one past the newest sample of the segment
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    float2 v[8];
    // common case: the whole 1024-sample segment lies inside this call's buffer and past the
    // stream's silent head -> plain 16-bit loads, no per-sample tests (wave-uniform branch)
    if (end - 1024 >= 0 && end <= s.n_samples && end - 1024 + s.global0 >= s.valid_from) {
        const short *src = s.pcm + (end - 1024) + 2 * lane;
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = make_float2(0.5f * (float)src[128 * r], 0.5f * (float)src[128 * r + 1]);
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const long p0 = end - 1024 + 2 * lane + 128 * r;
            v[r] = make_float2(0.5f * conv_sample(s, p0), 0.5f * conv_sample(s, p0 + 1));   // 0.5: split convention
        }
    }
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(spec, lane, v);
    wave_lds_fence();
    for (int f = 0; f < n_filters; f++) {
        const float2 *H = Hall + (size_t)f * 1024;
        float2 z[8], y[8];
        conv_mul_presplit_j<0>(spec, z, lane, wsp, H);
        conv_mul_presplit_j<1>(spec, z, lane, wsp, H);
        conv_mul_presplit_j<2>(spec, z, lane, wsp, H);
        conv_mul_presplit_j<3>(spec, z, lane, wsp, H);
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
        wave_lds_fence();
        wave_fft512<true>(y, lds, lane, tw);
        wave_lds_fence();
        // keep y[n_taps-1 .. 1023] / 1024 (:156-158): the samples go through LDS so that the
        // kept `block` of them leaves as consecutive 16-bit stores (128 B per wave instruction)
        // whatever the alignment of this block in the output plane
        short *o = out + (size_t)f * plane + e * block;
        float *pc = precast ? precast + (size_t)f * plane + e * block : nullptr;
        float *ys = reinterpret_cast<float *>(lds);                  // 1024 floats fit the wave's scratch
#pragma unroll
        for (int d = 0; d < 8; d++)
            *reinterpret_cast<float2 *>(ys + 2 * lane + 128 * d) = make_float2(y[d].x * (1.0f / 1024.0f), y[d].y * (1.0f / 1024.0f));
        wave_lds_fence();
        for (int i = lane; i < block; i += 64) {
            const float a = ys[i + n_taps - 1];
            o[i] = (short)cast_i16_bits(a);
            if (pc) pc[i] = a;
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------
// n_fft = 8192.  512 threads; thread t (wave w = t >> 6, lane l = t & 63).
// 4096-point complex DFT of z[n], n = 512 n1 + n2, k = k1 + 8 k2:
//   pass A (thread n2 = t):  A[k1][n2] = (sum_n1 z[512 n1 + n2] w_8^(n1 k1)) * w_4096^(n2 k1)
//   pass B (wave k1):        Z[k1 + 8 k2] = FFT512(A[k1][.])[k2]        (wave_fft512)
// LDS holds rows[k1][k2] with pitch kRow; a wave uses its own row as wave_fft512's scratch.
constexpr int kRow = 585;

template <bool INV>
__device__ __forceinline__ void wg_fft4096(float2 (&v)[8], float2 *rows, int t, const float2 *__restrict__ tw4096,
                                           const WaveTwiddles &tw)
{
    const int w = t >> 6, l = t & 63;
    dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        const float2 c = tw4096[(t * k) & 4095];
        v[k] = INV ? cmul_conj(v[k], c) : cmul(v[k], c);
    }
    __syncthreads();                                   // everyone is done with the rows' previous contents
#pragma unroll
    for (int k = 0; k < 8; k++) rows[k * kRow + t] = v[k];
    __syncthreads();
    float2 *mine = rows + w * kRow;
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = mine[l + 64 * r];
    wave_lds_fence();
    wave_fft512<INV>(v, mine, l, tw);
    wave_lds_fence();
#pragma unroll
    for (int d = 0; d < 8; d++) mine[l + 64 * d] = v[d];      // rows[k1][k2] = Z[k1 + 8 k2]
    __syncthreads();
}

__device__ __forceinline__ float2 row_at(const float2 *rows, int k) { return rows[(k & 7) * kRow + (k >> 3)]; }

#ifndef JDSP_CONV8192_MINWAVES
#define JDSP_CONV8192_MINWAVES 2
#endif

// split of the 4096-point packed transform + multiply by the filter spectrum + inverse
// pre-split for bin m and its partner m + 4096 (the inverse transform's input layout)
__device__ __forceinline__ void conv8192_bin(const float2 *rows, int m, const float2 *__restrict__ tw8192,
                                             float2 &xl, float2 &xh)
{
    const float2 zm = row_at(rows, m), zc = row_at(rows, (4096 - m) & 4095);
    const float2 ev = make_float2(zm.x + zc.x, zm.y - zc.y);
    const float2 od = make_float2(zm.y + zc.y, zc.x - zm.x);
    const float2 tt = cmul(tw8192[m], od);
    xl = cadd(ev, tt);                        // X[m]
    xh = csub(ev, tt);                        // X[m + 4096]
}

__device__ __forceinline__ float2 conv8192_mul(float2 xl, float2 xh, int m, const float2 *__restrict__ H,
                                               const float2 *__restrict__ tw8192)
{
    const float2 yl = cmul(xl, H[m]), yh = cmul(xh, H[m + 4096]);     // :150-151
    const float2 sm = cadd(yl, yh);
    const float2 df = cmul_conj(csub(yl, yh), tw8192[m]);
    return make_float2(sm.x - df.y, sm.y + df.x);
}

template <bool SINGLE>
__global__ __launch_bounds__(512, JDSP_CONV8192_MINWAVES) void fastconv8192_kernel(
    ConvStream s, long n_out_blocks, int first_block, int block, int n_taps, int n_filters,
    const float2 *__restrict__ Hall, const float2 *__restrict__ table, const float2 *__restrict__ tw4096,
    const float2 *__restrict__ tw8192, short *__restrict__ out, float *__restrict__ precast, long plane)
{
    __shared__ __attribute__((aligned(16))) float2 rows[8 * kRow];
    const int t = threadIdx.x, l = t & 63;
    const long e = blockIdx.x;
    const long end = (long)(first_block + e + 1) * block;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, l);
    float2 v[8];
    // common case (workgroup-uniform): the whole segment is inside this call's buffer, past the
    // stream's silent head and 4-byte aligned -> one dword per sample pair, no per-sample tests
    const long seg0 = end - 8192;
    if (seg0 >= 0 && end <= s.n_samples && seg0 + s.global0 >= s.valid_from && (((uintptr_t)(s.pcm + seg0)) & 3u) == 0) {
        const unsigned int *src = reinterpret_cast<const unsigned int *>(s.pcm + seg0) + t;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 p = unpack_i16x2(src[512 * r]);
            v[r] = make_float2(0.5f * p.x, 0.5f * p.y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const long p0 = seg0 + 2 * (t + 512 * r);
            v[r] = make_float2(0.5f * conv_sample(s, p0), 0.5f * conv_sample(s, p0 + 1));
        }
    }
    wg_fft4096<false>(v, rows, t, tw4096, tw);
    if (SINGLE) {
        float2 y[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int m = t + 512 * r;
            float2 xl, xh;
            conv8192_bin(rows, m, tw8192, xl, xh);
            y[r] = conv8192_mul(xl, xh, m, Hall, tw8192);
        }
        wg_fft4096<true>(y, rows, t, tw4096, tw);
    }
    float2 xl[SINGLE ? 1 : 8], xh[SINGLE ? 1 : 8];
    if (!SINGLE) {
#pragma unroll
        for (int r = 0; r < 8; r++) conv8192_bin(rows, t + 512 * r, tw8192, xl[r], xh[r]);
    }
    for (int f = 0; f < n_filters; f++) {
        if (!SINGLE) {
            float2 y[8];
#pragma unroll
            for (int r = 0; r < 8; r++) y[r] = conv8192_mul(xl[r], xh[r], t + 512 * r, Hall + (size_t)f * 8192, tw8192);
            wg_fft4096<true>(y, rows, t, tw4096, tw);
        }
        // rows[k1][k2] = z'[k1 + 8 k2] = (y[2n], y[2n+1]) * 8192, n = k1 + 8 k2; keep samples n_taps-1 .. 8191
        short *o = out + (size_t)f * plane + e * block;
        float *pc = precast ? precast + (size_t)f * plane + e * block : nullptr;
        for (int i = t; i < block; i += 512) {
            const int smp = i + n_taps - 1;
            const float2 zz = row_at(rows, smp >> 1);
            const float a = ((smp & 1) ? zz.y : zz.x) * (1.0f / 8192.0f);           // :157
            o[i] = (short)cast_i16_bits(a);
            if (pc) pc[i] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------
int launch_spectrum_to_f32(hipStream_t s, const double2 *in, float2 *out, long n)
{
    hipLaunchKernelGGL(spectrum_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fastconv(hipStream_t st, int n_fft, const ConvStream &s, long n_out_blocks, int first_block, int block,
                    int n_taps, int n_filters, const float2 *H, const float2 *table, const float2 *tw4096,
                    const float2 *tw8192, short *out, float *precast, long plane, short *hist_out)
{
    if (n_out_blocks > 0) {
        if (n_fft == 1024)
            hipLaunchKernelGGL(fastconv1024_kernel, dim3((unsigned)n_out_blocks), dim3(64), 0, st, s, n_out_blocks,
                               first_block, block, n_taps, n_filters, H, table, out, precast, plane);
        else if (n_filters == 1)
            hipLaunchKernelGGL(fastconv8192_kernel<true>, dim3((unsigned)n_out_blocks), dim3(512), 0, st, s, n_out_blocks,
                               first_block, block, n_taps, n_filters, H, table, tw4096, tw8192, out, precast, plane);
        else
            hipLaunchKernelGGL(fastconv8192_kernel<false>, dim3((unsigned)n_out_blocks), dim3(512), 0, st, s, n_out_blocks,
                               first_block, block, n_taps, n_filters, H, table, tw4096, tw8192, out, precast, plane);
    }
    if (s.hist_len > 0)
        hipLaunchKernelGGL(conv_hist_update_kernel, dim3((s.hist_len + 255) / 256), dim3(256), 0, st, s, hist_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void fill_conv_twiddles(float2 *tw4096, float2 *tw8192)
{
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < 4096; i++) {
        double a = -two_pi * i / 4096.0, b = -two_pi * i / 8192.0;
        tw4096[i] = make_float2((float)cos(a), (float)sin(a));
        tw8192[i] = make_float2((float)cos(b), (float)sin(b));
    }
}

}  // namespace jdsp
