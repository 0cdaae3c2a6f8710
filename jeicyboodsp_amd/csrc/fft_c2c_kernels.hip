// fft_c2c_kernels.hip -- FFTAlgorithm_ver2.cpp on the device: the Bitrev table
// (:186-202, bit-exact) and batched FFTProcess (:94-149) in double precision.
//
// One workgroup owns one n_fft-point transform held entirely in LDS
// (16 B per point: 128 KB at n_fft = 8192).  The structure follows the
// reference: bit-reversed gather through the Bitrev table, then log2(n)
// decimation-in-time stages.  The reference multiplies the upper half of each
// next-size group by its twiddle in a separate pass after the butterflies
// (:128-145); here that multiply is fused into the following stage's butterfly,
// which is the same arithmetic.  Twiddles come from a host-built table of the
// true pi (the reference truncates PI to 3.14159265358, a 1e-11 effect).
#include "jdsp_internal.h"

namespace jdsp {

// FFTAlgorithm_ver2.cpp:191-202 with the reference's 16-bit variables.
__device__ __forceinline__ short bitrev16(int k, int bits, int n_fft)
{
    short walk = (short)k;
    short rev = walk;
    for (int i = 1; i < bits; i++) {
        walk = (short)(walk >> 1);
        rev = (short)(unsigned short)(((unsigned)(unsigned short)rev) << 1);    // <<= 1 on the 16-bit pattern
        rev = (short)(rev | (walk & 1));
    }
    return (short)(rev & (n_fft - 1));
}

__global__ void bitrev_table_kernel(short *table, int n_fft, int bits)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_fft) table[k] = bitrev16(k, bits, n_fft);
}

// tw[j] = exp(-2*pi*i*j/n_fft), j < n_fft/2
__global__ __launch_bounds__(256) void fft_process_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                             int n_fft, int log2n, int forward,
                                                             const double2 *__restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *x = reinterpret_cast<double2 *>(smem_raw);
    const double2 *src = in + (size_t)blockIdx.x * n_fft;
    double2 *dst = out + (size_t)blockIdx.x * n_fft;
    // Bitrev gather (:204-205), done as a coalesced read + scattered LDS write: the table is an
    // involution for n_fft = 2^log2n, so x[rev[k]] = src[k] is the same permutation
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x)
        x[(unsigned short)bitrev16(k, log2n, n_fft)] = src[k];
    __syncthreads();
    const int half_n = n_fft >> 1;
    for (int s = 0; s < log2n; s++) {
        const int half = 1 << s;
        const int tstep = half_n >> s;                              // twiddle stride: n_fft / (2*half)
        for (int b = threadIdx.x; b < half_n; b += blockDim.x) {
            const int j = b & (half - 1);
            const int i0 = ((b >> s) << (s + 1)) + j;
            double2 w = tw[j * tstep];
            if (!forward) w.y = -w.y;
            const double2 u = x[i0], v = x[i0 + half];
            const double tr = w.x * v.x - w.y * v.y;
            const double ti = w.x * v.y + w.y * v.x;
            x[i0] = make_double2(u.x + tr, u.y + ti);
            x[i0 + half] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x) dst[k] = x[k];
}

// ---- n_fft = 512 (the reference's native BLOCK_LEN, FFT:16): one transform per wavefront -----------------
// Same decomposition as wave_fft512.h (three in-register radix-8 passes, two wave-private LDS exchanges,
// "lane + 64 r" layout in and out) in double precision: 8 KB in, 8 KB out per transform, all loads and
// stores 16 B per lane and coalesced, no workgroup barrier.  The DFT is the same; only the order of the
// additions differs from the reference's radix-2 DIT (1e-16-level rounding).  Twiddles are read from the
// c2c table: w_512^j = tw[j & 255], negated when j & 256.
struct cd {
    double x, y;
};
__device__ __forceinline__ cd cd_add(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd cd_sub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cd_mul(cd a, cd w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
template <bool INV> __device__ __forceinline__ cd cd_rot90(cd a) { return INV ? cd{-a.y, a.x} : cd{a.y, -a.x}; }
template <bool INV> __device__ __forceinline__ cd cd_rot45(cd a)
{
    const double c = 0.70710678118654752440;
    return INV ? cd{c * (a.x - a.y), c * (a.x + a.y)} : cd{c * (a.x + a.y), c * (a.y - a.x)};
}
template <bool INV> __device__ __forceinline__ cd cd_rot135(cd a)
{
    const double c = 0.70710678118654752440;
    return INV ? cd{-c * (a.x + a.y), c * (a.x - a.y)} : cd{c * (a.y - a.x), -c * (a.x + a.y)};
}
template <bool INV> __device__ __forceinline__ void cd_dft8(cd (&v)[8])
{
    cd a0 = cd_add(v[0], v[4]), a4 = cd_sub(v[0], v[4]);
    cd a1 = cd_add(v[1], v[5]), a5 = cd_rot45<INV>(cd_sub(v[1], v[5]));
    cd a2 = cd_add(v[2], v[6]), a6 = cd_rot90<INV>(cd_sub(v[2], v[6]));
    cd a3 = cd_add(v[3], v[7]), a7 = cd_rot135<INV>(cd_sub(v[3], v[7]));
    cd b0 = cd_add(a0, a2), b2 = cd_sub(a0, a2);
    cd b1 = cd_add(a1, a3), b3 = cd_rot90<INV>(cd_sub(a1, a3));
    cd b4 = cd_add(a4, a6), b6 = cd_sub(a4, a6);
    cd b5 = cd_add(a5, a7), b7 = cd_rot90<INV>(cd_sub(a5, a7));
    v[0] = cd_add(b0, b1); v[4] = cd_sub(b0, b1);
    v[2] = cd_add(b2, b3); v[6] = cd_sub(b2, b3);
    v[1] = cd_add(b4, b5); v[5] = cd_sub(b4, b5);
    v[3] = cd_add(b6, b7); v[7] = cd_sub(b6, b7);
}
template <bool INV> __device__ __forceinline__ cd w512(const double2 *__restrict__ tw, int j)
{
    const double2 t = tw[j & 255];
    const double s = (j & 256) ? -1.0 : 1.0;
    return {s * t.x, INV ? -s * t.y : s * t.y};
}
__device__ __forceinline__ void lds_fence_wave()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kFft512Lds = 8 * 73;                        // cd elements per wave (9,344 B), padded layouts
#ifndef JDSP_F64_SWIZZLE
#define JDSP_F64_SWIZZLE 1          // 1: unpadded XOR layouts, 512 elements (8,192 B) per wave: see stft1024_f64_v2_kernel
#endif

template <bool INV>
__global__ __launch_bounds__(64, JDSP_F64_SWIZZLE ? 5 : 4) void fft512_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, long batch,
                                                        const double2 *__restrict__ tw)
{
    // (JDSP_F64_SWIZZLE: the unpadded XOR layouts of stft1024_f64_v2_kernel, 8,192 B per wave, five waves per SIMD)
    __shared__ __attribute__((aligned(16))) cd lds[JDSP_F64_SWIZZLE ? 512 : kFft512Lds];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;            // XCD-aware order: neighbouring transforms share an XCD's L2
    const long t = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= batch) return;
    const double2 *src = in + t * 512 + lane;
    cd v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double2 q = src[64 * r];
        v[r] = {q.x, q.y};
    }
    cd_dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; k++) v[k] = cd_mul(v[k], w512<INV>(tw, lane * k));
#if JDSP_F64_SWIZZLE
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * 64 + (lane ^ (8 * ((k >> 1) & 1)))] = v[k];
    lds_fence_wave();
    {
        const int k1 = lane >> 3, base = k1 * 64, g = 8 * ((k1 >> 1) & 1);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = lds[base + (((lane & 7) + 8 * a) ^ g)];
    }
#else
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * 72 + lane] = v[k];
    lds_fence_wave();
    {
        const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = lds[base + 8 * a];
    }
#endif
    lds_fence_wave();
    cd_dft8<INV>(v);
#pragma unroll
    for (int c = 1; c < 8; c++) v[c] = cd_mul(v[c], w512<INV>(tw, 8 * (lane & 7) * c));
#if JDSP_F64_SWIZZLE
    {
        const int k1 = lane >> 3, b = lane & 7;
#pragma unroll
        for (int c = 0; c < 8; c++) lds[64 * k1 + ((8 * c + b) ^ k1)] = v[c];
    }
    lds_fence_wave();
    {
        const int k1 = lane & 7, c = lane >> 3;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = lds[64 * k1 + ((8 * c + b) ^ k1)];
    }
#else
    {
        const int base = (lane >> 3) * 73 + (lane & 7);
#pragma unroll
        for (int c = 0; c < 8; c++) lds[base + 8 * c] = v[c];
    }
    lds_fence_wave();
    {
        const int base = (lane & 7) * 73 + (lane >> 3) * 8;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = lds[base + b];
    }
#endif
    cd_dft8<INV>(v);
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 *dst = reinterpret_cast<f64x2 *>(out + t * 512 + lane);
#pragma unroll
    for (int d = 0; d < 8; d++) {
        f64x2 q = {v[d].x, v[d].y};
        __builtin_nontemporal_store(q, dst + 64 * d);
    }
}

// ---- DFTProcess / IDFTProcess / IFFTProcess (FFT:151-184): the definition-level O(N^2) sums ---------------
// One thread per output bin k, the i loop in the reference's order, every product, sum and quotient rounded
// on its own (no FMA contraction), the angle formed as the reference forms it -- ((2*PI)*i)*k/N with its
// PI 3.14159265358 (FFT:15) -- and the result ACCUMULATED into what the caller left in `out` (the reference
// adds into its output arrays and relies on the caller having zeroed them, :168,:178,:154).  Any n >= 1.
// KIND 0: DFTProcess, int16 in (:162-173); 1: IDFTProcess, unnormalised (:175-184); 2: IFFTProcess, each
// term `* 1 / (double)iFFTLen` (:151-160).
template <int KIND>
__global__ __launch_bounds__(256) void dft_direct_f64_kernel(const void *__restrict__ in_v, double2 *__restrict__ out, int n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double two_pi = __dmul_rn(2.0, 3.14159265358);
    const double dn = (double)n;
    double2 *dst = out + (size_t)blockIdx.y * n + k;
    double re = dst->x, im = dst->y;
    if (KIND == 0) {
        const short *in = reinterpret_cast<const short *>(in_v) + (size_t)blockIdx.y * n;
        for (int i = 0; i < n; i++) {
            const double a = __ddiv_rn(__dmul_rn(__dmul_rn(two_pi, (double)i), (double)k), dn);
            const double x = (double)in[i];
            re = __dadd_rn(re, __dmul_rn(x, cos(a)));
            im = __dadd_rn(im, __dmul_rn(x, -sin(a)));
        }
    } else {
        const double2 *in = reinterpret_cast<const double2 *>(in_v) + (size_t)blockIdx.y * n;
        for (int i = 0; i < n; i++) {
            const double a = __ddiv_rn(__dmul_rn(__dmul_rn(two_pi, (double)i), (double)k), dn);
            const double c = cos(a), s = sin(a);
            const double2 z = in[i];
            double tr = __dsub_rn(__dmul_rn(z.x, c), __dmul_rn(z.y, s));
            double ti = __dadd_rn(__dmul_rn(z.x, s), __dmul_rn(z.y, c));
            if (KIND == 2) { tr = __ddiv_rn(tr, dn); ti = __ddiv_rn(ti, dn); }     // (..) * 1 / (double)iFFTLen
            re = __dadd_rn(re, tr);
            im = __dadd_rn(im, ti);
        }
    }
    *dst = make_double2(re, im);
}

int launch_dft_direct_f64(hipStream_t stream, int kind, const void *in, double2 *inout, int n, long batch)
{
    if (batch <= 0 || n <= 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)batch), block(256);
    if (kind == 0) hipLaunchKernelGGL(dft_direct_f64_kernel<0>, grid, block, 0, stream, in, inout, n);
    else if (kind == 1) hipLaunchKernelGGL(dft_direct_f64_kernel<1>, grid, block, 0, stream, in, inout, n);
    else hipLaunchKernelGGL(dft_direct_f64_kernel<2>, grid, block, 0, stream, in, inout, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_bitrev_table(hipStream_t stream, short *table_dev, int n_fft, int bits)
{
    hipLaunchKernelGGL(bitrev_table_kernel, dim3((n_fft + 255) / 256), dim3(256), 0, stream, table_dev, n_fft, bits);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- STFT analysis in the reference's own precision ---------------------------------------------------------
// SS:218-230 for n_frames frames with every operation in FP64: frame f = pcm[hop f .. hop f + 1024) x Hamming (the
// reference's own doubles, PI 3.141592), the 1024-point real transform as a 512-point complex one of
// z[n] = x[2n] + j x[2n+1] (the radix-8 passes of fft512_f64_kernel) and the split
//     E = Z[m] + conj Z[512-m],  O = -j (Z[m] - conj Z[512-m]),  X[m] = (E + W^m O) / 2,  X[m+512] = (E - W^m O) / 2,
// all 1024 bins out as complex128: 2 KB in, 16 KB out per frame -- an HBM-bound kernel like the FP32 one, at twice
// its bytes.  table: [0, 1024) window as doubles, then 512 double2 W^m = exp(-2 pi j m / 1024).
// One wave walks `run` consecutive frames (the launch makes the batch one round of resident waves): the window,
// split-twiddle and transform-twiddle loads -- 40 KB per frame out of L1 / L2 against 16 KB of output -- are then
// loop-invariant.  312 us per 65,536 frames at one frame per wave, 263 at 2, 242 at 4, 224 at 16 (profiles/r02_stft_f64.txt).
__global__ __launch_bounds__(64) void stft1024_f64_kernel(const short *__restrict__ pcm, long n_frames, long hop,
                                                          const double *__restrict__ table, const double2 *__restrict__ tw,
                                                          double2 *__restrict__ out, int run)
{
    __shared__ __attribute__((aligned(16))) cd lds[kFft512Lds];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;            // XCD-aware order: neighbouring frames share an XCD's L2
    const long t0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * run;
    const double2 *win = reinterpret_cast<const double2 *>(table) + lane;
    const double2 *wsp = reinterpret_cast<const double2 *>(table + 1024) + lane;
#pragma unroll 1
  for (long t = t0; t < t0 + run && t < n_frames; t++) {
    const short *src = pcm + t * hop + 2 * lane;
    cd v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double2 w = win[64 * r];
        v[r] = {(double)src[128 * r] * w.x, (double)src[128 * r + 1] * w.y};
    }
    cd_dft8<false>(v);
#pragma unroll
    for (int k = 1; k < 8; k++) v[k] = cd_mul(v[k], w512<false>(tw, lane * k));
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * 72 + lane] = v[k];
    lds_fence_wave();
    {
        const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = lds[base + 8 * a];
    }
    lds_fence_wave();
    cd_dft8<false>(v);
#pragma unroll
    for (int c = 1; c < 8; c++) v[c] = cd_mul(v[c], w512<false>(tw, 8 * (lane & 7) * c));
    {
        const int base = (lane >> 3) * 73 + (lane & 7);
#pragma unroll
        for (int c = 0; c < 8; c++) lds[base + 8 * c] = v[c];
    }
    lds_fence_wave();
    {
        const int base = (lane & 7) * 73 + (lane >> 3) * 8;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = lds[base + b];
    }
    lds_fence_wave();
    cd_dft8<false>(v);                                   // v[d] = Z[lane + 64 d]
    // natural-order image (slot 512 = Z[0]) for the mirrored operands
#pragma unroll
    for (int d = 0; d < 8; d++) lds[lane + 64 * d] = v[d];
    if (lane == 0) lds[512] = v[0];
    lds_fence_wave();
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 *dst = reinterpret_cast<f64x2 *>(out + t * 1024 + lane);
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const cd zm = lds[512 - lane - 64 * d];
        const double2 w = wsp[64 * d];
        const cd e = {v[d].x + zm.x, v[d].y - zm.y};                  // Z + conj Zm
        const cd o = {v[d].y + zm.y, zm.x - v[d].x};                  // -j (Z - conj Zm)
        const cd p = cd_mul(o, cd{w.x, w.y});
        f64x2 lo = {0.5 * (e.x + p.x), 0.5 * (e.y + p.y)}, hi = {0.5 * (e.x - p.x), 0.5 * (e.y - p.y)};
        __builtin_nontemporal_store(lo, dst + 64 * d);
        __builtin_nontemporal_store(hi, dst + 64 * d + 512);
    }
    lds_fence_wave();                                    // the image is rewritten by the next frame's exchanges
  }
}

// Round 3: the same arithmetic at FOUR waves per SIMD instead of two.  The kernel above keeps 120 VGPRs of loop-
// invariant tables (7 + 7 transform twiddles, 8 window pairs, 8 split twiddles: 176 registers, two waves per SIMD) and
// so has two waves per SIMD to hide its loads behind -- loads that queue behind the chip's write stream (stft_kernels.hip,
// "read pass").  Here only ONE value of each twiddle family stays in registers and the others are its powers, formed
// per frame (FP64 issue has the slack: ~450 instructions per lane and frame = 48 us per 65,536 frames against a 166 us
// store floor; +90 for the powers):
//     pass 1   w512^(lane k)      = (w512^lane)^k                       k = 2..7 by six complex products
//     pass 2   w512^(8 j c)       = (w64^j)^c,  j = lane & 7            the same chain
//     split    W1024^(lane + 64d) = W1024^lane * W16^d                  W16^d are literals
// (a product's rounding is 1e-16: the result moves by < 1e-15 of the frame peak; the tests hold it to 1e-12 of the
// CPU restatement and 1e-9 of the reference's FFTProcess).  The window stays in registers.  An empty asm on the three base
// values per frame keeps the compiler from hoisting the powers back out of the loop.  The frame's samples are one
// dword per lane and row when pcm is 4-byte aligned and hop is even (DW), and the NEXT frame's are requested before
// this frame's sixteen stores.
__device__ __forceinline__ void cd_opaque(cd &a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }
__device__ __forceinline__ void cd_powers(const cd &b, cd (&p)[8])
{
    p[1] = b;
    p[2] = cd_mul(b, b);
    p[3] = cd_mul(p[2], b);
    p[4] = cd_mul(p[2], p[2]);
    p[5] = cd_mul(p[4], b);
    p[6] = cd_mul(p[3], p[3]);
    p[7] = cd_mul(p[4], p[3]);
}

#ifndef JDSP_F64_PLAIN_STORES
#define JDSP_F64_PLAIN_STORES 0
#endif
#ifndef JDSP_F64_LINEAR_MAP
#define JDSP_F64_LINEAR_MAP 1
#endif
// LOOP: a wave walks `run` > 1 frames (prefetch of the next frame, taken before this frame's stores); !LOOP: one frame per wave
template <bool DW, bool LOOP>
__global__ __launch_bounds__(64, (JDSP_F64_SWIZZLE && !LOOP) ? 5 : 4) void stft1024_f64_v2_kernel(const short *__restrict__ pcm, long n_frames, long hop,
                                                               const double *__restrict__ table,
                                                               const double2 *__restrict__ tw, double2 *__restrict__ out,
                                                               int run)
{
    // JDSP_F64_SWIZZLE: an UNPADDED 512-element scratch (8,192 B per wave: twenty waves per CU = five per SIMD instead of
    // the four that 9,344 B allow -- this kernel is bound by its store stream, and a fifth wave's stores in flight are
    // what it is short of).  Conflict-free for the 16-byte accesses (ds_write_b128: 8 x 8 contiguous lanes; ds_read_b128:
    // four groups of 16 lanes over 64 banks, MI355X_MICROARCH.md) by XOR instead of padding:
    //   first exchange   element (row k, position l)  at 64 k + (l ^ 8 ((k >> 1) & 1))
    //   second exchange  element (k1, c, b)           at 64 k1 + ((8 c + b) ^ k1)
    //   natural image    Z[i] at i, and Z[512 - i] read at (512 - i) & 511 (slot 0 is Z[0] = Z[512])
    constexpr bool SWZ = JDSP_F64_SWIZZLE && !LOOP;      // (the looping form is at its register limit without the XOR arithmetic)
    __shared__ __attribute__((aligned(16))) cd lds[SWZ ? 512 : kFft512Lds];
    const int lane = threadIdx.x;
    // Frames in dispatch order: behind the read pass the PCM comes out of the Infinity Cache whichever XCD asks, and the
    // eight XCDs then write ONE moving 128 KB window of the output instead of eight streams 128 MiB apart
    // (196-198 us against 204 for the XCD-contiguous order of the FP32 path's no-read-pass days: profiles/r03_stft_f64.txt)
#if JDSP_F64_LINEAR_MAP
    const long t0 = (long)blockIdx.x * run;
#else
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long t0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * run;
#endif
    if (t0 >= n_frames) return;
    const long t1 = LOOP ? (t0 + run < n_frames ? t0 + run : n_frames) : t0 + 1;
    double2 win[8];
    {
        const double2 *wp = reinterpret_cast<const double2 *>(table) + lane;
#pragma unroll
        for (int r = 0; r < 8; r++) win[r] = wp[64 * r];
    }
    cd b1 = w512<false>(tw, lane), b2 = w512<false>(tw, 8 * (lane & 7));
    cd b3;
    {
        const double2 q = (reinterpret_cast<const double2 *>(table + 1024))[lane];
        b3 = {q.x, q.y};
    }
    unsigned int raw[8];                                  // DW: the sample pair (2 lane + 128 r, + 1) of the current frame
    short raw16[16];
    auto fetch = [&](long t) {
        const short *src = pcm + t * hop + 2 * lane;
        if (DW) {
#pragma unroll
            for (int r = 0; r < 8; r++) raw[r] = *reinterpret_cast<const unsigned int *>(src + 128 * r);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) { raw16[2 * r] = src[128 * r]; raw16[2 * r + 1] = src[128 * r + 1]; }
        }
    };
    fetch(t0);
    unsigned int tk[8];                                   // DW: the frame's samples as taken from `raw` (see below)
#pragma unroll
    for (int r = 0; r < 8; r++) tk[r] = raw[r];
#pragma unroll 1
    for (long t = t0; t < t1; t++) {
        cd v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int s0 = DW ? (int)(short)(tk[r] & 0xffffu) : (int)raw16[2 * r];
            const int s1 = DW ? ((int)tk[r] >> 16) : (int)raw16[2 * r + 1];
            v[r] = {(double)s0 * win[r].x, (double)s1 * win[r].y};
        }
        if (LOOP && t + 1 < t1) fetch(t + 1);             // in flight across this frame's arithmetic
        cd_opaque(b1); cd_opaque(b2); cd_opaque(b3);
        cd_dft8<false>(v);
        {
            cd p[8];
            cd_powers(b1, p);
#pragma unroll
            for (int k = 1; k < 8; k++) v[k] = cd_mul(v[k], p[k]);
        }
        if constexpr (SWZ) {
#pragma unroll
            for (int k = 0; k < 8; k++) lds[k * 64 + (lane ^ (8 * ((k >> 1) & 1)))] = v[k];
            lds_fence_wave();
            const int k1 = lane >> 3, base = k1 * 64, g = 8 * ((k1 >> 1) & 1);
#pragma unroll
            for (int a = 0; a < 8; a++) v[a] = lds[base + (((lane & 7) + 8 * a) ^ g)];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) lds[k * 72 + lane] = v[k];
            lds_fence_wave();
            const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
            for (int a = 0; a < 8; a++) v[a] = lds[base + 8 * a];
        }
        lds_fence_wave();
        cd_dft8<false>(v);
        {
            cd p[8];
            cd_powers(b2, p);
#pragma unroll
            for (int c = 1; c < 8; c++) v[c] = cd_mul(v[c], p[c]);
        }
        if constexpr (SWZ) {
            {
                const int k1 = lane >> 3, b = lane & 7;
#pragma unroll
                for (int c = 0; c < 8; c++) lds[64 * k1 + ((8 * c + b) ^ k1)] = v[c];
            }
            lds_fence_wave();
            const int k1 = lane & 7, c = lane >> 3;
#pragma unroll
            for (int b = 0; b < 8; b++) v[b] = lds[64 * k1 + ((8 * c + b) ^ k1)];
        } else {
            {
                const int base = (lane >> 3) * 73 + (lane & 7);
#pragma unroll
                for (int c = 0; c < 8; c++) lds[base + 8 * c] = v[c];
            }
            lds_fence_wave();
            const int base = (lane & 7) * 73 + (lane >> 3) * 8;
#pragma unroll
            for (int b = 0; b < 8; b++) v[b] = lds[base + b];
        }
        lds_fence_wave();
        cd_dft8<false>(v);                                // v[d] = Z[lane + 64 d]
        // natural-order image (slot 512 = Z[0]) for the mirrored operands.  (Tried: the split for m < 256 only, every item
        // storing X[m], X[m + 512] AND their conjugates at 1024 - m and 512 - m -- half the split arithmetic, half this
        // image's LDS traffic.  236-242 us against 204: a wave-store whose addresses DESCEND with the lane number is
        // far slower than the ascending one, and un-reversing the lanes costs more than the image.
        // profiles/r03_stft_f64.txt.)
#pragma unroll
        for (int d = 0; d < 8; d++) lds[lane + 64 * d] = v[d];
        if constexpr (!SWZ) { if (lane == 0) lds[512] = v[0]; }
        lds_fence_wave();
        // the next frame's samples are taken BEFORE this frame's stores: vmcnt counts loads and stores together in issue
        // order and the loop's back edge waits for vmcnt(0), so taking them at the top of the next iteration waited for
        // these sixteen 1 KB stores to complete (see fastconv1024_pairs_kernel)
        if (LOOP && DW && t + 1 < t1) {
#pragma unroll
            for (int r = 0; r < 8; r++) { tk[r] = raw[r]; asm volatile("" : "+v"(tk[r])); }
            __builtin_amdgcn_sched_barrier(0);
        }
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        f64x2 *dst = reinterpret_cast<f64x2 *>(out + t * 1024 + lane);
        // W16^d = exp(-2 pi j d / 16)
        constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
        const cd w16[8] = {{1.0, 0.0}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.0, -1.0}, {-s1, -c1}, {-h, -h}, {-c1, -s1}};
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const cd zm = lds[SWZ ? ((512 - lane - 64 * d) & 511) : (512 - lane - 64 * d)];
            const cd w = d == 0 ? b3 : (d == 4 ? cd{b3.y, -b3.x} : cd_mul(b3, w16[d]));
            const cd e = {v[d].x + zm.x, v[d].y - zm.y};                  // Z + conj Zm
            const cd o = {v[d].y + zm.y, zm.x - v[d].x};                  // -j (Z - conj Zm)
            const cd pp = cd_mul(o, w);
            f64x2 lo = {0.5 * (e.x + pp.x), 0.5 * (e.y + pp.y)}, hi = {0.5 * (e.x - pp.x), 0.5 * (e.y - pp.y)};
#if JDSP_F64_PLAIN_STORES                                 /* timing-only A/B: cached stores instead of nontemporal ones */
            dst[64 * d] = lo;
            dst[64 * d + 512] = hi;
#else
            __builtin_nontemporal_store(lo, dst + 64 * d);
            __builtin_nontemporal_store(hi, dst + 64 * d + 512);
#endif
        }
        lds_fence_wave();                                 // the image is rewritten by the next frame's exchanges
    }
}

// variant 0: stft1024_f64_v2_kernel (default), 1: round 2's stft1024_f64_kernel.  fpw: frames one wave walks (0 = default)
int launch_stft1024_f64(hipStream_t stream, int n_cu, const short *pcm, long n_frames, long hop, const double *table,
                        const double2 *tw512, double2 *out, int variant, int fpw)
{
    if (n_frames <= 0) return 0;
    // variant 0: one frame per wave unless told otherwise -- short-lived waves behind the read pass, as in the FP32 path
    // (204 us per 65,536 cold frames against 220 for one round of resident waves: profiles/r03_stft_f64.txt)
    const long slots = 4096;
    const long run = fpw > 0 ? fpw : (variant == 1 ? (n_frames + slots - 1) / slots : 1);
    (void)n_cu;
    const long waves = (n_frames + run - 1) / run;
    const long grid = (waves + 7) / 8 * 8;
    if (variant == 1)
        hipLaunchKernelGGL(stft1024_f64_kernel, dim3((unsigned)grid), dim3(64), 0, stream, pcm, n_frames, hop, table, tw512,
                           out, (int)run);
    else {
        const bool dw = (((uintptr_t)pcm) & 3u) == 0 && (hop & 1) == 0;
#define JDSP_F64_LAUNCH(DW_, LOOP_) hipLaunchKernelGGL((stft1024_f64_v2_kernel<DW_, LOOP_>), dim3((unsigned)grid), dim3(64), 0, stream, pcm, \
                                                       n_frames, hop, table, tw512, out, (int)run)
        if (run > 1) { if (dw) JDSP_F64_LAUNCH(true, true); else JDSP_F64_LAUNCH(false, true); }
        else { if (dw) JDSP_F64_LAUNCH(true, false); else JDSP_F64_LAUNCH(false, false); }
#undef JDSP_F64_LAUNCH
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// host side of the table: the reference's window (SS:226) and the split twiddles
void fill_stft1024_f64_table(double *t)
{
    const double PI = 3.141592;                                      // SS:52
    for (int i = 0; i < 1024; i++) t[i] = 0.54 - 0.46 * cos(2 * PI * i / (1024 - 1));
    const double two_pi = 6.283185307179586476925286766559;
    for (int m = 0; m < 512; m++) {
        const double a = -two_pi * (double)m / 1024.0;
        t[1024 + 2 * m] = cos(a);
        t[1024 + 2 * m + 1] = sin(a);
    }
}

int launch_fft_process_f64(hipStream_t stream, const double2 *in, double2 *out, int n_fft, int log2n, long batch,
                           int forward, const double2 *tw)
{
    if (batch <= 0) return 0;
    if (n_fft == 512 && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        const long grid = (batch + 7) / 8 * 8;
        if (forward) hipLaunchKernelGGL(fft512_f64_kernel<false>, dim3((unsigned)grid), dim3(64), 0, stream, in, out, batch, tw);
        else hipLaunchKernelGGL(fft512_f64_kernel<true>, dim3((unsigned)grid), dim3(64), 0, stream, in, out, batch, tw);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const size_t lds = sizeof(double2) * (size_t)n_fft;
    int threads = n_fft / 2 < 256 ? (n_fft / 2 < 64 ? 64 : n_fft / 2) : 256;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fft_process_f64_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(fft_process_f64_kernel, dim3((unsigned)batch), dim3(threads), lds, stream, in, out, n_fft, log2n,
                       forward, tw);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void fill_c2c_twiddles(double2 *t, int n_fft)
{
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < n_fft / 2; j++) {
        double a = -two_pi * (double)j / (double)n_fft;
        t[j] = make_double2(cos(a), sin(a));
    }
}

}  // namespace jdsp
