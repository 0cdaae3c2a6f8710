// wave_fft512.h -- one 512-point complex FFT per 64-lane wavefront (gfx950).
//
// A 1024-point real frame is transformed as a 512-point complex FFT of
// z[n] = x[2n] + j x[2n+1] followed by a split step, so one wave64 owns one
// frame: 8 complex points per lane, three in-register radix-8 passes
// (512 = 8*8*8) and LDS exchanges between them.  No workgroup barrier is ever
// needed: the exchanges are wave-private and the LDS executes one wave's
// DS instructions in issue order.
//
// Index algebra (forward, w_N = exp(-2*pi*j/N)); n = 64*r + l, k = k1 + 8*k2:
//   pass 1  lane l        : A[k1]   = sum_r z[l+64r] w_8^(r k1);  A[k1] *= w_512^(l k1)
//   xchg 1  write k1*72+l ; lane l'=8*k1+b reads k1*72+8a+b          (72: bank spread)
//   pass 2  lane (k1,b)   : B[c]    = sum_a A[k1][8a+b] w_8^(a c);  B[c] *= w_64^(b c)
//   xchg 2  write k1*73+8c+b ; lane l''=k1+8c reads k1*73+8c+b'      (73: bank spread)
//   pass 3  lane (k1,c)   : Z[k1+8c+64d] = sum_b B[k1][c][b] w_8^(b d)
// so lane l ends up holding Z[l + 64 d] in v[d] -- the same "lane + 64*r"
// layout the input came in.  The inverse transform conjugates every twiddle.
#pragma once
#include <hip/hip_runtime.h>

namespace jdsp {

constexpr int kWaveLdsComplex = 8 * 73;          // 584 complex = 4672 B per wave
constexpr float kInvSqrt2 = 0.70710678118654752440f;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b)   // a * conj(b)
{
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// multiply by -j (forward) or +j (inverse)
template <bool INV> __device__ __forceinline__ float2 rot90(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// multiply by w_8^1 = (1-j)/sqrt2 (forward) or its conjugate (inverse)
template <bool INV> __device__ __forceinline__ float2 rot45(float2 a)
{
    return INV ? make_float2(kInvSqrt2 * (a.x - a.y), kInvSqrt2 * (a.x + a.y))
               : make_float2(kInvSqrt2 * (a.x + a.y), kInvSqrt2 * (a.y - a.x));
}
// multiply by w_8^3 = (-1-j)/sqrt2 (forward) or its conjugate (inverse)
template <bool INV> __device__ __forceinline__ float2 rot135(float2 a)
{
    return INV ? make_float2(-kInvSqrt2 * (a.x + a.y), kInvSqrt2 * (a.x - a.y))
               : make_float2(kInvSqrt2 * (a.y - a.x), -kInvSqrt2 * (a.x + a.y));
}

// In-register 8-point DFT, natural order in and out.
template <bool INV> __device__ __forceinline__ void dft8(float2 (&v)[8])
{
    float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), a5 = rot45<INV>(csub(v[1], v[5]));
    float2 a2 = cadd(v[2], v[6]), a6 = rot90<INV>(csub(v[2], v[6]));
    float2 a3 = cadd(v[3], v[7]), a7 = rot135<INV>(csub(v[3], v[7]));
    float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
    float2 b1 = cadd(a1, a3), b3 = rot90<INV>(csub(a1, a3));
    float2 b4 = cadd(a4, a6), b6 = csub(a4, a6);
    float2 b5 = cadd(a5, a7), b7 = rot90<INV>(csub(a5, a7));
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1);
    v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
    v[1] = cadd(b4, b5); v[5] = csub(b4, b5);
    v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}

// Orders this wave's LDS traffic for the compiler without any hardware wait:
// DS instructions of one wave execute in issue order, so a later ds_read from
// another lane's slot sees the earlier ds_write.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane twiddles, loaded once per wave from the handle's table:
//   t1[k-1] = w_512^(lane*k), t2[c-1] = w_64^((lane&7)*c)   (forward values)
struct WaveTwiddles {
    float2 t1[7];
    float2 t2[7];
};

// Table layout (float2): [0, 7*64) t1[k-1][lane], [448, 448+7*8) t2[c-1][b]
constexpr int kTwT1 = 0;
constexpr int kTwT2 = 7 * 64;
constexpr int kTwCount = 7 * 64 + 7 * 8;

__device__ __forceinline__ void load_wave_twiddles(WaveTwiddles &tw, const float2 *__restrict__ table, int lane)
{
#pragma unroll
    for (int k = 0; k < 7; k++) tw.t1[k] = table[kTwT1 + k * 64 + lane];
#pragma unroll
    for (int c = 0; c < 7; c++) tw.t2[c] = table[kTwT2 + c * 8 + (lane & 7)];
}

// v[r] = z[lane + 64 r] on entry, Z[lane + 64 d] on exit.  `lds` is this wave's
// private kWaveLdsComplex-element scratch.  Callers that reuse `lds` afterwards
// must put a wave_lds_fence() before their own writes.
template <bool INV>
__device__ __forceinline__ void wave_fft512(float2 (&v)[8], float2 *lds, int lane, const WaveTwiddles &tw)
{
    dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; k++) v[k] = INV ? cmul_conj(v[k], tw.t1[k - 1]) : cmul(v[k], tw.t1[k - 1]);
#pragma unroll
    for (int k = 0; k < 8; k++) lds[k * 72 + lane] = v[k];
    wave_lds_fence();
    {
        const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = lds[base + 8 * a];
    }
    wave_lds_fence();
    dft8<INV>(v);
#pragma unroll
    for (int c = 1; c < 8; c++) v[c] = INV ? cmul_conj(v[c], tw.t2[c - 1]) : cmul(v[c], tw.t2[c - 1]);
    {
        const int base = (lane >> 3) * 73 + (lane & 7);
#pragma unroll
        for (int c = 0; c < 8; c++) lds[base + 8 * c] = v[c];
    }
    wave_lds_fence();
    {
        const int base = (lane & 7) * 73 + (lane >> 3) * 8;
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = lds[base + b];
    }
    wave_lds_fence();
    dft8<INV>(v);
}

}  // namespace jdsp
