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
//   xchg 2  element (k1,c,b) at u(k1)+b+72c+(c&3), written by lane 8*k1+b, read by lane k1+8c (see xchg2_u below)
//   pass 3  lane (k1,c)   : Z[k1+8c+64d] = sum_b B[k1][c][b] w_8^(b d)
// so lane l ends up holding Z[l + 64 d] in v[d] -- the same "lane + 64*r"
// layout the input came in.  The inverse transform conjugates every twiddle.
#pragma once
#include <hip/hip_runtime.h>

namespace jdsp {

constexpr int kWaveLdsComplex = 8 * 73;          // 584 complex = 4672 B per wave
constexpr float kInvSqrt2 = 0.70710678118654752440f;

// ---- packed complex arithmetic ---------------------------------------------------------------------
// A complex number is one even-aligned VGPR pair (re, im); gfx950's VOP3P packed-FP32 instructions
// take, per source and per result half, a half selection (op_sel / op_sel_hi) and a negation
// (neg_lo / neg_hi).  That makes a + b, a - b, a +- j b and (1 +- j) a ONE v_pk_add_f32, a complex
// product v_pk_mul_f32 + v_pk_fma_f32, and the 1/sqrt2 of the odd eighth roots foldable into a
// v_pk_fma_f32.  The compiler does not form the mixed per-half selections (it falls back to two FMAs
// plus v_mov/v_pk_mov per product: 370 VALU instructions per frame against 205 spelled out), and at 4
// issue cycles per wave64 instruction the transform is VALU-issue-bound, so they are spelled out.
typedef float cfv __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cfv tv(float2 a) { cfv r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 fv(cfv r) { return make_float2(r.x, r.y); }

#define JDSP_PK_ADD(NAME, MODS)                                                                     \
    __device__ __forceinline__ float2 NAME(float2 a, float2 b)                                     \
    {                                                                                               \
        cfv r;                                                                                      \
        asm("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(r) : "v"(tv(a)), "v"(tv(b)));                   \
        return fv(r);                                                                               \
    }
JDSP_PK_ADD(cadd, "")                                                     // a + b
JDSP_PK_ADD(csub, "neg_lo:[0,1] neg_hi:[0,1]")                            // a - b
JDSP_PK_ADD(cadd_mj, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")         // a - j b = (a.x + b.y, a.y - b.x)
JDSP_PK_ADD(cadd_pj, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")         // a + j b = (a.x - b.y, a.y + b.x)
JDSP_PK_ADD(cadd_conj, "neg_hi:[0,1]")                                    // a + conj(b)
JDSP_PK_ADD(csub_conj_mj, "op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]")    // -j (a - conj(b)) = (a.y + b.y, b.x - a.x)
JDSP_PK_ADD(cconj_sub_j, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]")     // conj(a - j b) = (a.x + b.y, b.x - a.y)
#undef JDSP_PK_ADD

// d = c + s * a  /  c - s * a  with a real scale s held in both halves of `s2`, and the same with a
// rotated by -j or +j first
#define JDSP_PK_FMA(NAME, MODS)                                                                     \
    __device__ __forceinline__ float2 NAME(float2 a, cfv s2, float2 c)                             \
    {                                                                                               \
        cfv r;                                                                                      \
        asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(r) : "v"(tv(a)), "v"(s2), "v"(tv(c)));      \
        return fv(r);                                                                               \
    }
JDSP_PK_FMA(cfma, "")                                                                   // c + s a
JDSP_PK_FMA(cfnma, "neg_lo:[1,0,0] neg_hi:[1,0,0]")                                     // c - s a
JDSP_PK_FMA(cfma_mj, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")                 // c + s (-j a)
JDSP_PK_FMA(cfma_pj, "op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")                 // c + s (+j a)
#undef JDSP_PK_FMA

__device__ __forceinline__ cfv inv_sqrt2_pair() { cfv c = {kInvSqrt2, kInvSqrt2}; return c; }

__device__ __forceinline__ float2 cmul(float2 a, float2 w)             // a * w
{
    cfv t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(t) : "v"(tv(a)), "v"(tv(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(tv(a)), "v"(tv(w)), "v"(t));
    return fv(r);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 w)        // a * conj(w)
{
    cfv t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(tv(a)), "v"(tv(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(tv(a)), "v"(tv(w)), "v"(t));
    return fv(r);
}
// a + rot(b), a - rot(b) with rot = multiplication by -j (forward) or +j (inverse)
template <bool INV> __device__ __forceinline__ float2 cadd_rot90(float2 a, float2 b) { return INV ? cadd_pj(a, b) : cadd_mj(a, b); }
template <bool INV> __device__ __forceinline__ float2 csub_rot90(float2 a, float2 b) { return INV ? cadd_mj(a, b) : cadd_pj(a, b); }
// (1 - j) a (forward) or (1 + j) a (inverse): sqrt2 * w_8^1 a
template <bool INV> __device__ __forceinline__ float2 one_rot(float2 a) { return INV ? cadd_pj(a, a) : cadd_mj(a, a); }

// multiply by -j (forward) or +j (inverse)
template <bool INV> __device__ __forceinline__ float2 rot90(float2 a)
{
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// In-register 8-point DFT, natural order in and out: 26 packed instructions.  The odd eighth roots
// w_8 = (1 -+ j)/sqrt2 and w_8^3 = -(1 +- j)/sqrt2 enter as p5 = (1 -+ j) d1 and p7 = (1 +- j) d3 with
// the 1/sqrt2 folded into the last stage's FMAs: w_8 d1 + w_8^3 d3 = (p5 - p7)/sqrt2 and
// w_8 d1 - w_8^3 d3 = (p5 + p7)/sqrt2.
template <bool INV> __device__ __forceinline__ void dft8(float2 (&v)[8])
{
    const cfv c = inv_sqrt2_pair();
    const float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    const float2 a1 = cadd(v[1], v[5]), d1 = csub(v[1], v[5]);
    const float2 a2 = cadd(v[2], v[6]), d2 = csub(v[2], v[6]);
    const float2 a3 = cadd(v[3], v[7]), d3 = csub(v[3], v[7]);
    const float2 p5 = one_rot<INV>(d1), p7 = one_rot<!INV>(d3);
    const float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
    const float2 b1 = cadd(a1, a3), d13 = csub(a1, a3);
    const float2 b4 = cadd_rot90<INV>(a4, d2), b6 = csub_rot90<INV>(a4, d2);
    const float2 u = csub(p5, p7), w = cadd(p5, p7);
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1);
    v[2] = cadd_rot90<INV>(b2, d13); v[6] = csub_rot90<INV>(b2, d13);
    v[1] = cfma(u, c, b4); v[5] = cfnma(u, c, b4);
    v[3] = INV ? cfma_pj(w, c, b6) : cfma_mj(w, c, b6);
    v[7] = INV ? cfma_mj(w, c, b6) : cfma_pj(w, c, b6);
}

// The exchanges' LDS accesses, one ds_read_b64 / ds_write_b64 each.  Left to itself the compiler pairs neighbouring
// accesses into ds_read2_b64 / ds_write2_b64 -- and a ds_read2_b64 moves its 1,024 bytes in 8 LDS cycles over a 32-bank map,
// two ds_read_b64 theirs in 4 over 64 banks (MI355X_MICROARCH.md, LDS): half the read rate, and bank conflicts on a
// layout laid out for the 64-bank map (the 32 conflict cycles per transform SQ_LDS_BANK_CONFLICT kept showing after the
// round-3 layout).  A volatile access in the LDS address space is not paired; its waits stay counted (lgkmcnt(N)).
#ifndef JDSP_XCHG_UNPAIRED
#define JDSP_XCHG_UNPAIRED 1
#endif
typedef float xchg_v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile xchg_v2f xchg_lds_t;
__device__ __forceinline__ void xchg_st(float2 *lds, int i, float2 v)
{
#if JDSP_XCHG_UNPAIRED
    xchg_v2f t = {v.x, v.y};
    ((xchg_lds_t *)lds)[i] = t;
#else
    lds[i] = v;
#endif
}
__device__ __forceinline__ float2 xchg_ld(const float2 *lds, int i)
{
#if JDSP_XCHG_UNPAIRED
    const xchg_v2f t = ((xchg_lds_t *)lds)[i];
    return make_float2(t.x, t.y);
#else
    return lds[i];
#endif
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

// Second exchange, element (k1, c, b): written by lane 8 k1 + b from register c, read by lane k1 + 8 c into register b.
// Round 1-2 kept it at k1 * 73 + 8 c + b: conflict-free for the reads (ds_read_b64: 64 banks over 32 lanes) but not for
// the writes -- a ds_write_b64 is served 16 lanes at a time over 32 banks, and lanes (k1 odd, b = 7) and (k1 even,
// b = 0) of a group met in one bank: one extra LDS cycle per group, 4 per write, 96 per block of the HRIR convolver
// = all of its SQ_LDS_BANK_CONFLICT (16.7 % of its LDS cycles, profiles/r02_chains_sq_counters.txt).  The layout
//     u(k1) + b + 72 c + (c & 3),   u = 8 (k1 & 1) + 36 ((k1 >> 1) & 1) + 16 (k1 >> 2)      (max 574 < kWaveLdsComplex)
// is conflict-free on both sides under those two bank maps (MI355X_MICROARCH.md, LDS; found by exhaustive search over
// separable maps) and stays separable: the writer's lane part is u(k1) + b with the c part as instruction offsets, the
// reader's is u(k1) + 72 c + (c & 3) with b = 0..7 consecutive -- no per-register address arithmetic on either side.
#ifndef JDSP_XCHG2_PITCH73
#define JDSP_XCHG2_PITCH73 0        // 1: rounds 1-2's k1 * 73 + 8 c + b (A/B timing)
#endif
#if JDSP_XCHG2_PITCH73
__device__ __forceinline__ int xchg2_wbase(int lane) { return (lane >> 3) * 73 + (lane & 7); }
__device__ __forceinline__ int xchg2_rbase(int lane) { return (lane & 7) * 73 + (lane >> 3) * 8; }
constexpr int xchg2_coff(int c) { return 8 * c; }
#else
__device__ __forceinline__ int xchg2_u(int k1) { return 8 * (k1 & 1) + 36 * ((k1 >> 1) & 1) + 16 * (k1 >> 2); }
__device__ __forceinline__ int xchg2_wbase(int lane) { return xchg2_u(lane >> 3) + (lane & 7); }
__device__ __forceinline__ int xchg2_rbase(int lane) { const int c = lane >> 3; return xchg2_u(lane & 7) + 72 * c + (c & 3); }
constexpr int xchg2_coff(int c) { return 72 * c + (c & 3); }
#endif

// XOR-swizzled second exchange (conflict-free writes and reads).  Measured in steady state, A/B/A/B:
// the two-transform form gains 3 % with it (MFCC 96 -> 93 us), the one-transform kernels are unchanged or 2-3 %
// slower (its extra address arithmetic), so each form has its own switch.
#ifndef JDSP_XCHG2_SWIZZLE
#define JDSP_XCHG2_SWIZZLE 0
#endif
#ifndef JDSP_XCHG2_SWIZZLE_X2
#define JDSP_XCHG2_SWIZZLE_X2 0      // round 2 had 1 here (against the pitch-73 layout's write conflicts)
#endif
// Second exchange, swizzled form: element (k1, c, b) -- written by lane 8 k1 + b from register c, read by lane
// k1 + 8 c into register b -- lives at 16 (4 c + (k1 >> 1)) + (((2 b) | (k1 & 1)) ^ 2 (k1 >> 1) ^ 8 (c & 1)).
__device__ __forceinline__ int xchg2_write_base(int lane)
{
    const int k1 = lane >> 3, b = lane & 7;
    return 16 * (k1 >> 1) + (((2 * b) | (k1 & 1)) ^ (2 * (k1 >> 1)));
}
__device__ __forceinline__ int xchg2_read_row(int lane) { return 16 * (4 * (lane >> 3) + ((lane & 7) >> 1)); }
__device__ __forceinline__ int xchg2_read_xor(int lane)
{
    const int k1 = lane & 7, c = lane >> 3;
    return (k1 & 1) ^ (2 * (k1 >> 1)) ^ (8 * (c & 1));
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
    for (int k = 0; k < 8; k++) xchg_st(lds, k * 72 + lane, v[k]);
    wave_lds_fence();
    {
        const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = xchg_ld(lds, base + 8 * a);
    }
    wave_lds_fence();
    dft8<INV>(v);
#pragma unroll
    for (int c = 1; c < 8; c++) v[c] = INV ? cmul_conj(v[c], tw.t2[c - 1]) : cmul(v[c], tw.t2[c - 1]);
#if JDSP_XCHG2_SWIZZLE
    {
        const int wl = xchg2_write_base(lane);
#pragma unroll
        for (int c = 0; c < 8; c++) lds[64 * c + ((c & 1) ? (wl ^ 8) : wl)] = v[c];
    }
    wave_lds_fence();
    {
        const int rm = xchg2_read_row(lane), rx = xchg2_read_xor(lane);
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = lds[rm + ((2 * b) ^ rx)];
    }
#else
    {
        const int base = xchg2_wbase(lane);
#pragma unroll
        for (int c = 0; c < 8; c++) xchg_st(lds, base + xchg2_coff(c), v[c]);
    }
    wave_lds_fence();
    {
        const int base = xchg2_rbase(lane);
#pragma unroll
        for (int b = 0; b < 8; b++) v[b] = xchg_ld(lds, base + b);
    }
#endif
    wave_lds_fence();
    dft8<INV>(v);
}

// Two independent transforms in one wave, stage by stage: the same passes as wave_fft512 with both frames'
// LDS exchanges issued together, so each fence covers two transforms and the scheduler has two independent
// dependency chains to interleave.  `lds_a` / `lds_b`: two scratch regions of kWaveLdsComplex elements.
template <bool INV>
__device__ __forceinline__ void wave_fft512_x2(float2 (&a)[8], float2 (&b)[8], float2 *lds_a, float2 *lds_b, int lane,
                                               const WaveTwiddles &tw)
{
    dft8<INV>(a);
    dft8<INV>(b);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        a[k] = INV ? cmul_conj(a[k], tw.t1[k - 1]) : cmul(a[k], tw.t1[k - 1]);
        b[k] = INV ? cmul_conj(b[k], tw.t1[k - 1]) : cmul(b[k], tw.t1[k - 1]);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { xchg_st(lds_a, k * 72 + lane, a[k]); xchg_st(lds_b, k * 72 + lane, b[k]); }
    wave_lds_fence();
    {
        const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
        for (int q = 0; q < 8; q++) { a[q] = xchg_ld(lds_a, base + 8 * q); b[q] = xchg_ld(lds_b, base + 8 * q); }
    }
    wave_lds_fence();
    dft8<INV>(a);
    dft8<INV>(b);
#pragma unroll
    for (int c = 1; c < 8; c++) {
        a[c] = INV ? cmul_conj(a[c], tw.t2[c - 1]) : cmul(a[c], tw.t2[c - 1]);
        b[c] = INV ? cmul_conj(b[c], tw.t2[c - 1]) : cmul(b[c], tw.t2[c - 1]);
    }
#if JDSP_XCHG2_SWIZZLE_X2
    {
        const int wl = xchg2_write_base(lane);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int i = 64 * c + ((c & 1) ? (wl ^ 8) : wl);
            lds_a[i] = a[c];
            lds_b[i] = b[c];
        }
    }
    wave_lds_fence();
    {
        const int rm = xchg2_read_row(lane), rx = xchg2_read_xor(lane);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = rm + ((2 * q) ^ rx);
            a[q] = lds_a[i];
            b[q] = lds_b[i];
        }
    }
#else
    {
        const int base = xchg2_wbase(lane);
#pragma unroll
        for (int c = 0; c < 8; c++) { xchg_st(lds_a, base + xchg2_coff(c), a[c]); xchg_st(lds_b, base + xchg2_coff(c), b[c]); }
    }
    wave_lds_fence();
    {
        const int base = xchg2_rbase(lane);
#pragma unroll
        for (int q = 0; q < 8; q++) { a[q] = xchg_ld(lds_a, base + q); b[q] = xchg_ld(lds_b, base + q); }
    }
#endif
    wave_lds_fence();
    dft8<INV>(a);
    dft8<INV>(b);
}

// ---- the passes of wave_fft512 as separate pieces, and two transforms STAGGERED by half a stage -------------------
// wave_fft512_x2 issues both transforms' writes, then both transforms' reads, then waits: each fence covers two
// transforms, but the wave still has nothing to issue while its reads are in flight.  Here transform b runs half a
// stage behind a: a's exchange (8 writes, 8 reads) is in the LDS pipe while b's radix-8 pass and twiddles issue, and
// the other way round -- one wave keeps the VALU and the LDS pipe busy together (LDS returns in order, so the wait
// in front of a's next pass is a counted lgkmcnt that leaves b's exchange outstanding).  Separate scratch per
// transform; the unswizzled second exchange (its address arithmetic is free).
template <bool INV> __device__ __forceinline__ void fft512_pass1(float2 (&v)[8], const WaveTwiddles &tw)
{
    dft8<INV>(v);
#pragma unroll
    for (int k = 1; k < 8; k++) v[k] = INV ? cmul_conj(v[k], tw.t1[k - 1]) : cmul(v[k], tw.t1[k - 1]);
}
template <bool INV> __device__ __forceinline__ void fft512_pass2(float2 (&v)[8], const WaveTwiddles &tw)
{
    dft8<INV>(v);
#pragma unroll
    for (int c = 1; c < 8; c++) v[c] = INV ? cmul_conj(v[c], tw.t2[c - 1]) : cmul(v[c], tw.t2[c - 1]);
}
__device__ __forceinline__ void fft512_xchg1(float2 (&v)[8], float2 *lds, int lane)
{
#pragma unroll
    for (int k = 0; k < 8; k++) xchg_st(lds, k * 72 + lane, v[k]);
    wave_lds_fence();
    const int base = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
    for (int a = 0; a < 8; a++) v[a] = xchg_ld(lds, base + 8 * a);
}
__device__ __forceinline__ void fft512_xchg2(float2 (&v)[8], float2 *lds, int lane)
{
    const int wb = xchg2_wbase(lane);
#pragma unroll
    for (int c = 0; c < 8; c++) xchg_st(lds, wb + xchg2_coff(c), v[c]);
    wave_lds_fence();
    const int rb = xchg2_rbase(lane);
#pragma unroll
    for (int b = 0; b < 8; b++) v[b] = xchg_ld(lds, rb + b);
}

template <bool INV>
__device__ __forceinline__ void wave_fft512_x2_staggered(float2 (&a)[8], float2 (&b)[8], float2 *lds_a, float2 *lds_b,
                                                         int lane, const WaveTwiddles &tw)
{
    // sched_barrier: left to itself the scheduler sinks each exchange's reads down to the next exchange's writes (shorter
    // live ranges) and the wave then waits for them with nothing left to issue
    fft512_pass1<INV>(a, tw);
    fft512_xchg1(a, lds_a, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft512_pass1<INV>(b, tw);                 // under a's exchange
    fft512_xchg1(b, lds_b, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft512_pass2<INV>(a, tw);                 // under b's exchange
    wave_lds_fence();
    fft512_xchg2(a, lds_a, lane);
    __builtin_amdgcn_sched_barrier(0);
    fft512_pass2<INV>(b, tw);
    wave_lds_fence();
    fft512_xchg2(b, lds_b, lane);
    __builtin_amdgcn_sched_barrier(0);
    dft8<INV>(a);
    __builtin_amdgcn_sched_barrier(0);
    dft8<INV>(b);
    wave_lds_fence();
}

}  // namespace jdsp
