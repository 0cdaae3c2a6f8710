// frame_io.h -- shared pieces of the 1024-point real-frame kernels (gfx950):
// table layout, int16 unpacking, the forward split and the inverse pre-split.
#pragma once
#include "wave_fft512.h"

namespace jdsp {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// handle-owned table (float2 units), built by fill_stft1024_table():
//   [0, kTwCount)            wave twiddles (wave_fft512.h)
//   [kStftWin, +512)         0.5*w[2i], 0.5*w[2i+1]   Hamming, PI = 3.141592 as the reference
//   [kStftSplit, +512)       W^m = exp(-2*pi*j*m/1024)
//   [kStftWinFull, +512)     w[2i], w[2i+1]           (not halved; MFCC/debug)
constexpr int kStftWin = kTwCount;
constexpr int kStftSplit = kStftWin + 512;
constexpr int kStftWinFull = kStftSplit + 512;
constexpr int kStftTableCount = kStftWinFull + 512;

__device__ __forceinline__ float2 unpack_i16x2(unsigned int raw)
{
    return make_float2((float)(short)(raw & 0xffffu), (float)((int)raw >> 16));
}

// (short)double of the reference (e.g. SpectralSubtraction_final.cpp:252): truncate toward
// zero, keep the low 16 bits (what x86 does for in-int32-range values); NaN -> 0.
__device__ __forceinline__ unsigned int cast_i16_bits(float v)
{
    return (unsigned int)((int)v) & 0xffffu;
}
// two of them in one dword, a in the low half: the two truncations and ONE v_perm_b32 that picks the low 16 bits of each
// (cast | cast << 16 compiles to v_and + v_lshl_or: four instructions per sample pair instead of three)
__device__ __forceinline__ unsigned int cast_i16x2_bits(float a, float b)
{
    return __builtin_amdgcn_perm((unsigned int)(int)b, (unsigned int)(int)a, 0x05040100u);
}

// Forward split of the packed transform: Zh = FFT512(z)/2 (the 1/2 is folded into the
// window).  E = Zh[m] + conj(Zh[512-m]),  O = -j (Zh[m] - conj(Zh[512-m])),
// X[m] = E + W^m O,  X[m+512] = E - W^m O, with W^m = w_8^j * wsp (m = 128 j + 2 lane + e).
template <int J>
__device__ __forceinline__ void split_fwd(float2 zm, float2 zr, float2 wsp, float2 &lo, float2 &hi)
{
    const float2 e = cadd_conj(zm, zr);
    const float2 o = csub_conj_mj(zm, zr);
    const float2 t = cmul(wsp, o);
    if (J == 0) { lo = cadd(e, t); hi = csub(e, t); }
    if (J == 1) { const float2 p = one_rot<false>(t); lo = cfma(p, inv_sqrt2_pair(), e); hi = cfnma(p, inv_sqrt2_pair(), e); }
    if (J == 2) { lo = cadd_mj(e, t); hi = cadd_pj(e, t); }
    if (J == 3) { const float2 p = one_rot<true>(t); lo = cfnma(p, inv_sqrt2_pair(), e); hi = cfma(p, inv_sqrt2_pair(), e); }
}

// Inverse pre-split: from Y[m] and Y[m+512] of a Hermitian spectrum,
// Z'[m] = (Y[m] + Y[m+512]) + j (Y[m] - Y[m+512]) conj(W^m); then
// y[2n] + j y[2n+1] = IDFT512_unnormalised(Z')[n] / 1024.  With r = (Y[m] - Y[m+512]) conj(wsp) and
// conj(w_8^J) = 1, (1+j)/sqrt2, j, (-1+j)/sqrt2:  Z' = s + j r,  s - (1-j) r/sqrt2,  s - r,  s - (1+j) r/sqrt2.
template <int J>
__device__ __forceinline__ float2 presplit_inv(float2 ylo, float2 yhi, float2 wsp)
{
    const float2 s = cadd(ylo, yhi);
    const float2 r = cmul_conj(csub(ylo, yhi), wsp);
    if (J == 0) return cadd_pj(s, r);
    if (J == 1) return cfnma(one_rot<false>(r), inv_sqrt2_pair(), s);
    if (J == 2) return csub(s, r);
    return cfnma(one_rot<true>(r), inv_sqrt2_pair(), s);
}

// The natural-order image of a transform in LDS (element i = Z[i], read by the split steps), with
// Z[512] = Z[0] in slot 512 so that the mirrored operands need no index mask; the image needs 513 of the
// scratch's kWaveLdsComplex elements.
__device__ __forceinline__ void store_natural_image(float2 *img, int lane, const float2 (&v)[8])
{
#pragma unroll
    for (int d = 0; d < 8; d++) xchg_st(img, lane + 64 * d, v[d]);
    if (lane == 0) img[512] = v[0];
}
// Z[512 - m] and Z[511 - m]: two ADJACENT elements, consecutive across lanes -> one conflict-free
// ds_read2_b64.  (With the index masked to 511 they were two stride-2 reads, 2-way bank conflicts each:
// SQ_LDS_BANK_CONFLICT was 15-20 % of the LDS cycles of every transform kernel.)
__device__ __forceinline__ void load_mirror_pair(const float2 *img, int m, float2 &zr0, float2 &zr1)
{
    zr1 = img[511 - m];
    zr0 = img[512 - m];
}

// ---- the split steps without an LDS image: "lane + 64 d" bins --------------------------------------------------
// After wave_fft512 lane l holds Zh[l + 64 d] in v[d].  The split needs Zh[512 - m] next to Zh[m]: for m = l + 64 d
// that is register 7 - d of lane 64 - l (lane 0: its own register 8 - d; Zh[512] = Zh[0]) -- one ds_bpermute per
// dword, which moves data through the LDS crossbar without touching LDS memory.  Keeping the bins in this layout
// (X[l + 64 d] and X[l + 64 d + 512] in lane l) means the per-bin work and the inverse pre-split are register-only
// and Z'[l + 64 d] comes out exactly where the inverse transform takes its input: against the natural-order image
// (write 4 KB, read 8 KB, then write and read Z' again) a frame moves 16 KB less through the LDS pipe, which is
// these kernels' co-critical unit next to VALU issue.  Split twiddles: W^(l + 64 d), 8 per lane.
struct SplitTwiddles { float2 w[8]; };

__device__ __forceinline__ void load_split_twiddles(SplitTwiddles &t, const float2 *__restrict__ table, int lane);

__device__ __forceinline__ void mirror_fetch(const float2 (&v)[8], int lane, float2 (&zr)[8])
{
    const int addr = ((64 - lane) & 63) << 2;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const float2 src = v[7 - d];
        float2 r;
        r.x = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(src.x)));
        r.y = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(src.y)));
        const float2 own = v[(8 - d) & 7];                          // lane 0: 512 - 64 d = 64 (8 - d), its own register
        zr[d] = lane == 0 ? own : r;
    }
}

// The same operands through a natural-order LDS image instead: 8 ds_write_b64 + 8 ds_read_b64 (lane l reads slot
// 512 - l - 64 d: consecutive lanes, consecutive slots, conflict-free; slot 512 = slot 0, so lane 0 needs no special
// case).  A ds_bpermute_b32 costs the LDS pipe three times a ds_read_b64's worth per byte (tools/valu_rate.hip: 8.9
// issue slots against 16 x 1.6 for the selects it also needs), so this is the cheaper way to fetch the mirror --
// what is saved against the round-1 image is its second read (Zh[m] stays in registers) and the whole Z' image.
// `img` must not be in use: fence before this if the transform's last exchange may still be reading it.
__device__ __forceinline__ void mirror_fetch_lds(const float2 (&v)[8], float2 *img, int lane, float2 (&zr)[8])
{
    store_natural_image(img, lane, v);
    wave_lds_fence();
#pragma unroll
    for (int d = 0; d < 8; d++) zr[d] = xchg_ld(img, 512 - lane - 64 * d);
    wave_lds_fence();
}

// X[m] = E + W^m O, X[m + 512] = E - W^m O for m = lane + 64 d (the 1/2 is folded into the window as everywhere)
__device__ __forceinline__ void split_fwd_reg(const float2 (&v)[8], const float2 (&zr)[8], const SplitTwiddles &t,
                                              float2 (&lo)[8], float2 (&hi)[8])
{
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const float2 e = cadd_conj(v[d], zr[d]);
        const float2 o = csub_conj_mj(v[d], zr[d]);
        const float2 p = cmul(t.w[d], o);
        lo[d] = cadd(e, p);
        hi[d] = csub(e, p);
    }
}

// Z'[m] = (Y[m] + Y[m+512]) + j (Y[m] - Y[m+512]) conj(W^m), m = lane + 64 d: the inverse transform's input, in place
__device__ __forceinline__ float2 presplit_inv_reg(float2 ylo, float2 yhi, float2 w)
{
    const float2 s = cadd(ylo, yhi);
    const float2 r = cmul_conj(csub(ylo, yhi), w);
    return cadd_pj(s, r);
}

// ---- the split steps with every mirror pair owned by ONE lane ------------------------------------------------------
// split_fwd_reg hands each lane X[m] and X[m + 512] for its eight m = l + 64 d: all 1024 bins of the frame, i.e. every
// pair {X[k], X[1024 - k]} twice over (a real frame's spectrum is Hermitian), and a per-bin stage that keeps the spectrum
// Hermitian (a real gain that is the same for k and 1024 - k) pays for sixteen bins per lane where eight are distinct.
// Here lane l works on m = l + 64 d for d = 0..4 only -- the bins m and m + 512 of 320 values of m -- and gets the rest
// of the inverse transform's input from the symmetry: with s = Y[m] + Y[m+512], r = (Y[m] - Y[m+512]) conj(W^m),
//     Z'[m] = s + j r  (presplit_inv_reg)     and     Z'[512 - m] = conj(s - j r),
// the second going to whoever holds bin 512 - m (lane 64 - l, register 7 - d; lane 0: its own register 8 - d).  Lane l
// keeps Z'[l + 64 d] for d <= 4 and receives registers 5..7; the pairs of d = 3, 4 overlap (m = 192 + l and 256 + (64 - l)
// are mirrors) so that every lane runs the same five items and no lane needs a special case.  Five split / per-bin /
// pre-split items per lane instead of eight, ten per-bin evaluations instead of sixteen, ten noise values and five
// twiddles in registers instead of sixteen and eight; the LDS moves the same nine + eight vectors as mirror_fetch_lds.
struct PairTwiddles { float2 w[5]; };

__device__ __forceinline__ void load_pair_twiddles(PairTwiddles &t, const float2 *__restrict__ table, int lane)
{
#pragma unroll
    for (int d = 0; d < 5; d++) t.w[d] = table[kStftSplit + lane + 64 * d];
}

// zr[d] = Zh[512 - (lane + 64 d)], d = 0..4: registers 3..7 of the other lanes (and slot 512 = Zh[0]).  `img` must not
// be in use (fence before this if the transform's last exchange may still be reading it).
__device__ __forceinline__ void pair_fetch_lds(const float2 (&v)[8], float2 *img, int lane, float2 (&zr)[5])
{
#pragma unroll
    for (int d = 3; d < 8; d++) xchg_st(img, lane + 64 * d, v[d]);
    if (lane == 0) img[512] = v[0];
    wave_lds_fence();
#pragma unroll
    for (int d = 0; d < 5; d++) zr[d] = xchg_ld(img, 512 - lane - 64 * d);
    wave_lds_fence();
}

// One item: from Y[m], Y[m + 512] the two inverse-transform inputs Z'[m] and Z'[512 - m]
__device__ __forceinline__ void presplit_inv_pair(float2 ylo, float2 yhi, float2 w, float2 &zk, float2 &zmirror)
{
    const float2 s = cadd(ylo, yhi);
    const float2 r = cmul_conj(csub(ylo, yhi), w);
    zk = cadd_pj(s, r);
    zmirror = cconj_sub_j(s, r);
}

// y[0..4] hold Z'[lane + 64 d]; ret[d] = Z'[512 - lane - 64 d] for d = 0..3 go to their owners, and y[5..7] come back.
// (Slots 257..319 and 512 receive values nobody reads.)
__device__ __forceinline__ void pair_return_lds(const float2 (&ret)[4], float2 *img, int lane, float2 (&y)[8])
{
#pragma unroll
    for (int d = 0; d < 4; d++) xchg_st(img, 512 - lane - 64 * d, ret[d]);
    wave_lds_fence();
#pragma unroll
    for (int d = 5; d < 8; d++) y[d] = xchg_ld(img, lane + 64 * d);
    wave_lds_fence();
}

struct FrameTables {
    WaveTwiddles tw;
    float2 win[8];     // per lane: window pair of samples (2 lane + 128 r, +1), halved
    float2 wsp[2];     // W^(2 lane), W^(2 lane + 1)
};

__device__ __forceinline__ void load_frame_tables(FrameTables &t, const float2 *__restrict__ table, int lane)
{
    load_wave_twiddles(t.tw, table, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) t.win[r] = table[kStftWin + lane + 64 * r];
    t.wsp[0] = table[kStftSplit + 2 * lane];
    t.wsp[1] = table[kStftSplit + 2 * lane + 1];
}

__device__ __forceinline__ void load_split_twiddles(SplitTwiddles &t, const float2 *__restrict__ table, int lane)
{
#pragma unroll
    for (int d = 0; d < 8; d++) t.w[d] = table[kStftSplit + lane + 64 * d];
}

// Re-lays a half-frame out from the 16-byte-per-lane load image (lane holds samples
// 8 lane .. 8 lane + 7) to the transform's layout (lane gets the pairs 2 lane + 128 r, r < 4).
__device__ __forceinline__ void relayout_half(unsigned int *stage, int lane, u32x4 img, unsigned int *out4)
{
    reinterpret_cast<u32x4 *>(stage)[lane] = img;
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; r++) out4[r] = stage[lane + 64 * r];
    wave_lds_fence();
}

// BeamForming_MVDR_ver1.cpp's frame (:136-141,:195-196): [first 511 samples of the previous block,
// block, 0].  `img_prev` / `img_cur` are the two blocks as 16-byte-per-lane load images; they are
// laid side by side in LDS (stage: 1024 + 8 shorts) and every lane picks its eight sample pairs
// (2 lane + 128 r, +1), r < 8: position p reads the buffer at p for p < 511 and at p + 1 after.
__device__ __forceinline__ void mvdr_frame_pairs(unsigned int *stage32, int lane, u32x4 img_prev, u32x4 img_cur,
                                                 float2 (&v)[8], float scale)
{
    reinterpret_cast<u32x4 *>(stage32)[lane] = img_prev;
    reinterpret_cast<u32x4 *>(stage32)[64 + lane] = img_cur;
    if (lane == 0) stage32[512] = 0u;
    wave_lds_fence();
    // position p = 2 lane + 128 r: r < 3 lies below 510 for every lane, r = 3 reaches 510 in lane 63 only, r > 3 lies
    // above (and the pair at 1022 ends in the zero of stage32[512])
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i = lane + 64 * r;
        const unsigned int a = stage32[i];
        float x0, x1;
        if (r < 3) { x0 = (float)(short)(a & 0xffffu); x1 = (float)((int)a >> 16); }
        else {
            const unsigned int b = stage32[i + 1];
            if (r == 3) { x0 = (float)(short)(a & 0xffffu); x1 = lane == 63 ? (float)(short)(b & 0xffffu) : (float)((int)a >> 16); }
            else { x0 = (float)((int)a >> 16); x1 = (float)(short)(b & 0xffffu); }
        }
        v[r] = make_float2(scale * x0, scale * x1);
    }
    wave_lds_fence();
}

// Which estimate applies to block j, from plan_kernel's per-64-block summaries
// (denoise_kernels.hip): the count of latches at or before j.
__device__ __forceinline__ int version_of(const int *__restrict__ ver_base,
                                          const unsigned long long *__restrict__ snap_mask, long j)
{
    const unsigned long long m = snap_mask[j >> 6] & (~0ull >> (63 - (int)(j & 63)));
    return ver_base[j >> 6] + __popcll(m);
}


}  // namespace jdsp
