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

// Diagnostic build only (-DJDSP_STAMP, tools/wave_timeline.py): every wave of fastconv1024_pairs_kernel records when it
// started and ended (s_memrealtime: the 100 MHz constant clock) and where it ran (HW_ID: SIMD, CU, SE, XCC_ID).  The
// values go to a buffer nothing else reads; no output is computed from them.
#ifndef JDSP_STAMP
#define JDSP_STAMP 0
#endif
#if JDSP_STAMP
__device__ unsigned long long g_wave_stamp[3 * 8192];
#endif

__device__ __forceinline__ float conv_sample(const ConvStream &s, long pos)
{
    if (pos + s.global0 < s.valid_from) return 0.f;
    if (pos >= 0) return pos < s.n_samples ? (float)s.pcm[pos] : 0.f;
    const long h = pos + s.hist_len;
    return h >= 0 ? (float)s.hist[h] : 0.f;
}

__global__ void spectrum_to_f32_kernel(const double2 *__restrict__ in, float2 *__restrict__ out, long n, float scale)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2(scale * (float)in[i].x, scale * (float)in[i].y);   // scale: a power of two, exact
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
        for (int r = 0; r < 8; r++) v[r] = make_float2((float)src[128 * r], (float)src[128 * r + 1]);
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const long p0 = end - 1024 + 2 * lane + 128 * r;
            v[r] = make_float2(conv_sample(s, p0), conv_sample(s, p0 + 1));
        }
    }
    // the split's 1/2 and the inverse transform's 1/1024 (:156) are folded into H (a power of two: exact)
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
        for (int d = 0; d < 8; d++) *reinterpret_cast<float2 *>(ys + 2 * lane + 128 * d) = y[d];
        wave_lds_fence();
        // the kept samples leave as dwords (two samples) wherever the output address allows, with a single
        // 16-bit store at an odd head and an odd tail
        const int head = (int)((reinterpret_cast<uintptr_t>(o) >> 1) & 1);
        const int n_pairs = (block - head) >> 1;
        const float *yk = ys + n_taps - 1;
        if (lane == 0 && head) o[0] = (short)cast_i16_bits(yk[0]);
        if (lane == 1 && ((block - head) & 1)) o[block - 1] = (short)cast_i16_bits(yk[block - 1]);
        unsigned int *o32 = reinterpret_cast<unsigned int *>(o + head);
        for (int p = lane; p < n_pairs; p += 64) {
            const int i = head + 2 * p;
            o32[p] = cast_i16x2_bits(yk[i], yk[i + 1]);
        }
        if (pc)
            for (int i = lane; i < block; i += 64) pc[i] = yk[i];
        wave_lds_fence();
    }
}

// The same convolution with the spectrum kept in registers and every mirror pair of bins owned by one lane (frame_io.h,
// PairTwiddles): five split / multiply / pre-split items per lane and filter instead of eight, no natural-order image,
// no Z' image, and the kept samples leave straight from the inverse transform's registers -- lane l holds the sample
// pair (2 l + 128 d, +1) in y[d], the output address of sample n is obase + n, and whichever of the two pairings
// (n even | n odd first) is dword-aligned at this block's output address is stored as one dword per lane and register,
// the odd pairing taking its second half from the next lane (DPP wave_shl:1; lane 63: lane 0's next register).  Waves
// are persistent over blocks with the filters' 2 x 5 spectrum values per lane in registers.
// H spectra are Hermitian (real taps), so Y[1024 - k] = conj(Y[k]) holds exactly as the pair scheme assumes.
#ifndef JDSP_CONV1024_PAIRS
#define JDSP_CONV1024_PAIRS 1
#endif
#ifndef JDSP_CONV_PLAIN_STORES
#define JDSP_CONV_PLAIN_STORES 0
#endif
#ifndef JDSP_CONV_ABLATE
#define JDSP_CONV_ABLATE 0          // timing-only ablations of fastconv1024_pairs_kernel (tools/build_variant.sh): wrong results
#endif
#ifndef JDSP_CONV1024_GRID
#define JDSP_CONV1024_GRID (1024 * JDSP_CONV1024_MINWAVES)
#endif
// One register row of kept samples: lane l offers the pair (va, vb) = samples (na, na + 1), na = 2 l + 128 d + odd.  With
// N0 / N1 known when the kernel is built a row is kept whole (one dword store per lane, the usual case), not at all, or
// partly -- the rows holding the first and the last kept sample: per-lane tests there.  pc (tests only): pre-cast floats.
template <int N0, int N1, bool PC>
__device__ __forceinline__ void constexpr_row(int d, int odd, short *obase, unsigned int *p32, float *pc, int lane,
                                              float va, float vb)
{
    const int first = 128 * d + odd;                               // lane 0's na; d and odd are constants after unrolling
    if (first + 128 <= N0 || first >= N1) return;
    const int na = 2 * lane + first;
    if (first >= N0 && first + 128 <= N1) {
#if JDSP_CONV_ABLATE & 8                                          /* 8, timing-only: casts and packing stay, the row stores never execute */
        const unsigned int pk = cast_i16x2_bits(va, vb);
        if (pk == 0xdeadbeefu && __builtin_amdgcn_readfirstlane((int)pk) == 123) p32[64 * d] = pk;
        return;
#endif
#if JDSP_CONV_PLAIN_STORES                                        /* timing-only A/B: cached stores (L2 merges the partial lines of the odd-pitch rows) */
        p32[64 * d] = cast_i16x2_bits(va, vb);
#else
        __builtin_nontemporal_store(cast_i16x2_bits(va, vb), p32 + 64 * d);
#endif
        if (PC) { pc[na] = va; pc[na + 1] = vb; }
        return;
    }
    const bool ka = na >= N0 && na < N1, kb = na + 1 >= N0 && na + 1 < N1;
    if (ka && kb) p32[64 * d] = cast_i16x2_bits(va, vb);
    else if (ka) obase[na] = (short)cast_i16_bits(va);
    else if (kb) obase[na + 1] = (short)cast_i16_bits(vb);
    if (PC) {
        if (ka) pc[na] = va;
        if (kb) pc[na + 1] = vb;
    }
}

// N0 = n_taps - 1, BLOCK: compile-time, so that which register rows are kept whole, partly or not at all is decided
// when the kernel is built (instantiated for BASELINE config 2's 256 taps / 769-sample blocks; other shapes take
// fastconv1024_kernel).
#ifndef JDSP_CONV1024_H_IN_REGS
#define JDSP_CONV1024_H_IN_REGS 1
#endif
#ifndef JDSP_CONV1024_MINWAVES
#define JDSP_CONV1024_MINWAVES 3
#endif
#ifndef JDSP_CONV1024_X2
#define JDSP_CONV1024_X2 1          // 1: a filter pair's two inverse transforms staggered in one wave; 0: one after the other (round 2)
#endif
#if JDSP_CONV1024_X2 && !JDSP_CONV1024_H_IN_REGS
#error "JDSP_CONV1024_X2 takes the filter spectra from registers"
#endif
// PC: the pre-cast tap (tests only) is a build-time property of the kernel -- as a run-time test per register row it cut
// the output stage into two dozen basic blocks that the scheduler could not interleave with anything
template <int NF, int N0, int BLOCK, bool PC>
__global__ __launch_bounds__(64, JDSP_CONV1024_MINWAVES) void fastconv1024_pairs_kernel(ConvStream s, long n_out_blocks, int first_block,
                                                                const float2 *__restrict__ Hall,
                                                                const float2 *__restrict__ table, short *__restrict__ out,
                                                                float *__restrict__ precast, long plane,
                                                                short *__restrict__ hist_out)
{
    // NF == 2: the two ears' inverse transforms run staggered in one wave (wave_fft512.h), each in its own scratch
    __shared__ __attribute__((aligned(16))) float2 lds[(NF == 2 && JDSP_CONV1024_X2) ? 2 * kWaveLdsComplex : kWaveLdsComplex];
    const int lane = threadIdx.x;
    if ((long)blockIdx.x >= n_out_blocks) return;
#if JDSP_STAMP
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (blockIdx.x == gridDim.x - 1 && hist_out) {
        // the history the next call starts from (conv_hist_update_kernel's job, without its launch): the last
        // hist_len samples of [previous history | this call's samples]; hist_out is the other of the handle's two buffers
        for (int i = lane; i < s.hist_len; i += 64) {
            const long pos = s.n_samples - s.hist_len + i;
            short v = 0;
            if (pos >= 0) v = s.pcm[pos];
            else if (pos + s.hist_len >= 0) v = s.hist[pos + s.hist_len];
            hist_out[i] = v;
        }
    }
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    PairTwiddles pw;
    load_pair_twiddles(pw, table, lane);
    constexpr int block = BLOCK, n0 = N0, n1 = N0 + BLOCK;       // samples [n0, n1) of every segment are kept (:156-158)
    static_assert(n1 <= 1024, "segment of 1024 samples");
#if JDSP_CONV1024_H_IN_REGS
    float2 Hlo[NF][5], Hhi[NF][5];                              // the filters' spectrum values of this lane's ten bins
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int d = 0; d < 5; d++) {
            Hlo[f][d] = Hall[(size_t)f * 1024 + lane + 64 * d];
            Hhi[f][d] = Hall[(size_t)f * 1024 + lane + 64 * d + 512];
        }
#endif
    // Plain loads, no per-sample tests (but see the first block below).  A segment starts at a
    // multiple of BLOCK samples, so its sample pairs are 2-byte-aligned dwords (the hardware takes them as they are);
    // the next block's eight are requested before this block's arithmetic starts -- without that the wave spent 61 % of
    // its time in s_waitcnt (profiles/r02_fastconv_pairs.txt).
    typedef unsigned int u32_a2 __attribute__((aligned(2)));
    unsigned int cur[8], nxt[8];
    {
        const long start = (long)(first_block + blockIdx.x + 1) * block - 1024;
        if (start >= 0 && start + s.global0 >= s.valid_from) {
            const short *src = s.pcm + start + 2 * lane;
#pragma unroll
            for (int r = 0; r < 8; r++) nxt[r] = *reinterpret_cast<const u32_a2 *>(src + 128 * r);
        } else {
            // the first block or two of a call: the segment reaches into the previous call's history or the stream's
            // silent head (only a wave's FIRST block can: the ones it goes on to lie a whole grid further)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const long p0 = start + 2 * lane + 128 * r;
                const unsigned int a = (unsigned int)(int)conv_sample(s, p0) & 0xffffu, b = (unsigned int)(int)conv_sample(s, p0 + 1);
                nxt[r] = a | (b << 16);
            }
        }
    }
    // A SIMD arbitrates its resident waves by priority, then AGE: with equal priorities the oldest of the three waves a
    // SIMD holds runs nearly unimpeded and the youngest gets what is left -- of equal static shares a third finished at
    // 97 us, a third at 122 and a third at 150 (tools/wave_timeline.py, profiles/r03_wave_timeline.txt), the chip a third
    // empty for the last third of the launch.  Handing the blocks out at run time instead costs an atomic per block
    // (65,535 returning atomics on one line: 1,030 us).  So every wave walks the priority levels, one step per block:
    // priority outranks age, each wave does the same number of blocks at each level, and equal shares end together.
#ifndef JDSP_CONV_PRIO
#define JDSP_CONV_PRIO 1
#endif
    unsigned prio_step = JDSP_CONV_PRIO == 2 ? blockIdx.x % 3u : blockIdx.x >> 10;
    // vmcnt counts loads and stores together, in issue order, and across the loop's back edge the compiler waits for
    // vmcnt(0): taking the prefetched samples at the TOP of an iteration therefore waited for the previous block's two
    // dozen output stores to complete, every block.  They are taken (cur <- nxt) just BEFORE this block's stores instead:
    // the loads were issued a whole block's arithmetic earlier, and the stores then have until the next block's take.
#ifndef JDSP_CONV_TAKE_EARLY
#define JDSP_CONV_TAKE_EARLY 1
#endif
    constexpr bool take_early = JDSP_CONV_TAKE_EARLY && NF == 2 && JDSP_CONV1024_X2;   // (the one-filter form stores inside its filter loop)
    if (take_early) {
#pragma unroll
        for (int r = 0; r < 8; r++) cur[r] = nxt[r];
    }
    for (long e = blockIdx.x; e < n_out_blocks; e += gridDim.x) {
#if JDSP_CONV_PRIO
        {
            const unsigned lvl = prio_step % 3u;
            if (lvl == 0) __builtin_amdgcn_s_setprio(0);
            else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
            prio_step++;
        }
#endif
        if (!take_early) {
#pragma unroll
            for (int r = 0; r < 8; r++) cur[r] = nxt[r];
        }
        if (e + gridDim.x < n_out_blocks) {
            const short *src = s.pcm + ((long)(first_block + e + gridDim.x + 1) * block - 1024) + 2 * lane;
#pragma unroll
            for (int r = 0; r < 8; r++) nxt[r] = *reinterpret_cast<const u32_a2 *>(src + 128 * r);
        }
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = unpack_i16x2(cur[r]);
        wave_fft512<false>(v, lds, lane, tw);
        float2 zr[5], lo[5], hi[5];
        wave_lds_fence();
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 ev = cadd_conj(v[d], zr[d]);
            const float2 od = csub_conj_mj(v[d], zr[d]);
            const float2 p = cmul(pw.w[d], od);
            lo[d] = cadd(ev, p);
            hi[d] = csub(ev, p);
        }
        float2 yy[NF][8];
#if JDSP_CONV1024_X2
        if (NF == 2) {
            float2 ret[2][4];
#pragma unroll
            for (int f = 0; f < 2; f++)
#pragma unroll
                for (int d = 0; d < 5; d++) {
                    const float2 yl = cmul(lo[d], Hlo[f][d]), yh = cmul(hi[d], Hhi[f][d]);       // :150-151
                    if (d < 4) presplit_inv_pair(yl, yh, pw.w[d], yy[f][d], ret[f][d]);
                    else yy[f][d] = presplit_inv_reg(yl, yh, pw.w[d]);
                }
            float2 *lds_b = lds + (NF == 2 ? kWaveLdsComplex : 0);
            // both ears' mirror halves go to their owners in one round trip
#pragma unroll
            for (int d = 0; d < 4; d++) { xchg_st(lds, 512 - lane - 64 * d, ret[0][d]); xchg_st(lds_b, 512 - lane - 64 * d, ret[1][d]); }
            wave_lds_fence();
#pragma unroll
            for (int d = 5; d < 8; d++) { yy[0][d] = xchg_ld(lds, lane + 64 * d); yy[NF - 1][d] = xchg_ld(lds_b, lane + 64 * d); }
            wave_lds_fence();
            wave_fft512_x2_staggered<true>(yy[0], yy[NF - 1], lds, lds_b, lane, tw);
        }
#endif
        if (take_early) {
#pragma unroll
            for (int r = 0; r < 8; r++) { cur[r] = nxt[r]; asm volatile("" : "+v"(cur[r])); }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            float2 (&y)[8] = yy[f];
            if (!(NF == 2 && JDSP_CONV1024_X2)) {
                float2 ret[4];
#pragma unroll
                for (int d = 0; d < 5; d++) {
#if JDSP_CONV1024_H_IN_REGS
                    const float2 yl = cmul(lo[d], Hlo[f][d]), yh = cmul(hi[d], Hhi[f][d]);           // :150-151
#else
                    const float2 *H = Hall + (size_t)f * 1024 + lane;         // 16 KB for a filter pair: L1 / L2 hits
                    const float2 yl = cmul(lo[d], H[64 * d]), yh = cmul(hi[d], H[64 * d + 512]);     // :150-151
#endif
                    if (d < 4) presplit_inv_pair(yl, yh, pw.w[d], y[d], ret[d]);
                    else y[d] = presplit_inv_reg(yl, yh, pw.w[d]);
                }
                pair_return_lds(ret, lds, lane, y);
#if !(JDSP_CONV_ABLATE & 1)                                            /* 1, timing-only: no inverse transforms */
                wave_fft512<true>(y, lds, lane, tw);
#endif
                wave_lds_fence();
            }
            short *obase = out + (size_t)f * plane + e * block - n0;         // obase[n] = where sample n of the segment goes
            const int odd = (int)((reinterpret_cast<uintptr_t>(obase) >> 1) & 1);  // wave-uniform: which pairing is dword-aligned
            unsigned int *p32 = reinterpret_cast<unsigned int *>(obase + 2 * lane + odd);
            float *pc = PC ? precast + (size_t)f * plane + e * block - n0 : nullptr;
#if JDSP_CONV_ABLATE & 4                                               /* 4, timing-only: no output stores (one lane keeps the values alive) */
            if (y[0].x == 1.2345e30f && y[7].y == 5.4321e-30f && y[3].x == y[4].y && y[2].y == y[5].x && y[1].x == y[6].y) {
                out[lane] = (short)cast_i16_bits(y[0].x + y[1].x + y[2].x + y[3].x + y[4].x + y[5].x + y[6].x + y[7].x +
                                                 y[0].y + y[1].y + y[2].y + y[3].y + y[4].y + y[5].y + y[6].y + y[7].y);
            }
            continue;
#endif
#if JDSP_CONV_ABLATE & 2                                               /* 2, timing-only: aligned stores only */
            if (true) {
#pragma unroll
                for (int d = 2; d < 8; d++)
                    __builtin_nontemporal_store(cast_i16x2_bits(y[d].x, y[d].y),
                                                reinterpret_cast<unsigned int *>(out + (size_t)f * plane + e * block + odd) + lane + 64 * (d - 2));
            } else
#endif
            if (!odd) {
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    constexpr_row<n0, n1, PC>(d, 0, obase, p32, pc, lane, y[d].x, y[d].y);
                }
            } else {
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    // (y[n], y[n + 1]) for odd n = 2 lane + 1 + 128 d: the second half sits in the next lane's register
                    int nx = __builtin_amdgcn_update_dpp(0, __float_as_int(y[d].x), 0x130, 0xf, 0xf, true);   // wave_shl:1
                    if (d < 7) {
                        const int first = __builtin_amdgcn_readfirstlane(__float_as_int(y[d < 7 ? d + 1 : d].x));
                        nx = lane == 63 ? first : nx;
                    }
                    constexpr_row<n0, n1, PC>(d, 1, obase, p32, pc, lane, y[d].y, __int_as_float(nx));
                }
            }
        }
    }
#if JDSP_STAMP
    if (lane == 0 && blockIdx.x < 8192) {
        unsigned int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wave_stamp[3 * blockIdx.x] = stamp_t0;
        g_wave_stamp[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        g_wave_stamp[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

#if JDSP_STAMP
int read_wave_stamps(unsigned long long *host, int n)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_stamp), sizeof(unsigned long long) * 3 * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

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
int launch_spectrum_to_f32(hipStream_t s, const double2 *in, float2 *out, long n, float scale)
{
    hipLaunchKernelGGL(spectrum_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, scale);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fastconv(hipStream_t st, int n_fft, const ConvStream &s, long n_out_blocks, int first_block, int block,
                    int n_taps, int n_filters, const float2 *H, const float2 *table, const float2 *tw4096,
                    const float2 *tw8192, short *out, float *precast, long plane, short *hist_out)
{
    if (n_out_blocks > 0) {
        if (n_fft == 1024 && JDSP_CONV1024_PAIRS && (n_filters == 1 || n_filters == 2) && n_taps == 256 && block == 769) {
            // persistent waves (a wave's first block may reach into the history / the silent head: handled there)
            const unsigned grid = (unsigned)(n_out_blocks < JDSP_CONV1024_GRID ? n_out_blocks : JDSP_CONV1024_GRID);
            short *ho = s.hist_len > 0 ? hist_out : (short *)nullptr;
#define JDSP_LAUNCH_PAIRS(NF_, PC_) hipLaunchKernelGGL((fastconv1024_pairs_kernel<NF_, 255, 769, PC_>), dim3(grid), dim3(64), 0, st, s, \
                                                       n_out_blocks, first_block, H, table, out, precast, plane, ho)
            if (n_filters == 1) { if (precast) JDSP_LAUNCH_PAIRS(1, true); else JDSP_LAUNCH_PAIRS(1, false); }
            else { if (precast) JDSP_LAUNCH_PAIRS(2, true); else JDSP_LAUNCH_PAIRS(2, false); }
#undef JDSP_LAUNCH_PAIRS
            return hipGetLastError() == hipSuccess ? 0 : -1;
        } else if (n_fft == 1024)
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

// ---------------------------------------------------------------------------------------
// n_fft = 8192 with a block length that is a multiple of 512 (the reference-native 8192 / 7169 / 1024):
// the same linear convolution computed as a UNIFORMLY PARTITIONED overlap-save.  The 8192-point segment of
// every 1024-sample block overlaps its predecessor by 7/8, so the reference shape transforms every sample
// eight times; here the taps are cut into P = ceil(n_taps / 512) partitions of 512, every 512-sample
// sub-block of the stream is transformed ONCE (1024-point frame [previous sub-block, sub-block], the
// half-spectrum STFT kernel with a rectangular window), and output sub-block c is
//     y_c = IDFT( sum_p X[c - p] H_p )   (second half of the 1024-point result),
// the standard frequency-domain delay line.  Results equal the 8192-point formulation to FP32 rounding
// (same sums, associated differently) and are held to the same parity tests.
//   conv_stage_kernel   lays [history | this call's samples] out as one int16 run, with the reference's
//                       silent head (conv_sample) already applied
//   stft1024_hop512_half_kernel (stft_kernels.hip)   X rows, bins 0..512 at a pitch of 520
//   fastconv_upols_kernel   one wave per output sub-block: P multiply-accumulates per bin from L2-resident
//                       rows, Hermitian extension through LDS, pre-split, inverse wave FFT, cast
__global__ void conv_stage_kernel(ConvStream s, long lead, long n_total, short *__restrict__ staged,
                                  short *__restrict__ hist_out)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;         // eight samples per thread
    if (i >= n_total) return;
    const long pos = i - lead;
    const long hist0 = n_total - s.hist_len;                                  // the next call's history = the tail
    if (pos >= 0 && pos + 8 <= s.n_samples && pos + s.global0 >= s.valid_from && i + 8 <= n_total &&
        (((uintptr_t)(s.pcm + pos)) & 15u) == 0) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(s.pcm + pos);
        *reinterpret_cast<u32x4 *>(staged + i) = v;
        if (i + 8 > hist0) {
            const short *sv = reinterpret_cast<const short *>(&v);
            for (int k = 0; k < 8; k++)
                if (i + k >= hist0) hist_out[i + k - hist0] = sv[k];
        }
        return;
    }
    for (int k = 0; k < 8 && i + k < n_total; k++) {
        const short v = (short)conv_sample(s, pos + k);
        staged[i + k] = v;
        // the history keeps the samples themselves, silent head or not (conv_hist_update_kernel)
        if (i + k >= hist0) {
            const long p = pos + k;
            hist_out[i + k - hist0] = p >= 0 ? s.pcm[p] : (p + s.hist_len >= 0 ? s.hist[p + s.hist_len] : (short)0);
        }
    }
}

#ifndef JDSP_UPOLS_SIMPLE
#define JDSP_UPOLS_SIMPLE 0       // 1: the one-wave-per-sub-block kernel (kept for comparison)
#endif
#ifndef JDSP_UPOLS_UNROLL
#define JDSP_UPOLS_UNROLL 1       // partitions whose loads are in flight together
#endif
constexpr int kUpolsPitch = kUpolsRowPitch;   // 520 complex elements per spectrum row (513 used): 65 x 64 bytes

template <int J>
__device__ __forceinline__ void upols_presplit_j(const float2 *img, float2 *zout, int lane, const float2 *wsp)
{
    const int m = 128 * J + 2 * lane;
    const float4 yy = *reinterpret_cast<const float4 *>(&img[m]);
    float2 zr0, zr1;
    load_mirror_pair(img, m, zr0, zr1);                              // Y[512 - m], Y[511 - m]
    // Y[m + 512] = conj(Y[512 - m]): the output is real
    zout[2 * J] = presplit_inv<J>(make_float2(yy.x, yy.y), make_float2(zr0.x, -zr0.y), wsp[0]);
    zout[2 * J + 1] = presplit_inv<J>(make_float2(yy.z, yy.w), make_float2(zr1.x, -zr1.y), wsp[1]);
}

__global__ __launch_bounds__(64) void fastconv_upols_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Hp,
                                                            int n_part, int n_filters, long first_row, long n_sub,
                                                            const float2 *__restrict__ table, short *__restrict__ out,
                                                            float *__restrict__ precast, long plane)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                        // neighbours share an XCD's L2: they read
    const long t = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // n_part - 1 of the same rows
    if (t >= n_sub) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    for (int f = 0; f < n_filters; f++) {
        float2 acc[8], acc512 = make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = make_float2(0.f, 0.f);
        const float2 *xrow = X + (first_row + t) * kUpolsPitch;      // frame whose second half is this sub-block
        const float2 *hrow = Hp + (size_t)f * n_part * kUpolsPitch;
#pragma unroll JDSP_UPOLS_UNROLL
        for (int p = 0; p < n_part; p++, xrow -= kUpolsPitch, hrow += kUpolsPitch) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int m = 128 * j + 2 * lane;
                const float4 x = *reinterpret_cast<const float4 *>(xrow + m);
                const float4 h = *reinterpret_cast<const float4 *>(hrow + m);
                acc[2 * j] = cadd(acc[2 * j], cmul(make_float2(x.x, x.y), make_float2(h.x, h.y)));
                acc[2 * j + 1] = cadd(acc[2 * j + 1], cmul(make_float2(x.z, x.w), make_float2(h.z, h.w)));
            }
            if (lane == 0) acc512 = cadd(acc512, cmul(xrow[512], hrow[512]));
        }
        // natural-order image of Y[0..512]
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(acc[2 * j].x, acc[2 * j].y, acc[2 * j + 1].x, acc[2 * j + 1].y);
        if (lane == 0) lds[512] = acc512;
        wave_lds_fence();
        float2 z[8], y[8];
        upols_presplit_j<0>(lds, z, lane, wsp);
        upols_presplit_j<1>(lds, z, lane, wsp);
        upols_presplit_j<2>(lds, z, lane, wsp);
        upols_presplit_j<3>(lds, z, lane, wsp);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
        wave_lds_fence();
        wave_fft512<true>(y, lds, lane, tw);
        wave_lds_fence();
        // samples 512..1023 of the frame are the valid ones: y[4..7] hold the pairs 2 lane + 128 d
        unsigned int *o = reinterpret_cast<unsigned int *>(out + (size_t)f * plane + t * 512) + lane;
        float *pc = precast ? precast + (size_t)f * plane + t * 512 : nullptr;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const float2 v = make_float2(y[d + 4].x * (1.0f / 1024.0f), y[d + 4].y * (1.0f / 1024.0f));
            __builtin_nontemporal_store(cast_i16x2_bits(v.x, v.y), o + 64 * d);
            if (pc) *reinterpret_cast<float2 *>(pc + 2 * lane + 128 * d) = v;
        }
    }
}

// The same sums with the traffic cut down: one workgroup = 4 waves, one wave = kUpolsK consecutive output
// sub-blocks.  A row X[r] feeds sub-block t through H_(t-r), so a wave walking the rows it needs from the
// newest down uses every row for up to kUpolsK outputs (P + kUpolsK - 1 rows from L2 instead of
// kUpolsK * P), with the next row in flight while the current one is used, and the partition spectra sit in
// LDS, loaded once per workgroup.  (One wave per sub-block reading P rows and P partition spectra from L2
// ran at 22 TB/s of L2 traffic: 44 us for the native shape; profiles/r01_fastconv_partitioned.txt.)
#ifndef JDSP_UPOLS_SHAPE
#define JDSP_UPOLS_SHAPE 1        // 0: 4 waves x 4 outputs, 4 rows in flight (2 waves/SIMD); 1: 16 waves x 2 outputs, 2 rows
#endif                            //    in flight (4 waves/SIMD: the workgroup fills a CU, LDS = 60 KB of spectra + 73 KB scratch)
#if JDSP_UPOLS_SHAPE
constexpr int kUpolsWaves = 16, kUpolsK = 2, kUpolsDepth = 2;
#else
constexpr int kUpolsWaves = 4, kUpolsK = 4, kUpolsDepth = 4;
#endif

__global__ __launch_bounds__(kUpolsWaves * 64) void fastconv_upols4_kernel(const float2 *__restrict__ X, const float2 *__restrict__ Hp,
                                                              int n_part, int n_filters, long first_row, long n_sub,
                                                              long max_row, const float2 *__restrict__ table,
                                                              short *__restrict__ out, float *__restrict__ precast,
                                                              long plane)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *Hs = smem;                                                        // [n_part][520]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *lds = smem + (size_t)n_part * kUpolsPitch + (size_t)wave * kWaveLdsComplex;   // this wave's scratch
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long wg = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const long t0 = (wg * kUpolsWaves + wave) * kUpolsK;
    const bool live = t0 < n_sub;
    for (int f = 0; f < n_filters; f++) {
        {
            const float4 *src = reinterpret_cast<const float4 *>(Hp + (size_t)f * n_part * kUpolsPitch);
            float4 *dst = reinterpret_cast<float4 *>(Hs);
            for (int i = threadIdx.x; i < n_part * (kUpolsPitch / 2); i += kUpolsWaves * 64) dst[i] = src[i];
        }
        __syncthreads();
        if (live) {
            float2 acc[kUpolsK][8], acc512[kUpolsK];
#pragma unroll
            for (int k = 0; k < kUpolsK; k++) {
                acc512[k] = make_float2(0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 8; q++) acc[k][q] = make_float2(0.f, 0.f);
            }
            const long r_top = first_row + t0 + (kUpolsK - 1);
            auto row_of = [&](int q) {
                long r = r_top - q;
                return X + (r > max_row ? max_row : r) * kUpolsPitch;        // rows past the end feed no written output
            };
            // kUpolsDepth rows in flight: an iteration is ~100 VALU instructions, an L2 round trip several times that
            float4 xb[kUpolsDepth][4];
            float2 xb512[kUpolsDepth];
#pragma unroll
            for (int u = 0; u < kUpolsDepth; u++) {
                const float2 *xr = row_of(u);
#pragma unroll
                for (int j = 0; j < 4; j++) xb[u][j] = *reinterpret_cast<const float4 *>(xr + 128 * j + 2 * lane);
                xb512[u] = xr[512];
            }
            const int n_iter = n_part + kUpolsK - 1;
            for (int q0 = 0; q0 < n_iter; q0 += kUpolsDepth) {
#pragma unroll
                for (int u = 0; u < kUpolsDepth; u++) {
                    const int q = q0 + u;                                    // q >= n_iter: every p below is out of range
#pragma unroll
                    for (int k = 0; k < kUpolsK; k++) {
                        const int p = q - (kUpolsK - 1) + k;                 // wave-uniform
                        if (p < 0 || p >= n_part) continue;
                        const float2 *hr = Hs + p * kUpolsPitch;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float4 h = *reinterpret_cast<const float4 *>(hr + 128 * j + 2 * lane);
                            const float4 x = xb[u][j];
                            acc[k][2 * j] = cadd(acc[k][2 * j], cmul(make_float2(x.x, x.y), make_float2(h.x, h.y)));
                            acc[k][2 * j + 1] = cadd(acc[k][2 * j + 1], cmul(make_float2(x.z, x.w), make_float2(h.z, h.w)));
                        }
                        acc512[k] = cadd(acc512[k], cmul(xb512[u], hr[512]));
                    }
                    if (q + kUpolsDepth < n_iter) {                          // refill this slot
                        const float2 *xr = row_of(q + kUpolsDepth);
#pragma unroll
                        for (int j = 0; j < 4; j++) xb[u][j] = *reinterpret_cast<const float4 *>(xr + 128 * j + 2 * lane);
                        xb512[u] = xr[512];
                    }
                }
            }
            WaveTwiddles tw;
            load_wave_twiddles(tw, table, lane);
            const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
#pragma unroll
            for (int k = 0; k < kUpolsK; k++) {
                const long t = t0 + k;
                if (t >= n_sub) break;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) =
                        make_float4(acc[k][2 * j].x, acc[k][2 * j].y, acc[k][2 * j + 1].x, acc[k][2 * j + 1].y);
                if (lane == 0) lds[512] = acc512[k];
                wave_lds_fence();
                float2 z[8], y[8];
                upols_presplit_j<0>(lds, z, lane, wsp);
                upols_presplit_j<1>(lds, z, lane, wsp);
                upols_presplit_j<2>(lds, z, lane, wsp);
                upols_presplit_j<3>(lds, z, lane, wsp);
                wave_lds_fence();
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y);
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
                wave_lds_fence();
                wave_fft512<true>(y, lds, lane, tw);
                wave_lds_fence();
                unsigned int *o = reinterpret_cast<unsigned int *>(out + (size_t)f * plane + t * 512) + lane;
                float *pc = precast ? precast + (size_t)f * plane + t * 512 : nullptr;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const float2 v = make_float2(y[d + 4].x * (1.0f / 1024.0f), y[d + 4].y * (1.0f / 1024.0f));
                    __builtin_nontemporal_store(cast_i16x2_bits(v.x, v.y), o + 64 * d);
                    if (pc) *reinterpret_cast<float2 *>(pc + 2 * lane + 128 * d) = v;
                }
            }
        }
        __syncthreads();                                                     // Hs is rewritten for the next filter
    }
}

// bins 0..512 of `rows` 1024-point FP64 spectra -> float rows at the partitioned convolver's pitch
__global__ void spectrum_rows_to_f32_kernel(const double2 *__restrict__ in, float2 *__restrict__ out, long rows)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * kUpolsPitch) return;
    const long r = i / kUpolsPitch;
    const int b = (int)(i % kUpolsPitch);
    out[i] = b <= 512 ? make_float2((float)in[r * 1024 + b].x, (float)in[r * 1024 + b].y) : make_float2(0.f, 0.f);
}

int launch_spectrum_rows_to_f32(hipStream_t s, const double2 *in, float2 *out, long rows)
{
    const long n = rows * kUpolsPitch;
    hipLaunchKernelGGL(spectrum_rows_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, rows);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fastconv_upols(hipStream_t st, const ConvStream &s, long n_out_blocks, int first_block, int block, int n_part,
                          int n_filters, const float2 *Hp, const float2 *rect_table, short *staged, float2 *X,
                          short *out, float *precast, long plane, short *hist_out)
{
    const long lead = 512L * n_part;
    const long n_total = lead + s.n_samples;
    const long n_frames = n_total / 512 - 1;                          // frame f = staged[512 f, 512 f + 1024)
    if (n_out_blocks > 0) {
        hipLaunchKernelGGL(conv_stage_kernel, dim3((unsigned)((n_total / 8 + 256) / 256)), dim3(256), 0, st, s, lead, n_total,
                           staged, s.hist_len > 0 ? hist_out : nullptr);
        if (launch_stft1024_half(st, staged, n_frames, X, kUpolsPitch, rect_table)) return -1;
        const long n_sub = n_out_blocks * (block / 512);
        // output sub-block t = call-local samples [first_block * block + 512 t, +512) = staged sub-block
        // n_part + first_block * block / 512 + t = the second half of frame (that index - 1)
        const long first_row = n_part + (long)first_block * (block / 512) - 1;
#if JDSP_UPOLS_SIMPLE
        const long grid = (n_sub + 7) / 8 * 8;
        hipLaunchKernelGGL(fastconv_upols_kernel, dim3((unsigned)grid), dim3(64), 0, st, X, Hp, n_part, n_filters, first_row,
                           n_sub, rect_table, out, precast, plane);
#else
        const long per_wg = kUpolsWaves * kUpolsK;
        const long grid = ((n_sub + per_wg - 1) / per_wg + 7) / 8 * 8;
        const size_t lds = ((size_t)n_part * kUpolsPitch + kUpolsWaves * (size_t)kWaveLdsComplex) * sizeof(float2);
        // more than 64 KB of dynamic LDS needs the attribute (per device: set on every launch, it is a host-side flag)
        if (hipFuncSetAttribute((const void *)fastconv_upols4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((16 * (size_t)kUpolsPitch + kUpolsWaves * (size_t)kWaveLdsComplex) * sizeof(float2))) != hipSuccess)
            return -1;
        hipLaunchKernelGGL(fastconv_upols4_kernel, dim3((unsigned)grid), dim3(kUpolsWaves * 64), lds, st, X, Hp, n_part, n_filters,
                           first_row, n_sub, n_frames - 1, rect_table, out, precast, plane);
#endif
    }
    if (s.hist_len > 0 && n_out_blocks <= 0)                           // otherwise the staging kernel wrote it
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
