// mvdr_kernels.hip -- BeamForming_MVDR_ver1.cpp (stretch) on gfx950: two
// microphones, one real 2x2 "spatial correlation" accumulated over non-voice runs and over ALL
// bins (:263-268), per-bin weights w = R^-1 c / (c^H R^-1 c) (:170-171), both spectra weighted
// (with the reference's in-place overwrite, :180-183), summed, inverse-transformed; samples
// 511..1022 of every frame are emitted (:192-194).
//
//   vad_kernel (denoise_kernels.hip)   :207-242 energy-only decision, window offset 511
//   plan_kernel (denoise_kernels.hip)  main()'s run counter :191-219; every event changes R
//   mvdr_corr_kernel                   :244-270 per event: the four sums over 1024 bins (by Parseval: two energies)
//   mvdr_prefix_kernel                 running R after each event (FP64)
//   mvdr_kernel                        :124-205 one block per wavefront
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

__device__ __forceinline__ u32x4 mvdr_load_block(const short *__restrict__ pcm, long n_blocks,
                                                 const short *__restrict__ prev, long j, int lane)
{
    u32x4 zero = {0u, 0u, 0u, 0u};
    if (j >= 0 && j < n_blocks) return reinterpret_cast<const u32x4 *>(pcm + j * 512)[lane];
    if (j == -1) return reinterpret_cast<const u32x4 *>(prev)[lane];
    return zero;
}

// forward transform of 8 float2 (already scaled by 0.5) -> X[m], X[m+512] for m = 128 j + 2 lane + e
__device__ __forceinline__ void spectrum_of(float2 (&v)[8], float2 *lds, int lane, const WaveTwiddles &tw,
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

// sum over the wave, valid in lane 63 (quad swaps, row mirrors, row broadcasts: five DPP steps on both halves of the double)
__device__ __forceinline__ double wave_sum_f64(double v)
{
#define JDSP_DPP_ADD64(CTRL, ROWS)                                                                             \
    {                                                                                                          \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, true);               \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, true);               \
        v += __hiloint2double(hi, lo);                                                                         \
    }
    JDSP_DPP_ADD64(0xB1, 0xf)      // quad_perm [1,0,3,2]
    JDSP_DPP_ADD64(0x4E, 0xf)      // quad_perm [2,3,0,1]
    JDSP_DPP_ADD64(0x141, 0xf)     // row_half_mirror: sums of 8
    JDSP_DPP_ADD64(0x140, 0xf)     // row_mirror: sums of 16
    JDSP_DPP_ADD64(0x142, 0xa)     // row_bcast15 into rows 1 and 3
    JDSP_DPP_ADD64(0x143, 0xc)     // row_bcast31 into rows 2 and 3
#undef JDSP_DPP_ADD64
    return v;
}

// :244-270 -- frame = [block j-1, block j] of each channel, no window, through two FP64 transforms, then over ALL 1024 bins
//   R[0][0] += |L_k|^2 / N,   R[0][1] += (-Re L_k Im R_k + Im L_k Re R_k) / N,   R[1][0] likewise,   R[1][1] += |R_k|^2 / N.
// The frames are real, so the spectra are Hermitian: the cross terms of bins k and N - k are each other's negatives (and
// those of bins 0 and N / 2 are zero) -- the reference's R[0][1], R[1][0] receive nothing but its transform's rounding,
// 1e-17 of R[0][0] -- and by Parseval sum_k |L_k|^2 / N = sum_n l_n^2, an integer below 2^41.  So an event's contribution
// is (sum l^2, 0, 0, sum r^2), exact, with no transform: it agrees with the FP64 restatement of the reference to 1e-15
// (tests/test_mvdr_gpu.py) where two FP32 transforms per event agreed to 1e-7 and cost 182 us per 65,536 events.
__device__ __forceinline__ unsigned long long sum_squares_i16x8(u32x4 w)
{
    unsigned long long acc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int a = (short)(w[q] & 0xffffu), b = (int)w[q] >> 16;
        acc += (unsigned int)(a * a) + (unsigned int)(b * b);             // <= 2^31: fits
    }
    return acc;
}

__global__ __launch_bounds__(64) void mvdr_corr_kernel(const short *__restrict__ left, const short *__restrict__ right,
                                                       long n_blocks, const MvdrState *__restrict__ st_in,
                                                       const int *__restrict__ events,
                                                       const DenoisePlan *__restrict__ plan,
                                                       double *__restrict__ delta,
                                                       const int *__restrict__ range, long ext0)
{
    // range (device, or NULL: every event) = {first, one past last} event this launch handles;
    // ext0 = global index of the first block the pcm pointers hold (sharded runs)
    const int lane = threadIdx.x;
    const int e_lo = range ? range[0] : 0, e_hi = range ? range[1] : plan->n_events;
    for (int e = e_lo + blockIdx.x; e < e_hi; e += gridDim.x) {
        const long j = events[e] - ext0;
        const unsigned long long el = sum_squares_i16x8(mvdr_load_block(left, n_blocks, st_in->prev_l, j - 1, lane)) +
                                      sum_squares_i16x8(mvdr_load_block(left, n_blocks, st_in->prev_l, j, lane));
        const unsigned long long er = sum_squares_i16x8(mvdr_load_block(right, n_blocks, st_in->prev_r, j - 1, lane)) +
                                      sum_squares_i16x8(mvdr_load_block(right, n_blocks, st_in->prev_r, j, lane));
        // wave sums by DPP moves (lane 63 ends up with the total); integers below 2^41: exact in FP64
        const double s00 = wave_sum_f64((double)el), s11 = wave_sum_f64((double)er);
        if (lane == 63) *reinterpret_cast<double4 *>(delta + (size_t)(e - e_lo) * 4) = make_double4(s00, 0.0, 0.0, s11);
    }
}

// R after event e: rver[e+1] = rver[e] + delta[e]; rver[0] = the matrix carried in (the state's,
// or for a sharded run the sum of the earlier ranks' totals).  total (may be NULL) = sum of delta.
// Two launches over tiles of 1024 events: an LDS scan inside each tile (coalesced 32-byte rows, the tile's sum to
// tile_sums), then each tile's base = matrix carried in + the sums of the tiles before it.  A stream that is half pauses
// has tens of thousands of events per call: one workgroup walking them took 218 us for 65,536 (every thread's run of
// events a cache line apart from its neighbour's).  (FP64 sums, so the grouping moves nothing above 1e-16.)
constexpr int kPrefixGrid = 64;

__global__ __launch_bounds__(1024) void mvdr_prefix_kernel(const double *__restrict__ delta,
                                                           const DenoisePlan *__restrict__ plan,
                                                           const int *__restrict__ range, const double *__restrict__ r_in,
                                                           const double *__restrict__ sums_all, int rank, MvdrState *st_out,
                                                           double *__restrict__ rver, double *__restrict__ total,
                                                           double *__restrict__ tile_sums)
{
    __shared__ double part[2][4][1024];
    const int t = threadIdx.x;
    const int n = range ? range[1] - range[0] : plan->n_events;
    if (n <= 64) {
        // a handful of events (a loud stream): four threads walk them, in the reference's own order of additions
        if (t >= 4 || blockIdx.x) return;
        double acc = r_in ? r_in[t] : 0.0;
        if (sums_all)
            for (int q = 0; q < rank; q++) acc += sums_all[q * 4 + t];
        if (rver) rver[t] = acc;
        double sum = 0.0;
        for (int e = 0; e < n; e++) {
            acc += delta[(size_t)e * 4 + t];
            sum += delta[(size_t)e * 4 + t];
            if (rver) rver[(size_t)(e + 1) * 4 + t] = acc;
        }
        if (st_out) st_out->corr[t] = acc;
        if (total) total[t] = sum;
        return;
    }
    const int n_tiles = (n + 1023) >> 10;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int e = tile * 1024 + t;
        const double4 d = e < n ? *reinterpret_cast<const double4 *>(delta + (size_t)e * 4) : make_double4(0.0, 0.0, 0.0, 0.0);
        part[0][0][t] = d.x; part[0][1][t] = d.y; part[0][2][t] = d.z; part[0][3][t] = d.w;
        __syncthreads();
        int cur = 0;
        for (int o = 1; o < 1024; o <<= 1) {                          // inclusive scan over the tile
#pragma unroll
            for (int c = 0; c < 4; c++) part[cur ^ 1][c][t] = part[cur][c][t] + (t >= o ? part[cur][c][t - o] : 0.0);
            cur ^= 1;
            __syncthreads();
        }
        if (rver && e < n)
            *reinterpret_cast<double4 *>(rver + (size_t)(e + 1) * 4) =
                make_double4(part[cur][0][t], part[cur][1][t], part[cur][2][t], part[cur][3][t]);
        if (t < 4) tile_sums[tile * 4 + t] = part[cur][t][1023];
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void mvdr_prefix_bases_kernel(const DenoisePlan *__restrict__ plan,
                                                                 const int *__restrict__ range, const double *__restrict__ r_in,
                                                                 const double *__restrict__ sums_all, int rank,
                                                                 MvdrState *st_out, double *__restrict__ rver,
                                                                 double *__restrict__ total,
                                                                 const double *__restrict__ tile_sums)
{
    __shared__ double red[4][1024];
    const int t = threadIdx.x;
    const int n = range ? range[1] - range[0] : plan->n_events;
    if (n <= 64) return;
    const int n_tiles = (n + 1023) >> 10;
    double r0[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        r0[c] = r_in ? r_in[c] : 0.0;
        if (sums_all)
            for (int q = 0; q < rank; q++) r0[c] += sums_all[q * 4 + c];
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = t; k < tile; k += 1024) {
            const double4 d = *reinterpret_cast<const double4 *>(tile_sums + (size_t)k * 4);
            s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) red[c][t] = s[c];
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (t < o) {
#pragma unroll
                for (int c = 0; c < 4; c++) red[c][t] += red[c][t + o];
            }
            __syncthreads();
        }
        const double before[4] = {red[0][0], red[1][0], red[2][0], red[3][0]};      // sum of the tiles before this one
        __syncthreads();
        const int e = tile * 1024 + t;
        if (rver) {
            if (tile == 0 && t == 0) *reinterpret_cast<double4 *>(rver) = make_double4(r0[0], r0[1], r0[2], r0[3]);
            if (e < n) {
                double4 *row = reinterpret_cast<double4 *>(rver + (size_t)(e + 1) * 4);
                const double4 loc = *row;
                *row = make_double4((r0[0] + before[0]) + loc.x, (r0[1] + before[1]) + loc.y, (r0[2] + before[2]) + loc.z,
                                    (r0[3] + before[3]) + loc.w);
            }
        }
        if (tile == n_tiles - 1 && t < 4) {
            const double tot = before[t] + tile_sums[tile * 4 + t];
            if (st_out) st_out->corr[t] = r0[t] + tot;
            if (total) total[t] = tot;
        }
    }
}

__global__ __launch_bounds__(64) void mvdr_kernel(const short *__restrict__ left, const short *__restrict__ right,
                                                  long n_blocks, long calls_before,
                                                  const MvdrState *__restrict__ st_in, MvdrState *st_out,
                                                  const int *__restrict__ ver_base,
                                                  const unsigned long long *__restrict__ snap_mask,
                                                  const double *__restrict__ rver, const double2 *__restrict__ steer,
                                                  const float2 *__restrict__ table, short *__restrict__ out,
                                                  float *__restrict__ precast, DenoiseShard sh)
{
    // One 8 KB buffer, used in turn as: transform scratch (first 584 elements) with the frame-assembly stage in
    // its tail, the merged 1024-bin spectrum, and the scratch of the inverse transform.  (Three separate arrays
    // were 15 KB per wave and capped the CU at 10 waves.)
    __shared__ __attribute__((aligned(16))) float2 buf[1024];
    float2 *lds = buf, *merged = buf;
    unsigned int *stage32 = reinterpret_cast<unsigned int *>(buf + 600);      // 520 dwords of the 848 after element 600
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long j = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (j >= n_blocks) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    const float2 wsp[2] = {table[kStftSplit + 2 * lane], table[kStftSplit + 2 * lane + 1]};
    const bool have_prev = calls_before + j > 0;

    float2 llo[8], lhi[8], rlo[8], rhi[8], v[8];
    // keep buffer of a stream's very first block: zeros (:130-131); otherwise the previous block
    const long jp = have_prev ? j - 1 : -2;
    mvdr_frame_pairs(stage32, lane, mvdr_load_block(left, n_blocks, st_in->prev_l, jp, lane),
                     mvdr_load_block(left, n_blocks, st_in->prev_l, j, lane), v, 0.5f);
    spectrum_of(v, lds, lane, tw, wsp, llo, lhi);
    mvdr_frame_pairs(stage32, lane, mvdr_load_block(right, n_blocks, st_in->prev_r, jp, lane),
                     mvdr_load_block(right, n_blocks, st_in->prev_r, j, lane), v, 0.5f);
    spectrum_of(v, lds, lane, tw, wsp, rlo, rhi);

    wave_lds_fence();                      // the split reads of the second spectrum are done: `merged` may overwrite
    // mxAutoCorr.inverse() (:170) for the matrix in effect at this block
    int ver = version_of(ver_base, snap_mask, j + sh.ver_block_off);
    if (sh.ver_row_off) ver -= *sh.ver_row_off;
    if (ver < 0) ver = 0;
    const double *R = rver + (size_t)ver * 4;
    const double a = R[0], b = R[1], c = R[2], d = R[3];
    const double invdet = 1.0 / (a * d - b * c);
    const double i00 = d * invdet, i01 = -b * invdet, i10 = -c * invdet, i11 = a * invdet;
#pragma unroll
    for (int q = 0; q < 8; q++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int bin = 128 * (q >> 1) + 2 * lane + (q & 1) + 512 * h;
            const double2 s1 = steer[bin];                            // (cos, sin)(2 PI i fs/N dTime)  :164-165
            double w0r = i00 + i01 * s1.x, w0i = i01 * s1.y;          // R^-1 c, c = (1, s1)
            double w1r = i10 + i11 * s1.x, w1i = i11 * s1.y;
            const double dr = w0r + (s1.x * w1r + s1.y * w1i);        // c^H (R^-1 c)
            const double di = w0i + (s1.x * w1i - s1.y * w1r);
            // :171, a complex quotient: one reciprocal of |d|^2 instead of four divisions (FP64: the last-place
            // difference is nine orders of magnitude inside the parity bar)
            // v_rcp_f64 and one Newton step (the compiler's IEEE division is ~18 FP64 instructions, a third of
            // this loop); 1 / 0 kept as the division gives it
            const double dn = dr * dr + di * di;
            double rn = __builtin_amdgcn_rcp(dn);
            rn = fma(rn, fma(-dn, rn, 1.0), rn);
            rn = dn == 0.0 ? (double)INFINITY : rn;
            const double t0r = (w0r * dr + w0i * di) * rn, t0i = (w0i * dr - w0r * di) * rn;
            const double t1r = (w1r * dr + w1i * di) * rn, t1i = (w1i * dr - w1r * di) * rn;
            const float lw0 = (float)t0r, lw1 = (float)-t0i, rw0 = (float)t1r, rw1 = (float)-t1i;   // :175-178 conjugates
            float2 L = h ? lhi[q] : llo[q], Rr = h ? rhi[q] : rlo[q];
            // :180-183 -- the imaginary part is formed from the ALREADY OVERWRITTEN real part
            L.x = L.x * lw0 - L.y * lw1;
            L.y = L.x * lw1 + L.y * lw0;
            Rr.x = Rr.x * rw0 - Rr.y * rw1;
            Rr.y = Rr.x * rw1 + Rr.y * rw0;
            merged[bin] = make_float2(L.x + Rr.x, L.y + Rr.y);        // :184-185
        }
    }
    wave_lds_fence();
    // The reference keeps only the real part of the inverse transform (:193) = the inverse
    // transform of the Hermitian part of the merged spectrum.
    float2 z[8];
#define JDSP_HERM(J)                                                                                  \
    {                                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 2; e++) {                                               \
            const int m = 128 * J + 2 * lane + e;                                                     \
            const float2 a0 = merged[m], a1 = merged[(1024 - m) & 1023];                              \
            const float2 b0 = merged[m + 512], b1 = merged[512 - m];                                  \
            const float2 ylo = make_float2(0.5f * (a0.x + a1.x), 0.5f * (a0.y - a1.y));               \
            const float2 yhi = make_float2(0.5f * (b0.x + b1.x), 0.5f * (b0.y - b1.y));               \
            z[2 * J + e] = presplit_inv<J>(ylo, yhi, wsp[e]);                                         \
        }                                                                                             \
    }
    JDSP_HERM(0) JDSP_HERM(1) JDSP_HERM(2) JDSP_HERM(3)
#undef JDSP_HERM
    wave_lds_fence();                      // every lane has read `merged`: the same memory becomes the inverse's scratch
#pragma unroll
    for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4 *>(&lds[128 * q + 2 * lane]) = make_float4(z[2 * q].x, z[2 * q].y, z[2 * q + 1].x, z[2 * q + 1].y);
    wave_lds_fence();
    float2 y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, tw);

    const long first_emit = sh.emit_from;                           // :201-204: the first call's block is dropped
#if JDSP_MVDR_ABLATE == 3                                   // timing only: one store per lane
    if (y[0].x + y[3].y + y[5].x + y[7].y == 1.2345f) out[lane] = 1;
    if (false) {
#else
    if (j >= first_emit && j < sh.emit_to) {
#endif
        short *o = out + (j - first_emit) * 512;
        float *pc = precast ? precast + (j - first_emit) * 512 : nullptr;
#pragma unroll
        for (int dd = 0; dd < 8; dd++) {
            const int i0 = 2 * lane + 128 * dd - 511;                // :193 rgsOutputBuffer[i] = y[i + 511] / 1024
            const float s0 = y[dd].x * (1.0f / 1024.0f), s1 = y[dd].y * (1.0f / 1024.0f);
            if (i0 >= 0 && i0 < 512) { o[i0] = (short)cast_i16_bits(s0); if (pc) pc[i0] = s0; }
            if (i0 + 1 >= 0 && i0 + 1 < 512) { o[i0 + 1] = (short)cast_i16_bits(s1); if (pc) pc[i0 + 1] = s1; }
        }
    }
    if (j == n_blocks - 1) {
        reinterpret_cast<u32x4 *>(st_out->prev_l)[lane] = reinterpret_cast<const u32x4 *>(left + j * 512)[lane];
        reinterpret_cast<u32x4 *>(st_out->prev_r)[lane] = reinterpret_cast<const u32x4 *>(right + j * 512)[lane];
    }
}

// ---- the same block with the spectra in registers and the weights out of a table --------------------------------
// Where mvdr_kernel's time goes (profiles/r02_mvdr_pairs.txt): its ~230 LDS instructions per block (the weighted
// 1024-bin spectrum written to LDS and read back four values per output bin for its Hermitian part, :193) are NOT the
// bound -- the pair-owned form below halves them for no gain -- the per-bin FP64 weights are: 104 of 253 us, although
// they change only when the matrix does.  So:
//  * mvdr_weights_kernel: the weights of every (version, bin) of the call once, as float4 (the four values :175-178
//    cast them to), when the call has at most kMvdrTableVersions - 1 events (otherwise the block kernel computes its
//    own, as before: a stream that is mostly pauses has a new version for nearly every block);
//  * mvdr_pairs_kernel: lane l owns the bins m, m + 512 of m = l + 64 d, d < 5 (frame_io.h, pair-owned split);
//    X[1024 - m] = conj X[m] and X[512 - m] = conj X[m + 512] of a real frame are in the same lane, so the four weighted
//    values an output pair needs (bins m, 1024 - m, m + 512, 512 - m, each with its own steering phase) are formed
//    where they are used, the Hermitian parts and the pre-split follow in registers, and only the mirror operands and
//    Z'[512 - m] cross lanes.
#ifndef JDSP_MVDR_PAIRS
#define JDSP_MVDR_PAIRS 1
#endif
#ifndef JDSP_MVDR_EARLY_LOADS
#define JDSP_MVDR_EARLY_LOADS 1
#endif
#ifndef JDSP_MVDR_ABLATE
#define JDSP_MVDR_ABLATE 0      // 1..4: timing-only builds that drop one part of mvdr_pairs_kernel (tools/build_variant.sh)
#endif
struct MvdrInv { double i00, i01, i10, i11; };

__device__ __forceinline__ MvdrInv mvdr_inverse(const double *__restrict__ R)     // mxAutoCorr.inverse() (:170)
{
    const double a = R[0], b = R[1], c = R[2], d = R[3];
    const double invdet = 1.0 / (a * d - b * c);
    MvdrInv iv;
    iv.i00 = d * invdet; iv.i01 = -b * invdet; iv.i10 = -c * invdet; iv.i11 = a * invdet;
    return iv;
}

// one bin's weights: w = R^-1 c / (c^H R^-1 c), c = (1, s1) (:164-171), as the four floats of :175-178
__device__ __forceinline__ float4 mvdr_bin_weights(const MvdrInv &iv, double2 s1)
{
    const double w0r = iv.i00 + iv.i01 * s1.x, w0i = iv.i01 * s1.y;          // R^-1 c
    const double w1r = iv.i10 + iv.i11 * s1.x, w1i = iv.i11 * s1.y;
    const double dr = w0r + (s1.x * w1r + s1.y * w1i);                        // c^H (R^-1 c)
    const double di = w0i + (s1.x * w1i - s1.y * w1r);
    const double dn = dr * dr + di * di;
    double rn = __builtin_amdgcn_rcp(dn);                                     // v_rcp_f64 + one Newton step; 1 / 0 as the division gives it
    rn = fma(rn, fma(-dn, rn, 1.0), rn);
    rn = dn == 0.0 ? (double)INFINITY : rn;
    const double t0r = (w0r * dr + w0i * di) * rn, t0i = (w0i * dr - w0r * di) * rn;
    const double t1r = (w1r * dr + w1i * di) * rn, t1i = (w1i * dr - w1r * di) * rn;
    return make_float4((float)t0r, (float)-t0i, (float)t1r, (float)-t1i);     // lw0, lw1, rw0, rw1: the conjugates
}

// :180-185 -- the imaginary part is formed from the ALREADY OVERWRITTEN real part
__device__ __forceinline__ float2 mvdr_apply_weights(float4 w, float2 L, float2 Rr)
{
    L.x = L.x * w.x - L.y * w.y;
    L.y = L.x * w.y + L.y * w.x;
    Rr.x = Rr.x * w.z - Rr.y * w.w;
    Rr.y = Rr.x * w.w + Rr.y * w.z;
    return make_float2(L.x + Rr.x, L.y + Rr.y);
}

// Weights from the table or computed per block?  A version's row costs 1,024 evaluations and 16 KB written and read
// back; computing costs 1,280 evaluations per block.  The table wins while versions are few against blocks (under a
// quarter here) or few outright (under 1,024: the rows stay in L2).
__device__ __forceinline__ bool mvdr_tabled(int n_events, long n_blocks)
{
    return n_events < 1024 || (n_events < kMvdrTableVersions && 4L * n_events < n_blocks);
}

__global__ __launch_bounds__(256) void mvdr_weights_kernel(const DenoisePlan *__restrict__ plan, const double *__restrict__ rver,
                                                           const double2 *__restrict__ steer, float4 *__restrict__ wtab,
                                                           long n_blocks)
{
    const int n_events = plan->n_events;
    if (!mvdr_tabled(n_events, n_blocks)) return;                          // the block kernel computes its own
    const int bin = threadIdx.x + 256 * (blockIdx.x & 3);
    const double2 s1 = steer[bin];
    for (int ver = blockIdx.x >> 2; ver <= n_events; ver += gridDim.x >> 2)
        wtab[(size_t)ver * 1024 + bin] = mvdr_bin_weights(mvdr_inverse(rver + (size_t)ver * 4), s1);
}

// ---- the beamformer's frame straight from the two blocks' dwords (round 3): no LDS staging ----------------------------
// frame = [first 511 samples of the previous block, block, 0] (BF:136-141,:195-196); lane l transforms the sample pairs
// (2 l + 128 r, + 1), r < 8.  With P / C the previous / current block as 256 dwords each:
//   r < 3, and r = 3 for l < 63:  the pair is the dword P[l + 64 r]                                  (positions < 510)
//   r = 3, l = 63              :  (P[255] low half, C[0] low half)                                   (positions 510, 511)
//   r >= 4                     :  position p = 2 l + 128 r holds block sample 2 l + 128 (r - 4) + 1: the pair is
//                                 (high half of C[k], low half of C[k + 1]), k = l + 64 (r - 4) -- C[k + 1] is the next lane's
//                                 dword (DPP wave_shl:1; lane 63: lane 0's next register, and nothing after C[255]: the 0)
// Eight coalesced dword loads per channel, four DPP moves and four v_alignbit: the 16-byte-per-lane load images went
// through LDS for this (2 ds_write_b128 + 13 ds_read_b32 per channel and two fences).
__device__ __forceinline__ void mvdr_load_block32(const short *__restrict__ pcm, long n_blocks, const short *__restrict__ prev,
                                                  long j, int lane, unsigned int (&d)[4])
{
    const unsigned int *src = nullptr;
    if (j >= 0 && j < n_blocks) src = reinterpret_cast<const unsigned int *>(pcm + j * 512);
    else if (j == -1) src = reinterpret_cast<const unsigned int *>(prev);
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = src ? src[lane + 64 * r] : 0u;
}

__device__ __forceinline__ void mvdr_frame_pairs_direct(const unsigned int (&P)[4], const unsigned int (&C)[4], int lane,
                                                        float2 (&v)[8], float scale)
{
    const unsigned int c0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)C[0]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned int pr = P[r];
        if (r == 3) pr = lane == 63 ? ((P[3] & 0xffffu) | (c0 << 16)) : P[3];
        const float2 a = unpack_i16x2(pr);
        v[r] = make_float2(scale * a.x, scale * a.y);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        unsigned int nx = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)C[r], 0x130, 0xf, 0xf, true);   // wave_shl:1, lane 63 <- 0
        if (r < 3) {
            const unsigned int first = (unsigned int)__builtin_amdgcn_readfirstlane((int)C[r < 3 ? r + 1 : r]);
            nx = lane == 63 ? first : nx;
        }
        const float2 a = unpack_i16x2(__builtin_amdgcn_alignbit(nx, C[r], 16));   // (C >> 16) | (next << 16)
        v[r + 4] = make_float2(scale * a.x, scale * a.y);
    }
}

#ifndef JDSP_MVDR_DIRECT
#define JDSP_MVDR_DIRECT 1      // 1: frames from dwords in registers, both forward transforms staggered in one wave, dword output stores
#endif
__global__ __launch_bounds__(64) void mvdr_pairs_kernel(const short *__restrict__ left, const short *__restrict__ right,
                                                        long n_blocks, long calls_before,
                                                        const MvdrState *__restrict__ st_in, MvdrState *st_out,
                                                        const int *__restrict__ ver_base,
                                                        const unsigned long long *__restrict__ snap_mask,
                                                        const double *__restrict__ rver, const double2 *__restrict__ steer,
                                                        const float2 *__restrict__ table, short *__restrict__ out,
                                                        float *__restrict__ precast, DenoiseShard sh,
                                                        const DenoisePlan *__restrict__ plan, const float4 *__restrict__ wtab)
{
#if JDSP_MVDR_DIRECT
    __shared__ __attribute__((aligned(16))) float2 lds2[2][kWaveLdsComplex];
    float2 *lds = lds2[0];
#else
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    __shared__ __attribute__((aligned(16))) unsigned int stage32[528];
#endif
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;
    const long j = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (j >= n_blocks) return;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    PairTwiddles pw;
    load_pair_twiddles(pw, table, lane);
    const bool have_prev = calls_before + j > 0;
    const long jp = have_prev ? j - 1 : -2;          // keep buffer of a stream's very first block: zeros (:130-131)

    float2 llo[5], lhi[5], rlo[5], rhi[5], v[8], zr[5];
#if JDSP_MVDR_DIRECT
    {
        unsigned int lp[4], lc[4], rp[4], rc[4];
        mvdr_load_block32(left, n_blocks, st_in->prev_l, jp, lane, lp);
        mvdr_load_block32(left, n_blocks, st_in->prev_l, j, lane, lc);
        mvdr_load_block32(right, n_blocks, st_in->prev_r, jp, lane, rp);
        mvdr_load_block32(right, n_blocks, st_in->prev_r, j, lane, rc);
        float2 vr[8], zq[5];
        mvdr_frame_pairs_direct(lp, lc, lane, v, 0.5f);
        mvdr_frame_pairs_direct(rp, rc, lane, vr, 0.5f);
        wave_fft512_x2_staggered<false>(v, vr, lds2[0], lds2[1], lane, tw);
        // both channels' mirror operands in one round trip
#pragma unroll
        for (int d = 3; d < 8; d++) { xchg_st(lds2[0], lane + 64 * d, v[d]); xchg_st(lds2[1], lane + 64 * d, vr[d]); }
        if (lane == 0) { lds2[0][512] = v[0]; lds2[1][512] = vr[0]; }
        wave_lds_fence();
#pragma unroll
        for (int d = 0; d < 5; d++) { zr[d] = xchg_ld(lds2[0], 512 - lane - 64 * d); zq[d] = xchg_ld(lds2[1], 512 - lane - 64 * d); }
        wave_lds_fence();
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
            const float2 t = cmul(pw.w[d], o);
            llo[d] = cadd(e, t);
            lhi[d] = csub(e, t);
            const float2 e2 = cadd_conj(vr[d], zq[d]), o2 = csub_conj_mj(vr[d], zq[d]);
            const float2 t2 = cmul(pw.w[d], o2);
            rlo[d] = cadd(e2, t2);
            rhi[d] = csub(e2, t2);
        }
    }
#else
#if JDSP_MVDR_EARLY_LOADS
    // the right channel's blocks are requested before the left channel's transform (whose LDS fences would hold the loads back)
    const u32x4 r_prev = mvdr_load_block(right, n_blocks, st_in->prev_r, jp, lane);
    const u32x4 r_cur = mvdr_load_block(right, n_blocks, st_in->prev_r, j, lane);
#endif
    mvdr_frame_pairs(stage32, lane, mvdr_load_block(left, n_blocks, st_in->prev_l, jp, lane),
                     mvdr_load_block(left, n_blocks, st_in->prev_l, j, lane), v, 0.5f);
    wave_fft512<false>(v, lds, lane, tw);
    wave_lds_fence();
    pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
    for (int d = 0; d < 5; d++) {
        const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
        const float2 t = cmul(pw.w[d], o);
        llo[d] = cadd(e, t);
        lhi[d] = csub(e, t);
    }
#if JDSP_MVDR_ABLATE == 2                                   // timing only: no second forward transform
#pragma unroll
    for (int d = 0; d < 5; d++) { rlo[d] = lhi[d]; rhi[d] = llo[d]; }
#else
#if JDSP_MVDR_EARLY_LOADS
    mvdr_frame_pairs(stage32, lane, r_prev, r_cur, v, 0.5f);
#else
    mvdr_frame_pairs(stage32, lane, mvdr_load_block(right, n_blocks, st_in->prev_r, jp, lane),
                     mvdr_load_block(right, n_blocks, st_in->prev_r, j, lane), v, 0.5f);
#endif
    wave_fft512<false>(v, lds, lane, tw);
    wave_lds_fence();
    pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
    for (int d = 0; d < 5; d++) {
        const float2 e = cadd_conj(v[d], zr[d]), o = csub_conj_mj(v[d], zr[d]);
        const float2 t = cmul(pw.w[d], o);
        rlo[d] = cadd(e, t);
        rhi[d] = csub(e, t);
    }
#endif
#endif   // JDSP_MVDR_DIRECT
    // the matrix in effect at this block (looking it up ahead of the transforms instead measured the same)
#if JDSP_MVDR_ABLATE == 4                                   // timing only: no version lookup
    int ver = 0;
    const bool tabled = true;
#else
    int ver = version_of(ver_base, snap_mask, j + sh.ver_block_off);
    if (sh.ver_row_off) ver -= *sh.ver_row_off;
    if (ver < 0) ver = 0;
    const bool tabled = wtab && mvdr_tabled(plan->n_events, n_blocks);     // wave-uniform
#endif
    const float4 *wrow = wtab + (size_t)ver * 1024;
    MvdrInv iv = {0.0, 0.0, 0.0, 0.0};
    if (!tabled) iv = mvdr_inverse(rver + (size_t)ver * 4);
    float2 y[8], ret[4];
#pragma unroll
    for (int d = 0; d < 5; d++) {
        const int m = lane + 64 * d;
        const int b1 = (1024 - m) & 1023, b3 = 512 - m;
        float4 w0, w1, w2, w3;
#if JDSP_MVDR_ABLATE == 1                                   // timing only: no table loads
        if (tabled) { w0 = w1 = w2 = w3 = make_float4(0.5f, 0.25f, 0.5f, -0.25f); }
#else
        if (tabled) { w0 = wrow[m]; w1 = wrow[b1]; w2 = wrow[m + 512]; w3 = wrow[b3]; }
#endif
        else {
            w0 = mvdr_bin_weights(iv, steer[m]); w1 = mvdr_bin_weights(iv, steer[b1]);
            w2 = mvdr_bin_weights(iv, steer[m + 512]); w3 = mvdr_bin_weights(iv, steer[b3]);
        }
        const float2 cl = make_float2(llo[d].x, -llo[d].y), cr = make_float2(rlo[d].x, -rlo[d].y);   // X[1024 - m] = conj X[m]
        const float2 ch = make_float2(lhi[d].x, -lhi[d].y), cs = make_float2(rhi[d].x, -rhi[d].y);   // X[512 - m] = conj X[m + 512]
        const float2 a0 = mvdr_apply_weights(w0, llo[d], rlo[d]), a1 = mvdr_apply_weights(w1, cl, cr);
        const float2 g0 = mvdr_apply_weights(w2, lhi[d], rhi[d]), g1 = mvdr_apply_weights(w3, ch, cs);
        // The reference keeps only the real part of the inverse transform (:193) = the inverse transform of the
        // Hermitian part of the weighted spectrum.
        const float2 ylo = make_float2(0.5f * (a0.x + a1.x), 0.5f * (a0.y - a1.y));
        const float2 yhi = make_float2(0.5f * (g0.x + g1.x), 0.5f * (g0.y - g1.y));
        if (d < 4) presplit_inv_pair(ylo, yhi, pw.w[d], y[d], ret[d]);
        else y[d] = presplit_inv_reg(ylo, yhi, pw.w[d]);
    }
    pair_return_lds(ret, lds, lane, y);
    wave_fft512<true>(y, lds, lane, tw);

    const long first_emit = sh.emit_from;                           // :201-204: the first call's block is dropped
#if JDSP_MVDR_DIRECT
    if (j >= first_emit && j < sh.emit_to) {
        // :193 rgsOutputBuffer[i] = y[i + 511] / 1024: output sample i = n - 511.  Lane l holds y[n], y[n + 1] for n = 2 l + 128 dd;
        // the dword-aligned output pairs are (n odd, n + 1): this lane's .y and the next lane's .x (lane 63: lane 0's next
        // register) -- five dword stores per block instead of sixteen 2-byte ones.
        unsigned int *o32 = reinterpret_cast<unsigned int *>(out + (j - first_emit) * 512);
        float *pc = precast ? precast + (j - first_emit) * 512 : nullptr;
#pragma unroll
        for (int dd = 3; dd < 8; dd++) {
            float nx = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y[dd].x), 0x130, 0xf, 0xf, true));   // wave_shl:1
            if (dd < 7) {
                const float first = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(y[dd < 7 ? dd + 1 : dd].x)));
                nx = lane == 63 ? first : nx;
            }
            const float s0 = y[dd].y * (1.0f / 1024.0f), s1 = nx * (1.0f / 1024.0f);
            const int i0 = 2 * lane + 128 * dd - 510;                // even; the pair (i0, i0 + 1)
            if (i0 >= 0 && i0 < 512) {
                o32[i0 >> 1] = cast_i16x2_bits(s0, s1);
                if (pc) { pc[i0] = s0; pc[i0 + 1] = s1; }
            }
        }
    }
#else
    if (j >= first_emit && j < sh.emit_to) {
        short *o = out + (j - first_emit) * 512;
        float *pc = precast ? precast + (j - first_emit) * 512 : nullptr;
#pragma unroll
        for (int dd = 0; dd < 8; dd++) {
            const int i0 = 2 * lane + 128 * dd - 511;                // :193 rgsOutputBuffer[i] = y[i + 511] / 1024
            const float s0 = y[dd].x * (1.0f / 1024.0f), s1 = y[dd].y * (1.0f / 1024.0f);
            if (i0 >= 0 && i0 < 512) { o[i0] = (short)cast_i16_bits(s0); if (pc) pc[i0] = s0; }
            if (i0 + 1 >= 0 && i0 + 1 < 512) { o[i0 + 1] = (short)cast_i16_bits(s1); if (pc) pc[i0 + 1] = s1; }
        }
    }
#endif
    if (j == n_blocks - 1) {
        reinterpret_cast<u32x4 *>(st_out->prev_l)[lane] = reinterpret_cast<const u32x4 *>(left + j * 512)[lane];
        reinterpret_cast<u32x4 *>(st_out->prev_r)[lane] = reinterpret_cast<const u32x4 *>(right + j * 512)[lane];
    }
}

static void launch_mvdr_prefix(hipStream_t s, const double *delta, const DenoisePlan *plan, const int *range,
                               const double *r_in, const double *sums_all, int rank, MvdrState *st_out, double *rver,
                               double *total, double *tile_sums)
{
    hipLaunchKernelGGL(mvdr_prefix_kernel, dim3(kPrefixGrid), dim3(1024), 0, s, delta, plan, range, r_in, sums_all, rank, st_out,
                       rver, total, tile_sums);
    hipLaunchKernelGGL(mvdr_prefix_bases_kernel, dim3(kPrefixGrid), dim3(1024), 0, s, plan, range, r_in, sums_all, rank, st_out,
                       rver, total, (const double *)tile_sums);
}

int launch_mvdr(hipStream_t s, const short *left, const short *right, long n_blocks, long calls_before,
                const MvdrState *st_in, MvdrState *st_out, const int *events, const DenoisePlan *plan,
                const int *ver_base, const unsigned long long *snap_mask, double *delta, double *rver,
                const double2 *steer, const float2 *table, short *out, float *precast, float4 *wtab,
                double *tile_sums)
{
    if (n_blocks <= 0) return 0;
    const long g1 = n_blocks < 2048 ? n_blocks : 2048;
    hipLaunchKernelGGL(mvdr_corr_kernel, dim3((unsigned)g1), dim3(64), 0, s, left, right, n_blocks, st_in, events, plan,
                       delta, (const int *)nullptr, 0L);
    launch_mvdr_prefix(s, delta, plan, nullptr, st_in->corr, nullptr, 0, st_out, rver, nullptr, tile_sums);
    DenoiseShard sh;
    sh.ver_block_off = 0;
    sh.ver_row_off = nullptr;
    sh.emit_from = calls_before >= 1 ? 0 : 1;
    sh.emit_to = n_blocks;
    const long grid = (n_blocks + 7) / 8 * 8;
#if JDSP_MVDR_PAIRS
    if (wtab)                                            // wtab: min(blocks + 1, kMvdrTableVersions) x 1024 float4, or NULL (no table)
        hipLaunchKernelGGL(mvdr_weights_kernel, dim3(4096), dim3(256), 0, s, plan, rver, steer, wtab, n_blocks);     // 1,024 versions per pass
    hipLaunchKernelGGL(mvdr_pairs_kernel, dim3((unsigned)grid), dim3(64), 0, s, left, right, n_blocks, calls_before, st_in,
                       st_out, ver_base, snap_mask, rver, steer, table, out, precast, sh, plan, (const float4 *)wtab);
#else
    (void)wtab;
    hipLaunchKernelGGL(mvdr_kernel, dim3((unsigned)grid), dim3(64), 0, s, left, right, n_blocks, calls_before, st_in,
                       st_out, ver_base, snap_mask, rver, steer, table, out, precast, sh);
#endif
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// EstimateSpatialCorrMtx on its own: the deltas of every listed event and their sum (no state touched).
int launch_mvdr_corr_total(hipStream_t s, const short *left, const short *right, long n_blocks, const MvdrState *st_in,
                           const int *events, const DenoisePlan *plan, const float2 *table, double *delta, double *total,
                           double *tile_sums)
{
    if (n_blocks <= 0) return 0;
    const long g1 = n_blocks < 2048 ? n_blocks : 2048;
    hipLaunchKernelGGL(mvdr_corr_kernel, dim3((unsigned)g1), dim3(64), 0, s, left, right, n_blocks, st_in, events, plan,
                       delta, (const int *)nullptr, 0L);
    launch_mvdr_prefix(s, delta, plan, nullptr, nullptr, nullptr, 0, nullptr, nullptr, total, tile_sums);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ProcessMVDR on its own: rver / ver_base / snap_mask are the caller's (jdsp_mvdr_apply: one matrix for every block).
int launch_mvdr_apply(hipStream_t s, const short *left, const short *right, long n_blocks, long calls_before,
                      const MvdrState *st_in, MvdrState *st_out, const int *ver_base, const unsigned long long *snap_mask,
                      const double *rver, const double2 *steer, const float2 *table, short *out, float *precast)
{
    if (n_blocks <= 0) return 0;
    DenoiseShard sh;
    sh.ver_block_off = 0;
    sh.ver_row_off = nullptr;
    sh.emit_from = calls_before >= 1 ? 0 : 1;
    sh.emit_to = n_blocks;
    const long grid = (n_blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(mvdr_kernel, dim3((unsigned)grid), dim3(64), 0, s, left, right, n_blocks, calls_before, st_in,
                       st_out, ver_base, snap_mask, rver, steer, table, out, precast, sh);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- sharded run (multi-GPU): see include/jdsp.h "sharded MVDR" --------------------------------
__global__ void mvdr_event_range_kernel(const int *__restrict__ events, const DenoisePlan *__restrict__ plan,
                                        const int *__restrict__ ver_base,
                                        const unsigned long long *__restrict__ snap_mask, long b0, long b1,
                                        int *__restrict__ range)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = plan->n_events;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (events[mid] < b0) lo = mid + 1; else hi = mid; }
    range[0] = lo;
    hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (events[mid] < b1) lo = mid + 1; else hi = mid; }
    range[1] = lo;
    range[2] = b0 > 0 ? version_of(ver_base, snap_mask, b0 - 1) : 0;
}

int launch_mvdr_shard_summary(hipStream_t s, const short *left_ext, const short *right_ext, long n_ext, long ext0,
                              long b0, long b1, const MvdrState *zero_state, const int *events,
                              const DenoisePlan *plan, const int *ver_base, const unsigned long long *snap_mask,
                              const float2 *table, int *range, double *delta, double *total, double *tile_sums)
{
    hipLaunchKernelGGL(mvdr_event_range_kernel, dim3(1), dim3(64), 0, s, events, plan, ver_base, snap_mask, b0, b1, range);
    const long own = b1 - b0;
    const long g1 = own < 2048 ? (own > 0 ? own : 1) : 2048;
    hipLaunchKernelGGL(mvdr_corr_kernel, dim3((unsigned)g1), dim3(64), 0, s, left_ext, right_ext, n_ext, zero_state, events,
                       plan, delta, (const int *)range, ext0);
    launch_mvdr_prefix(s, delta, plan, range, nullptr, nullptr, 0, nullptr, nullptr, total, tile_sums);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_mvdr_shard_finish(hipStream_t s, const short *left_ext, const short *right_ext, long n_ext, long ext0,
                             long b0, long b1, const MvdrState *zero_state, MvdrState *scratch_state,
                             const DenoisePlan *plan, const int *ver_base, const unsigned long long *snap_mask,
                             const int *range, const double *delta, const double *sums_all, int rank, double *rver,
                             const double2 *steer, const float2 *table, short *out, float *precast, double *tile_sums)
{
    launch_mvdr_prefix(s, delta, plan, range, nullptr, sums_all, rank, nullptr, rver, nullptr, tile_sums);
    DenoiseShard sh;
    sh.ver_block_off = ext0;
    sh.ver_row_off = range + 2;
    const long lo = b0 > 1 ? b0 : 1;                                  // :201-204: global block 0 emits nothing
    sh.emit_from = lo - ext0;
    sh.emit_to = b1 - ext0;
    if (sh.emit_to <= sh.emit_from) return 0;
    const long grid = (n_ext + 7) / 8 * 8;
    hipLaunchKernelGGL(mvdr_kernel, dim3((unsigned)grid), dim3(64), 0, s, left_ext, right_ext, n_ext, ext0, zero_state,
                       scratch_state, ver_base, snap_mask, rver, steer, table, out, precast, sh);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
