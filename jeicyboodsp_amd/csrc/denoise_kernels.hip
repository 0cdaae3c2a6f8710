// denoise_kernels.hip -- SpectralSubtraction_final.cpp / WienerFilter_final.cpp on gfx950.
//
// The reference runs one 512-sample block at a time through a chain of functions with
// static state (SS:92-113).  Here a whole batch of blocks goes through five launches:
//
//   vad_kernel          A11  VoiceActivityDetection per block (SS:121-156): integer/FP64, bit-exact
//   plan_kernel         A13  main()'s run-length counter (SS:98-109) as prefix scans over 64-block
//                            bit masks: the list of blocks that call EstimateNoiseSpectrum (run
//                            length >= 2) and, per block, which latched estimate (run length == 10,
//                            SS:189-193) is current
//   noise_accum_kernel  A12  |FFT(window * [previous block, block])| of the listed blocks (SS:168-183) folded into one
//                            affine map of the running average (SS:182-187) per chunk of events, in registers
//   noise_combine_kernel A12 the chunk maps composed: the average entering every chunk, the latched rows, the state
//   denoise_run_kernel  A8/A9/A10  window -> FFT -> gain -> IFFT -> overlap-add -> (short), fused; one wavefront
//                            walks a run of consecutive output blocks and recomputes one halo frame
// (512-point frames: vad_*<4>, noise_accum512_kernel, denoise512_run_kernel; sharded runs use the same kernels on a
// rank's range of events.)
//
// All of the reference's statics live in DenoiseState (ping-ponged between calls so a launch
// never reads state it is also writing), which makes batches of any size -- down to the
// reference's one block per call -- produce the same stream.
#include "frame_io.h"
#include "jdsp_internal.h"

namespace jdsp {

// ---------------------------------------------------------------------------------------
// A11.  frame = [zeros(512), block] because the keep buffer is never updated (SS:154 is
// unreachable); s = (short)(x * w) truncates; E = sum s^2 / 1024; Z counts s[i]*x[i+1] < 0
// (the next sample is not windowed yet, SS:139).  E > 700 <=> sum s^2 > 716800 exactly.
#ifndef JDSP_VAD_BLOCKS_PER_WAVE
#define JDSP_VAD_BLOCKS_PER_WAVE 4
#endif
// blocks per wave: the FP64 window slice (64 B per lane) is loaded once per wave.  4: 66 registers, seven waves per SIMD,
// 13.0 us per 65,536 blocks; 8: 86 registers, 15.0 us; 2: 13.4; 1: 18.9; 16: 278 registers, 28.8 (profiles/r02_denoise_ab.txt)
constexpr int kVadBlocksPerWave = JDSP_VAD_BLOCKS_PER_WAVE;

// SPL = samples per lane = BLOCK_LEN / 64: 8 for the reference's 512-sample blocks (SS:54), 4 for 256-sample blocks
// (FFT_PROCESSING_SIZE 512, BASELINE config 3 as worded).  dEnergy = sum / (2 BLOCK_LEN) > 700 (SS:143,147,48).
template <int SPL>
__global__ __launch_bounds__(64) void vad_kernel(const short *__restrict__ pcm, long n_blocks,
                                                 const double *__restrict__ w_hi, int use_zcr,
                                                 unsigned char *__restrict__ flags,
                                                 long long *__restrict__ dbg_energy, int *__restrict__ dbg_zcr)
{
    static_assert(SPL == 8 || SPL == 4, "block of 512 or 256 samples");
    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * kVadBlocksPerWave;
    if (b0 >= n_blocks) return;
    double w[SPL];
#pragma unroll
    for (int k = 0; k < SPL; k++) w[k] = w_hi[SPL * lane + k];
    u32x4 img[kVadBlocksPerWave];
#pragma unroll
    for (int i = 0; i < kVadBlocksPerWave; i++) {
        const long b = b0 + i < n_blocks ? b0 + i : n_blocks - 1;
        if (SPL == 8) {
            img[i] = reinterpret_cast<const u32x4 *>(pcm + b * 512)[lane];
        } else {
            const uint2 h = reinterpret_cast<const uint2 *>(pcm + b * 256)[lane];
            img[i].x = h.x; img[i].y = h.y; img[i].z = 0u; img[i].w = 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < kVadBlocksPerWave; i++) {
        const long b = b0 + i;
        if (b >= n_blocks) break;
        int x[9];
        x[0] = (short)(img[i].x & 0xffffu); x[1] = (int)img[i].x >> 16;
        x[2] = (short)(img[i].y & 0xffffu); x[3] = (int)img[i].y >> 16;
        x[4] = (short)(img[i].z & 0xffffu); x[5] = (int)img[i].z >> 16;
        x[6] = (short)(img[i].w & 0xffffu); x[7] = (int)img[i].w >> 16;
        x[SPL] = __shfl_down(x[0], 1);                 // first sample of the next lane
        if (lane == 63) x[SPL] = 0;                    // the sample past the block does not exist: frame[N], defined 0
        // s and x are 16-bit quantities: 24-bit multiplies (full rate; the 32-bit v_mul_lo is quarter rate) are
        // exact, and four squares (< 2^30 each) fit an unsigned 32-bit partial sum
        long long e = 0;
        unsigned int q4 = 0;
        int z = 0;
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            const int s = (int)((double)x[k] * w[k]);  // (short)(short * double), in range
            q4 += (unsigned int)__mul24(s, s);
            if ((k & 3) == 3) { e += (long long)q4; q4 = 0; }
            z += (__mul24(s, x[k + 1]) < 0) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            e += __shfl_xor(e, o);
            z += __shfl_xor(z, o);
        }
        if (lane == 0) {
            // SS:147 with THRESHOLD_OF_ENERGY 700, _ZCR 200; BeamForming_MVDR_ver1.cpp:233 tests the energy only
            flags[b] = (e > 700LL * 128 * SPL || (use_zcr && z < 200)) ? 1 : 0;
            if (dbg_energy) dbg_energy[b] = e;
            if (dbg_zcr) dbg_zcr[b] = z;
        }
    }
}

// The same decision without the trace (dbg_energy / dbg_zcr are NULL in every batched chain): what costs in the
// kernel above is not the arithmetic but the two wave reductions -- __shfl_xor is a ds_bpermute, 8.9 issue slots
// each on this chip (tools/valu_rate.hip), eighteen of them per block.  Only the FLAG is needed here, so
//   * the zero-crossing count never exists per lane: every sample position's comparison goes to an SGPR pair and
//     the scalar unit counts its bits (s_bcnt1), beside the vector pipe;
//   * a lane's energy is clamped to 2^20 per four samples before the wave sum: the sum then fits 32 bits, and it
//     exceeds 700 * n exactly when the true sum does (a clamped lane alone is already above the threshold);
//   * the one 32-bit wave sum is five DPP adds (quad swaps, row mirrors, row broadcasts) and a v_readlane.
// Flags are bit-identical to the traced kernel's (tests/test_denoise_gpu.py checks both against the CPU restatement).
__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v)
{
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);    // row_half_mirror: sums of 8
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);    // row_mirror: sums of 16
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);    // row_bcast15 into rows 1 and 3
    v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);    // row_bcast31 into rows 2 and 3
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

template <int SPL>
__global__ __launch_bounds__(64) void vad_flags_kernel(const short *__restrict__ pcm, long n_blocks,
                                                       const double *__restrict__ w_hi, int use_zcr,
                                                       unsigned char *__restrict__ flags)
{
    static_assert(SPL == 8 || SPL == 4, "block of 512 or 256 samples");
    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * kVadBlocksPerWave;
    if (b0 >= n_blocks) return;
    double w[SPL];
#pragma unroll
    for (int k = 0; k < SPL; k++) w[k] = w_hi[SPL * lane + k];
    u32x4 img[kVadBlocksPerWave];
#pragma unroll
    for (int i = 0; i < kVadBlocksPerWave; i++) {
        const long b = b0 + i < n_blocks ? b0 + i : n_blocks - 1;
        if (SPL == 8) {
            img[i] = reinterpret_cast<const u32x4 *>(pcm + b * 512)[lane];
        } else {
            const uint2 h = reinterpret_cast<const uint2 *>(pcm + b * 256)[lane];
            img[i].x = h.x; img[i].y = h.y; img[i].z = 0u; img[i].w = 0u;
        }
    }
    unsigned int voice_bits = 0;                                   // wave-uniform
#pragma unroll
    for (int i = 0; i < kVadBlocksPerWave; i++) {
        int x[9];
        x[0] = (short)(img[i].x & 0xffffu); x[1] = (int)img[i].x >> 16;
        x[2] = (short)(img[i].y & 0xffffu); x[3] = (int)img[i].y >> 16;
        x[4] = (short)(img[i].z & 0xffffu); x[5] = (int)img[i].z >> 16;
        x[6] = (short)(img[i].w & 0xffffu); x[7] = (int)img[i].w >> 16;
        x[SPL] = __builtin_amdgcn_update_dpp(0, x[0], 0x130, 0xf, 0xf, true);    // wave_shl:1 -- lane i takes lane i+1; lane 63 gets 0
        unsigned int qa = 0, qb = 0;
        int z = 0;                                                  // wave-uniform: counted on the scalar unit
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            const int sv = (int)((double)x[k] * w[k]);              // (short)(short * double), in range
            if (k < 4) qa += (unsigned int)__mul24(sv, sv); else qb += (unsigned int)__mul24(sv, sv);
            z += __popcll(__ballot(__mul24(sv, x[k + 1]) < 0));
        }
        const unsigned int cap = 1u << 20;
        const unsigned int e = wave_sum_u32((qa < cap ? qa : cap) + (qb < cap ? qb : cap));
        if (e > 700u * 128u * SPL || (use_zcr && z < 200)) voice_bits |= 1u << i;
    }
    if (lane < kVadBlocksPerWave && b0 + lane < n_blocks) flags[b0 + lane] = (voice_bits >> lane) & 1u;
}

// ---------------------------------------------------------------------------------------
// A13 bookkeeping.  One workgroup of 1024 threads; each thread owns 64 consecutive blocks,
// whose voice flags it packs into one 64-bit mask, so every pass after the single load runs
// in registers.  Batches longer than 65,536 blocks are walked tile by tile with the carries
// (run length, event and latch counts) kept in LDS.
//
// Outputs: events[e] / ev_n[e] = block index and run length of the e-th EstimateNoiseSpectrum
// call (run length >= 2, SS:103-105); per 64-block slice s, ver_base[s] = number of latches
// (run length == 10, SS:189) before the slice and snap_mask[s] = which of its blocks latch, so
//     version(j) = ver_base[j >> 6] + popcount(snap_mask[j >> 6] & bits 0..(j & 63))
// (the estimate latched AT block j already applies to block j: main() estimates first).
struct RunSummary { int has_voice, trailing; };

__device__ __forceinline__ RunSummary combine(RunSummary a, RunSummary b)
{
    RunSummary r;
    r.has_voice = a.has_voice | b.has_voice;
    r.trailing = b.has_voice ? b.trailing : a.trailing + b.trailing;
    return r;
}

__global__ __launch_bounds__(1024) void plan_kernel(const unsigned char *__restrict__ flags, long n_blocks,
                                                    const int *__restrict__ run_len_in_p, DenoiseState *st_out,
                                                    int *__restrict__ ver_base,
                                                    unsigned long long *__restrict__ snap_mask,
                                                    int *__restrict__ events, int *__restrict__ ev_n,
                                                    DenoisePlan *plan, int latch_run, int *run_len_out)
{
    // latch_run > 0: the estimate changes when the run length equals it (SS:189: 10);
    // latch_run == 0: it changes at every event (BeamForming_MVDR_ver1.cpp:201 accumulates each time)
    __shared__ RunSummary sum[1024];
    __shared__ int cnt_ev[1024], cnt_sn[1024];
    __shared__ int carry_run, carry_ev, carry_sn;
    __shared__ RunSummary wave_sum[16];
    __shared__ int wave_ev[16], wave_sn[16];
    const int t = threadIdx.x;
    const int run_len_in = *run_len_in_p;
    if (t == 0) { carry_run = run_len_in; carry_ev = 0; carry_sn = 0; }
    __syncthreads();

    for (long tile0 = 0; tile0 < n_blocks; tile0 += 65536) {
        const long a = tile0 + (long)t * 64;
        long left = n_blocks - a;
        const int cnt = left >= 64 ? 64 : (left > 0 ? (int)left : 0);
        unsigned long long V = 0;                               // bit i: block a+i is voice
        if (cnt == 64 && (((uintptr_t)(flags + a)) & 15u) == 0) {
            const uint4 *p = reinterpret_cast<const uint4 *>(flags + a);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 w = p[q];
                const unsigned int d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // the four 0/1 bytes of d[k] -> 4 adjacent bits
                    const unsigned int nib = ((d[k] & 0x01010101u) * 0x01020408u) >> 24;
                    V |= (unsigned long long)(nib & 15u) << (16 * q + 4 * k);
                }
            }
        } else {
            for (int i = 0; i < cnt; i++) V |= (unsigned long long)(flags[a + i] != 0) << i;
        }
        RunSummary mine;
        mine.has_voice = V != 0;
        mine.trailing = V ? (cnt - 1) - (63 - __clzll(V)) : cnt;
        // inclusive scan of the run summaries over the 1024 threads: shuffles inside each wave, one pass over the
        // 16 wave totals, two barriers (a Hillis-Steele scan through LDS took twenty)
        {
            const int lane = t & 63, wv = t >> 6;
            RunSummary v = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                RunSummary u;
                u.has_voice = __shfl_up(v.has_voice, o, 64);
                u.trailing = __shfl_up(v.trailing, o, 64);
                if (lane >= o) v = combine(u, v);
            }
            if (lane == 63) wave_sum[wv] = v;
            __syncthreads();
            if (wv > 0) {
                RunSummary pre = wave_sum[0];
                for (int q = 1; q < wv; q++) pre = combine(pre, wave_sum[q]);
                v = combine(pre, v);
            }
            sum[t] = v;
            __syncthreads();
        }
        const int run_in = carry_run;
        int r = run_in;                                          // run length entering my slice
        if (t > 0) r = sum[t - 1].has_voice ? sum[t - 1].trailing : run_in + sum[t - 1].trailing;
        const int r_start = r;
        // event / latch masks of my slice, bit-parallel.  N = noise bits.  An event (run length >= 2) is a noise
        // bit whose predecessor is noise too -- bit 0's predecessor is the incoming run.  A latch (run length ==
        // latch_run) is the last of latch_run noise bits that follow a voice bit inside the slice, or bit
        // latch_run - r_start - 1 when everything up to it is noise (the run came in from before the slice).
        const unsigned long long full = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
        const unsigned long long N = ~V & full;
        const unsigned long long E = N & ((N << 1) | (r_start >= 1 ? 1ull : 0ull));
        unsigned long long S = E;                                // latch_run == 0: every event changes the estimate
        if (latch_run > 0) {
            unsigned long long A = N;
            for (int k = 1; k < latch_run && k < 64; k++) A &= N << k;
            S = latch_run < 64 ? (A & (V << latch_run)) : 0ull;
            const int ic = latch_run - r_start - 1;
            if (ic >= 0 && ic < cnt) {
                const unsigned long long upto = (2ull << ic) - 1ull;        // bits 0..ic (all ones for ic = 63)
                if ((N & upto) == upto) S |= 1ull << ic;
            }
        }
        const int ne = __popcll(E), ns = __popcll(S);
        {
            const int lane = t & 63, wv = t >> 6;
            int ve = ne, vs = ns;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int ue = __shfl_up(ve, o, 64), us = __shfl_up(vs, o, 64);
                if (lane >= o) { ve += ue; vs += us; }
            }
            if (lane == 63) { wave_ev[wv] = ve; wave_sn[wv] = vs; }
            __syncthreads();
            for (int q = 0; q < wv; q++) { ve += wave_ev[q]; vs += wave_sn[q]; }
            cnt_ev[t] = ve;
            cnt_sn[t] = vs;
            __syncthreads();
        }
        if (cnt > 0) {
            ver_base[a >> 6] = carry_sn + cnt_sn[t] - ns;
            snap_mask[a >> 6] = S;
        }
        {
            // The event list, written by the wave together: for each of its 64 slices in turn, lane i takes bit i of the
            // slice's event mask.  (Every thread walking its own slice's bits wrote up to 64 scattered 4-byte stores
            // per lane and instruction: 68 us of this kernel on an all-quiet 65,536-block call, now a handful of
            // contiguous stores per slice.)
            const int eo_mine = carry_ev + cnt_ev[t] - ne;
            const int lane = t & 63;
            const long a_wave = tile0 + (long)(t - lane) * 64;
            unsigned long long todo = __ballot(E != 0ull);                  // slices of this wave that hold events
            for (; todo; todo &= todo - 1ull) {
                const int sl = __ffsll((long long)todo) - 1;                   // wave-uniform
                const unsigned int e_lo = __builtin_amdgcn_readlane((int)(unsigned int)E, sl);
                const unsigned int e_hi = __builtin_amdgcn_readlane((int)(unsigned int)(E >> 32), sl);
                const unsigned long long Es = ((unsigned long long)e_hi << 32) | e_lo;
                const unsigned int v_lo = __builtin_amdgcn_readlane((int)(unsigned int)V, sl);
                const unsigned int v_hi = __builtin_amdgcn_readlane((int)(unsigned int)(V >> 32), sl);
                const unsigned long long Vs = ((unsigned long long)v_hi << 32) | v_lo;
                const int rs = __builtin_amdgcn_readlane(r_start, sl), eos = __builtin_amdgcn_readlane(eo_mine, sl);
                if ((Es >> lane) & 1ull) {
                    const unsigned long long lower = (1ull << lane) - 1ull;
                    const int idx = eos + __popcll(Es & lower);
                    const unsigned long long below = Vs & lower;            // voice bits before this one (this bit is noise)
                    events[idx] = (int)(a_wave + (long)sl * 64 + lane);
                    ev_n[idx] = below ? lane - (63 - __clzll((long long)below)) : rs + lane + 1;
                }
            }
        }
        __syncthreads();
        if (t == 0) {
            carry_run = sum[1023].has_voice ? sum[1023].trailing : run_in + sum[1023].trailing;
            carry_ev += cnt_ev[1023];
            carry_sn += cnt_sn[1023];
        }
        __syncthreads();
    }
    if (t == 0) {
        if (st_out) st_out->run_len = carry_run;
        if (run_len_out) *run_len_out = carry_run;
        plan->n_events = carry_ev;
        plan->n_snap = carry_sn;
    }
}

// ---------------------------------------------------------------------------------------
// The same block in the transform's own layout: q[r] = the sample pair (2 lane + 128 r, +1), four coalesced
// dword loads, nothing to re-lay out.
__device__ __forceinline__ void load_block_pairs(const short *__restrict__ pcm, long n_blocks,
                                                 const DenoiseState *__restrict__ st_in, long j, int lane,
                                                 unsigned int (&q)[4])
{
    const unsigned int *src = nullptr;
    if (j >= 0 && j < n_blocks) src = reinterpret_cast<const unsigned int *>(pcm + j * 512);
    else if (j == -1) src = reinterpret_cast<const unsigned int *>(st_in->prev);
#pragma unroll
    for (int r = 0; r < 4; r++) q[r] = src ? src[lane + 64 * r] : 0u;
}

// window -> forward transform -> natural-order image of Zh in LDS
__device__ __forceinline__ void forward_to_lds(const unsigned int *raw, const FrameTables &t, float2 *lds, int lane)
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 s = unpack_i16x2(raw[r]);
        v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
    }
    wave_fft512<false>(v, lds, lane, t.tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
}

// ---------------------------------------------------------------------------------------
// A12 (SS:165-193), the noise average: at every EstimateNoiseSpectrum event  A += |X|;  from run length 3 on  A /= 2;  at
// run length 10 the average is latched as the estimate.  Over the events of a call that is a chain of affine maps
// A <- h (A + |X|), h in {1, 1/2}: sequential by definition, and speech is 40-60 % pauses -- a 65,536-block call can hold
// tens of thousands of events.  (Walked one event after the other, one thread per bin, that chain took 0.26 us per event:
// 8.9 ms per 65,536 blocks of a stream that is half pauses, against 0.1 ms for everything else;
// tools/denoise_events_probe.py.)  So the chain is cut into chunks:
//   noise_accum_kernel    one wave per chunk of C consecutive events (C = 1 up to kNoiseChunks events, else
//                         ceil(events / grid)): transforms each event's frame, keeps the chunk's map
//                         A -> alpha A + beta[bin] in registers (alpha a power of two), and writes it once; at a latch it
//                         writes the map so far into the latched row.  No magnitude ever goes to memory.
//   noise_combine_kernel  16 bins per workgroup, 64 groups of chunks per bin: composes each group's maps, walks the 64
//                         group maps from the carried-in average, expands to the average entering every chunk, and
//                         completes the latched rows:  row = alpha_so_far * A_entering_chunk + beta_so_far.
// Multiplying by a power of two commutes with rounding, so with one event per chunk and one chunk per group -- up to 64
// events per call, e.g. any per-block or small-batch streaming use -- the result is bit-identical to the sequential
// walk; beyond that the additions associate differently (1e-7 relative, the same size as FP32 against the reference's FP64).
struct ChunkGeom { int per_chunk, n_chunks; };
__device__ __forceinline__ ChunkGeom chunk_geom(int n_events, int grid)
{
    ChunkGeom g;
    g.per_chunk = n_events > grid ? (n_events + grid - 1) / grid : 1;
    g.n_chunks = (n_events + g.per_chunk - 1) / g.per_chunk;
    return g;
}

#ifndef JDSP_ACCUM_PAIRS
#define JDSP_ACCUM_PAIRS 1
#endif
__global__ __launch_bounds__(64) void noise_accum_kernel(const short *__restrict__ pcm, long n_blocks,
                                                         const DenoiseState *__restrict__ st_in,
                                                         const int *__restrict__ events, const int *__restrict__ ev_n,
                                                         const DenoisePlan *__restrict__ plan,
                                                         const int *__restrict__ ver_base,
                                                         const unsigned long long *__restrict__ snap_mask,
                                                         const float2 *__restrict__ table, NoiseAccum acc,
                                                         float *__restrict__ noise_rows, int latch_run,
                                                         const int *__restrict__ range, long ext0)
{
    // range (device, or NULL: every event of the plan) = {first event, one past the last, latches before the first} of a
    // sharded run; ext0 = global index of the first block `pcm` holds (sharded runs; st_in is then NULL and never needed)
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const int e_base = range ? range[0] : 0, row_off = range ? range[2] : 0;
    const int n_events = range ? range[1] - range[0] : plan->n_events;
    const ChunkGeom cg = chunk_geom(n_events, (int)gridDim.x);
    const int chunk = blockIdx.x;
    if (chunk >= cg.n_chunks) return;
    const int e0 = chunk * cg.per_chunk;
    const int e1 = e0 + cg.per_chunk < n_events ? e0 + cg.per_chunk : n_events;
    FrameTables t;
    load_frame_tables(t, table, lane);
#if JDSP_ACCUM_PAIRS
    // Pair-owned bins (frame_io.h): |X| of the bins m, m + 512 of m = lane + 64 d, d < 5; the bins 1024 - m and 512 - m
    // are their mirrors (a real frame), so ten magnitudes per lane and event instead of sixteen.  A row is stored as the
    // ten own values plus the mirrors of m = 1..192 (the items of d = 3, 4 mirror each other: those bins are stored by
    // their owners only).
    PairTwiddles pw;
    load_pair_twiddles(pw, table, lane);
    constexpr int ND = 5;
#else
    SplitTwiddles sw;
    load_split_twiddles(sw, table, lane);
    constexpr int ND = 8;
#endif
    float blo[ND], bhi[ND], alpha = 1.0f;                        // beta[lane + 64 d], beta[lane + 64 d + 512]
    auto store_row = [&](float *row) {
#pragma unroll
        for (int d = 0; d < ND; d++) { row[lane + 64 * d] = blo[d]; row[lane + 64 * d + 512] = bhi[d]; }
#if JDSP_ACCUM_PAIRS
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int m = lane + 64 * d;
            if (m >= 1 && m <= 192) { row[1024 - m] = blo[d]; row[512 - m] = bhi[d]; }
        }
#endif
    };
#pragma unroll
    for (int d = 0; d < ND; d++) blo[d] = bhi[d] = 0.0f;
    for (int e = e0; e < e1; e++) {
        const long jg = events[e_base + e], j = jg - ext0;
        const int n = ev_n[e_base + e];
        unsigned int prev[4], cur[4];
        load_block_pairs(pcm, n_blocks, st_in, j - 1, lane, prev);   // rgssKeepBuffer (SS:165-170)
        load_block_pairs(pcm, n_blocks, st_in, j, lane, cur);
        float2 v[8], zr[ND], lo[ND], hi[ND];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float2 s0 = unpack_i16x2(prev[r]), s1 = unpack_i16x2(cur[r]);
            v[r] = make_float2(s0.x * t.win[r].x, s0.y * t.win[r].y);
            v[r + 4] = make_float2(s1.x * t.win[r + 4].x, s1.y * t.win[r + 4].y);
        }
        wave_fft512<false>(v, lds, lane, t.tw);
        wave_lds_fence();
#if JDSP_ACCUM_PAIRS
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const float2 ev = cadd_conj(v[d], zr[d]), od = csub_conj_mj(v[d], zr[d]);
            const float2 tw = cmul(pw.w[d], od);
            lo[d] = cadd(ev, tw);
            hi[d] = csub(ev, tw);
        }
#else
        mirror_fetch_lds(v, lds, lane, zr);
        split_fwd_reg(v, zr, sw, lo, hi);
#endif
        const float h = n >= 3 ? 0.5f : 1.0f;                    // SS:182-187
#pragma unroll
        for (int d = 0; d < ND; d++) {
            // hardware square root (1 ulp; sqrtf() expands to ~12 instructions of scaling and fix-up per value, a third
            // of this loop)
            blo[d] = (blo[d] + __builtin_amdgcn_sqrtf(lo[d].x * lo[d].x + lo[d].y * lo[d].y)) * h;
            bhi[d] = (bhi[d] + __builtin_amdgcn_sqrtf(hi[d].x * hi[d].x + hi[d].y * hi[d].y)) * h;
        }
        alpha *= h;
        if (n == latch_run) {                                    // SS:189-193
            const int row = version_of(ver_base, snap_mask, jg) - row_off;
            store_row(noise_rows + (size_t)row * 1024);
            if (lane == 0) { acc.lat_alpha[row] = alpha; acc.lat_chunk[row] = chunk; }
        }
    }
    store_row(acc.chunk_beta + (size_t)chunk * 1024);
    if (lane == 0) acc.chunk_alpha[chunk] = alpha;
}

// grid: n_bins / kCombineBins workgroups of 1024 threads = kCombineBins bins x kCombineGroups chunk groups;
// accum_grid = noise_accum's grid size
#ifndef JDSP_COMBINE_BINS
#define JDSP_COMBINE_BINS 16
#endif
constexpr int kCombineBins = JDSP_COMBINE_BINS, kCombineGroups = 1024 / JDSP_COMBINE_BINS;
// Sharded runs (range != NULL: {first event, one past the last, latches before, latches inside} of this rank): the
// average starts from a_in (NULL: zero), summary (or NULL) receives the rank's composed map {alpha, beta[1024]}, last
// (or NULL) {number of latches, the last latched row}; complete_rows = 0 leaves the latched rows' partial maps alone
// (the first of a rank's two passes: its a_in is not known yet).
__global__ __launch_bounds__(1024) void noise_combine_kernel(const DenoisePlan *__restrict__ plan,
                                                             const DenoiseState *__restrict__ st_in, DenoiseState *st_out,
                                                             NoiseAccum acc, float *__restrict__ noise_rows, int accum_grid,
                                                             const int *__restrict__ range, const float *__restrict__ a_in,
                                                             float *__restrict__ summary, float *__restrict__ last,
                                                             int complete_rows)
{
    __shared__ float ga[kCombineGroups][kCombineBins], gb[kCombineGroups][kCombineBins];
    const int bl = threadIdx.x % kCombineBins, g = threadIdx.x / kCombineBins;
    const int bin = blockIdx.x * kCombineBins + bl;
    const int n_snap = range ? range[3] : plan->n_snap;
    const ChunkGeom cg = chunk_geom(range ? range[1] - range[0] : plan->n_events, accum_grid);
    const int per_group = (cg.n_chunks + kCombineGroups - 1) / kCombineGroups;
    const int c0 = g * per_group < cg.n_chunks ? g * per_group : cg.n_chunks;
    const int c1 = c0 + per_group < cg.n_chunks ? c0 + per_group : cg.n_chunks;
    // The two walks over the group's chunks are chains of one FMA per chunk -- but of one LOAD per chunk too, and a
    // load at a time is a memory latency per chunk (49 us for 4,096 chunks: profiles/r02_denoise_events.txt).  The
    // loads do not depend on the chain: sixteen chunks' values are requested together, then applied in order.
    constexpr int kBatch = 16;
    {
        float a = 1.0f, b = 0.0f;                                 // this group's chunks composed
        for (int cb = c0; cb < c1; cb += kBatch) {
            float al[kBatch], be[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const int c = cb + u < c1 ? cb + u : c1 - 1;
                al[u] = acc.chunk_alpha[c];
                be[u] = acc.chunk_beta[(size_t)c * 1024 + bin];
            }
#pragma unroll
            for (int u = 0; u < kBatch; u++)
                if (cb + u < c1) { b = al[u] * b + be[u]; a *= al[u]; }
        }
        ga[g][bl] = a;
        gb[g][bl] = b;
    }
    __syncthreads();
    if (summary && blockIdx.x == 0 && threadIdx.x == 0) {
        float p = 1.0f;
        for (int q = 0; q < kCombineGroups; q++) p *= ga[q][0];
        summary[0] = p;                                          // alpha = 2^-(number of halvings)
    }
    float A = a_in ? a_in[bin] : (st_in ? st_in->avg[bin] : 0.0f);
    for (int q = 0; q < g; q++) A = ga[q][bl] * A + gb[q][bl];   // the average entering this group
    for (int cb = c0; cb < c1; cb += kBatch) {
        float al[kBatch], be[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const int c = cb + u < c1 ? cb + u : c1 - 1;
            al[u] = acc.chunk_alpha[c];
            be[u] = acc.chunk_beta[(size_t)c * 1024 + bin];
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++)
            if (cb + u < c1) {
                acc.a_start[(size_t)(cb + u) * 1024 + bin] = A;
                A = al[u] * A + be[u];
            }
    }
    if (g == kCombineGroups - 1) {                               // groups past the last chunk are identities
        if (st_out) st_out->avg[bin] = A;
        if (summary) summary[1 + bin] = A;
    }
    if (g == 0 && st_in) noise_rows[bin] = st_in->noise[bin];    // row 0: the estimate carried in
    __syncthreads();                                             // a_start is read back by other threads of this workgroup
    if (complete_rows)
        for (int r = 1 + g; r <= n_snap; r += kCombineGroups) {
            float *row = noise_rows + (size_t)r * 1024 + bin;
            *row = acc.lat_alpha[r] * acc.a_start[(size_t)acc.lat_chunk[r] * 1024 + bin] + *row;
        }
    __syncthreads();
    if (g == 0) {
        if (st_out) st_out->noise[bin] = n_snap > 0 ? noise_rows[(size_t)n_snap * 1024 + bin] : st_in->noise[bin];
        if (last) {
            last[1 + bin] = n_snap > 0 ? noise_rows[(size_t)n_snap * 1024 + bin] : 0.0f;
            if (bin == 0) last[0] = (float)n_snap;
        }
    }
}

// ---------------------------------------------------------------------------------------
// A8 (SS:233-242): Y = (|X| - N) e^{j phase(X)}; no clamp at zero; X == 0 -> phase 0 -> (-N, 0).
// A9 (WF:196-213): Y = |X| (1 - min(1, N^2/|X|^2)) e^{j phase(X)}; 0/0 stays NaN as in the reference.
#ifndef JDSP_GAIN_SELECT
#define JDSP_GAIN_SELECT 1
#endif
template <int MODE>
__device__ __forceinline__ float2 apply_gain(float2 x, float n)
{
    // hardware reciprocal / reciprocal square root (1 ulp): the bar is 1e-5, not IEEE division
    const float p = x.x * x.x + x.y * x.y;
    if (MODE == 0) {
#if JDSP_GAIN_SELECT
        const float g = 1.0f - n * __frsqrt_rn(p);           // (|X| - N) / |X|; p == 0 gives inf / NaN, replaced below
        const bool zero = p == 0.0f;
        return make_float2(zero ? -n : x.x * g, zero ? 0.0f : x.y * g);
#else
        if (p == 0.0f) return make_float2(-n, 0.0f);
        const float g = 1.0f - n * __frsqrt_rn(p);           // (|X| - N) / |X|
        return make_float2(x.x * g, x.y * g);
#endif
    } else {
        float r = (n * n) * __builtin_amdgcn_rcpf(p);                    // 0 * inf = NaN keeps the reference's 0/0
        if (r >= 1.0f) r = 1.0f;
        const float g = 1.0f - r;
        return make_float2(x.x * g, x.y * g);
    }
}

template <int MODE, int J>
__device__ __forceinline__ void gain_presplit_j(const float2 *lds, float2 *zout, int lane, const float2 *wsp,
                                                const float *__restrict__ noise)
{
    const int m = 128 * J + 2 * lane;
    const float4 zz = *reinterpret_cast<const float4 *>(&lds[m]);
    float2 zr0, zr1;
    load_mirror_pair(lds, m, zr0, zr1);
    const float2 nlo = *reinterpret_cast<const float2 *>(noise + m);
    const float2 nhi = *reinterpret_cast<const float2 *>(noise + m + 512);
    float2 lo0, hi0, lo1, hi1;
    split_fwd<J>(make_float2(zz.x, zz.y), zr0, wsp[0], lo0, hi0);
    split_fwd<J>(make_float2(zz.z, zz.w), zr1, wsp[1], lo1, hi1);
    lo0 = apply_gain<MODE>(lo0, nlo.x); hi0 = apply_gain<MODE>(hi0, nhi.x);
    lo1 = apply_gain<MODE>(lo1, nlo.y); hi1 = apply_gain<MODE>(hi1, nhi.y);
    zout[2 * J] = presplit_inv<J>(lo0, hi0, wsp[0]);
    zout[2 * J + 1] = presplit_inv<J>(lo1, hi1, wsp[1]);
}

// One frame: samples -> y = IDFT(gain(DFT(window * frame))) / 1024, returned in the
// transform layout: y[d] = (y[2 lane + 128 d], y[2 lane + 128 d + 1]).
template <int MODE>
__device__ __forceinline__ void denoise_frame(const unsigned int *raw, const FrameTables &t, float2 *lds, int lane,
                                              const float *__restrict__ noise, float2 (&y)[8])
{
    forward_to_lds(raw, t, lds, lane);
    float2 z[8];
    gain_presplit_j<MODE, 0>(lds, z, lane, t.wsp, noise);
    gain_presplit_j<MODE, 1>(lds, z, lane, t.wsp, noise);
    gain_presplit_j<MODE, 2>(lds, z, lane, t.wsp, noise);
    gain_presplit_j<MODE, 3>(lds, z, lane, t.wsp, noise);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 4; j++)
        *reinterpret_cast<float4 *>(&lds[128 * j + 2 * lane]) = make_float4(z[2 * j].x, z[2 * j].y, z[2 * j + 1].x, z[2 * j + 1].y);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, t.tw);
    // the reference's 1/N after FFTW's unnormalised inverse (SS:248); a power of two, exact
#pragma unroll
    for (int d = 0; d < 8; d++) y[d] = make_float2(y[d].x * (1.0f / 1024.0f), y[d].y * (1.0f / 1024.0f));
    wave_lds_fence();
}

#ifndef JDSP_DENOISE_REGSPLIT
#define JDSP_DENOISE_REGSPLIT 1       // 1: split / pre-split in registers (mirror operands by ds_bpermute); 0: LDS images
#endif
#if JDSP_DENOISE_REGSPLIT
// The same frame with the bins kept in "lane + 64 d" registers from the forward transform to the inverse one:
// no natural-order image, no Z' image, three fences fewer (frame_io.h).  noise[m] is read at m = lane + 64 d.
template <int MODE>
__device__ __forceinline__ void denoise_frame_reg(const unsigned int *raw, const FrameTables &t, const SplitTwiddles &sw,
                                                  float2 *lds, int lane, const float *__restrict__ noise, float2 (&y)[8])
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 s = unpack_i16x2(raw[r]);
        v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
    }
    wave_fft512<false>(v, lds, lane, t.tw);
    float2 zr[8], lo[8], hi[8];
    mirror_fetch(v, lane, zr);
    split_fwd_reg(v, zr, sw, lo, hi);
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const int m = lane + 64 * d;
        lo[d] = apply_gain<MODE>(lo[d], noise[m]);
        hi[d] = apply_gain<MODE>(hi[d], noise[m + 512]);
        y[d] = presplit_inv_reg(lo[d], hi[d], sw.w[d]);
    }
    wave_lds_fence();                                            // the forward transform's last exchange reads are done
    wave_fft512<true>(y, lds, lane, t.tw);
    // the reference's 1/N after FFTW's unnormalised inverse (SS:248); a power of two, exact
#pragma unroll
    for (int d = 0; d < 8; d++) y[d] = make_float2(y[d].x * (1.0f / 1024.0f), y[d].y * (1.0f / 1024.0f));
    wave_lds_fence();
}
#endif

// The noise row of a frame held in registers: nlo[d] = N[lane + 64 d], nhi[d] = N[lane + 64 d + 512].  An estimate
// changes at most every tenth block (SS:189) and usually far less often, so a wave walking a run of blocks reloads these
// sixteen values only when the row pointer changes instead of fetching 4 KB per frame through L2.
struct NoiseRegs { float lo[8], hi[8]; };
__device__ __forceinline__ void load_noise_regs(NoiseRegs &n, const float *__restrict__ row, int lane)
{
#pragma unroll
    for (int d = 0; d < 8; d++) { n.lo[d] = row[lane + 64 * d]; n.hi[d] = row[lane + 64 * d + 512]; }
}

#ifndef JDSP_DENOISE_MIRROR_LDS
#define JDSP_DENOISE_MIRROR_LDS 1     // 1: mirror operands through a natural-order LDS image; 0: by ds_bpermute
#endif
template <int MODE>
__device__ __forceinline__ void denoise_frame_nreg(const unsigned int *raw, const FrameTables &t, const SplitTwiddles &sw,
                                                   float2 *lds, int lane, const NoiseRegs &n, float2 (&y)[8])
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 s = unpack_i16x2(raw[r]);
        v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
    }
    wave_fft512<false>(v, lds, lane, t.tw);
    float2 zr[8], lo[8], hi[8];
#if JDSP_DENOISE_MIRROR_LDS
    wave_lds_fence();                                            // the transform's last exchange reads are done
    mirror_fetch_lds(v, lds, lane, zr);
#else
    mirror_fetch(v, lane, zr);
#endif
    split_fwd_reg(v, zr, sw, lo, hi);
    // (Tried: skipping the two X == 0 selects per bin of spectral subtraction unless a wave-wide test finds a zero
    // bin -- a second copy of the gain loop behind a uniform branch.  134.6 us against 120.1 us per 65,536 blocks:
    // the doubled loop body costs far more than the 32 selects it saves.  profiles/r02_denoise_ab.txt.)
    {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            lo[d] = apply_gain<MODE>(lo[d], n.lo[d]);
            hi[d] = apply_gain<MODE>(hi[d], n.hi[d]);
            y[d] = presplit_inv_reg(lo[d], hi[d], sw.w[d]);
        }
    }
    wave_lds_fence();                                            // the forward transform's last exchange reads are done
    wave_fft512<true>(y, lds, lane, t.tw);
    // the reference's 1/N after FFTW's unnormalised inverse (SS:248); a power of two, exact
#pragma unroll
    for (int d = 0; d < 8; d++) y[d] = make_float2(y[d].x * (1.0f / 1024.0f), y[d].y * (1.0f / 1024.0f));
    wave_lds_fence();
}

// The frame with every mirror pair of bins owned by one lane (frame_io.h, PairTwiddles): five items per lane, the gain
// evaluated on the 640 bins m, m + 512 (m = lane + 64 d, d < 5) and the rest of the inverse transform's input taken
// from the Hermitian symmetry -- which holds because the gain is real and N[1024 - k] = N[k] up to the rounding of the
// estimate (|X[k]| and |X[1024 - k]| of a real frame, 1e-7 apart in FP32; the reference's own FFTW spectrum is
// symmetric to 1e-16).  The reference's 1/1024 after the inverse transform (SS:248, a power of two: exact wherever it
// is applied) is folded into the gain, so the noise values arrive pre-multiplied by it.
#ifndef JDSP_DENOISE_PAIRS
#define JDSP_DENOISE_PAIRS 1
#endif
#ifndef JDSP_DENOISE_ABLATE
#define JDSP_DENOISE_ABLATE 0       // timing-only ablations of denoise_frame_pairs (tools/build_variant.sh): wrong results
#endif
struct NoisePairRegs { float lo[5], hi[5]; };
template <int MODE>
__device__ __forceinline__ void load_noise_pair_regs(NoisePairRegs &n, const float *__restrict__ row, int lane)
{
    // spectral subtraction: N / 1024; Wiener: N as is (its gain is a ratio)
    const float sc = MODE == 0 ? (1.0f / 1024.0f) : 1.0f;
#pragma unroll
    for (int d = 0; d < 5; d++) { n.lo[d] = sc * row[lane + 64 * d]; n.hi[d] = sc * row[lane + 64 * d + 512]; }
}

// apply_gain<MODE>(x, n) / 2^LOG2N (1024 or 512) with n as load_noise_pair_regs left it
template <int MODE, int LOG2N = 10>
__device__ __forceinline__ float2 apply_gain_scaled(float2 x, float n)
{
    const float c = 1.0f / (float)(1 << LOG2N);
    const float p = x.x * x.x + x.y * x.y;
    if (MODE == 0) {
        const float g = c - n * __frsqrt_rn(p);              // (|X| - N) / (1024 |X|); p == 0 gives inf / NaN, replaced below
        const bool zero = p == 0.0f;
        return make_float2(zero ? -n : x.x * g, zero ? 0.0f : x.y * g);
    } else {
        // v_rcp_f32 (1 ulp; __frcp_rn() expands to the IEEE division sequence, a dozen instructions per bin)
        float r = (n * n) * __builtin_amdgcn_rcpf(p);                    // 0 * inf = NaN keeps the reference's 0/0
        if (r >= 1.0f) r = 1.0f;
        const float g = c - c * r;
        return make_float2(x.x * g, x.y * g);
    }
}

template <int MODE>
__device__ __forceinline__ void denoise_frame_pairs(const unsigned int *raw, const FrameTables &t, const PairTwiddles &pw,
                                                    float2 *lds, int lane, const NoisePairRegs &n, float2 (&y)[8])
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float2 s = unpack_i16x2(raw[r]);
        v[r] = make_float2(s.x * t.win[r].x, s.y * t.win[r].y);
    }
    wave_fft512<false>(v, lds, lane, t.tw);
    float2 zr[5], ret[4];
    wave_lds_fence();                                            // the transform's last exchange reads are done
#if JDSP_DENOISE_ABLATE & 2                                           /* timing-only: no mirror fetch / return */
#pragma unroll
    for (int d = 0; d < 5; d++) zr[d] = v[7 - d];
#else
    pair_fetch_lds(v, lds, lane, zr);
#endif
#pragma unroll
    for (int d = 0; d < 5; d++) {
        const float2 e = cadd_conj(v[d], zr[d]);
        const float2 o = csub_conj_mj(v[d], zr[d]);
        const float2 p = cmul(pw.w[d], o);
#if JDSP_DENOISE_ABLATE & 1                                           /* timing-only: no gain */
        const float2 lo = cadd(e, p), hi = csub(e, p);
#else
        const float2 lo = apply_gain_scaled<MODE>(cadd(e, p), n.lo[d]);      // Y[m] / 1024
        const float2 hi = apply_gain_scaled<MODE>(csub(e, p), n.hi[d]);      // Y[m + 512] / 1024
#endif
        if (d < 4) {
            presplit_inv_pair(lo, hi, pw.w[d], y[d], ret[d]);
        } else {
            y[d] = presplit_inv_reg(lo, hi, pw.w[d]);
        }
    }
#if JDSP_DENOISE_ABLATE & 2
#pragma unroll
    for (int d = 5; d < 8; d++) y[d] = ret[d - 5];
#else
    pair_return_lds(ret, lds, lane, y);
#endif
#if !(JDSP_DENOISE_ABLATE & 4)                                        /* 4, timing-only: no inverse transform */
    wave_fft512<true>(y, lds, lane, t.tw);
#endif
    wave_lds_fence();
}

#ifndef JDSP_DENOISE_MINWAVES
#define JDSP_DENOISE_MINWAVES 3
#endif

// Which noise row block j (local index) uses.  Single-GPU: the plan's version.  Sharded
// (DenoiseShard): the plan is indexed by GLOBAL block number and the local row table starts at
// the estimate in effect at the block before the shard (row 0).
__device__ __forceinline__ const float *noise_row(const float *__restrict__ rows, const int *__restrict__ ver_base,
                                                  const unsigned long long *__restrict__ snap_mask, long j,
                                                  const DenoiseShard &sh)
{
    int v = version_of(ver_base, snap_mask, j + sh.ver_block_off);
    if (sh.ver_row_off) v -= *sh.ver_row_off;
    if (v < 0) v = 0;
    return rows + (size_t)v * 1024;
}
// The same with *sh.ver_row_off read once by the caller (a persistent wave's loop: read inside it, that load is a
// vector load with an s_waitcnt vmcnt(0) behind it -- which also waits for the block prefetch issued just before)
__device__ __forceinline__ const float *noise_row_at(const float *__restrict__ rows, const int *__restrict__ ver_base,
                                                     const unsigned long long *__restrict__ snap_mask, long j,
                                                     long ver_block_off, int row_off)
{
    int v = version_of(ver_base, snap_mask, j + ver_block_off) - row_off;
    if (v < 0) v = 0;
    return rows + (size_t)v * 1024;
}

template <int MODE, int K>
__global__ __launch_bounds__(64, JDSP_DENOISE_MINWAVES) void denoise_kernel(
    const short *__restrict__ pcm, long n_blocks, long calls_before, const DenoiseState *__restrict__ st_in,
    DenoiseState *st_out, const int *__restrict__ ver_base, const unsigned long long *__restrict__ snap_mask,
    const float *__restrict__ noise_rows, const float2 *__restrict__ table, short *__restrict__ out,
    float *__restrict__ precast, DenoiseShard sh)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                // XCD-aware chunk order (speed only)
    const long j0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * K;
    if (j0 >= n_blocks) return;

    unsigned int half[K + 2][4];                              // blocks j0-2 .. j0+K-1, pairs 2 lane + 128 r
#pragma unroll
    for (int h = 0; h < K + 2; h++) load_block_pairs(pcm, n_blocks, st_in, j0 - 2 + h, lane, half[h]);
    FrameTables t;
    load_frame_tables(t, table, lane);
#if JDSP_DENOISE_REGSPLIT
    SplitTwiddles sw;
    load_split_twiddles(sw, table, lane);
#define JDSP_DN_FRAME(RAW, NOISE, Y) denoise_frame_reg<MODE>(RAW, t, sw, lds, lane, NOISE, Y)
#else
#define JDSP_DN_FRAME(RAW, NOISE, Y) denoise_frame<MODE>(RAW, t, lds, lane, NOISE, Y)
#endif

    const long first_emit = sh.emit_from;                     // SS:260-263: calls 1 and 2 emit nothing
    unsigned int raw[8];
    float2 tail[4], y[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { raw[r] = half[0][r]; raw[r + 4] = half[1][r]; }
    if (j0 == 0) {
        // rgsdOveraped[512..1023] carried over from the previous call
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = *reinterpret_cast<const float2 *>(&st_in->tail[2 * lane + 128 * d]);
    } else if (calls_before + j0 - 1 == 0) {
        // the very first call of a stream only stashes its block (SS:211-216): no output, empty overlap
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = make_float2(0.f, 0.f);
    } else {
        JDSP_DN_FRAME(raw, noise_row(noise_rows, ver_base, snap_mask, j0 - 1, sh), y);   // halo frame
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = y[d + 4];
    }

#pragma unroll
    for (int i = 0; i < K; i++) {
        const long j = j0 + i;
        if (j >= n_blocks) break;
#pragma unroll
        for (int r = 0; r < 4; r++) { raw[r] = raw[r + 4]; raw[r + 4] = half[i + 2][r]; }
        if (calls_before + j == 0) {
#pragma unroll
            for (int d = 0; d < 8; d++) y[d] = make_float2(0.f, 0.f);
        } else {
            JDSP_DN_FRAME(raw, noise_row(noise_rows, ver_base, snap_mask, j, sh), y);
        }
        float2 o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            o[d] = make_float2(tail[d].x + y[d].x, tail[d].y + y[d].y);      // SS:248 overlap-add
            tail[d] = y[d + 4];                                               // SS:255-256
        }
        if (j >= first_emit && j < sh.emit_to) {
            const long oi = j - first_emit;
            unsigned int *dst = reinterpret_cast<unsigned int *>(out + oi * 512) + lane;
#pragma unroll
            for (int d = 0; d < 4; d++)                      // four coalesced 256-byte stores, in the layout as is
                __builtin_nontemporal_store(cast_i16x2_bits(o[d].x, o[d].y), dst + 64 * d);
            if (precast) {
#pragma unroll
                for (int d = 0; d < 4; d++) *reinterpret_cast<float2 *>(precast + oi * 512 + 2 * lane + 128 * d) = o[d];
            }
        }
        if (j == n_blocks - 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) reinterpret_cast<unsigned int *>(st_out->prev)[lane + 64 * r] = half[i + 2][r];   // SS:257
#pragma unroll
            for (int d = 0; d < 4; d++) *reinterpret_cast<float2 *>(&st_out->tail[2 * lane + 128 * d]) = tail[d];
        }
    }
}

// The same work as denoise_kernel<MODE, K> with the run length chosen at launch instead of at compile time: one wave
// walks `run` consecutive blocks (one recomputed halo frame in front), and the launch picks `run` so that the whole
// batch is ONE round of resident waves (3 per SIMD at this register budget).  With K = 8 a 65,536-block batch is
// 8,192 waves = 2.67 rounds of the 3,072 resident ones, i.e. 3 rounds of 9 frames = 27 frame times per wave slot for
// 24 of work; with run = 22 it is 2,979 waves, one round, 23 frame times -- and the halo overhead falls from 1/8 to
// 1/22.  A wave needs the next block's samples only one iteration later, so they are loaded at the top of the
// iteration that precedes their use (4 dwords per lane) instead of all K + 2 blocks up front (40 VGPRs at K = 8).
#ifndef JDSP_DENOISE_RESIDENT
#define JDSP_DENOISE_RESIDENT (JDSP_DENOISE_PAIRS ? 4 : 3)      // waves per SIMD the run kernel's register budget allows
#endif
#ifndef JDSP_DENOISE_TAKE_EARLY
#define JDSP_DENOISE_TAKE_EARLY 1                                // 1: the prefetched block is taken before this block's stores, not after them
#endif
#ifndef JDSP_DENOISE_PRIO
#define JDSP_DENOISE_PRIO 1                                      // 1: the run kernels' waves rotate through the priority levels
#endif
template <int MODE>
__global__ __launch_bounds__(64, JDSP_DENOISE_RESIDENT) void denoise_run_kernel(
    const short *__restrict__ pcm, long n_blocks, long calls_before, const DenoiseState *__restrict__ st_in,
    DenoiseState *st_out, const int *__restrict__ ver_base, const unsigned long long *__restrict__ snap_mask,
    const float *__restrict__ noise_rows, const float2 *__restrict__ table, short *__restrict__ out,
    float *__restrict__ precast, DenoiseShard sh, int run)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                // XCD-aware run order (speed only)
    const long j0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * run;
    if (j0 >= n_blocks) return;
    const long j1 = j0 + run < n_blocks ? j0 + run : n_blocks;

    FrameTables t;
    load_frame_tables(t, table, lane);
#if JDSP_DENOISE_PAIRS
    PairTwiddles sw;
    load_pair_twiddles(sw, table, lane);
    NoisePairRegs nz;
#define JDSP_RUN_LOAD_NOISE(ROW) load_noise_pair_regs<MODE>(nz, ROW, lane)
#define JDSP_RUN_FRAME() denoise_frame_pairs<MODE>(raw, t, sw, lds, lane, nz, y)
#else
    SplitTwiddles sw;
    load_split_twiddles(sw, table, lane);
    NoiseRegs nz;
#define JDSP_RUN_LOAD_NOISE(ROW) load_noise_regs(nz, ROW, lane)
#define JDSP_RUN_FRAME() denoise_frame_nreg<MODE>(raw, t, sw, lds, lane, nz, y)
#endif

    unsigned int raw[8], nxt[4];
    float2 tail[4], y[8];
    const float *cur_row = nullptr;
    const int row_off = sh.ver_row_off ? *sh.ver_row_off : 0;          // once: see noise_row_at
    load_block_pairs(pcm, n_blocks, st_in, j0 - 2, lane, nxt);
#pragma unroll
    for (int r = 0; r < 4; r++) raw[r] = nxt[r];
    load_block_pairs(pcm, n_blocks, st_in, j0 - 1, lane, nxt);
#pragma unroll
    for (int r = 0; r < 4; r++) raw[r + 4] = nxt[r];
    load_block_pairs(pcm, n_blocks, st_in, j0, lane, nxt);
    if (j0 == 0) {
        // rgsdOveraped[512..1023] carried over from the previous call
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = *reinterpret_cast<const float2 *>(&st_in->tail[2 * lane + 128 * d]);
    } else if (calls_before + j0 - 1 == 0) {
        // the very first call of a stream only stashes its block (SS:211-216): no output, empty overlap
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = make_float2(0.f, 0.f);
    } else {
        cur_row = noise_row_at(noise_rows, ver_base, snap_mask, j0 - 1, sh.ver_block_off, row_off);
        JDSP_RUN_LOAD_NOISE(cur_row);
        JDSP_RUN_FRAME();                                          // halo frame
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = y[d + 4];
    }
    const long first_emit = sh.emit_from;                     // SS:260-263: calls 1 and 2 emit nothing
    // every wave walks the priority levels, one step per block (see fastconv1024_pairs_kernel: a SIMD serves equal-priority
    // waves by age, and equal shares then finish far apart)
    unsigned prio_step = blockIdx.x >> 10;
    // vmcnt counts loads and stores together, in issue order, and across the loop's back edge the compiler waits for
    // vmcnt(0): taking the prefetched block at the TOP of an iteration waited for the previous block's output stores to
    // complete, every block.  It is taken (tk <- nxt) just before this block's stores instead (see fastconv1024_pairs_kernel).
    unsigned int tk[4];
#pragma unroll
    for (int r = 0; r < 4; r++) tk[r] = nxt[r];
    for (long j = j0; j < j1; j++) {
#if JDSP_DENOISE_PRIO
        switch (prio_step++ & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        unsigned int cur[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { cur[r] = tk[r]; raw[r] = raw[r + 4]; raw[r + 4] = tk[r]; }
        if (j + 1 < j1) load_block_pairs(pcm, n_blocks, st_in, j + 1, lane, nxt);      // needed one iteration from now
        if (calls_before + j == 0) {
#pragma unroll
            for (int d = 0; d < 8; d++) y[d] = make_float2(0.f, 0.f);
        } else {
            const float *row = noise_row_at(noise_rows, ver_base, snap_mask, j, sh.ver_block_off, row_off);
            if (row != cur_row) {                                // wave-uniform: a new estimate was latched
                cur_row = row;
                JDSP_RUN_LOAD_NOISE(row);
            }
            JDSP_RUN_FRAME();
        }
        float2 o[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            o[d] = make_float2(tail[d].x + y[d].x, tail[d].y + y[d].y);      // SS:248 overlap-add
            tail[d] = y[d + 4];                                               // SS:255-256
        }
#if JDSP_DENOISE_TAKE_EARLY
#pragma unroll
        for (int r = 0; r < 4; r++) { tk[r] = nxt[r]; asm volatile("" : "+v"(tk[r])); }
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (j >= first_emit && j < sh.emit_to) {
            const long oi = j - first_emit;
            unsigned int *dst = reinterpret_cast<unsigned int *>(out + oi * 512) + lane;
#pragma unroll
            for (int d = 0; d < 4; d++)                      // four coalesced 256-byte stores, in the layout as is
                __builtin_nontemporal_store(cast_i16x2_bits(o[d].x, o[d].y), dst + 64 * d);
            if (precast) {
#pragma unroll
                for (int d = 0; d < 4; d++) *reinterpret_cast<float2 *>(precast + oi * 512 + 2 * lane + 128 * d) = o[d];
            }
        }
        if (j == n_blocks - 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) reinterpret_cast<unsigned int *>(st_out->prev)[lane + 64 * r] = cur[r];   // SS:257
#pragma unroll
            for (int d = 0; d < 4; d++) *reinterpret_cast<float2 *>(&st_out->tail[2 * lane + 128 * d]) = tail[d];
        }
#if !JDSP_DENOISE_TAKE_EARLY
#pragma unroll
        for (int r = 0; r < 4; r++) tk[r] = nxt[r];
#endif
    }
}

#undef JDSP_RUN_LOAD_NOISE
#undef JDSP_RUN_FRAME

// ---------------------------------------------------------------------------------------
// Multi-GPU sharding of the noise estimate.  Rank r owns blocks [b0, b1) of the
// global stream.  The run-length plan is replicated (every rank runs plan_kernel over the
// all-gathered voice flags); the running average A (SS:182-187) is an affine recurrence
// A <- a*A + b over the events, so each rank reduces its own events to one (alpha, beta[1024])
// pair, the pairs are all-gathered, and every rank folds the pairs of the ranks before it to get
// the A it starts from.  Latched estimates (SS:189-193) are then absolute, and only the last
// one per rank has to travel.
__global__ void event_range_kernel(const int *__restrict__ events, const DenoisePlan *__restrict__ plan,
                                   const int *__restrict__ ver_base, const unsigned long long *__restrict__ snap_mask,
                                   long b0, long b1, int *__restrict__ range)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = plan->n_events;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (events[mid] < b0) lo = mid + 1; else hi = mid; }
    range[0] = lo;
    hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (events[mid] < b1) lo = mid + 1; else hi = mid; }
    range[1] = lo;
    range[2] = b0 > 0 ? version_of(ver_base, snap_mask, b0 - 1) : 0;     // latches before the shard = row offset
    range[3] = (b1 > b0 ? version_of(ver_base, snap_mask, b1 - 1) : range[2]) - range[2];   // latches inside it
}

// A entering rank `rank` = the pairs of the ranks before it applied in order.
__global__ __launch_bounds__(256) void fold_summaries_kernel(const float *__restrict__ all, int rank, float *__restrict__ a_in)
{
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.f;
    for (int q = 0; q < rank; q++) a = all[(size_t)q * 1025] * a + all[(size_t)q * 1025 + 1 + bin];
    a_in[bin] = a;
}

// rows[0] = the estimate in effect at the block before the shard: the last latch of the nearest
// earlier rank that has one (zeros if none: rgdEstimatedNS starts at 0, SS:70).
__global__ __launch_bounds__(256) void select_row0_kernel(const float *__restrict__ last_all, int rank, float *__restrict__ rows)
{
    const int bin = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    for (int q = rank - 1; q >= 0; q--)
        if (last_all[(size_t)q * 1025] > 0.f) { v = last_all[(size_t)q * 1025 + 1 + bin]; break; }
    rows[bin] = v;
}

// A rank's events through the chunked average (noise_accum_kernel / noise_combine_kernel) in two passes: the first
// leaves the rank's composed map in `summary` (and every latched row as its partial map), the second -- once the
// maps of the ranks before it are known -- completes the rows from the average this rank starts from.
static int shard_accum_grid(long own) { return own < kNoiseChunks ? (int)(own > 0 ? own : 1) : kNoiseChunks; }

int launch_shard_summary(hipStream_t s, const short *pcm_ext, long n_ext, long ext0, long b0, long b1,
                         const int *events, const int *ev_n, const DenoisePlan *plan, const int *ver_base,
                         const unsigned long long *snap_mask, const float2 *table, int *range, const NoiseAccum &acc,
                         float *rows, float *summary)
{
    hipLaunchKernelGGL(event_range_kernel, dim3(1), dim3(64), 0, s, events, plan, ver_base, snap_mask, b0, b1, range);
    const int grid = shard_accum_grid(b1 - b0);
    hipLaunchKernelGGL(noise_accum_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm_ext, n_ext, (const DenoiseState *)nullptr,
                       events, ev_n, plan, ver_base, snap_mask, table, acc, rows, 10, (const int *)range, ext0);
    hipLaunchKernelGGL(noise_combine_kernel, dim3(1024 / kCombineBins), dim3(1024), 0, s, plan, (const DenoiseState *)nullptr,
                       (DenoiseState *)nullptr, acc, rows, grid, (const int *)range, (const float *)nullptr, summary,
                       (float *)nullptr, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_shard_rows(hipStream_t s, const float *summaries_all, int rank, long b0, long b1, const DenoisePlan *plan,
                      const int *range, const NoiseAccum &acc, float *a_in, float *rows, float *last)
{
    hipLaunchKernelGGL(fold_summaries_kernel, dim3(4), dim3(256), 0, s, summaries_all, rank, a_in);
    hipLaunchKernelGGL(noise_combine_kernel, dim3(1024 / kCombineBins), dim3(1024), 0, s, plan, (const DenoiseState *)nullptr,
                       (DenoiseState *)nullptr, acc, rows, shard_accum_grid(b1 - b0), range, (const float *)a_in,
                       (float *)nullptr, last, 1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_shard_row0(hipStream_t s, const float *last_all, int rank, float *rows)
{
    hipLaunchKernelGGL(select_row0_kernel, dim3(4), dim3(256), 0, s, last_all, rank, rows);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------
int launch_vad(hipStream_t s, const short *pcm, long n_blocks, const double *w_hi, int use_zcr, unsigned char *flags,
               long long *dbg_energy, int *dbg_zcr)
{
    if (n_blocks <= 0) return 0;
    const dim3 grid((unsigned)((n_blocks + kVadBlocksPerWave - 1) / kVadBlocksPerWave));
    if (!dbg_energy && !dbg_zcr)
        hipLaunchKernelGGL(vad_flags_kernel<8>, grid, dim3(64), 0, s, pcm, n_blocks, w_hi, use_zcr, flags);
    else
        hipLaunchKernelGGL(vad_kernel<8>, grid, dim3(64), 0, s, pcm, n_blocks, w_hi, use_zcr, flags, dbg_energy, dbg_zcr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// 256-sample blocks (frames of 512): w_hi = the second half of Hamming(512), 256 doubles
int launch_vad256(hipStream_t s, const short *pcm, long n_blocks, const double *w_hi, unsigned char *flags,
                  long long *dbg_energy, int *dbg_zcr, int use_zcr)
{
    if (n_blocks <= 0) return 0;
    const dim3 grid((unsigned)((n_blocks + kVadBlocksPerWave - 1) / kVadBlocksPerWave));
    if (!dbg_energy && !dbg_zcr)
        hipLaunchKernelGGL(vad_flags_kernel<4>, grid, dim3(64), 0, s, pcm, n_blocks, w_hi, use_zcr, flags);
    else
        hipLaunchKernelGGL(vad_kernel<4>, grid, dim3(64), 0, s, pcm, n_blocks, w_hi, use_zcr, flags, dbg_energy, dbg_zcr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_denoise_plan(hipStream_t s, const unsigned char *flags, long n_blocks, const DenoiseState *st_in,
                        DenoiseState *st_out, int *ver_base, unsigned long long *snap_mask, int *events, int *ev_n,
                        DenoisePlan *plan)
{
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, s, flags, n_blocks, &st_in->run_len, st_out, ver_base,
                       snap_mask, events, ev_n, plan, 10, (int *)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the same scan for callers with their own state layout (mvdr_kernels.hip)
int launch_run_plan(hipStream_t s, const unsigned char *flags, long n_blocks, const int *run_len_in, int *run_len_out,
                    int latch_run, int *ver_base, unsigned long long *snap_mask, int *events, int *ev_n,
                    DenoisePlan *plan)
{
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, s, flags, n_blocks, run_len_in, (DenoiseState *)nullptr,
                       ver_base, snap_mask, events, ev_n, plan, latch_run, run_len_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_noise_estimate(hipStream_t s, const short *pcm, long n_blocks, const DenoiseState *st_in,
                          DenoiseState *st_out, const int *events, const int *ev_n, const DenoisePlan *plan,
                          const int *ver_base, const unsigned long long *snap_mask, const float2 *table,
                          const NoiseAccum &acc, float *noise_rows)
{
    const int grid = n_blocks < kNoiseChunks ? (int)(n_blocks > 0 ? n_blocks : 1) : kNoiseChunks;
    if (n_blocks > 0)
        hipLaunchKernelGGL(noise_accum_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, st_in, events, ev_n, plan,
                           ver_base, snap_mask, table, acc, noise_rows, 10, (const int *)nullptr, 0L);
    hipLaunchKernelGGL(noise_combine_kernel, dim3(1024 / kCombineBins), dim3(1024), 0, s, plan, st_in, st_out, acc, noise_rows, grid,
                       (const int *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr, 1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int MODE, int K>
static void launch_dn(hipStream_t s, const short *pcm, long n_blocks, long calls_before, const DenoiseState *st_in,
                      DenoiseState *st_out, const int *ver_base, const unsigned long long *snap_mask,
                      const float *noise_rows, const float2 *table, short *out, float *precast, const DenoiseShard &sh)
{
    long grid = ((n_blocks + K - 1) / K + 7) / 8 * 8;
    hipLaunchKernelGGL((denoise_kernel<MODE, K>), dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before,
                       st_in, st_out, ver_base, snap_mask, noise_rows, table, out, precast, sh);
}

int launch_denoise(hipStream_t s, int mode, int k_opt, int n_cu, const short *pcm, long n_blocks, long calls_before,
                   const DenoiseState *st_in, DenoiseState *st_out, const int *ver_base,
                   const unsigned long long *snap_mask, const float *noise_rows, const float2 *table, short *out,
                   float *precast, const DenoiseShard *shard)
{
    if (n_blocks <= 0) return 0;
    DenoiseShard sh;
    if (shard) sh = *shard;
    else {
        sh.ver_block_off = 0;
        sh.ver_row_off = nullptr;
        sh.emit_from = calls_before >= 2 ? 0 : 2 - calls_before;
        sh.emit_to = n_blocks;
    }
    if (k_opt == 0) {
        // one round of resident waves: JDSP_DENOISE_RESIDENT per SIMD, 4 SIMDs per CU; never fewer than 4 blocks
        // per wave (a quarter of halo overhead at most)
        const long slots = (long)(n_cu > 0 ? n_cu : 256) * 4 * JDSP_DENOISE_RESIDENT;
        long waves = (n_blocks + 3) / 4;
        if (waves > slots) waves = slots;
        long run = (n_blocks + waves - 1) / waves;
#ifdef JDSP_DENOISE_RUN_BLOCKS
        run = JDSP_DENOISE_RUN_BLOCKS;                             // tools/build_variant.sh: run-length sweeps only
#endif
        waves = (n_blocks + run - 1) / run;
        const long grid = (waves + 7) / 8 * 8;
        if (mode == 0)
            hipLaunchKernelGGL(denoise_run_kernel<0>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                               st_out, ver_base, snap_mask, noise_rows, table, out, precast, sh, (int)run);
        else
            hipLaunchKernelGGL(denoise_run_kernel<1>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                               st_out, ver_base, snap_mask, noise_rows, table, out, precast, sh, (int)run);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
#define JDSP_DN(M, KK) launch_dn<M, KK>(s, pcm, n_blocks, calls_before, st_in, st_out, ver_base, snap_mask, noise_rows, table, out, precast, sh)
    if (mode == 0) {
        switch (k_opt) {
        case 1: JDSP_DN(0, 1); break;
        case 2: JDSP_DN(0, 2); break;
        case 4: JDSP_DN(0, 4); break;
        default: JDSP_DN(0, 8); break;
        }
    } else {
        switch (k_opt) {
        case 1: JDSP_DN(1, 1); break;
        case 2: JDSP_DN(1, 2); break;
        case 4: JDSP_DN(1, 4); break;
        default: JDSP_DN(1, 8); break;
        }
    }
#undef JDSP_DN
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// =======================================================================================
// FFT_PROCESSING_SIZE 512, BLOCK_LEN = KEEP_LEN 256 -- the same programs with the macros of SS:53-55 / WF:42-44 at
// half their values, which is how BASELINE config 3 words the workload ("512-pt STFT 50 % hop").
//
// A 512-sample real frame is half of the wave transform's work, so TWO consecutive frames ride one 512-point
// complex transform: z[n] = a[n] + j b[n] (both already windowed), Z = FFT512(z), and
//     A[k] = (Z[k] + conj Z[512-k]) / 2,      B[k] = -j (Z[k] - conj Z[512-k]) / 2
// are the two frames' spectra (the 1/2 is folded into the window table, as everywhere).  Every lane owns the bin
// pairs (k, 512-k), k = lane + 64 q, so A[512-k] = conj A[k] costs nothing; the per-bin gain (SS:233-242 /
// WF:196-213) is applied to both frames and to both bins of a pair with their own N[k], N[512-k]; the gained
// spectra go back into ONE inverse transform as Y = Ya + j Yb, whose real and imaginary parts are the two frames'
// time signals (both spectra are Hermitian).  Overlap-add with hop 256 pairs "lane + 64 d" registers exactly as
// the 1024-point kernel does with hop 512: y[d], d < 4, is the first half of a frame, y[d + 4] its second half.
//
// One wave owns 2 kDn512Pairs - 1 consecutive blocks: pairs (j0-1, j0), (j0+1, j0+2), ...; the first frame of the
// first pair is the halo that rebuilds the overlap tail (its output block belongs to the previous wave).
constexpr int kDn512Pairs = 4;
constexpr int kDn512BlocksPerWave = 2 * kDn512Pairs - 1;

// Sample s of this call's stream; s in [-256, 0) is the previous call's last block (state), anything else
// outside the call is silence.  s = base + lane + 64 t with base a multiple of 256: every branch is wave-uniform.
__device__ __forceinline__ float dn512_sample(const short *__restrict__ pcm, long n_samples,
                                              const DenoiseState *__restrict__ st_in, long s)
{
    if (s >= 0) return s < n_samples ? (float)pcm[s] : 0.f;
    if (s >= -256 && st_in) return (float)st_in->prev[256 + s];
    return 0.f;
}

// z = (a + j b) * window -> natural-order image of Zh = FFT512(z) in LDS (slot 512 = slot 0)
__device__ __forceinline__ void dn512_forward(const float (&xa)[8], const float (&xb)[8], const float (&win)[8],
                                              const WaveTwiddles &tw, float2 *lds, int lane)
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = make_float2(xa[r] * win[r], xb[r] * win[r]);
    wave_fft512<false>(v, lds, lane, tw);
    store_natural_image(lds, lane, v);
    wave_lds_fence();
}

// A12 on 512-point frames, chunked like noise_accum_kernel; events e and e + 1 of a chunk share a transform (frame pair
// a + j b).  |X[512 - k]| = |X[k]| of a real frame: lane l keeps the maps of bins l + 64 q (q < 4) and writes each to its
// mirror bin as well (the carried-in average is symmetric for the same reason); bin 256 is lane 0's.
__global__ __launch_bounds__(64) void noise_accum512_kernel(const short *__restrict__ pcm, long n_blocks,
                                                            const DenoiseState *__restrict__ st_in,
                                                            const int *__restrict__ events, const int *__restrict__ ev_n,
                                                            const DenoisePlan *__restrict__ plan,
                                                            const int *__restrict__ ver_base,
                                                            const unsigned long long *__restrict__ snap_mask,
                                                            const float2 *__restrict__ table,
                                                            const float *__restrict__ win512, NoiseAccum acc,
                                                            float *__restrict__ noise_rows, int latch_run,
                                                            const int *__restrict__ range, long ext0)
{
    // range / ext0: a sharded run's slice of the event list and the global index of pcm's first block, as in
    // noise_accum_kernel (st_in is then NULL: a rank's events never reach back past its halo)
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const int e_base = range ? range[0] : 0, row_off = range ? range[2] : 0;
    const int n_events = range ? range[1] - range[0] : plan->n_events;
    const ChunkGeom cg = chunk_geom(n_events, (int)gridDim.x);
    const int chunk = blockIdx.x;
    if (chunk >= cg.n_chunks) return;
    const int e0 = chunk * cg.per_chunk;
    const int e1 = e0 + cg.per_chunk < n_events ? e0 + cg.per_chunk : n_events;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = win512[lane + 64 * r];
    const long n_samples = n_blocks * 256;
    float b[4] = {0.f, 0.f, 0.f, 0.f}, b256 = 0.f, alpha = 1.0f;
    auto put = [&](float *row) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = lane + 64 * q;
            row[k] = b[q];
            row[(512 - k) & 511] = b[q];
        }
        if (lane == 0) row[256] = b256;
    };
    auto step = [&](const float (&m)[4], float m256, int e) {
        const int n = ev_n[e_base + e];
        const float h = n >= 3 ? 0.5f : 1.0f;                    // SS:182-187
#pragma unroll
        for (int q = 0; q < 4; q++) b[q] = (b[q] + m[q]) * h;
        b256 = (b256 + m256) * h;
        alpha *= h;
        if (n == latch_run) {                                    // SS:189-193
            const int row = version_of(ver_base, snap_mask, events[e_base + e]) - row_off;
            put(noise_rows + (size_t)row * 1024);
            if (lane == 0) { acc.lat_alpha[row] = alpha; acc.lat_chunk[row] = chunk; }
        }
    };
    for (int e = e0; e < e1; e += 2) {
        const int ea = e, eb = e + 1 < e1 ? e + 1 : e;
        const long sa = ((long)events[e_base + ea] - ext0 - 1) * 256, sb = ((long)events[e_base + eb] - ext0 - 1) * 256;   // [previous block, block] (SS:165-170)
        float xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            xa[r] = dn512_sample(pcm, n_samples, st_in, sa + lane + 64 * r);
            xb[r] = dn512_sample(pcm, n_samples, st_in, sb + lane + 64 * r);
        }
        dn512_forward(xa, xb, win, tw, lds, lane);
        float ma[4], mb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = lane + 64 * q;
            const float2 zk = lds[k], zm = lds[512 - k];
            const float2 A = cadd_conj(zk, zm), B = csub_conj_mj(zk, zm);
            ma[q] = __builtin_amdgcn_sqrtf(A.x * A.x + A.y * A.y);      // hardware square root, 1 ulp (see noise_accum_kernel)
            mb[q] = __builtin_amdgcn_sqrtf(B.x * B.x + B.y * B.y);
        }
        const float2 z = lds[256];
        wave_lds_fence();
        step(ma, fabsf(2.f * z.x), ea);
        if (eb != ea) step(mb, fabsf(2.f * z.y), eb);
    }
    put(acc.chunk_beta + (size_t)chunk * 1024);
    if (lane == 0) acc.chunk_alpha[chunk] = alpha;
}

template <int MODE>
__device__ __forceinline__ void dn512_pair(const float (&xa)[8], const float (&xb)[8], const float (&win)[8],
                                           const WaveTwiddles &tw, float2 *lds, int lane,
                                           const float *__restrict__ na, const float *__restrict__ nb, float2 (&y)[8])
{
    dn512_forward(xa, xb, win, tw, lds, lane);
    float2 yka[4], ykb[4], yma[4], ymb[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int k = lane + 64 * q, m = (512 - k) & 511;
        const float2 zk = lds[k], zm = lds[512 - k];
        const float2 A = cadd_conj(zk, zm), B = csub_conj_mj(zk, zm);
        // bin k of both frames, and bin 512-k (= the conjugates) with ITS noise values
        yka[q] = apply_gain<MODE>(A, na[k]); ykb[q] = apply_gain<MODE>(B, nb[k]);
        yma[q] = apply_gain<MODE>(A, na[m]); ymb[q] = apply_gain<MODE>(B, nb[m]);
    }
    float2 a256, b256;
    {
        const float2 z = lds[256];                                // self-mirrored bin: A = 2 Re z, B = 2 Im z, both real
        a256 = apply_gain<MODE>(make_float2(2.f * z.x, 0.f), na[256]);
        b256 = apply_gain<MODE>(make_float2(2.f * z.y, 0.f), nb[256]);
    }
    // A frame with a non-finite bin (Wiener's 0/0 on an all-zero frame before any estimate, WF:204) has an all-NaN
    // inverse transform in the reference.  Here it must not poison the frame it shares the transform with: its
    // spectrum goes in as zero and its samples come out as NaN.
    bool bad_a = false, bad_b = false;
    if (MODE == 1) {
        float ta = a256.x * 0.f, tb = b256.x * 0.f;               // NaN iff NaN or inf
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ta += (yka[q].x + yka[q].y + yma[q].x + yma[q].y) * 0.f;
            tb += (ykb[q].x + ykb[q].y + ymb[q].x + ymb[q].y) * 0.f;
        }
        bad_a = __ballot(ta != ta) != 0ull;
        bad_b = __ballot(tb != tb) != 0ull;
    }
    float2 yk[4], ym[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float2 ak = bad_a ? make_float2(0.f, 0.f) : yka[q], am = bad_a ? make_float2(0.f, 0.f) : yma[q];
        const float2 bk = bad_b ? make_float2(0.f, 0.f) : ykb[q], bm = bad_b ? make_float2(0.f, 0.f) : ymb[q];
        yk[q] = cadd_pj(ak, bk);                                  // Ya[k] + j Yb[k]
        const float2 t = cadd_mj(am, bm);                         // conj(Ya) + j conj(Yb) = conj(Ya - j Yb)
        ym[q] = make_float2(t.x, -t.y);
    }
    const float2 y256 = cadd_pj(bad_a ? make_float2(0.f, 0.f) : a256, bad_b ? make_float2(0.f, 0.f) : b256);
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int k = lane + 64 * q;
        lds[512 - k] = ym[q];                                     // k = 0 lands in the spare slot 512
        lds[k] = yk[q];
    }
    if (lane == 0) lds[256] = y256;
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = lds[lane + 64 * r];
    wave_lds_fence();
    wave_fft512<true>(y, lds, lane, tw);
    // the reference's 1/N after FFTW's unnormalised inverse (SS:248) with N = 512; a power of two, exact
#pragma unroll
    for (int d = 0; d < 8; d++)
        y[d] = make_float2(bad_a ? __builtin_nanf("") : y[d].x * (1.0f / 512.0f), bad_b ? __builtin_nanf("") : y[d].y * (1.0f / 512.0f));
    wave_lds_fence();
}

template <int MODE>
__global__ __launch_bounds__(64, 3) void denoise512_kernel(
    const short *__restrict__ pcm, long n_blocks, long calls_before, const DenoiseState *__restrict__ st_in,
    DenoiseState *st_out, const int *__restrict__ ver_base, const unsigned long long *__restrict__ snap_mask,
    const float *__restrict__ noise_rows, const float2 *__restrict__ table, const float *__restrict__ win512,
    short *__restrict__ out, float *__restrict__ precast, DenoiseShard sh)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                // XCD-aware chunk order (speed only)
    const long j0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * kDn512BlocksPerWave;
    if (j0 >= n_blocks) return;
    const long n_samples = n_blocks * 256;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = win512[lane + 64 * r];

    // samples base + lane + 64 t: frame (j0 - 1 + 2p) starts at t = 8 p, frame (j0 + 2p) at t = 8 p + 4
    float xs[8 * kDn512Pairs + 4];
    const long base = (j0 - 2) * 256;
#pragma unroll
    for (int t = 0; t < 8 * kDn512Pairs + 4; t++) xs[t] = dn512_sample(pcm, n_samples, st_in, base + lane + 64 * t);

    float tail[4];
    const long first_emit = sh.emit_from;
#pragma unroll
    for (int p = 0; p < kDn512Pairs; p++) {
        const long ja = j0 - 1 + 2 * p, jb = ja + 1;
        if (ja >= n_blocks) break;
        float xa[8], xb[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { xa[r] = xs[8 * p + r]; xb[r] = xs[8 * p + 4 + r]; }
        float2 y[8];
        dn512_pair<MODE>(xa, xb, win, tw, lds, lane, noise_row(noise_rows, ver_base, snap_mask, ja >= 0 ? ja : 0, sh),
                         noise_row(noise_rows, ver_base, snap_mask, jb < n_blocks ? jb : n_blocks - 1, sh), y);
        // the very first call of a stream only stashes its block (SS:211-216): no transform, empty overlap
        const bool a_void = calls_before + ja <= 0, b_void = calls_before + jb <= 0;
        float oa[4], ob[4];
        if (p == 0) {
            // frame ja = j0 - 1 is the halo: its block belongs to the previous wave (or call); only its second half counts
            if (j0 == 0) {
#pragma unroll
                for (int d = 0; d < 4; d++) tail[d] = st_in->tail[lane + 64 * d];       // rgsdOveraped carried over
            } else {
#pragma unroll
                for (int d = 0; d < 4; d++) tail[d] = a_void ? 0.f : y[d + 4].x;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                oa[d] = tail[d] + (a_void ? 0.f : y[d].x);                              // SS:248 overlap-add
                tail[d] = a_void ? 0.f : y[d + 4].x;                                    // SS:255-256
            }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) ob[d] = tail[d] + (b_void ? 0.f : y[d].y);
        float tail_b[4];
#pragma unroll
        for (int d = 0; d < 4; d++) tail_b[d] = b_void ? 0.f : y[d + 4].y;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const long j = half ? jb : ja;
            if (half == 0 && p == 0) continue;
            if (j >= n_blocks) continue;
            const float *o = half ? ob : oa;
            if (j >= first_emit && j < sh.emit_to) {
                const long oi = j - first_emit;
#pragma unroll
                for (int d = 0; d < 4; d++) out[oi * 256 + lane + 64 * d] = (short)cast_i16_bits(o[d]);
                if (precast) {
#pragma unroll
                    for (int d = 0; d < 4; d++) precast[oi * 256 + lane + 64 * d] = o[d];
                }
            }
            if (j == n_blocks - 1) {                                                    // SS:257 and the overlap carried out
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    st_out->prev[lane + 64 * d] = pcm[j * 256 + lane + 64 * d];
                    st_out->tail[lane + 64 * d] = half ? tail_b[d] : tail[d];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) tail[d] = tail_b[d];
    }
}

// The same frames with the spectrum in registers and every mirror pair of bins owned by one lane (frame_io.h,
// pair_fetch_lds / pair_return_lds): after the forward transform lane l works on k = l + 64 d, d < 5 -- A[k], B[k] from
// Z[k] and Z[512 - k], ONE gain per frame and bin (N[512 - k] = N[k]: the estimate of a real frame's spectrum is
// symmetric by construction, noise_accum512_kernel writes both from one value), Z'[k] = Ya + j Yb kept and
// Z'[512 - k] = conj(Ya - j Yb) sent to its owner.  Ten gains per lane and frame pair instead of eighteen, no natural-
// order image and no Y image; the wave is persistent over `run` consecutive blocks (run odd: pairs (j0 - 1, j0),
// (j0 + 1, j0 + 2), ...), keeps both frames' noise values in registers until the estimate changes, and requests the
// next pair's two new blocks before this pair's arithmetic.  The reference's 1/512 (SS:248) is folded into the gain.
template <int MODE>
__device__ __forceinline__ void load_noise512_regs(float (&n)[5], const float *__restrict__ row, int lane)
{
    const float sc = MODE == 0 ? (1.0f / 512.0f) : 1.0f;
#pragma unroll
    for (int d = 0; d < 5; d++) n[d] = sc * row[(lane + 64 * d) & 511];   // d = 4: bins 256..319 (bin k and 512 - k share a value)
}

// waves per SIMD: spectral subtraction fits 128 registers; Wiener with its NaN isolation does not (spills at 128:
// 97 us against 89 us at three waves, profiles/r02_denoise512_run.txt)
#define JDSP_DENOISE512_WAVES(MODE) ((MODE) == 0 ? 4 : 3)
template <int MODE>
__global__ __launch_bounds__(64, JDSP_DENOISE512_WAVES(MODE)) void denoise512_run_kernel(
    const short *__restrict__ pcm, long n_blocks, long calls_before, const DenoiseState *__restrict__ st_in,
    DenoiseState *st_out, const int *__restrict__ ver_base, const unsigned long long *__restrict__ snap_mask,
    const float *__restrict__ noise_rows, const float2 *__restrict__ table, const float *__restrict__ win512,
    short *__restrict__ out, float *__restrict__ precast, DenoiseShard sh, int run)
{
    __shared__ __attribute__((aligned(16))) float2 lds[kWaveLdsComplex];
    const int lane = threadIdx.x;
    const long per_xcd = (gridDim.x + 7) >> 3;                // XCD-aware run order (speed only)
    const long j0 = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * run;
    if (j0 >= n_blocks) return;
    const long j1 = j0 + run < n_blocks ? j0 + run : n_blocks;
    const long n_samples = n_blocks * 256;
    WaveTwiddles tw;
    load_wave_twiddles(tw, table, lane);
    float win[8];
#pragma unroll
    for (int r = 0; r < 8; r++) win[r] = win512[lane + 64 * r];

    // blocks ja - 1, ja, ja + 1 of the current pair (ja, ja + 1), samples lane + 64 t of each
    // (the next pair's blocks stay raw halfwords until the rotation at the bottom of the loop: a conversion next to the
    // loads sits inside their wave-uniform branch, and the s_waitcnt with it -- the "prefetch" then costs a full memory
    // round trip per pair before the transforms start)
    float xk[3][4];
    int nx[2][4];
    if (j0 >= 2 && j0 + 1 <= n_blocks) {                      // all three inside this call's buffer (wave-uniform): twelve
        const short *src = pcm + (j0 - 2) * 256 + lane;       // loads in flight together; guarded, each waits for itself
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) xk[b][t] = (float)src[256 * b + 64 * t];
    } else {
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) xk[b][t] = dn512_sample(pcm, n_samples, st_in, (j0 - 2 + b) * 256 + lane + 64 * t);
    }
    const int row_off = sh.ver_row_off ? *sh.ver_row_off : 0;
    float na[5], nb[5], tail[4];
    const float *row_a = nullptr, *row_b = nullptr;
    const long first_emit = sh.emit_from;
    bool halo = true;                                         // frame j0 - 1 only rebuilds the overlap tail
    unsigned prio_step = blockIdx.x >> 10;                    // as in denoise_run_kernel: one priority step per frame pair
    for (long ja = j0 - 1; ja < j1; ja += 2) {
#if JDSP_DENOISE_PRIO
        switch (prio_step++ & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
#endif
        const long jb = ja + 1;
        if (ja + 2 < j1) {                                    // the next pair's two new blocks, needed one iteration from now
            if (ja + 2 >= 0 && ja + 4 <= n_blocks) {            // both inside this call's buffer (wave-uniform)
                const short *src = pcm + (ja + 2) * 256 + lane;
#pragma unroll
                for (int t = 0; t < 4; t++) { nx[0][t] = src[64 * t]; nx[1][t] = src[256 + 64 * t]; }
            } else {
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) nx[b][t] = (int)dn512_sample(pcm, n_samples, st_in, (ja + 2 + b) * 256 + lane + 64 * t);
            }
        }
        {
            const float *ra = noise_row_at(noise_rows, ver_base, snap_mask, ja >= 0 ? ja : 0, sh.ver_block_off, row_off);
            const float *rb = noise_row_at(noise_rows, ver_base, snap_mask, jb < n_blocks ? jb : n_blocks - 1, sh.ver_block_off, row_off);
            if (ra != row_a) { row_a = ra; load_noise512_regs<MODE>(na, ra, lane); }     // wave-uniform: a new estimate was latched
            if (rb != row_b) { row_b = rb; load_noise512_regs<MODE>(nb, rb, lane); }
        }
        float2 v[8], y[8];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            v[r] = make_float2(xk[0][r] * win[r], xk[1][r] * win[r]);
            v[r + 4] = make_float2(xk[1][r] * win[r + 4], xk[2][r] * win[r + 4]);
        }
        wave_fft512<false>(v, lds, lane, tw);
        float2 zr[5], ya[5], yb[5], ret[4];
        wave_lds_fence();
        pair_fetch_lds(v, lds, lane, zr);
#pragma unroll
        for (int d = 0; d < 5; d++) {
            ya[d] = apply_gain_scaled<MODE, 9>(cadd_conj(v[d], zr[d]), na[d]);        // frame a, bin lane + 64 d
            yb[d] = apply_gain_scaled<MODE, 9>(csub_conj_mj(v[d], zr[d]), nb[d]);     // frame b
        }
        // A frame with a non-finite bin (Wiener's 0/0 on an all-zero frame before any estimate, WF:204) has an all-NaN
        // inverse transform in the reference.  Here it must not poison the frame it shares the transform with: its
        // spectrum goes in as zero and its samples come out as NaN.
        bool bad_a = false, bad_b = false;
        if (MODE == 1) {
            float ta = 0.f, tb = 0.f;                                 // NaN iff some bin is NaN or inf
#pragma unroll
            for (int d = 0; d < 5; d++) { ta += (ya[d].x + ya[d].y) * 0.f; tb += (yb[d].x + yb[d].y) * 0.f; }
            bad_a = __ballot(ta != ta) != 0ull;
            bad_b = __ballot(tb != tb) != 0ull;
        }
#pragma unroll
        for (int d = 0; d < 5; d++) {
            const float2 ak = bad_a ? make_float2(0.f, 0.f) : ya[d], bk = bad_b ? make_float2(0.f, 0.f) : yb[d];
            y[d] = cadd_pj(ak, bk);                                   // Z'[k] = Ya[k] + j Yb[k]
            if (d < 4) ret[d] = cconj_sub_j(ak, bk);                  // Z'[512 - k] = conj(Ya[k]) + j conj(Yb[k])
        }
        pair_return_lds(ret, lds, lane, y);
        wave_fft512<true>(y, lds, lane, tw);
        wave_lds_fence();
        if (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 8; d++) y[d] = make_float2(bad_a ? __builtin_nanf("") : y[d].x, bad_b ? __builtin_nanf("") : y[d].y);
        }
        // the very first call of a stream only stashes its block (SS:211-216): no transform, empty overlap
        const bool a_void = calls_before + ja <= 0, b_void = calls_before + jb <= 0;
        float oa[4], ob[4], tail_b[4];
        if (halo) {
            if (j0 == 0) {
#pragma unroll
                for (int d = 0; d < 4; d++) tail[d] = st_in->tail[lane + 64 * d];       // rgsdOveraped carried over
            } else {
#pragma unroll
                for (int d = 0; d < 4; d++) tail[d] = a_void ? 0.f : y[d + 4].x;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 4; d++) {
                oa[d] = tail[d] + (a_void ? 0.f : y[d].x);                              // SS:248 overlap-add
                tail[d] = a_void ? 0.f : y[d + 4].x;                                    // SS:255-256
            }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            ob[d] = tail[d] + (b_void ? 0.f : y[d].y);
            tail_b[d] = b_void ? 0.f : y[d + 4].y;
        }
#if JDSP_DENOISE_TAKE_EARLY
        if (ja + 2 < j1) {                                        // the next pair's samples are taken before this pair's stores
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int t = 0; t < 4; t++) asm volatile("" : "+v"(nx[b][t]));
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const long j = half ? jb : ja;
            if ((half == 0 && halo) || j >= n_blocks) continue;
            const float *o = half ? ob : oa;
            if (j >= first_emit && j < sh.emit_to) {
                const long oi = j - first_emit;
#pragma unroll
                for (int d = 0; d < 4; d++) out[oi * 256 + lane + 64 * d] = (short)cast_i16_bits(o[d]);
                if (precast) {
#pragma unroll
                    for (int d = 0; d < 4; d++) precast[oi * 256 + lane + 64 * d] = o[d];
                }
            }
            if (j == n_blocks - 1) {                                                    // SS:257 and the overlap carried out
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    st_out->prev[lane + 64 * d] = pcm[j * 256 + lane + 64 * d];
                    st_out->tail[lane + 64 * d] = half ? tail_b[d] : tail[d];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {
            tail[d] = tail_b[d];
            xk[0][d] = xk[2][d];
            xk[1][d] = (float)nx[0][d];
            xk[2][d] = (float)nx[1][d];
        }
        halo = false;
    }
}

int launch_noise_estimate512(hipStream_t s, const short *pcm, long n_blocks, const DenoiseState *st_in,
                             DenoiseState *st_out, const int *events, const int *ev_n, const DenoisePlan *plan,
                             const int *ver_base, const unsigned long long *snap_mask, const float2 *table,
                             const float *win512, const NoiseAccum &acc, float *noise_rows)
{
    const int grid = n_blocks < kNoiseChunks ? (int)(n_blocks > 0 ? n_blocks : 1) : kNoiseChunks;
    if (n_blocks > 0)
        hipLaunchKernelGGL(noise_accum512_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, st_in, events, ev_n,
                           plan, ver_base, snap_mask, table, win512, acc, noise_rows, 10, (const int *)nullptr, 0L);
    // bins 0..511 only (rows keep the 1024-float pitch of the 1024-point path)
    hipLaunchKernelGGL(noise_combine_kernel, dim3(512 / kCombineBins), dim3(1024), 0, s, plan, st_in, st_out, acc, noise_rows, grid,
                       (const int *)nullptr, (const float *)nullptr, (float *)nullptr, (float *)nullptr, 1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// sharded runs on 512-point frames: launch_shard_summary / launch_shard_rows with the 512-point accumulate kernel and
// the combine kernel over bins 0..511 (rows and summaries keep the 1024-float pitch; bins 512.. are never read)
int launch_shard_summary512(hipStream_t s, const short *pcm_ext, long n_ext, long ext0, long b0, long b1,
                            const int *events, const int *ev_n, const DenoisePlan *plan, const int *ver_base,
                            const unsigned long long *snap_mask, const float2 *table, const float *win512, int *range,
                            const NoiseAccum &acc, float *rows, float *summary)
{
    hipLaunchKernelGGL(event_range_kernel, dim3(1), dim3(64), 0, s, events, plan, ver_base, snap_mask, b0, b1, range);
    const int grid = shard_accum_grid(b1 - b0);
    hipLaunchKernelGGL(noise_accum512_kernel, dim3((unsigned)grid), dim3(64), 0, s, pcm_ext, n_ext, (const DenoiseState *)nullptr,
                       events, ev_n, plan, ver_base, snap_mask, table, win512, acc, rows, 10, (const int *)range, ext0);
    hipLaunchKernelGGL(noise_combine_kernel, dim3(512 / kCombineBins), dim3(1024), 0, s, plan, (const DenoiseState *)nullptr,
                       (DenoiseState *)nullptr, acc, rows, grid, (const int *)range, (const float *)nullptr, summary,
                       (float *)nullptr, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_shard_rows512(hipStream_t s, const float *summaries_all, int rank, long b0, long b1, const DenoisePlan *plan,
                         const int *range, const NoiseAccum &acc, float *a_in, float *rows, float *last)
{
    hipLaunchKernelGGL(fold_summaries_kernel, dim3(2), dim3(256), 0, s, summaries_all, rank, a_in);
    hipLaunchKernelGGL(noise_combine_kernel, dim3(512 / kCombineBins), dim3(1024), 0, s, plan, (const DenoiseState *)nullptr,
                       (DenoiseState *)nullptr, acc, rows, shard_accum_grid(b1 - b0), range, (const float *)a_in,
                       (float *)nullptr, last, 1);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_shard_row0_512(hipStream_t s, const float *last_all, int rank, float *rows)
{
    hipLaunchKernelGGL(select_row0_kernel, dim3(2), dim3(256), 0, s, last_all, rank, rows);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_denoise512(hipStream_t s, int mode, int n_cu, const short *pcm, long n_blocks, long calls_before,
                      const DenoiseState *st_in, DenoiseState *st_out, const int *ver_base,
                      const unsigned long long *snap_mask, const float *noise_rows, const float2 *table,
                      const float *win512, short *out, float *precast, const DenoiseShard *shard)
{
    if (n_blocks <= 0) return 0;
    DenoiseShard sh;
    if (shard) sh = *shard;
    else {
        sh.ver_block_off = 0;
        sh.ver_row_off = nullptr;
        sh.emit_from = calls_before >= 2 ? 0 : 2 - calls_before;
        sh.emit_to = n_blocks;
    }
#ifndef JDSP_DENOISE512_RUN
#define JDSP_DENOISE512_RUN 1
#endif
    if (JDSP_DENOISE512_RUN) {
        // one round of resident waves (4 per SIMD); an odd number of blocks per wave, 7 at least
        const long slots = (long)(n_cu > 0 ? n_cu : 256) * 4 * JDSP_DENOISE512_WAVES(mode);
        long run = (n_blocks + slots - 1) / slots;
        if (run < 7) run = 7;
        run |= 1;
        const long waves = (n_blocks + run - 1) / run;
        const long grid = (waves + 7) / 8 * 8;
        if (mode == 0)
            hipLaunchKernelGGL(denoise512_run_kernel<0>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                               st_out, ver_base, snap_mask, noise_rows, table, win512, out, precast, sh, (int)run);
        else
            hipLaunchKernelGGL(denoise512_run_kernel<1>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                               st_out, ver_base, snap_mask, noise_rows, table, win512, out, precast, sh, (int)run);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const long waves = (n_blocks + kDn512BlocksPerWave - 1) / kDn512BlocksPerWave;
    const long grid = (waves + 7) / 8 * 8;
    if (mode == 0)
        hipLaunchKernelGGL(denoise512_kernel<0>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                           st_out, ver_base, snap_mask, noise_rows, table, win512, out, precast, sh);
    else
        hipLaunchKernelGGL(denoise512_kernel<1>, dim3((unsigned)grid), dim3(64), 0, s, pcm, n_blocks, calls_before, st_in,
                           st_out, ver_base, snap_mask, noise_rows, table, win512, out, precast, sh);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
