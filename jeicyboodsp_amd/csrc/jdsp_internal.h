// jdsp_internal.h -- shared declarations of libjdsp (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jdsp.h"

struct jdsp_ctx {
    int device = 0;
    int n_cu = 0;
    size_t hbm_bytes = 0;
    char name[64] = {0};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // the stream work is enqueued on
    hipEvent_t switch_ev = nullptr;    // orders the old stream's work before the new stream's (jdsp_set_stream)
    std::string error;
    int opt_stft_fpw = 0;              // 0 = auto
    int opt_stft_window = 0;           // 0 Hamming (the reference), 1 Hann -- jdsp_stft_* only
    int opt_stft_read_pass = -1;       // read-only pass that pulls the PCM into the Infinity Cache before the transform:
                                       // 0 never, 1 always, -1 auto (large hop-512 batches); stft_kernels.hip
    int opt_stft_touch_wg = 0;         // tuning: workgroups per CU of that pass (0 = default)
    int opt_stft_f64_kernel = 0;       // 0: stft1024_f64_v2_kernel, 1: round 2's stft1024_f64_kernel (A/B)
    int opt_stft_f64_fpw = 0;          // frames one wave of the FP64 analysis walks (0 = one round of resident waves)
    float2 *stft1024_table_hann = nullptr, *win512_hann = nullptr;
    float2 *stft1024_table_rect = nullptr;   // rectangular window: the partitioned convolver's forward frames
    // device tables, created on first use
    float2 *stft1024_table = nullptr;
    float2 *win512 = nullptr;          // halved Hamming-512 pairs
    double2 *c2c_tw[16] = {nullptr};   // by log2(n_fft)
    double *stft_f64_table = nullptr;  // FP64 STFT: window + split twiddles (fft_c2c_kernels.hip)
    float2 *conv_tw4096 = nullptr, *conv_tw8192 = nullptr;
    double *vad_w_hi = nullptr;        // second half of the FP64 Hamming window
    double *vad_w_ex[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // jdsp_vad_blocks_ex: [variant][block 512 | 256]
    // pinned-host pipeline of jdsp_stft_i16: copy-in / compute / copy-out on three streams
    hipStream_t pipe_in = nullptr, pipe_out = nullptr;
    hipEvent_t pipe_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *pipe_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // in[2], out[2]
    size_t pipe_cap[4] = {0, 0, 0, 0};
};

namespace jdsp {

// All of the reference's static state for one SS/Wiener stream (device memory).
struct DenoiseState {
    int run_len;          // main(): iNumOfIteration                      SS:72,99,108
    int pad[3];
    float avg[1024];      // EstimateNoiseSpectrum: rgsdAveragedNS         SS:161
    float noise[1024];    // main(): rgdEstimatedNS                        SS:70
    short prev[512];      // the keep buffers = previous input block       SS:164,208
    float tail[512];      // rgsdOveraped[0..511] after the shift          SS:209,255
};
struct DenoisePlan { int n_events, n_snap, pad0, pad1; };

// Workspace of the chunked noise average (denoise_kernels.hip, noise_accum_kernel): one affine map per chunk of events
constexpr int kNoiseChunks = 4096;         // most chunks a call is cut into = noise_accum's largest grid (4 waves per SIMD)
struct NoiseAccum {
    float *chunk_alpha;    // [kNoiseChunks]
    float *chunk_beta;     // [kNoiseChunks][1024]
    float *a_start;        // [kNoiseChunks][1024]   the average entering each chunk
    float *lat_alpha;      // [rows]                 per latched row: the chunk's alpha up to and including the latch
    int *lat_chunk;        // [rows]                 ... and its chunk
};

// How denoise_kernel maps local block indices onto the (possibly global) plan and what it emits.
struct DenoiseShard {
    long ver_block_off;        // global index of local block 0 (0 when not sharded)
    const int *ver_row_off;    // device int: latches before the shard, subtracted from the version (or NULL)
    long emit_from, emit_to;   // local block range written to `out`
};

// BeamForming_MVDR_ver1.cpp's state between calls (device memory)
constexpr int kMvdrTableVersions = 8192;     // 2-microphone MVDR: calls with fewer events than this get their weights from a table (16 KB per version)
constexpr int kMvnChunks = 128;            // chunks the n-microphone covariance update cuts a call's events into

struct MvdrState {
    int run_len;          // main(): iNumOfIteration             MVDR:59,99,108
    int pad[3];
    double corr[4];       // rgdSpatialCorr, row-major             MVDR:57
    short prev_l[512];    // previous block of each channel (temp buffers :56, keep buffers :130-131)
    short prev_r[512];
};

// Device-side view of one MFCC configuration (passed by value to the kernel).
struct MfccDev {
    int win_len, hop, n_chan, n_cep, bin_stride;   // bin_stride 2: 512-point bins out of the 1024-point transform
    int n_bins;
    float preemph;
    const float2 *window;        // [512] halved Hamming pairs over win_len, zero beyond
    const float *mel_fb;         // [512] rgdFilterBank as float (zero beyond n_bins)
    const int *mel_k;            // [512] rgdFiBins
    // lane-per-index form of the same filterbank (mfcc_x2_kernel): lane L sums the bins [seg[L].x, +seg[L].y)
    // (at most piece_len, all with rgdFiBins value seg[L].z); seg_wc[q * 64 + L] = {w, 1 - w of bin 2 q; w, 1 - w of bin 2 q + 1},
    // both ZERO past the piece's last bin (eight dwordx4 loads per lane instead of thirty-two dwords, and each
    // (w, 1 - w) an aligned register pair for the packed multiply-add); seg_ok = it fits 64 lanes
    const int4 *seg;
    const float4 *seg_wc;
    int piece_len;               // 8, 12 or 16: no piece is longer (the smallest for which the pieces fit 64 lanes)
    int seg_ok;
    // per channel ch: lanes [x, x + y) hold pieces with index ch (their `hi` parts), lanes [z, z + w) pieces with index
    // ch + 1 (their `lo` parts); chan_ok = no channel needs more than four of either
    const int4 *chan_src;
    int chan_ok;
    const double *dct;           // [max(n_chan, 40)][32]: sqrt(2/C) cos(PI i (k-0.5)/C), zero past n_cep and past n_chan
    const double *lifter_w;      // [32]: 1 + L/2 sin(PI i / L)
};

int fail(jdsp_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess);

#define JDSP_HIP(ctx, call)                                             \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return ::jdsp::fail((ctx), JDSP_EHIP, #call, e_); \
    } while (0)

// stft_kernels.hip
int stft1024_table_count();
void fill_stft1024_table(float2 *host_table, int window_kind);
int launch_stft1024(hipStream_t stream, int n_cu, int fpw_opt, const short *pcm, long n_frames, long hop, float2 *spec,
                    const float2 *table, int read_pass = 0, int touch_wg_per_cu = 0);


int launch_stft1024_half(hipStream_t stream, const short *pcm, long n_frames, float2 *spec, long pitch,
                         const float2 *table);
int launch_stft512(hipStream_t stream, int n_cu, const short *pcm, long n_frames, long hop, float2 *spec,
                   const float2 *table, const float2 *win512);
void fill_win512(float2 *w, int window_kind);

// fft_c2c_kernels.hip
int launch_bitrev_table(hipStream_t stream, short *table_dev, int n_fft, int bits);
int launch_fft_process_f64(hipStream_t stream, const double2 *in, double2 *out, int n_fft, int log2n, long batch,
                           int forward, const double2 *tw);
void fill_c2c_twiddles(double2 *t, int n_fft);
int launch_stft1024_f64(hipStream_t stream, int n_cu, const short *pcm, long n_frames, long hop, const double *table,
                        const double2 *tw512, double2 *out, int variant, int fpw);
int launch_pcm_read_pass(hipStream_t stream, int n_cu, int wg_per_cu, const short *pcm, long n_samples);
void fill_stft1024_f64_table(double *t);          // [1024 window doubles][512 double2 split twiddles]
int launch_dft_direct_f64(hipStream_t stream, int kind, const void *in, double2 *inout, int n, long batch);


// denoise_kernels.hip
int launch_vad(hipStream_t s, const short *pcm, long n_blocks, const double *w_hi, int use_zcr, unsigned char *flags,
               long long *dbg_energy, int *dbg_zcr);
int launch_vad256(hipStream_t s, const short *pcm, long n_blocks, const double *w_hi, unsigned char *flags,
                  long long *dbg_energy, int *dbg_zcr, int use_zcr = 1);
int launch_noise_estimate512(hipStream_t s, const short *pcm, long n_blocks, const DenoiseState *st_in,
                             DenoiseState *st_out, const int *events, const int *ev_n, const DenoisePlan *plan,
                             const int *ver_base, const unsigned long long *snap_mask, const float2 *table,
                             const float *win512, const NoiseAccum &acc, float *noise_rows);
int launch_denoise512(hipStream_t s, int mode, int n_cu, const short *pcm, long n_blocks, long calls_before,
                      const DenoiseState *st_in, DenoiseState *st_out, const int *ver_base,
                      const unsigned long long *snap_mask, const float *noise_rows, const float2 *table,
                      const float *win512, short *out, float *precast, const DenoiseShard *shard = nullptr);
int launch_shard_summary512(hipStream_t s, const short *pcm_ext, long n_ext, long ext0, long b0, long b1,
                            const int *events, const int *ev_n, const DenoisePlan *plan, const int *ver_base,
                            const unsigned long long *snap_mask, const float2 *table, const float *win512, int *range,
                            const NoiseAccum &acc, float *rows, float *summary);
int launch_shard_rows512(hipStream_t s, const float *summaries_all, int rank, long b0, long b1, const DenoisePlan *plan,
                         const int *range, const NoiseAccum &acc, float *a_in, float *rows, float *last);
int launch_shard_row0_512(hipStream_t s, const float *last_all, int rank, float *rows);
int launch_run_plan(hipStream_t s, const unsigned char *flags, long n_blocks, const int *run_len_in, int *run_len_out,
                    int latch_run, int *ver_base, unsigned long long *snap_mask, int *events, int *ev_n,
                    DenoisePlan *plan);
int launch_denoise_plan(hipStream_t s, const unsigned char *flags, long n_blocks, const DenoiseState *st_in,
                        DenoiseState *st_out, int *ver_base, unsigned long long *snap_mask, int *events, int *ev_n,
                        DenoisePlan *plan);
int launch_noise_estimate(hipStream_t s, const short *pcm, long n_blocks, const DenoiseState *st_in,
                          DenoiseState *st_out, const int *events, const int *ev_n, const DenoisePlan *plan,
                          const int *ver_base, const unsigned long long *snap_mask, const float2 *table,
                          const NoiseAccum &acc, float *noise_rows);
int launch_denoise(hipStream_t s, int mode, int k_opt, int n_cu, const short *pcm, long n_blocks, long calls_before,
                   const DenoiseState *st_in, DenoiseState *st_out, const int *ver_base,
                   const unsigned long long *snap_mask, const float *noise_rows, const float2 *table, short *out,
                   float *precast, const DenoiseShard *shard = nullptr);
int launch_shard_summary(hipStream_t s, const short *pcm_ext, long n_ext, long ext0, long b0, long b1,
                         const int *events, const int *ev_n, const DenoisePlan *plan, const int *ver_base,
                         const unsigned long long *snap_mask, const float2 *table, int *range, const NoiseAccum &acc,
                         float *rows, float *summary);
int launch_shard_rows(hipStream_t s, const float *summaries_all, int rank, long b0, long b1, const DenoisePlan *plan,
                      const int *range, const NoiseAccum &acc, float *a_in, float *rows, float *last);
int launch_shard_row0(hipStream_t s, const float *last_all, int rank, float *rows);
int ensure_stft1024_table(jdsp_ctx *ctx);
int ensure_vad_window(jdsp_ctx *ctx);
// FP64 Hamming(2 block_len) over the positions a block occupies in the VAD's frame: keep + i, keep = block_len (SS:127-128)
// or block_len - 1 (BeamForming_MVDR_ver1.cpp:37,213-214); cached per context
int ensure_vad_window_ex(jdsp_ctx *ctx, int variant, int block_len, const double **w);
// fastconv_kernels.hip
// (fastconv) Sample `pos` of this call's stream (pos < 0: history carried in the handle).  Samples of
// the first n_hist blocks of a stream never reach the transform in the reference (its queue
// holds uninitialised malloc() blocks for them, :120): they are defined as zero.
struct ConvStream {
    const short *pcm;        // this call's samples
    const short *hist;       // last hist_len samples before this call
    long n_samples;          // in pcm
    long global0;            // global index of pcm[0]
    long valid_from;         // global index of the first sample that exists for the convolver
    int hist_len;
};

int launch_spectrum_to_f32(hipStream_t s, const double2 *in, float2 *out, long n, float scale);
int launch_fastconv(hipStream_t st, int n_fft, const ConvStream &s, long n_out_blocks, int first_block, int block,
                    int n_taps, int n_filters, const float2 *H, const float2 *table, const float2 *tw4096,
                    const float2 *tw8192, short *out, float *precast, long plane, short *hist_out);
void fill_conv_twiddles(float2 *tw4096, float2 *tw8192);
constexpr int kUpolsRowPitch = 520;      // = kUpolsPitch in fastconv_kernels.hip
int launch_spectrum_rows_to_f32(hipStream_t s, const double2 *in, float2 *out, long rows);
int launch_fastconv_upols(hipStream_t st, const ConvStream &s, long n_out_blocks, int first_block, int block, int n_part,
                          int n_filters, const float2 *Hp, const float2 *rect_table, short *staged, float2 *X,
                          short *out, float *precast, long plane, short *hist_out);
int ensure_stft1024_table_rect(jdsp_ctx *ctx);
// mvdr_kernels.hip
int launch_mvdr(hipStream_t s, const short *left, const short *right, long n_blocks, long calls_before,
                const MvdrState *st_in, MvdrState *st_out, const int *events, const DenoisePlan *plan,
                const int *ver_base, const unsigned long long *snap_mask, double *delta, double *rver,
                const double2 *steer, const float2 *table, short *out, float *precast, float4 *wtab,
                double *tile_sums);
int launch_mvdr_corr_total(hipStream_t s, const short *left, const short *right, long n_blocks, const MvdrState *st_in,
                           const int *events, const DenoisePlan *plan, const float2 *table, double *delta, double *total,
                           double *tile_sums);
int launch_mvdr_apply(hipStream_t s, const short *left, const short *right, long n_blocks, long calls_before,
                      const MvdrState *st_in, MvdrState *st_out, const int *ver_base, const unsigned long long *snap_mask,
                      const double *rver, const double2 *steer, const float2 *table, short *out, float *precast);
int launch_mvdr_shard_summary(hipStream_t s, const short *left_ext, const short *right_ext, long n_ext, long ext0,
                              long b0, long b1, const MvdrState *zero_state, const int *events,
                              const DenoisePlan *plan, const int *ver_base, const unsigned long long *snap_mask,
                              const float2 *table, int *range, double *delta, double *total, double *tile_sums);
int launch_mvdr_shard_finish(hipStream_t s, const short *left_ext, const short *right_ext, long n_ext, long ext0,
                             long b0, long b1, const MvdrState *zero_state, MvdrState *scratch_state,
                             const DenoisePlan *plan, const int *ver_base, const unsigned long long *snap_mask,
                             const int *range, const double *delta, const double *sums_all, int rank, double *rver,
                             const double2 *steer, const float2 *table, short *out, float *precast, double *tile_sums);
// mvdrn_kernels.hip
int launch_mvdrn(hipStream_t s, const short *pcm, long chan_stride, int n_mics, long n_blocks, long calls_before,
                 const short *prev_in, short *prev_out, const int *events, const DenoisePlan *plan, const int *ver_base,
                 const unsigned long long *snap_mask, float2 *spec, const double2 *cov_in, double2 *cov_out,
                 const double2 *steer, double loading, float2 *weights, const float2 *table, short *out, float *precast, double2 *chunk_ws, int chunk_cap);
int launch_mvdrn512(hipStream_t s, const short *pcm, long chan_stride, int n_mics, long n_blocks, long calls_before,
                    const short *prev_in, short *prev_out, const int *events, const DenoisePlan *plan, const int *ver_base,
                    const unsigned long long *snap_mask, float2 *spec, const double2 *cov_in, double2 *cov_out,
                    const double2 *steer, double loading, float2 *weights, const float2 *table, short *out, float *precast, double2 *chunk_ws, int chunk_cap);
// pitch_kernels.hip
int launch_pitch(hipStream_t s, const short *pcm, long n_blocks, const short *prev_block, const float2 *table, int *arg,
                 float *rmax, float *autocorr);
// mfcc_kernels.hip
// ---- GMM / HMM (gmm_kernels.hip) ----
// packed per-GMM record (doubles): alpa[4], mean[4][4], var[4][4], coef[4][4], eig[4][12][4], and for the
// fused evaluation nhiv[4][4] = -0.5 / var and cprod[4] = the product of a mixture's four coef
constexpr int kGmmAlpa = 0, kGmmMean = 4, kGmmVar = 20, kGmmCoef = 36, kGmmEig = 52, kGmmNhiv = 244, kGmmCprod = 260,
              kGmmRecord = 264;
constexpr int kGmmMaxClasses = 256;
int launch_gmm_score(hipStream_t stream, const double *feats, long n_frames, const long long *utt_first, long n_utts,
                     const double *gmm, int n_classes, int fused, double *scores, int *best);
int launch_hmm_viterbi(hipStream_t stream, const double *feats, long n_frames, const long long *utt_first, long n_utts,
                       const double *gmm, const double *log_trans, int n_models, int fused, double log_init, double *b,
                       double *scores, int *best, int *path, double *trellis);
int launch_mfcc(hipStream_t s, const short *pcm, const long long *starts, long n_frames, const MfccDev &p,
                const float2 *table, double *feats, int *redo);

}  // namespace jdsp

struct jdsp_denoise {
    jdsp_ctx *ctx = nullptr;
    int mode = 0;
    long calls = 0;                       // blocks consumed so far (the reference's call counters)
    jdsp::DenoiseState *st[2] = {nullptr, nullptr};
    int cur = 0;                          // st[cur] is the state the next call reads
    double *w_hi = nullptr;               // second half of the FP64 Hamming window (VAD)
    long cap_blocks = 0;                  // workspace capacity (plan arrays)
    unsigned char *flags = nullptr;
    int *ev_n = nullptr, *ver_base = nullptr, *events = nullptr;
    unsigned long long *snap_mask = nullptr;
    long long *dbg_energy = nullptr;
    int *dbg_zcr = nullptr;
    jdsp::DenoisePlan *plan = nullptr;
    float *rows = nullptr;                // [cap_rows][1024] latched estimates of the call (row 0: the one carried in)
    long cap_rows = 0;                    // rows of `rows` / entries of acc.lat_*
    jdsp::NoiseAccum acc = {nullptr, nullptr, nullptr, nullptr, nullptr};
    long last_blocks = 0;
    int opt_k = 0;
    int opt_vad_trace = 0;                // 1: keep every block's energy sum and ZCR for jdsp_denoise_vad_trace (slower VAD kernel)
    int last_trace_valid = 0;             // the option's value when the last call ran: what jdsp_denoise_vad_trace may hand out
    int n_fft = 1024, block = 512;        // FFT_PROCESSING_SIZE, BLOCK_LEN = KEEP_LEN (SS:53-55); 512 / 256 also built
    double *w_hi256 = nullptr;            // 512-point frames: second half of the FP64 Hamming(512) (VAD)
    float *win512h = nullptr;             // 512-point frames: 0.5 * Hamming(512), natural order
    // sharded (multi-GPU) run in progress: jdsp_denoise_shard_*
    long sh_ext0 = 0, sh_b0 = 0, sh_b1 = 0, sh_total = 0;
    const int16_t *sh_pcm = nullptr;
    int *sh_range = nullptr;              // device: {first event, one past last event, latches before the shard}
    float *sh_a_in = nullptr;             // device: [1024]
    int *sh_zero_run = nullptr;           // device: a zero (run length entering a fresh global stream)
};

struct jdsp_mfcc {
    jdsp_ctx *ctx = nullptr;
    jdsp_mfcc_cfg cfg;
    jdsp::MfccDev dev;
    void *blob = nullptr;                 // one device allocation holding every table
    int *redo = nullptr;                  // 512-FFT configurations: {count, frame pairs to recompute apart} (mfcc512_run_kernel)
    long redo_cap = 0;                    // pairs it holds
    std::vector<double> mel_freqs, fbank;
    std::vector<int> fi_bins;
    // FP64 tables of the separately callable sub-steps (stage_api.hip), built on first use
    void *stage_blob = nullptr;
    const double *stage_fb = nullptr, *stage_cos = nullptr, *stage_lift = nullptr;
    const int *stage_fi = nullptr;
};

struct jdsp_gmm {
    jdsp_ctx *ctx = nullptr;
    int n_classes = 0;
    int fused = 0;                        // "evaluation" option: 0 the reference's operation order, 1 fused
    double *records = nullptr;            // [n_classes][kGmmRecord]
};

struct jdsp_hmm {
    jdsp_ctx *ctx = nullptr;
    int n_models = 0;
    int fused = 0;                        // "evaluation" option, as jdsp_gmm
    double *records = nullptr;            // [n_models * 6][kGmmRecord]
    double *log_trans = nullptr;          // [n_models][6][6], log() taken on the host
    double *emission = nullptr;           // scratch [frames][n_models * 6], grown on demand
    long emission_frames = 0;
};

struct jdsp_fastconv {
    jdsp_ctx *ctx = nullptr;
    int n_fft = 0, n_taps = 0, n_filters = 0, block = 0, n_hist = 0;
    float2 *H = nullptr;                  // [n_filters][n_fft]
    // uniformly partitioned path (n_fft 8192, block % 512 == 0): see fastconv_kernels.hip
    int n_part = 0;                       // 0: not used
    float2 *Hp = nullptr;                 // [n_filters][n_part][520]: bins 0..512 of every 512-tap partition
    short *staged = nullptr;              // [512 n_part + samples of a call]
    float2 *X = nullptr;                  // [frames][520]
    long ws_samples = 0;                  // samples per call the workspace is sized for
    short *hist[2] = {nullptr, nullptr};  // last n_taps-1 samples of the stream, ping-pong
    int cur = 0;
    long calls = 0;                       // blocks consumed so far (siNumOfCount)
};

struct jdsp_mvdr {
    jdsp_ctx *ctx = nullptr;
    double d_time = 0;
    long calls = 0;
    jdsp::MvdrState *st[2] = {nullptr, nullptr};
    int cur = 0;
    jdsp::DenoisePlan *plan = nullptr;
    double2 *steer = nullptr;             // [1024] steering vector's second component per bin
    double *w_vad = nullptr;              // Hamming[511 .. 1022] in FP64
    long cap_blocks = 0;
    unsigned char *flags = nullptr;
    int *events = nullptr, *ev_n = nullptr, *ver_base = nullptr;
    unsigned long long *snap_mask = nullptr;
    double *delta = nullptr, *rver = nullptr;
    double *tile_sums = nullptr;          // [cap_blocks / 1024 + 1][4] sums of the prefix pass's tiles of 1024 events
    float4 *wtab = nullptr;               // [min(cap_blocks + 1, kMvdrTableVersions)][1024] per-version weights (mvdr_weights_kernel)
    // sharded (multi-GPU) run in progress
    long sh_ext0 = 0, sh_b0 = 0, sh_b1 = 0, sh_total = 0;
    const int16_t *sh_left = nullptr, *sh_right = nullptr;
    int *sh_range = nullptr;              // device: {first event, one past last, versions before the shard}
    int *sh_zero_run = nullptr;
};

struct jdsp_mvdrn {
    jdsp_ctx *ctx = nullptr;
    int n_mics = 0;
    int n_fft = 1024, block = 512, n_bins = 513;   // FFT_PROCESSING_LEN, BLOCK_LEN, bins kept (1024 / 512 / 513 or 512 / 256 / 257)
    double loading = 0;
    long calls = 0;
    int cur = 0;
    double2 *cov[2] = {nullptr, nullptr};     // [513][64] per-bin covariance, ping-pong
    short *prev[2] = {nullptr, nullptr};      // [8][512] previous block per microphone
    int *run_len[2] = {nullptr, nullptr};
    jdsp::DenoisePlan *plan = nullptr;
    double2 *steer = nullptr;                 // [513][8]
    double *w_vad = nullptr;
    long cap_blocks = 0;
    unsigned char *flags = nullptr;
    int *events = nullptr, *ev_n = nullptr, *ver_base = nullptr;
    unsigned long long *snap_mask = nullptr;
    float2 *spec = nullptr, *weights = nullptr;
    double2 *chunk_ws = nullptr;              // [2][chunk_cap][n_bins][64]: per-chunk covariance sums and entering matrices (sized with the workspace)
    int chunk_cap = 0;
};
