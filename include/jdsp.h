/*
 * jdsp.h -- C ABI of the MI355X (gfx950) engine behind JeicybooDSP's FFT-based
 * spectral path.  Plain C, plain pointers and sizes; no C++/torch types.
 *
 * The reference (phoenix163/JeicybooDSP) has no plugin/FFI layer: its seam is
 * the per-block free function each program's main() calls.  Every entry point
 * below names the reference function(s) (file:line) whose work it performs for
 * a whole batch of blocks at once; jeicyboodsp_amd/compat/ keeps the
 * reference's own per-block C++ signatures on top of this ABI, and
 * INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success and a negative JDSP_E* code on
 *     failure, never throws; jdsp_last_error() gives the text for the handle.
 *   - a jdsp_ctx is confined to one host thread / one stream of audio at a
 *     time (the reference's functions are non-re-entrant: static state in
 *     every per-block function, e.g. SpectralSubtraction_final.cpp:202,208-209;
 *     here that state lives in the handle).
 *   - "_dev" entry points take DEVICE pointers and only enqueue work on the
 *     handle's HIP stream: no allocation and no synchronisation once the
 *     handle is warm, i.e. after one call of that entry point (constant tables
 *     are built on first use) and after the matching *_reserve() where the
 *     entry has a workspace (jdsp_denoise_reserve, jdsp_hmm_reserve).  A warm
 *     "_dev" call can be captured into a hipGraph (bench.py does that).
 *     The plain entry points take HOST pointers, copy in, run, copy out and
 *     synchronise.
 *   - spectra are interleaved (re, im) float pairs, full length n_fft per frame
 *     exactly like the reference's fftw_complex[FFT_PROCESSING_SIZE] buffers
 *     (SpectralSubtraction_final.cpp:205-206), in single precision.
 */
#ifndef JDSP_H
#define JDSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: jdsp_vad_blocks_ex added; jdsp_denoise_apply and jdsp_denoise_shard_* accept 512-point streams;
 *    jdsp_denoise_vad_trace's energies / counts follow the option's value at the time of the traced call;
 *    jdsp_set_option("stft.read_pass") accepts -1 / 0 / 1 only.  (1: rounds 1-2.) */
#define JDSP_ABI_VERSION 2

enum {
    JDSP_OK = 0,
    JDSP_EINVAL = -1,   /* bad argument / unsupported size */
    JDSP_EHIP = -2,     /* a HIP runtime call failed */
    JDSP_ENOMEM = -3,
    JDSP_ENODEV = -4    /* no gfx950 device / device ordinal out of range */
};

typedef struct jdsp_ctx jdsp_ctx;
typedef struct { float re, im; } jdsp_c32;

/* ---- handle ---------------------------------------------------------------- */
int jdsp_abi_version(void);
/* Creates a handle on HIP device `device`.  Fails with JDSP_ENODEV when there
 * is no GPU: there is no CPU fallback in this library. */
int jdsp_create(int device, jdsp_ctx **out);
int jdsp_destroy(jdsp_ctx *ctx);
const char *jdsp_last_error(const jdsp_ctx *ctx);   /* ctx may be NULL: global creation error */
/* Enqueue on a caller-owned hipStream_t (e.g. the caller framework's current
 * stream).  The value is used as given: NULL is HIP's default (null) stream.
 * jdsp_use_own_stream() goes back to the handle's private non-blocking stream.
 * When the stream actually changes, everything the handle (and its stream objects:
 * create/reset memsets, state carried between process calls) has enqueued on the
 * old stream is ordered before what follows on the new one (event record + wait;
 * no host synchronisation).  Exception: if either stream is being captured into a
 * hipGraph no event is inserted and that ordering is the caller's. */
int jdsp_set_stream(jdsp_ctx *ctx, void *hip_stream);
int jdsp_use_own_stream(jdsp_ctx *ctx);
/* Tuning knobs, by name; unknown names return JDSP_EINVAL.
 *   "stft.frames_per_wave"  consecutive frames one wavefront owns (0 = auto)
 *   "stft.read_pass"        -1 (default, auto) / 0 / 1: before the hop-512 transform of a large batch, one
 *                           read-only launch streams the PCM into the 256 MiB Infinity Cache, slab by slab, so
 *                           that HBM sees a read stream and then a write stream instead of their 1 : 8 mix
 *                           (65,536 frames whose PCM is in HBM: 98 us against 147 us; PCM that is already
 *                           cache-resident pays 9 us for nothing -- 0 turns the pass off).  Results are identical.
 *                           The FP64 analysis (jdsp_stft_i16_f64*) takes the same pass.
 *   "stft.f64_kernel"       0 (default): the FP64 analysis at four waves per SIMD (twiddles as powers of one value
 *                           per family); 1: round 2's kernel (all tables in registers, two waves per SIMD) -- A/B
 *   "stft.f64_frames_per_wave"  frames one wave of the FP64 analysis walks (0 = one round of resident waves)
 *   "stft.window"           window of jdsp_stft_*: 0 = the reference's Hamming
 *                           0.54-0.46cos(2*3.141592*i/(n-1)) (default), 1 = Hann 0.5-0.5cos(same).
 *                           The denoise / MFCC / pitch chains always use what the reference uses. */
int jdsp_set_option(jdsp_ctx *ctx, const char *name, long value);
int jdsp_synchronize(jdsp_ctx *ctx);
/* Device properties the host side sizes launches with. */
int jdsp_device_info(jdsp_ctx *ctx, int *n_cu, size_t *hbm_bytes, char *name, size_t name_len);

/* Device memory helpers for callers without their own allocator. */
int jdsp_malloc(jdsp_ctx *ctx, size_t bytes, void **dev_ptr);
int jdsp_free(jdsp_ctx *ctx, void *dev_ptr);
int jdsp_memcpy_h2d(jdsp_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int jdsp_memcpy_d2h(jdsp_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
/* Page-locked host memory.  Host-pointer entry points given PINNED buffers (these, or any
 * hipHostMalloc/hipHostRegister memory) overlap the PCIe copies with the kernels: jdsp_stft_i16
 * then streams the batch through in chunks on three HIP streams (copy in | transform | copy out). */
int jdsp_host_alloc(jdsp_ctx *ctx, size_t bytes, void **host_ptr);
int jdsp_host_free(jdsp_ctx *ctx, void *host_ptr);

/* ---- FFTAlgorithm_ver2.cpp -------------------------------------------------- */
/* Bitrev table (FFTAlgorithm_ver2.cpp:186-202), computed on the device with the
 * reference's 16-bit shift/or loop; bit count from block_len (:188), mask with
 * n_fft-1 (:202).  table_host: n_fft int16.  Bit-exact. */
int jdsp_bitrev_table(jdsp_ctx *ctx, int n_fft, int block_len, int16_t *table_host);
/* Batched FFTProcess (FFTAlgorithm_ver2.cpp:94-149): `batch` independent
 * n_fft-point complex transforms, interleaved double (COMPLEX, :20-22) in and
 * out, forward != 0 -> exp(-j..), unnormalised both ways.  Host pointers. */
int jdsp_fft_process_f64(jdsp_ctx *ctx, const double *in_host, double *out_host,
                         int n_fft, long batch, int forward);
int jdsp_fft_process_f64_dev(jdsp_ctx *ctx, const double *in_dev, double *out_dev,
                             int n_fft, long batch, int forward);

/* DFTProcess (FFTAlgorithm_ver2.cpp:162-173), IDFTProcess (:175-184) and IFFTProcess (:151-160): the
 * reference's definition-level O(N^2) transforms, evaluated on the device as the reference writes them
 * (angle ((2*PI)*i)*k/N with its PI 3.14159265358, terms added in i order, nothing fused) and, like the
 * reference, ACCUMULATED into the output the caller passes in (a caller that wants the plain transform
 * zeroes it first, as the reference's commented-out call sites :74,:76 would have to).  Any n >= 1 --
 * these are not restricted to powers of two.  `batch` independent transforms of length n, back to back.
 * in: int16[n] per transform for JDSP_DFT_I16, interleaved double (COMPLEX) otherwise. */
enum { JDSP_DFT_I16 = 0, JDSP_IDFT = 1, JDSP_IDFT_OVER_N = 2 };
int jdsp_dft_direct_f64_dev(jdsp_ctx *ctx, int kind, const void *in_dev, double *inout_dev, int n, long batch);
int jdsp_dft_direct_f64(jdsp_ctx *ctx, int kind, const void *in_host, double *inout_host, int n, long batch);

/* ---- STFT analysis: framing + Hamming + forward transform -------------------- */
/* Replaces, for n_frames frames at once, SpectralSubtraction_final.cpp:218-230
 * (== WienerFilter_final.cpp:181-193, noise path SS:168-180): frame f =
 * pcm[hop*f .. hop*f+n_fft) * (0.54-0.46cos(2*3.141592*i/(n_fft-1))) -> unnormalised
 * forward DFT, all n_fft bins.  pcm must hold hop*(n_frames-1)+n_fft samples.
 * Supported: n_fft = 1024 with hop = 512 (reference-native, the headline configuration: pcm
 * 16-byte aligned takes the fast kernel), n_fft = 1024 with any hop >= 1, and n_fft = 512 with
 * any hop (BASELINE config 3 as written; computed as the even bins of the zero-padded
 * 1024-point transform). */
int jdsp_stft_i16_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames,
                      int n_fft, int hop, jdsp_c32 *spec_dev);
/* Half-spectrum variant (n_fft 1024, hop 512, Hamming): bins 0..512 only, row f at
 * spec_dev + f * row_pitch complex64 (row_pitch >= 513; 513 = dense rows of 4,104 B instead of
 * 8,192).  The reference keeps all 1024 bins, so this is an extra, measured separately
 * (5,128 algorithmic bytes per frame). */
int jdsp_stft_half_i16_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames, jdsp_c32 *spec_dev, long row_pitch);
int jdsp_stft_i16(jdsp_ctx *ctx, const int16_t *pcm_host, long n_samples,
                  int n_fft, int hop, jdsp_c32 *spec_host, long *n_frames_out);
/* The same analysis in the reference's own precision (SS:218-230 computes in double): FP64 window, FP64 transform,
 * spec = n_frames x 1024 interleaved (re, im) doubles (16,384 B per frame).  n_fft = 1024, any hop >= 1.  Agrees with
 * the reference's FFTProcess on the same windowed frames to ~1e-11 of the frame peak (FFTProcess's truncated PI,
 * FFT:15); the FP32 entries above agree with this one to ~5e-7. */
int jdsp_stft_i16_f64_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames, int n_fft, int hop, double *spec_dev);
int jdsp_stft_i16_f64(jdsp_ctx *ctx, const int16_t *pcm_host, long n_samples, int n_fft, int hop, double *spec_host,
                      long *n_frames_out);

/* ---- spectral subtraction / Wiener filter ------------------------------------- */
/* One jdsp_denoise object holds everything the reference keeps in static locals
 * for one audio stream: main()'s run-length counter and noise estimate
 * (SpectralSubtraction_final.cpp:70-72), EstimateNoiseSpectrum's running
 * average (:161), the keep buffers (:164,:208), the overlap buffer (:209) and
 * the call counters (:202).  Feeding a stream in batches of any size -- down to
 * the reference's one 512-sample block per call -- yields the same output.
 *
 * jdsp_denoise_process* replaces, for n_blocks blocks, one iteration each of
 * main()'s loop: VoiceActivityDetection (SS:121-156 / WF:261-296),
 * EstimateNoiseSpectrum (SS:159-198 / WF:120-159) and SpectralSubtraction
 * (SS:201-264) or WienerFiltering (WF:162-235).  Like the reference, the first
 * two blocks of a stream produce no output (SS:260-263): *n_out_blocks =
 * jdsp_denoise_blocks_out().  out gets n_out blocks of 512 int16; precast (may
 * be NULL) gets the same samples before the (short) cast, as float. */
enum { JDSP_SPECSUB = 0, JDSP_WIENER = 1 };
typedef struct jdsp_denoise jdsp_denoise;
int jdsp_denoise_create(jdsp_ctx *ctx, int mode, jdsp_denoise **out);
/* The same object with the reference's macros as parameters: FFT_PROCESSING_SIZE = n_fft, BLOCK_LEN =
 * KEEP_LEN = hop (SS:53-55 / WF:42-44 are 1024 / 512 / 512 = jdsp_denoise_create; BASELINE config 3 words the workload
 * "512-pt STFT 50 % hop" = (512, 256)).  Blocks are then hop samples long everywhere below, the noise estimate has
 * n_fft entries, and the thresholds stay the reference's (energy 700, ZCR 200, latch at run length 10).  Supported:
 * (1024, 512) and (512, 256), on every entry below (batched, per-block jdsp_denoise_apply, sharded). */
int jdsp_denoise_create_cfg(jdsp_ctx *ctx, int mode, int n_fft, int hop, jdsp_denoise **out);
int jdsp_denoise_block_len(const jdsp_denoise *h);
int jdsp_denoise_destroy(jdsp_denoise *h);
int jdsp_denoise_reset(jdsp_denoise *h);                       /* back to a fresh stream */
/* "blocks_per_wave": 0 (default: one round of resident waves walks the batch) or 1 / 2 / 4 / 8 (the compile-time
 * variants, kept for A/B timing); "vad_trace": 1 keeps every block's energy sum and zero-crossing count for
 * jdsp_denoise_vad_trace (a slower VAD kernel: two exact wave reductions per block); 0 (default) keeps the flags only. */
int jdsp_denoise_set_option(jdsp_denoise *h, const char *name, long value);
long jdsp_denoise_blocks_out(const jdsp_denoise *h, long n_blocks);
/* Sizes the device workspace for batches of up to max_blocks (the only call
 * that allocates; process() calls it on demand). */
int jdsp_denoise_reserve(jdsp_denoise *h, long max_blocks);
int jdsp_denoise_process_dev(jdsp_denoise *h, const int16_t *pcm_dev, long n_blocks, int16_t *out_dev,
                             float *precast_dev, long *n_out_blocks);
int jdsp_denoise_process(jdsp_denoise *h, const int16_t *pcm_host, long n_blocks, int16_t *out_host,
                         float *precast_host, long *n_out_blocks);
/* Current rgdEstimatedNS (n_fft doubles, host).  Synchronises. */
int jdsp_denoise_noise(jdsp_denoise *h, double *noise_host);
/* VoiceActivityDetection results of the first n blocks of the last process call:
 * voice flag, sum of squared truncated samples (dEnergy*1024, SS:135) and dZCR (SS:140).
 * Any pointer may be NULL.  The flags are always there; energies and counts only when the "vad_trace" option was
 * set WHEN THAT CALL RAN (JDSP_EINVAL otherwise -- since ABI version 2; setting it afterwards does not make a
 * trace appear).  Synchronises. */
int jdsp_denoise_vad_trace(jdsp_denoise *h, long n, uint8_t *voice_host, int64_t *energy_sum_host,
                           int32_t *zcr_host);

/* Sharded denoise: one rank's share of ONE global stream of n_total blocks (multi-GPU,
 * SURVEY §8e).  The rank owns global blocks [b0, b1); pcm_ext_dev holds global blocks
 * [ext0, b1) with ext0 = max(b0 - 2, 0) (two halo blocks rebuild the overlap tail); blocks are
 * jdsp_denoise_block_len() samples (512, or 256 on 512-point frames: the exchanged buffers keep
 * their 1025-float size, entries past n_fft are unused).  Between
 * the steps the caller all-gathers three small buffers across ranks (any transport; RCCL in
 * jeicyboodsp_amd/sharding.py):
 *   1. shard_vad      -> flags_own (b1-b0 bytes)          all-gather -> flags_all (n_total bytes)
 *   2. shard_summary  -> summary (1025 floats: the affine map A <- a*A + b of this rank's
 *                         noise frames, SS:182-187)        all-gather -> summaries_all (world*1025)
 *   3. shard_rows     -> last (1025 floats: latch count and last latched estimate, SS:189-193)
 *                                                          all-gather -> last_all (world*1025)
 *   4. shard_finish   -> out: the emitted blocks among [max(b0,2), b1), shard_blocks_out() of them
 * Concatenating the ranks' outputs gives the single-GPU stream (int16 within +-1 LSB). */
int jdsp_denoise_shard_vad_dev(jdsp_denoise *h, const int16_t *pcm_ext_dev, long ext0, long b0, long b1, long n_total,
                               uint8_t *flags_own_dev);
int jdsp_denoise_shard_summary_dev(jdsp_denoise *h, const uint8_t *flags_all_dev, float *summary_dev);
int jdsp_denoise_shard_rows_dev(jdsp_denoise *h, const float *summaries_all_dev, int world, int rank, float *last_dev);
long jdsp_denoise_shard_blocks_out(const jdsp_denoise *h);
int jdsp_denoise_shard_finish_dev(jdsp_denoise *h, const float *last_all_dev, int world, int rank, int16_t *out_dev,
                                  float *precast_dev, long *n_out_blocks);

/* The reference's two helper functions on their own, for callers that keep main()'s
 * structure (jeicyboodsp_amd/compat): */
/* VoiceActivityDetection (SS:121-156) for n_blocks blocks of 512 host samples; outputs as
 * jdsp_denoise_vad_trace.  Any output pointer may be NULL. */
int jdsp_vad_blocks(jdsp_ctx *ctx, const int16_t *pcm_host, long n_blocks, uint8_t *voice_host,
                    int64_t *energy_sum_host, int32_t *zcr_host);
/* The same function at the other shapes it exists in.  variant JDSP_VAD_DENOISE: SS:121-156 = WF:261-296, frame
 * [zeros(block_len), block], voice <=> energy > 700 or ZCR < 200.  JDSP_VAD_MVDR: BeamForming_MVDR_ver1.cpp:207-242,
 * frame [zeros(block_len - 1), block, 0] (KEEP_LEN 511, :37), voice <=> energy > 700 (:233; the zero-crossing count
 * is still computed and returned, as the reference prints it).  block_len 512 (the reference's BLOCK_LEN) or 256
 * (FFT_PROCESSING_SIZE 512: BASELINE configs 3 and 5 as worded).  energy_sum = the integer sum of squares; the
 * reference's dEnergy is energy_sum / (2 block_len). */
#define JDSP_VAD_DENOISE 0
#define JDSP_VAD_MVDR 1
int jdsp_vad_blocks_ex(jdsp_ctx *ctx, int variant, int block_len, const int16_t *pcm_host, long n_blocks,
                       uint8_t *voice_host, int64_t *energy_sum_host, int32_t *zcr_host);
/* SpectralSubtraction / WienerFiltering (SS:201-264 / WF:162-235) with the CALLER's
 * pdEstimatedNoiseSpec (n_fft doubles, host) instead of the handle's own VAD + estimate: only
 * the keep buffer, the overlap buffer and the call counter of the handle are used. */
int jdsp_denoise_apply(jdsp_denoise *h, const int16_t *pcm_host, long n_blocks, const double *noise_host,
                       int16_t *out_host, float *precast_host, long *n_out_blocks);

/* ---- overlap-save fast convolution --------------------------------------------- */
/* Fast_Convolution_Based_3DAudio_Impl.cpp.  jdsp_fastconv_create replaces main()'s filter
 * set-up (:82-84) and the per-block FFT of the filter (:140,:143): taps is n_filters rows of
 * n_taps doubles (the reference: one row, rgdFirLPF_coefficients[7169] of FilterCoefficient.h);
 * n_fft is 8192 (reference-native) or 1024; block = n_fft - n_taps + 1 samples per call
 * (1024 native).  jdsp_fastconv_process* replaces AnalySisFreqDomain (:102-177) for n_blocks
 * blocks: like the reference, the first hist_blocks = ceil((n_taps-1)/block) blocks of a
 * stream produce no output (:119-123) and never reach the transform (the reference queues
 * uninitialised buffers for them, :120 -- defined as silence here).  out: n_filters planes of
 * n_out*block int16, plane-major; precast (may be NULL): the same values before the (short)
 * cast.  The handle carries the last n_taps-1 samples between calls. */
typedef struct jdsp_fastconv jdsp_fastconv;
int jdsp_fastconv_create(jdsp_ctx *ctx, const double *taps, int n_taps, int n_filters, int n_fft,
                         jdsp_fastconv **out);
int jdsp_fastconv_destroy(jdsp_fastconv *h);
int jdsp_fastconv_reset(jdsp_fastconv *h);
/* Multi-GPU / random access: pretend `blocks_consumed` blocks of the stream have already been
 * processed (history = silence).  A rank that owns output blocks [e0, e1) of a stream feeds
 * input blocks [e0 + hist_blocks - hist_blocks', ...): see sharding.fastconv_sharded -- it
 * prepends hist_blocks halo blocks and drops the outputs they produce. */
int jdsp_fastconv_set_position(jdsp_fastconv *h, long blocks_consumed);
int jdsp_fastconv_block_len(const jdsp_fastconv *h);
int jdsp_fastconv_hist_blocks(const jdsp_fastconv *h);
long jdsp_fastconv_blocks_out(const jdsp_fastconv *h, long n_blocks);
/* Sizes the workspace for calls of up to n_blocks blocks ahead of time (the 8192-point configuration with a
 * block length that is a multiple of 512 runs as a partitioned convolution with a spectrum workspace). */
int jdsp_fastconv_reserve(jdsp_fastconv *h, long n_blocks);
int jdsp_fastconv_process_dev(jdsp_fastconv *h, const int16_t *pcm_dev, long n_blocks, int16_t *out_dev,
                              float *precast_dev, long *n_out_blocks);
int jdsp_fastconv_process(jdsp_fastconv *h, const int16_t *pcm_host, long n_blocks, int16_t *out_host,
                          float *precast_host, long *n_out_blocks);

/* ---- FFT autocorrelation pitch --------------------------------------------------- */
/* PitchEstimation_method1.cpp: CalcPitch (:69-116) for n_blocks blocks of 512 samples.
 * Frame b = [block b-1, block b] with no window; block -1 is prev_block (NULL: zeros, the
 * reference's initial keep buffer, :74).  arg[b] = the lag the reference prints as
 * "Estimation arg" (:109; pitch = 16000/arg), rmax[b] = the autocorrelation there;
 * autocorr (may be NULL) = r[0..511] per block (:95-97).  pcm must be 16-byte aligned.
 * AnalysisAdditiveWhiteGaussianNoise.cpp:98-133 (the analysis half of that program) is this same chain
 * without the arg-max: its dAutoCorrelation (:122-124) is the autocorr output here. */
int jdsp_pitch_autocorr_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_blocks, const int16_t *prev_block_dev,
                            int32_t *arg_dev, float *rmax_dev, float *autocorr_dev);
int jdsp_pitch_autocorr(jdsp_ctx *ctx, const int16_t *pcm_host, long n_blocks, const int16_t *prev_block_host,
                        int32_t *arg_host, float *rmax_host, float *autocorr_host);

/* ---- two-microphone MVDR beamformer ----------------------------------------------- */
/* BeamForming_MVDR_ver1.cpp.  One jdsp_mvdr object = the statics of one stereo stream
 * (rgdSpatialCorr :57, the run counter :59, the keep/temp buffers :56,:130-131, iNumOfCount
 * :137).  d_time is ProcessMVDR's dTime (:124; the reference's main() passes
 * (DISTANCE_OF_MIC/SPEED_OF_SOUND)*sin(0) = 0, :60,:121).  jdsp_mvdr_process* replaces, for
 * n_blocks blocks of 512 samples per channel, one iteration each of main()'s loop (:169-231):
 * VoiceActivityDetection on the left channel (:207-242), EstimateSpatialCorrMtx (:244-270) and
 * ProcessMVDR (:124-205).  The first block of a stream produces no output (:201-204); until the
 * correlation matrix is invertible the reference's samples are NaN, cast to 0 here.
 * jdsp_mvdr_corr: current rgdSpatialCorr, row-major (4 doubles, host; synchronises). */
typedef struct jdsp_mvdr jdsp_mvdr;
int jdsp_mvdr_create(jdsp_ctx *ctx, double d_time, jdsp_mvdr **out);
int jdsp_mvdr_destroy(jdsp_mvdr *h);
int jdsp_mvdr_reset(jdsp_mvdr *h);
long jdsp_mvdr_blocks_out(const jdsp_mvdr *h, long n_blocks);
int jdsp_mvdr_process_dev(jdsp_mvdr *h, const int16_t *left_dev, const int16_t *right_dev, long n_blocks,
                          int16_t *out_dev, float *precast_dev, long *n_out_blocks);
int jdsp_mvdr_process(jdsp_mvdr *h, const int16_t *left_host, const int16_t *right_host, long n_blocks,
                      int16_t *out_host, float *precast_host, long *n_out_blocks);
int jdsp_mvdr_corr(jdsp_mvdr *h, double *corr4_host);
/* The program's two helper functions on their own, for callers that keep main()'s structure -- or do not follow
 * it (jeicyboodsp_amd/compat: EstimateSpatialCorrMtx, ProcessMVDR with the reference's signatures):
 *   jdsp_mvdr_estimate_corr  EstimateSpatialCorrMtx (:244-270) for n_frames frames of 1024 samples per channel
 *                            ([previous block, block], rgsTempBufferL/R), every frame's contribution ADDED to the
 *                            caller's rgdSpatialCorr (4 doubles, row-major).  Touches no stream state.
 *   jdsp_mvdr_apply          ProcessMVDR (:124-205) for n_blocks blocks with the CALLER's rgdSpatialCorr instead of
 *                            the handle's own VAD + accumulation: only the two keep buffers (:130-131) and the call
 *                            counter (:137) of the handle are used and advanced. */
int jdsp_mvdr_estimate_corr(jdsp_mvdr *h, const int16_t *left_frames_host, const int16_t *right_frames_host, long n_frames,
                            double *corr4_inout_host);
int jdsp_mvdr_apply(jdsp_mvdr *h, const int16_t *left_host, const int16_t *right_host, long n_blocks,
                    const double *corr4_host, int16_t *out_host, float *precast_host, long *n_out_blocks);
/* Sharded MVDR (multi-GPU): the rank owns global blocks [b0, b1) of a stream of n_total
 * blocks; the ext buffers hold global blocks [ext0, b1), ext0 = max(b0 - 1, 0).  shard_vad ->
 * flags_own; all-gather -> flags_all; shard_summary -> sum4 (this rank's contribution to
 * rgdSpatialCorr); all-gather -> sums_all (world*4 doubles); shard_finish -> the emitted blocks
 * among [max(b0,1), b1).  rgdSpatialCorr is a running SUM (:263-268), so the matrix entering a
 * rank is the sum of the earlier ranks' contributions: the reduction SURVEY §8e asks for. */
int jdsp_mvdr_shard_vad_dev(jdsp_mvdr *h, const int16_t *left_ext_dev, const int16_t *right_ext_dev, long ext0, long b0,
                            long b1, long n_total, uint8_t *flags_own_dev);
int jdsp_mvdr_shard_summary_dev(jdsp_mvdr *h, const uint8_t *flags_all_dev, double *sum4_dev);
long jdsp_mvdr_shard_blocks_out(const jdsp_mvdr *h);
int jdsp_mvdr_shard_finish_dev(jdsp_mvdr *h, const double *sums_all_dev, int world, int rank, int16_t *out_dev,
                               float *precast_dev, long *n_out_blocks);

/* ---- MVDR generalised to n microphones, per-bin covariance (BASELINE config 5) ------------- */
/* NOT a reference function: the reference has 2 microphones and one real 2x2 matrix for all
 * bins (jdsp_mvdr above is that algorithm).  This keeps its framing, VAD, run counter and weight
 * formula and makes the correlation a Hermitian n_mics x n_mics matrix PER BIN, accumulated over
 * the same noise frames, R_k += X_k X_k^H / 1024; w_k = R_k^-1 c_k / (c_k^H R_k^-1 c_k) with
 * c_k[m] = exp(j 2 PI f_k delays_s[m]); y = IDFT(w^H X).  loading >= 0 adds
 * loading * trace(R_k)/n_mics to the diagonal (with 0, R_k is singular until n_mics noise frames
 * have been seen, and the output is NaN -> 0 like the reference's before its first estimate).
 * pcm: n_mics planes, chan_stride samples apart, each n_blocks*512 samples. 2 <= n_mics <= 8. */
typedef struct jdsp_mvdrn jdsp_mvdrn;
int jdsp_mvdrn_create(jdsp_ctx *ctx, int n_mics, const double *delays_s, double loading, jdsp_mvdrn **out);
/* The same with FFT_PROCESSING_LEN = n_fft: 1024 (= jdsp_mvdrn_create: the reference's frame) or 512 (BASELINE config 5
 * as worded, "8-mic array, 512-pt STFT": blocks of 256 samples, KEEP_LEN 255, frames [first 255 samples of the previous
 * block, block, 0], 257 bins at k * 16000 / 512 Hz, R_k += X X^H / 512, samples 255..510 out).  Blocks are then n_fft / 2
 * samples everywhere below. */
int jdsp_mvdrn_create_cfg(jdsp_ctx *ctx, int n_mics, const double *delays_s, double loading, int n_fft, jdsp_mvdrn **out);
int jdsp_mvdrn_block_len(const jdsp_mvdrn *h);
int jdsp_mvdrn_destroy(jdsp_mvdrn *h);
int jdsp_mvdrn_reset(jdsp_mvdrn *h);
long jdsp_mvdrn_blocks_out(const jdsp_mvdrn *h, long n_blocks);
int jdsp_mvdrn_process_dev(jdsp_mvdrn *h, const int16_t *pcm_dev, long chan_stride, long n_blocks, int16_t *out_dev,
                           float *precast_dev, long *n_out_blocks);
int jdsp_mvdrn_process(jdsp_mvdrn *h, const int16_t *pcm_host, long chan_stride, long n_blocks, int16_t *out_host,
                       float *precast_host, long *n_out_blocks);

/* ---- MFCC ---------------------------------------------------------------------- */
/* MFCCFeatureExtraction_auto_version1.cpp.  The #defines at :23-33 become a runtime
 * configuration; jdsp_mfcc_native_cfg() fills in the reference's values
 * (window 1024, hop 512, 1024-FFT, first 512 bins, 38 channels over 0..22050 Hz,
 * 12 cepstra c1..c12, lifter 22, pre-emphasis 0.96).  n_fft may be 1024 or 512
 * (bins used: n_fft/2), win_len <= n_fft, n_chan <= 64, n_cep <= 32. */
typedef struct {
    int win_len, hop, n_fft, n_chan, n_cep, lifter;
    double half_rate, preemph;
} jdsp_mfcc_cfg;
typedef struct jdsp_mfcc jdsp_mfcc;
int jdsp_mfcc_native_cfg(jdsp_mfcc_cfg *cfg);
/* Runs MelFilterBankInit (:118-152) on the host for cfg and uploads the tables. */
int jdsp_mfcc_create(jdsp_ctx *ctx, const jdsp_mfcc_cfg *cfg, jdsp_mfcc **out);
int jdsp_mfcc_destroy(jdsp_mfcc *h);
/* MelFilterBankInit's three arrays (host): rgdMelFreqs[n_chan+1], rgdFiBins[n_fft/2],
 * rgdFilterBank[n_fft/2].  Any pointer may be NULL. */
int jdsp_mfcc_tables(const jdsp_mfcc *h, double *mel_freqs, int *fi_bins, double *fbank);
/* MFCCFeatureExtraction (:194-231) for n_frames frames: frame j = win_len samples at
 * pcm[frame_start[j]] (frame_start NULL: hop*j), x[0] of every frame is 0 as at :208.
 * feats: n_frames * n_cep doubles -- the reference's on-disk vector format (:99). */
int jdsp_mfcc_frames_dev(jdsp_mfcc *h, const int16_t *pcm_dev, const int64_t *frame_start_dev, long n_frames,
                         double *feats_dev);
int jdsp_mfcc_frames(jdsp_mfcc *h, const int16_t *pcm_host, long n_samples, const int64_t *frame_start_host,
                     long n_frames, double *feats_host);

/* The sub-steps of MFCCFeatureExtraction as functions of their own, for callers that keep the reference's
 * structure (jeicyboodsp_amd/compat: MelFilterBank, DCT, Liftering with the reference's signatures).  FP64 and the
 * reference's operation order; n_rows independent rows per call; host pointers.
 *   MelFilterBank (:154-174): abs = |X| of the first n_fft/2 bins per row -> mel = ln of the n_chan channel sums
 *   DCT           (:176-183): cep[n_cep] += sqrt(2/C) sum_k mel[k-1] cos(PI i (k-0.5)/C) -- ACCUMULATES into cep like
 *                             the reference (its caller zeroes dMFCCFeature first, :224)
 *   Liftering     (:185-192): cep[i-1] *= 1 + 0.5 L sin(PI i / L), in place */
int jdsp_mfcc_melfilterbank(jdsp_mfcc *h, const double *abs_host, long n_rows, double *mel_host);
int jdsp_mfcc_dct(jdsp_mfcc *h, const double *mel_host, long n_rows, double *cep_inout_host);
int jdsp_mfcc_liftering(jdsp_mfcc *h, double *cep_inout_host, long n_rows);

/* ---- GMM scoring / HMM recursion on MFCC vectors ------------- */
/* GMMAlgorithm_Test_Auto_ver2.cpp and Viterbi_version1.cpp consume the 12-double vectors the MFCC program
 * writes (MFCC:99); here they are read where jdsp_mfcc_frames_dev left them, in HBM.  The parameter records
 * are the programs' own (GMMTest:29-34; Viterbi:30-40), byte for byte, so a parameter file (GMMTest:76,
 * Viterbi:80) can be read straight into an array of them.  Everything is FP64 like the reference.
 * A batch is a set of utterances: utterance u owns the vectors utt_first[u] .. utt_first[u+1]-1
 * (n_utts + 1 offsets, non-decreasing). */
typedef struct {
    double alpa[4];
    double mean[4][12];
    double covariance[4][12][12];
    double eigenVector[4][12][4];
} jdsp_gmm_param;
typedef struct {
    jdsp_gmm_param gMMParam[6];
    double transProb[6][6];
} jdsp_hmm_param;

typedef struct jdsp_gmm jdsp_gmm;
/* classes: n_classes (1..256) records; what probability() reads of them (GMMTest:216-235) is packed and
 * uploaded: eigenVector, mean[k][0..3], covariance[k][i][i] for i < 4. */
int jdsp_gmm_create(jdsp_ctx *ctx, const jdsp_gmm_param *classes, int n_classes, jdsp_gmm **out);
int jdsp_gmm_destroy(jdsp_gmm *h);
/* "evaluation": 0 (default) evaluates probability() in the reference's operation order (un-fused products
 * and sums, IEEE division, four exp per mixture: GMMTest:228-233); 1 fuses it (FMA projection, -0.5/var
 * precomputed, one exp per mixture) -- about a third of the instructions, equal to a few 1e-16 relative
 * except where a mixture density is itself denormal. */
int jdsp_gmm_set_option(jdsp_gmm *h, const char *name, long value);
/* Recognition() (GMMTest:151-162) of every utterance against every class and main()'s arg-max
 * (GMMTest:113-127: `dMax < score`, first maximum wins).  scores: [n_utts][n_classes] doubles,
 * best (may be NULL): [n_utts] 0-based class indices.  feats: n_frames vectors, 16-byte aligned; offsets
 * outside [0, n_frames] are clamped on the device (nothing outside feats is ever read).  The host entry takes
 * n_frames from utt_first_host[n_utts] and requires utt_first_host[0] == 0. */
int jdsp_gmm_score_dev(jdsp_gmm *h, const double *feats_dev, long n_frames, const int64_t *utt_first_dev,
                       long n_utts, double *scores_dev, int *best_dev);
int jdsp_gmm_score(jdsp_gmm *h, const double *feats_host, const int64_t *utt_first_host, long n_utts,
                   double *scores_host, int *best_host);

/* probability() (GMMAlgorithm_Test_Auto_ver2.cpp:164-236, the live branch :216-235; = Viterbi_version1.cpp:248-267)
 * on its own: the density of n 12-double vectors under ONE mixture component given as the reference passes it --
 * pdMean (12 doubles, first 4 read), rgdCovariance[12][12] (diagonal entries 0..3 read), rgdEigenVector[12][4].
 * FP64, the reference's operation order.  Host pointers. */
int jdsp_gmm_probability(jdsp_ctx *ctx, const double *feats_host, long n, const double *mean12, const double *cov144,
                         const double *eig48, double *prob_host);

typedef struct jdsp_hmm jdsp_hmm;
/* models: n_models (1..1024) six-state records; log(transProb) is taken here, on the host (Viterbi:196). */
int jdsp_hmm_create(jdsp_ctx *ctx, const jdsp_hmm_param *models, int n_models, jdsp_hmm **out);
int jdsp_hmm_destroy(jdsp_hmm *h);
/* "evaluation": 0 (default) / 1, as jdsp_gmm_set_option, for the state densities. */
int jdsp_hmm_set_option(jdsp_hmm *h, const char *name, long value);
/* Sizes the emission scratch (n_frames * n_models * 6 doubles) ahead of time, e.g. before a graph capture. */
int jdsp_hmm_reserve(jdsp_hmm *h, long n_frames);
/* HMMRecognition() (Viterbi:157-246) as the reference computes it, quirks included (see
 * oracle/jdsp_oracle.h: log of the accumulated log probability at :196; the "decoding result" is the
 * per-frame arg-max state; the returned value is the frame-1 maximum).  n_frames = total vectors; every
 * utt_first offset must lie in [0, n_frames].  scores (may be NULL): [n_utts][n_models]; best (may be NULL):
 * [n_utts] arg-max model (Viterbi:119-126); path (may be NULL): [n_models][n_frames] states, 0 at each
 * utterance's first frame; trellis (may be NULL): [n_models][6][n_frames] = sProb.dHMMProb. */
int jdsp_hmm_viterbi_dev(jdsp_hmm *h, const double *feats_dev, long n_frames, const int64_t *utt_first_dev,
                         long n_utts, double *scores_dev, int *best_dev, int *path_dev, double *trellis_dev);
/* host entry: utt_first_host[0] must be 0 and utt_first_host[n_utts] is the total number of vectors */
int jdsp_hmm_viterbi(jdsp_hmm *h, const double *feats_host, const int64_t *utt_first_host, long n_utts,
                     double *scores_host, int *best_host, int *path_host, double *trellis_host);

#ifdef __cplusplus
}
#endif
#endif /* JDSP_H */
