/*
 * jdsp_oracle.h -- CPU restatement of the JeicybooDSP FFT-based spectral path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product path (jeicyboodsp_amd/libjdsp.so) never links or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference checkout).  Arithmetic is IEEE double like the reference.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - orc_bitrev_table / orc_fft_process / orc_dft_process / orc_idft_process /
 *     orc_ifft_process / orc_fft_roundtrip_i16 are pinned BIT-EXACTLY against
 *     FFTAlgorithm_ver2.cpp compiled from where it lies (oracle/_ref, recipe in
 *     oracle/Makefile) and against the committed fixtures tests/golden/fftalg_*.
 *   - the four application chains (spectral subtraction, Wiener, fast
 *     convolution, MFCC) call FFTW3, which is absent from the reference tree and
 *     from this image, so the reference programs are unbuildable here and those
 *     chains are "parity unpinned" at the FFTW boundary: the transform is taken
 *     as FFTW's documented contract (unnormalised c2c DFT, sign -1 forward),
 *     orc_dft_c2c, itself cross-checked against the pinned orc_fft_process.
 */
#ifndef JDSP_ORACLE_H
#define JDSP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* FFTAlgorithm_ver2.cpp:20-22 (COMPLEX) / fftw_complex = double[2] */
typedef struct { double re, im; } orc_cplx;

/* ---- FFTAlgorithm_ver2.cpp ------------------------------------------------ */
/* :186-207 Bitrev: table only.  bits come from block_len (the reference takes
 * them from BLOCK_LEN, not from the transform length: :188). */
void orc_bitrev_table(int n_fft, int block_len, short *table);
/* :94-149 FFTProcess (forward != 0: sign -1).  block_len as above (512 native). */
void orc_fft_process(const orc_cplx *in, orc_cplx *out, int n_fft, int forward, int block_len);
/* :162-173 DFTProcess (accumulates into out) */
void orc_dft_process(const short *in, orc_cplx *out, int n_fft);
/* :175-184 IDFTProcess (unnormalised, accumulates) */
void orc_idft_process(const orc_cplx *in, orc_cplx *out, int n_fft);
/* :151-160 IFFTProcess (1/N-normalised O(N^2) IDFT, accumulates) */
void orc_ifft_process(const orc_cplx *in, orc_cplx *out, int n_fft);
/* :62-86 main loop body: int16 -> FFT -> IFFT -> (short)(re/N), n_blocks blocks of n_fft */
void orc_fft_roundtrip_i16(const short *pcm, int n_blocks, int n_fft, short *out);

/* ---- FFTW contract (third-party, absent): unnormalised c2c DFT ------------- */
/* sign = -1 forward, +1 backward.  n must be a power of two. */
void orc_dft_c2c(const orc_cplx *in, orc_cplx *out, int n, int sign);

/* ---- shared pieces of the application programs ---------------------------- */
/* SpectralSubtraction_final.cpp:226 etc.: 0.54-0.46cos(2*PI*i/(n-1)), PI=3.141592 */
void orc_hamming(int n, double *w);
/* SS:218-230 / WF:181-193 framing + window + forward transform for a whole
 * stream: frame f = samples [hop*f, hop*f+n), n_frames frames, full spectrum. */
void orc_stft(const short *pcm, long n_frames, int n, int hop, orc_cplx *spec);
/* SS:121-156 == WF:261-296 VoiceActivityDetection on one block (keep buffer is
 * always zero: the update at :154 is unreachable).  The out-of-bounds read of
 * frame[n] at i == n-1 (:139) is defined here as 0.  Returns 1 = voice.
 * energy/zcr may be NULL. */
int orc_vad_block(const short *block, int block_len, double *energy, int *zcr);
/* BeamForming_MVDR_ver1.cpp:207-242: the same function over [zeros(n_fft/2 - 1), block(n_fft/2), 0], energy decides */
int orc_mvdr_vad_block(const short *block, int n_fft, double *energy, int *zcr);

/* SS:62-119,159-264 / WF:52-235: the whole per-block state machine. */
typedef struct orc_denoise orc_denoise;
enum { ORC_SPECSUB = 0, ORC_WIENER = 1 };
orc_denoise *orc_denoise_create(int mode);
void orc_denoise_destroy(orc_denoise *s);
/* One iteration of main's while-loop for one 512-sample block.  Returns 1 when
 * the reference would fwrite() out_block.  ola_out (may be NULL) receives the
 * 512 pre-cast doubles behind out_block. */
int orc_denoise_block(orc_denoise *s, const short *in_block, short *out_block, double *ola_out);
/* Current noise estimate (1024 doubles) and the flag/run length of the last block. */
const double *orc_denoise_noise(const orc_denoise *s);
int orc_denoise_last_voice(const orc_denoise *s);
/* Whole stream: n_blocks blocks in, returns blocks written (n_blocks-2). */
long orc_denoise_stream(int mode, const short *pcm, long n_blocks, short *out, double *ola_out);
/* The same with the reference's macros as parameters: BLOCK_LEN = KEEP_LEN = block_len (a power of two <= 512),
 * FFT_PROCESSING_SIZE = 2 block_len (SS:53-55 are 512 / 512 / 1024; BASELINE config 3 words them 256 / 256 / 512).
 * The thresholds (SS:48-49: energy 700, ZCR 200) and NOISE_ESTIMATION_FRAMECOUNT stay as they are. */
orc_denoise *orc_denoise_create2(int mode, int block_len);
long orc_denoise_stream2(int mode, int block_len, const short *pcm, long n_blocks, short *out, double *ola_out);

/* Fast_Convolution_Based_3DAudio_Impl.cpp:102-177 overlap-save convolver,
 * generalised: n_fft transform, n_taps filter, block = n_fft - n_taps + 1
 * (native 8192 / 7169 / 1024).  The reference's first (n_fft/block - 1)
 * queued blocks are uninitialised heap (:120); they are defined as ZERO here,
 * which makes emitted block e the convolution of the stream starting at block
 * (n_hist) with zero history.  Returns blocks written (n_blocks - n_hist). */
long orc_fastconv_stream(const short *pcm, long n_blocks, const double *taps, int n_taps,
                         int n_fft, short *out, double *pre_cast);

/* MFCCFeatureExtraction_auto_version1.cpp:118-231, parametrised.  Native:
 * win 1024, hop 512, n_fft 1024, n_bins 512, 38 channels, half_rate 22050,
 * 12 cepstra, lifter 22, pre-emphasis 0.96. */
typedef struct {
    int win_len, hop, n_fft, n_bins, n_chan, n_cep, lifter;
    double half_rate, preemph;
} orc_mfcc_cfg;
void orc_mfcc_native_cfg(orc_mfcc_cfg *c);
/* :118-152 MelFilterBankInit: mel_freqs[n_chan+1], fi_bins[n_bins], fbank[n_bins] */
void orc_mel_init(const orc_mfcc_cfg *c, double *mel_freqs, int *fi_bins, double *fbank);
/* :154-174 MelFilterBank(dAbs, dMelFiltered): mag[n_bins] -> ln of the n_chan channel sums.
 * :176-183 DCT(dMelFiltered, dMFCCFeature): ACCUMULATES into cep[n_cep].  :185-192 Liftering(cep), in place. */
void orc_mel_filterbank(const orc_mfcc_cfg *c, const int *fi_bins, const double *fbank, const double *mag, double *mel);
void orc_dct(const orc_mfcc_cfg *c, const double *mel, double *cep);
void orc_liftering(const orc_mfcc_cfg *c, double *cep);
/* :194-231 per frame: frame = win_len int16 samples -> n_cep doubles.
 * x[0] stays 0 (:208 starts at i=1). */
void orc_mfcc_frame(const orc_mfcc_cfg *c, const int *fi_bins, const double *fbank,
                    const short *frame, double *cep);
/* :86-104 one file as the first of a run: n_blocks blocks of 2*hop samples
 * (keep starts at zero, very first vector skipped) -> 2*n_blocks-1 vectors.
 * Only valid for win_len == 2*hop.  Returns vectors written. */
long orc_mfcc_stream(const orc_mfcc_cfg *c, const short *pcm, long n_blocks, double *feats);

/* PitchEstimation_method1.cpp:69-116 CalcPitch for a whole stream of 512-sample blocks
 * (SURVEY §8f rank 2): frame = [previous block (zeros first), block], no window;
 * r = IDFT(|DFT(frame)|^2).re / 1024; scan i = 511 .. 101 with `>=` (ties: smallest lag wins).
 * arg[b], rmax[b] per block; autocorr (may be NULL) gets r[0..511] per block. */
void orc_pitch_stream(const short *pcm, long n_blocks, int *arg, double *rmax, double *autocorr);

/* BeamForming_MVDR_ver1.cpp: the whole main() loop (:169-231) for two
 * channels of 512-sample blocks: energy VAD on the left channel (:207-242), spatial correlation
 * accumulated over non-voice runs (:244-270), ProcessMVDR (:124-205) with steering delay
 * d_time (the reference: 0, :60,:121).  Returns blocks written (n_blocks-1).  corr4 (may be NULL)
 * receives the final rgdSpatialCorr in row-major order; corr_trace (may be NULL) receives the
 * 2x2 matrix in effect for every block (4 doubles per block).  Until the matrix is
 * invertible the reference's output is NaN; (short)NaN is 0 here as everywhere. */
long orc_mvdr_stream(const short *left, const short *right, long n_blocks, double d_time,
                     short *out, double *pre_cast, double *corr4, double *corr_trace);
/* The same program function by function, for callers that do not follow main()'s protocol:
 * EstimateSpatialCorrMtx (:244-270) on one 1024-sample [previous block, block] frame per channel, ADDED to R4
 * (row-major rgdSpatialCorr); ProcessMVDR (:124-205) for one block with the caller's R4 (its statics -- the two
 * 511-sample keep buffers and the call counter -- live in orc_mvdr).  Returns 1 from the second call on. */
typedef struct orc_mvdr orc_mvdr;
void orc_mvdr_estimate(const short *temp_l, const short *temp_r, double *R4);
orc_mvdr *orc_mvdr_create(void);
void orc_mvdr_destroy(orc_mvdr *s);
int orc_mvdr_process_block(orc_mvdr *st, const short *left, const short *right, double d_time, const double *R4,
                           short *out, double *pre_cast);

/* BASELINE config 5 / SURVEY §8f rank 3: the MVDR beamformer generalised to n_mics <= 8 microphones
 * with a PER-BIN n_mics x n_mics covariance.  NOT what the reference computes (it has 2 mics and
 * one real 2x2 matrix summed over all bins, see orc_mvdr_stream): there is no reference for this
 * function, it defines the generalisation ("parity unpinned" by construction).  Kept from the
 * reference: 1024-point frames [first 511 samples of the previous block, block, 0] (:136-141,
 * :195-196), the energy VAD on microphone 0 (:207-242), the run counter and the
 * [previous block, block] estimation frames (:191-211,:250-253), R_k += X_k X_k^H / 1024 per noise
 * frame, w_k = R_k^-1 c_k / (c_k^H R_k^-1 c_k) (:170-171) with c_k[m] = exp(j 2 PI f_k delay[m])
 * (signed frequency, so the output is real), y = IDFT(w^H X), samples 511..1022 emitted, first
 * block dropped (:201-204).  loading >= 0 adds loading * trace(R_k)/n_mics to R_k's diagonal.
 * pcm: n_mics planes of chan_stride samples.  Returns blocks written. */
long orc_mvdrn_stream(const short *pcm, long chan_stride, int n_mics, long n_blocks, const double *delays,
                      double loading, short *out, double *pre_cast);
/* The same with FFT_PROCESSING_LEN = n_fft (1024 as above, or 512 = BASELINE config 5 as worded: blocks of 256,
 * KEEP_LEN 255, frames [first 255 samples of the previous block, block, 0], R_k += X X^H / 512, samples 255..510
 * emitted, f_k = k * 16000 / 512; thresholds unchanged). */
long orc_mvdrn_stream2(const short *pcm, long chan_stride, int n_mics, long n_blocks, const double *delays,
                       double loading, int n_fft, short *out, double *pre_cast);

/* GMM scoring and the HMM recursion on MFCC vectors.
 * Both programs use Eigen only for one (1 x 12)(12 x 4) product per call (GMMAlgorithm_Test_Auto_ver2.cpp:228,
 * Viterbi_version1.cpp:260); Eigen is not in this image, so these programs cannot be compiled here and the
 * product is restated as the in-order sum over the 12 inputs ("parity unpinned at the Eigen boundary": an
 * Eigen build may associate that sum differently, an O(1e-16) relative effect).
 * The structs are the programs' own parameter records (GMMTest:29-34, Viterbi:30-40), byte for byte: the
 * parameter files are arrays of them (GMMTest:76, Viterbi:80). */
typedef struct {
    double alpa[4];
    double mean[4][12];
    double covariance[4][12][12];
    double eigenVector[4][12][4];
} orc_gmm_param;
typedef struct {
    orc_gmm_param gMMParam[6];
    double transProb[6][6];
} orc_hmm_param;

/* probability() (GMMTest:216-235 = Viterbi:248-267): project the 12 features on 4 principal axes, product of
 * four univariate normal densities with mean[i], covariance[i][i], i < 4 (PI = 3.141592). */
double orc_gmm_probability(const double *x, const double *mean, const double *cov /*[12][12]*/,
                           const double *eig /*[12][4]*/);
/* Recognition() (GMMTest:151-162): mean over the frames of log(sum_k alpa[k] * probability_k). */
double orc_gmm_recognition(const double *feats, long n_frames, const orc_gmm_param *g);
/* The class loop of main() (GMMTest:113-127): scores[u] for every class and the arg-max with the reference's
 * `dMax < score` update rule (first maximum wins; a NaN never replaces, a leading NaN is never replaced). */
int orc_gmm_classify(const double *feats, long n_frames, const orc_gmm_param *classes, int n_classes,
                     double *scores);
/* HMMRecognition() (Viterbi:157-246) as written, including what it does rather than what a Viterbi decoder
 * would: the recursion adds log(previous accumulated LOG probability) (:196), so a negative accumulated value
 * turns the trellis into NaN from the second frame on; the back-pointer is overwritten by the next frame's
 * arg-max (:212-224), so the "decoding result" is the per-frame arg-max state for frames 1..n-1 (frame 0 stays
 * 0) and the returned value is the largest trellis entry of frame 1 (0 when n_frames < 2).  The one-past-the-end
 * write of dDecodingReslt[iFileLen-1] (:164,:223) lands in path[n_frames-1] here (path has n_frames entries).
 * trellis (may be NULL): [6][n_frames] as sProb.dHMMProb. */
double orc_hmm_viterbi(const double *feats, long n_frames, const orc_hmm_param *h, int *path, double *trellis);

#ifdef __cplusplus
}
#endif
#endif
