// jeicyboo_compat.cpp -- see jeicyboo_compat.h.
#include "jeicyboo_compat.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

int g_block_len = 512;
int g_frame_count = 0;        // iFrameCount of the denoise stream in progress (0: none yet)
int g_device = -1;
jdsp_ctx *g_ctx = nullptr;
jdsp_denoise *g_dn[2] = {nullptr, nullptr};
jdsp_fastconv *g_conv = nullptr;
jdsp_mfcc *g_mfcc = nullptr;

// EstimateNoiseSpectrum's statics (SS:161,164)
double g_avg[1024];
short g_est_keep[512];
// MFCCFeatureExtraction's static (MFCC:198)
short g_mfcc_keep[512];
// CalcPitch's static keep buffer (Pitch1:74) and last result
short g_pitch_keep[512];
int g_pitch_arg = 0;
double g_pitch_max = 0;

[[noreturn]] void die(const char *what, jdsp_ctx *ctx)
{
    fprintf(stderr, "jeicyboo_compat: %s: %s\n", what, jdsp_last_error(ctx));
    abort();
}

#define CK(call)                                  \
    do {                                          \
        if ((call) != JDSP_OK) die(#call, g_ctx); \
    } while (0)

}  // namespace

double rgdFilterBank[512] = {0};
int rgdFiBins[512] = {0};
double rgdMelFreqs[39] = {0};

void JeicybooSetBlockLen(int block_len) { g_block_len = block_len; }
void JeicybooSetDevice(int d) { g_device = d; }

jdsp_ctx *JeicybooContext(void)
{
    if (!g_ctx) {
        int dev = g_device;
        if (dev < 0) {
            const char *e = getenv("JDSP_DEVICE");
            dev = e ? atoi(e) : 0;
        }
        if (jdsp_create(dev, &g_ctx) != JDSP_OK) die("jdsp_create", nullptr);
    }
    return g_ctx;
}

void JeicybooResetStreams(void)
{
    for (auto &h : g_dn) { if (h) jdsp_denoise_destroy(h); h = nullptr; }
    if (g_conv) jdsp_fastconv_destroy(g_conv);
    g_conv = nullptr;
    if (g_mfcc) jdsp_mfcc_destroy(g_mfcc);
    g_mfcc = nullptr;
    g_frame_count = 0;
    memset(g_avg, 0, sizeof(g_avg));
    memset(g_est_keep, 0, sizeof(g_est_keep));
    memset(g_mfcc_keep, 0, sizeof(g_mfcc_keep));
    memset(g_pitch_keep, 0, sizeof(g_pitch_keep));
}

// ---- FFTAlgorithm_ver2.cpp ----------------------------------------------------------------
void FFTProcess(COMPLEX *in, COMPLEX *out, int n, bool dir)
{
    CK(jdsp_fft_process_f64(JeicybooContext(), (const double *)in, (double *)out, n, 1, dir ? 1 : 0));
}

void Bitrev(COMPLEX *in, short *bits, int n, COMPLEX *out)
{
    CK(jdsp_bitrev_table(JeicybooContext(), n, g_block_len, bits));
    for (int k = 0; k < n; k++) out[k] = in[bits[k]];                    // :204-205
}

// The O(N^2) family: evaluated on the device as the reference writes the sums (any n, accumulating into the
// caller's output like the reference, :168,:178,:154) -- jdsp_dft_direct_f64.
void DFTProcess(short *in, COMPLEX *out, int n)
{
    CK(jdsp_dft_direct_f64(JeicybooContext(), JDSP_DFT_I16, in, (double *)out, n, 1));
}

void IDFTProcess(COMPLEX *in, COMPLEX *out, int n)
{
    CK(jdsp_dft_direct_f64(JeicybooContext(), JDSP_IDFT, in, (double *)out, n, 1));
}

void IFFTProcess(COMPLEX *in, COMPLEX *out, int n)
{
    CK(jdsp_dft_direct_f64(JeicybooContext(), JDSP_IDFT_OVER_N, in, (double *)out, n, 1));
}

// ---- SpectralSubtraction_final.cpp / WienerFilter_final.cpp --------------------------------
// iFrameCount is BLOCK_LEN = KEEP_LEN (SS:53-54) and FFT_PROCESSING_SIZE is twice that (SS:55): 512 / 1024 as the
// reference defines them, or 256 / 512 (BASELINE config 3: "512-pt STFT 50 % hop").  A stream keeps the size of its
// first call; JeicybooResetStreams() forgets it.
static int frame_count(const char *who, int n)
{
    if (n != 512 && n != 256) { fprintf(stderr, "%s: iFrameCount must be 512 or 256\n", who); abort(); }
    if (g_frame_count && n != g_frame_count) { fprintf(stderr, "%s: iFrameCount changed inside a stream\n", who); abort(); }
    g_frame_count = n;
    return n;
}

bool VoiceActivityDetection(short *block, int n)
{
    frame_count("VoiceActivityDetection", n);
    uint8_t v = 0;
    CK(jdsp_vad_blocks_ex(JeicybooContext(), JDSP_VAD_DENOISE, n, block, 1, &v, nullptr, nullptr));
    return v != 0;
}

void EstimateNoiseSpectrum(short *temp, int iter, short *in, double *noise, int n)
{
    frame_count("EstimateNoiseSpectrum", n);
    const int N = 2 * n;
    if (iter == 2) memcpy(g_est_keep, temp, sizeof(short) * n);                           // SS:165-167
    short frame[1024];
    memcpy(frame, g_est_keep, sizeof(short) * n);
    memcpy(frame + n, in, sizeof(short) * n);
    static jdsp_c32 spec[1024];
    long nf = 0;
    CK(jdsp_stft_i16(JeicybooContext(), frame, N, N, n, spec, &nf));                      // SS:168-180 on the GPU
    for (int i = 0; i < N; i++) {                                                         // SS:182-187
        g_avg[i] += sqrt((double)spec[i].re * spec[i].re + (double)spec[i].im * spec[i].im);
        if (iter >= 3) g_avg[i] /= 2.0;
    }
    if (iter == 10) memcpy(noise, g_avg, sizeof(double) * N);                             // SS:189-193
    memcpy(g_est_keep, in, sizeof(short) * n);                                            // SS:195
}

static bool denoise_one(int mode, short *in, double *noise, short *out, int n)
{
    frame_count("SpectralSubtraction/WienerFiltering", n);
    jdsp_ctx *ctx = JeicybooContext();
    if (!g_dn[mode]) CK(jdsp_denoise_create_cfg(ctx, mode, 2 * n, n, &g_dn[mode]));
    long n_out = 0;
    short tmp[512];
    CK(jdsp_denoise_apply(g_dn[mode], in, 1, noise, tmp, nullptr, &n_out));
    if (n_out == 1) memcpy(out, tmp, sizeof(short) * n);
    return n_out == 1;                                                                    // SS:260-263
}

bool SpectralSubtraction(short *in, double *noise, short *out, int n) { return denoise_one(JDSP_SPECSUB, in, noise, out, n); }
bool WienerFiltering(short *in, double *noise, short *out, int n) { return denoise_one(JDSP_WIENER, in, noise, out, n); }

// ---- Fast_Convolution_Based_3DAudio_Impl.cpp ------------------------------------------------
bool AnalySisFreqDomain(short *in, short *out, int n, double (*filter)[2])
{
    jdsp_ctx *ctx = JeicybooContext();
    if (!g_conv) {
        const int n_fft = 8192;                                   // FFT_PROCESSING_SIZE (:48)
        const int n_taps = n_fft - n + 1;                         // FILTER_LENGTH 7169 for BLOCK_SIZE 1024
        std::vector<double> taps(n_taps);
        for (int i = 0; i < n_taps; i++) taps[i] = filter[i][0];  // :82-84
        CK(jdsp_fastconv_create(ctx, taps.data(), n_taps, 1, n_fft, &g_conv));
    }
    if (n != jdsp_fastconv_block_len(g_conv)) { fprintf(stderr, "AnalySisFreqDomain: iFrameCount changed\n"); abort(); }
    long n_out = 0;
    std::vector<short> tmp(n);
    CK(jdsp_fastconv_process(g_conv, in, 1, tmp.data(), nullptr, &n_out));
    if (n_out == 1) memcpy(out, tmp.data(), sizeof(short) * n);
    return n_out == 1;                                            // :122,:176
}

// ---- MFCCFeatureExtraction_auto_version1.cpp ------------------------------------------------
void MelFilterBankInit()
{
    jdsp_ctx *ctx = JeicybooContext();
    if (!g_mfcc) {
        jdsp_mfcc_cfg cfg;
        jdsp_mfcc_native_cfg(&cfg);
        CK(jdsp_mfcc_create(ctx, &cfg, &g_mfcc));
    }
    CK(jdsp_mfcc_tables(g_mfcc, rgdMelFreqs, rgdFiBins, rgdFilterBank));
}

bool MFCCFeatureExtraction(short *in, double (*feat)[12])
{
    if (!g_mfcc) MelFilterBankInit();
    short buf[512 + 1024];                                        // rgsProcessingBuffer (:199,:203-204)
    memcpy(buf, g_mfcc_keep, sizeof(g_mfcc_keep));
    memcpy(buf + 512, in, sizeof(short) * 1024);
    CK(jdsp_mfcc_frames(g_mfcc, buf, 1536, nullptr, 2, &feat[0][0]));   // frames at offsets 0 and 512 (:205)
    memcpy(g_mfcc_keep, in + 512, sizeof(g_mfcc_keep));           // :228
    return true;
}

// The three sub-steps with the reference's own signatures (MFCC:40-42): a caller that keeps
// MFCCFeatureExtraction's body (its |X| loop, memset, then these three calls, :218-226) links unchanged.
void MelFilterBank(double *dAbs, double *dMelFiltered)                                   // :154
{
    if (!g_mfcc) MelFilterBankInit();
    CK(jdsp_mfcc_melfilterbank(g_mfcc, dAbs, 1, dMelFiltered));                         // reads dAbs[0..511], writes 38
}

void DCT(double *dMelFiltered, double *dMFCCFeature)                                     // :176 (accumulates)
{
    if (!g_mfcc) MelFilterBankInit();
    CK(jdsp_mfcc_dct(g_mfcc, dMelFiltered, 1, dMFCCFeature));
}

void Liftering(double *dMFCCFeature)                                                     // :185
{
    if (!g_mfcc) MelFilterBankInit();
    CK(jdsp_mfcc_liftering(g_mfcc, dMFCCFeature, 1));
}

// ---- PitchEstimation_method1.cpp ------------------------------------------------------------
void CalcPitch(short *in, int n)
{
    if (n != 512) { fprintf(stderr, "CalcPitch: iFrameCount must be 512\n"); abort(); }
    int32_t arg = 0;
    float rmax = 0;
    CK(jdsp_pitch_autocorr(JeicybooContext(), in, 1, g_pitch_keep, &arg, &rmax, nullptr));
    g_pitch_arg = arg;
    g_pitch_max = rmax;
    printf("Estimation arg %d , dMin %f pitch %f \n", arg, (double)rmax, (16000.0 / (double)arg));   // :109
    memcpy(g_pitch_keep, in, sizeof(g_pitch_keep));                                                   // :112
}
int JeicybooLastPitchArg(void) { return g_pitch_arg; }
double JeicybooLastPitchMax(void) { return g_pitch_max; }

// ---- AnalysisAdditiveWhiteGaussianNoise.cpp:98-133 --------------------------------------------
static short g_awgn_keep[512];                     // rgssKeepBuffer (:103)
static double g_awgn_autocorr[512];
void AnalysisAdditiveWhiteGaussianNoise(short *noise, int n)
{
    if (n != 512) { fprintf(stderr, "AnalysisAdditiveWhiteGaussianNoise: iFrameCount must be 512\n"); abort(); }
    int32_t arg = 0;
    float rmax = 0, r[512];
    CK(jdsp_pitch_autocorr(JeicybooContext(), noise, 1, g_awgn_keep, &arg, &rmax, r));
    for (int i = 0; i < 512; i++) g_awgn_autocorr[i] = r[i];                                    // :122-124
    memcpy(g_awgn_keep, noise, sizeof(g_awgn_keep));                                            // :129
}
const double *JeicybooLastAutoCorrelation(void) { return g_awgn_autocorr; }

// ---- GMMAlgorithm_Test_Auto_ver2.cpp / Viterbi_version1.cpp -----------------------------------
static std::vector<double> gather_rows(double **rows, int n)
{
    std::vector<double> x((size_t)(n > 0 ? n : 0) * 12);
    for (int i = 0; i < n; i++) memcpy(&x[(size_t)i * 12], rows[i], 12 * sizeof(double));   // dpTestBuf[i][0..11]
    return x;
}

double probability(double *pdFeature, double *pdMean, double rgdCovariance[12][12], double rgdEigenVector[12][4])   // GMMTest:164
{
    double p = 0;
    CK(jdsp_gmm_probability(JeicybooContext(), pdFeature, 1, pdMean, &rgdCovariance[0][0], &rgdEigenVector[0][0], &p));
    return p;
}

double Recognition(double **rows, GMMParameter *param, int n)
{
    jdsp_gmm *h = nullptr;
    CK(jdsp_gmm_create(JeicybooContext(), param, 1, &h));
    const std::vector<double> x = gather_rows(rows, n);
    const int64_t first[2] = {0, n > 0 ? n : 0};
    double score = 0;
    CK(jdsp_gmm_score(h, x.data(), first, 1, &score, nullptr));
    jdsp_gmm_destroy(h);
    return score;
}

double HMMRecognition(double **rows, HMMParameter *param, int n)
{
    jdsp_hmm *h = nullptr;
    CK(jdsp_hmm_create(JeicybooContext(), param, 1, &h));
    const std::vector<double> x = gather_rows(rows, n);
    const int64_t first[2] = {0, n > 0 ? n : 0};
    const size_t m = (size_t)(n > 0 ? n : 1);
    std::vector<int> path(m, 0);
    std::vector<double> trellis(6 * m, 0.0);
    double score = 0;
    CK(jdsp_hmm_viterbi(h, x.data(), first, 1, &score, nullptr, path.data(), trellis.data()));
    jdsp_hmm_destroy(h);
    for (int i = n - 1; i > 0; i--) printf("max accumulated prob %f \n", trellis[(size_t)path[i] * m + i]);   // :222
    printf("decoding result ! \n");                                                                          // :227
    for (int i = 0; i < n - 1; i++) printf("%d ,", path[i]);                                                  // :228-230
    printf("\n");
    return score;
}
