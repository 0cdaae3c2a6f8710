// jeicyboo_compat.h -- the reference's own per-block C++ function signatures, implemented
// over the C ABI (include/jdsp.h) so that a reference main() can be relinked against the
// MI355X engine unchanged.  One call = one small GPU launch: this is the latency mode.  The
// throughput path is the batched ABI (see drivers.cpp and INTEGRATION.md).
//
// Every declaration cites the reference function it replaces.  State the reference keeps in
// static locals lives in handles owned by this translation unit; like the reference, these
// functions are not re-entrant.
#ifndef JEICYBOO_COMPAT_H
#define JEICYBOO_COMPAT_H

#include "../../include/jdsp.h"

// FFTAlgorithm_ver2.cpp:20-22
#ifndef JEICYBOO_NO_COMPLEX_TYPEDEF
typedef struct { double real, imag; } COMPLEX;
#endif

// The reference hard-wires these as macros; here they are run-time settings with the same defaults.
void JeicybooSetBlockLen(int block_len);                 // FFTAlgorithm_ver2.cpp:16  BLOCK_LEN 512 (Bitrev's bit count, :188)
void JeicybooSetDevice(int hip_device);                  // which GPU the implicit context uses (default 0 / $JDSP_DEVICE)
void JeicybooResetStreams(void);                         // forget all per-stream state (= restarting the reference program)
jdsp_ctx *JeicybooContext(void);

// ---- FFTAlgorithm_ver2.cpp:24-28 --------------------------------------------------------
void FFTProcess(COMPLEX *cpFftInput, COMPLEX *cpFftOutput, int iFFTLen, bool bDir);          // :94
void IFFTProcess(COMPLEX *cpFftOutput, COMPLEX *cpFftInput, int iFFTLen);                    // :151 (accumulates, 1/N)
void DFTProcess(short *spInputBuffer, COMPLEX *cpFftOutput, int iFFTLen);                    // :162 (accumulates)
void IDFTProcess(COMPLEX *cpFftOutput, COMPLEX *cpFftInput, int iFFTLen);                    // :175 (accumulates)
void Bitrev(COMPLEX *cpFftInput, short *psBit, int iFFTLen, COMPLEX *cpFftBitRevInput);      // :186

// ---- SpectralSubtraction_final.cpp:58-60 / WienerFilter_final.cpp:48-50 ------------------
bool VoiceActivityDetection(short *rgsInputBuffer, int iFrameCount);                          // SS:121 / WF:261
void EstimateNoiseSpectrum(short *rgsTempBuffer, int iNumOfIteration, short *psInputBuffer,
                           double *pdEstimatedNoiseSpec, int iFrameCount);                    // SS:159 / WF:120
bool SpectralSubtraction(short *psInputBuffer, double *pdEstimatedNoiseSpec, short *psOutputBuffer,
                         int iFrameCount);                                                    // SS:201
bool WienerFiltering(short *psInputBuffer, double *pdEstimatedNoiseSpec, short *psOutputBuffer,
                     int iFrameCount);                                                        // WF:162

// ---- Fast_Convolution_Based_3DAudio_Impl.cpp:51 -------------------------------------------
// fcFilterBefFFT is the reference's fftw_complex[8192] (= double[8192][2]) holding the taps in
// the real parts (:82-84); the filter length is FFT_PROCESSING_SIZE - iFrameCount + 1.
bool AnalySisFreqDomain(short *psInputBuffer, short *psOutputBuffer, int iFrameCount, double (*fcFilterBefFFT)[2]);  // :102

// ---- MFCCFeatureExtraction_auto_version1.cpp:34-42 ----------------------------------------
extern double rgdFilterBank[512];      // :34
extern int rgdFiBins[512];             // :35
extern double rgdMelFreqs[38 + 1];     // :36
void MelFilterBankInit();                                                                     // :118
bool MFCCFeatureExtraction(short *rgsInputBuffer, double (*dMFCCFeature)[12]);                // :194
// The sub-steps on their own (MFCCFeatureExtraction's kernel fuses them; these are for callers that keep its body):
void MelFilterBank(double *dAbs, double *dMelFiltered);                                      // :154  dAbs[512] -> 38 ln sums
void DCT(double *dMelFiltered, double *dMFCCFeature);                                        // :176  ACCUMULATES into 12
void Liftering(double *dMFCCFeature);                                                        // :185  in place

// ---- PitchEstimation_method1.cpp:31 --------------------------------------------------------
// Prints the reference's "Estimation arg %d , dMin %f pitch %f" line (:109); the lag and the
// autocorrelation value of the last call are also available programmatically.
void CalcPitch(short *psInputBuffer, int iFrameCount);                                          // :69
int JeicybooLastPitchArg(void);
double JeicybooLastPitchMax(void);

// ---- AnalysisAdditiveWhiteGaussianNoise.cpp:37 (the analysis half, :98-133) ------------------
// Same chain as CalcPitch without the arg-max: frame [previous noise block, block], |X|^2, inverse, /1024.
// The reference discards dAutoCorrelation (a local, :104); here the last call's 512 lags can be read back.
// The generation half (time-seeded std::normal_distribution, GetTickCount, :85-96,:136-145) is not part of
// the spectral path and is not provided.
void AnalysisAdditiveWhiteGaussianNoise(short *psNoiseBuffer, int iFrameCount);                // :98
const double *JeicybooLastAutoCorrelation(void);                                                // dAutoCorrelation[512]

// ---- GMMAlgorithm_Test_Auto_ver2.cpp:29-34,:44 / Viterbi_version1.cpp:30-40,:49 -------------
// The parameter records are the C ABI's (same layout as the reference's structs).  One call scores one
// utterance against one record; the class loop of main() (GMMTest:113-127) calls Recognition once per
// class -- jdsp_gmm_score does all classes and all utterances in one launch.  probability() (GMMTest:43,:164)
// keeps its signature too (one vector, one mixture component per call); HMMRecognition prints what the reference prints
// (Viterbi:222-231), the decoded states as integers (the reference passes doubles to %d there).
#ifndef JEICYBOO_NO_GMM_TYPEDEFS
typedef jdsp_gmm_param GMMParameter;
typedef jdsp_hmm_param HMMParameter;
#endif
double probability(double *pdFeature, double *pdMean, double rgdCovariance[12][12],
                   double rgdEigenVector[12][4]);                                               // GMMTest:43,:164
double Recognition(double **dpTestBuf, GMMParameter *pGmmParameter, int iFileLen);              // GMMTest:151
double HMMRecognition(double **dpTestBuf, HMMParameter *pHmmParameter, int iFileLen);           // Viterbi:157

#endif
