// compat_selftest.cpp -- drives the reference-signature functions exactly the way the
// reference main()s do (one block per call) and dumps what they return, for tests/ to compare
// with the CPU checker.  usage: compat_selftest <ss|wf|noise|vad|conv|mfcc|mfccsteps|pitch|awgn|fft|dft|gmm|prob|hmm> in.raw out.bin [taps.f64 | params.bin]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jeicyboo_compat.h"

static std::vector<short> slurp(const char *p)
{
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(1); }
    std::vector<short> v;
    short s;
    while (fread(&s, 2, 1, f) == 1) v.push_back(s);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 1;
    const char *what = argv[1];
    std::vector<short> pcm = slurp(argv[2]);
    FILE *out = fopen(argv[3], "wb");
    if (!strcmp(what, "ss") || !strcmp(what, "wf")) {
        // SpectralSubtraction_final.cpp:92-113 verbatim in structure; argv[4] = BLOCK_LEN (default 512, or 256)
        const int B = argc > 4 ? atoi(argv[4]) : 512;
        short temp[512] = {0}, ob[512] = {0};
        double noise[1024] = {0};
        int iter = 0;
        for (size_t b = 0; b + B <= pcm.size(); b += B) {
            short *in = &pcm[b];
            if (!VoiceActivityDetection(in, B)) {
                iter++;
                if (iter == 1) memcpy(temp, in, sizeof(short) * B);
                else if (iter > 1) EstimateNoiseSpectrum(temp, iter, in, noise, B);
            } else iter = 0;
            bool ok = !strcmp(what, "ss") ? SpectralSubtraction(in, noise, ob, B) : WienerFiltering(in, noise, ob, B);
            if (ok) fwrite(ob, 2, B, out);
        }
    } else if (!strcmp(what, "noise")) {
        // EstimateNoiseSpectrum by name (SS:159-198): main()'s VAD / run-length logic around it (SS:98-109), and
        // pdEstimatedNoiseSpec dumped every time it is latched (iNumOfIteration == 10, :189-193); argv[4] = BLOCK_LEN
        const int B = argc > 4 ? atoi(argv[4]) : 512;
        short temp[512] = {0};
        double noise[1024] = {0};
        int iter = 0;
        for (size_t b = 0; b + B <= pcm.size(); b += B) {
            short *in = &pcm[b];
            if (!VoiceActivityDetection(in, B)) {
                iter++;
                if (iter == 1) memcpy(temp, in, sizeof(short) * B);
                else if (iter > 1) {
                    EstimateNoiseSpectrum(temp, iter, in, noise, B);
                    if (iter == 10) fwrite(noise, 8, 2 * B, out);
                }
            } else iter = 0;
        }
    } else if (!strcmp(what, "vad")) {
        // VoiceActivityDetection (SS:121-156) alone, one byte per block; argv[4] = BLOCK_LEN (default 512)
        const int B = argc > 4 ? atoi(argv[4]) : 512;
        for (size_t b = 0; b + B <= pcm.size(); b += B) {
            const unsigned char v = VoiceActivityDetection(&pcm[b], B) ? 1 : 0;
            fwrite(&v, 1, 1, out);
        }
    } else if (!strcmp(what, "conv")) {
        static double filt[8192][2];
        FILE *tf = fopen(argv[4], "rb");
        double v; int i = 0;
        while (fread(&v, 8, 1, tf) == 1 && i < 8192) filt[i++][0] = v;
        fclose(tf);
        short ob[1024];
        for (size_t b = 0; b + 1024 <= pcm.size(); b += 1024)
            if (AnalySisFreqDomain(&pcm[b], ob, 1024, filt)) fwrite(ob, 2, 1024, out);
    } else if (!strcmp(what, "mfcc")) {
        double feat[2][12];
        int it = 0;
        MelFilterBankInit();
        for (size_t b = 0; b + 1024 <= pcm.size(); b += 1024, it++)
            if (MFCCFeatureExtraction(&pcm[b], feat))
                for (int i = 0; i < 2; i++)
                    if (!(it == 0 && i == 0)) fwrite(feat[i], 8, 12, out);          // MFCC:94-101
    } else if (!strcmp(what, "pitch")) {
        for (size_t b = 0; b + 512 <= pcm.size(); b += 512) {
            CalcPitch(&pcm[b], 512);
            int a = JeicybooLastPitchArg();
            fwrite(&a, 4, 1, out);
        }
    } else if (!strcmp(what, "awgn")) {
        for (size_t b = 0; b + 512 <= pcm.size(); b += 512) {
            AnalysisAdditiveWhiteGaussianNoise(&pcm[b], 512);
            fwrite(JeicybooLastAutoCorrelation(), 8, 512, out);
        }
    } else if (!strcmp(what, "gmm") || !strcmp(what, "hmm")) {
        // in.raw = raw double[12] vectors of one utterance (an .mfc file), argv[4] = parameter records
        if (argc < 5) return 1;
        const size_t n = pcm.size() * sizeof(short) / sizeof(double) / 12;
        std::vector<double *> rows(n);
        for (size_t i = 0; i < n; i++) rows[i] = reinterpret_cast<double *>(pcm.data()) + 12 * i;
        FILE *fp = fopen(argv[4], "rb");
        if (!fp) { perror(argv[4]); return 1; }
        if (!strcmp(what, "gmm")) {
            // the class loop of GMMAlgorithm_Test_Auto_ver2.cpp:113-127
            GMMParameter g;
            double dMax = 0;
            int dArg = 0, u = 0;
            while (fread(&g, sizeof(g), 1, fp) == 1) {
                const double s = Recognition(rows.data(), &g, (int)n);
                if (u == 0) { dMax = s; dArg = 0; }
                else if (dMax < s) { dMax = s; dArg = u; }
                fwrite(&s, 8, 1, out);
                u++;
            }
            const double arg = dArg;
            fwrite(&arg, 8, 1, out);
        } else {
            HMMParameter hmm;
            if (fread(&hmm, sizeof(hmm), 1, fp) != 1) return 1;
            const double s = HMMRecognition(rows.data(), &hmm, (int)n);
            fwrite(&s, 8, 1, out);
        }
        fclose(fp);
    } else if (!strcmp(what, "mfccsteps")) {
        // in.raw = raw double rows of 512 magnitudes (dAbs).  Per row, the tail of MFCCFeatureExtraction's body
        // (:223-226): MelFilterBank, memset, DCT, Liftering -- and once more with a PRE-FILLED feature array to show
        // that DCT accumulates.  Output per row: 38 + 12 + 12 doubles.
        const size_t rows = pcm.size() * sizeof(short) / sizeof(double) / 512;
        MelFilterBankInit();
        for (size_t r = 0; r < rows; r++) {
            double *dAbs = reinterpret_cast<double *>(pcm.data()) + 512 * r;
            double mel[38], feat[12], pre[12];
            MelFilterBank(dAbs, mel);
            memset(feat, 0, sizeof(feat));
            DCT(mel, feat);
            Liftering(feat);
            for (int i = 0; i < 12; i++) pre[i] = 100.0 + i;
            DCT(mel, pre);
            fwrite(mel, 8, 38, out);
            fwrite(feat, 8, 12, out);
            fwrite(pre, 8, 12, out);
        }
    } else if (!strcmp(what, "prob")) {
        // in.raw = raw double[12] vectors, argv[4] = GMMParameter records: probability() of every vector under every
        // mixture of the first record, the way Recognition's inner loop calls it (GMMTest:155-157)
        if (argc < 5) return 1;
        const size_t n = pcm.size() * sizeof(short) / sizeof(double) / 12;
        FILE *fp = fopen(argv[4], "rb");
        GMMParameter g;
        if (!fp || fread(&g, sizeof(g), 1, fp) != 1) return 1;
        fclose(fp);
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) {
                const double p = probability(reinterpret_cast<double *>(pcm.data()) + 12 * i, g.mean[k], g.covariance[k], g.eigenVector[k]);
                fwrite(&p, 8, 1, out);
            }
    } else if (!strcmp(what, "dft")) {
        // DFTProcess / IDFTProcess / IFFTProcess (FFTAlgorithm_ver2.cpp:151-184) the way the commented-out
        // call sites (:74,:76) use them, n = argv[4] (default 512; any n, not only powers of two).  The output
        // arrays are PRE-FILLED with (k, -2k) instead of zeros: the reference accumulates into them.
        const int n = argc > 4 ? atoi(argv[4]) : 512;
        std::vector<COMPLEX> spec(n), acc(n);
        for (size_t blk = 0; blk + n <= pcm.size(); blk += n) {
            for (int k = 0; k < n; k++) { spec[k].real = k; spec[k].imag = -2.0 * k; }
            DFTProcess(&pcm[blk], spec.data(), n);
            fwrite(spec.data(), sizeof(COMPLEX), n, out);
            for (int k = 0; k < n; k++) { acc[k].real = k; acc[k].imag = -2.0 * k; }
            IDFTProcess(spec.data(), acc.data(), n);
            fwrite(acc.data(), sizeof(COMPLEX), n, out);
            for (int k = 0; k < n; k++) { acc[k].real = k; acc[k].imag = -2.0 * k; }
            IFFTProcess(spec.data(), acc.data(), n);
            fwrite(acc.data(), sizeof(COMPLEX), n, out);
        }
    } else if (!strcmp(what, "fft")) {
        std::vector<COMPLEX> a(512), b(512);
        short bits[512];
        for (size_t blk = 0; blk + 512 <= pcm.size(); blk += 512) {
            for (int i = 0; i < 512; i++) { a[i].real = pcm[blk + i]; a[i].imag = 0; }
            FFTProcess(a.data(), b.data(), 512, true);
            fwrite(b.data(), sizeof(COMPLEX), 512, out);
        }
        Bitrev(a.data(), bits, 512, b.data());
        fwrite(bits, 2, 512, out);
    }
    fclose(out);
    return 0;
}
