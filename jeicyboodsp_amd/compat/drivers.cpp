// drivers.cpp -- the reference programs' command lines over the BATCHED C ABI: each program
// reads its whole input like the reference's fread() loop would, makes ONE engine call per
// stream and writes the reference's output format.  Built as nine executables (Makefile):
//
//   jdsp_fftalg   in.wav  out.raw            FFTAlgorithm_ver2.cpp main()            (:30-92)
//   jdsp_specsub  in.raw  out.raw            SpectralSubtraction_final.cpp main()    (:62-119)
//   jdsp_wiener   in.raw  out.raw            WienerFilter_final.cpp main()           (:52-118)
//   jdsp_conv3d   in.wav  out.raw  taps.f64  Fast_Convolution_Based_3DAudio_Impl.cpp main() (:53-100);
//                                            taps.f64 = raw little-endian doubles (the reference
//                                            compiles FilterCoefficient.h in; n_taps = size/8)
//   jdsp_mfcc     list.txt                   MFCCFeatureExtraction_auto_version1.cpp main() (:44-114)
//   jdsp_mvdr     left.wav right.wav out.raw BeamForming_MVDR_ver1.cpp main()        (:47-122)
//   jdsp_pitch1   in.wav                     PitchEstimation_method1.cpp main()      (:33-67); prints like :109
//   jdsp_gmmtest  test_list.txt params.bin   GMMAlgorithm_Test_Auto_ver2.cpp main()  (:46-149)
//   jdsp_viterbi  test_list.txt params.bin   Viterbi_version1.cpp main()             (:51-155)
//
// File conventions kept from the reference: raw little-endian int16 PCM; a 44-byte WAV header
// is skipped by fftalg/conv3d/mfcc (FFT:59, 3D:79, MFCC:83) and NOT by specsub/wiener
// (SS:89, WF:81 commented out); a short final fread() is processed with the stale tail of the
// previous block still in the buffer (e.g. SS:94: the loop only stops when fread returns 0).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jdsp.h"

static jdsp_ctx *g_ctx = nullptr;
static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, jdsp_last_error(g_ctx));
    exit(2);
}
#define CK(call) do { if ((call) != JDSP_OK) die(#call); } while (0)

// The reference's `while (fread(buf, 2, BLOCK, f) != 0)` loop: returns whole blocks, the last
// one completed with whatever the buffer still held.
static std::vector<short> read_blocks(FILE *f, int block, long header_bytes)
{
    std::vector<short> all, buf(block, 0);
    if (header_bytes) { std::vector<char> h(header_bytes); if (fread(h.data(), 1, header_bytes, f) == 0) return all; }
    while (fread(buf.data(), sizeof(short), block, f) != 0) all.insert(all.end(), buf.begin(), buf.end());
    return all;
}

static FILE *open_or_die(const char *p, const char *mode)
{
    FILE *f = fopen(p, mode);
    if (!f) { fprintf(stderr, "%s File Open Error: %s\n", mode[0] == 'r' ? "Read" : "Write", p); exit(1); }
    return f;
}

static int run_fftalg(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: jdsp_fftalg in.wav out.raw\n"); return 1; }
    FILE *in = open_or_die(argv[1], "rb"), *out = open_or_die(argv[2], "wb");
    const int N = 512;                                                     // BLOCK_LEN (:16)
    std::vector<short> pcm = read_blocks(in, N, 44);
    const long nb = (long)pcm.size() / N;
    std::vector<double> a(2 * pcm.size(), 0.0), b(2 * pcm.size());
    for (size_t i = 0; i < pcm.size(); i++) a[2 * i] = pcm[i];            // :68-70
    CK(jdsp_fft_process_f64(g_ctx, a.data(), b.data(), N, nb, 1));        // :75
    CK(jdsp_fft_process_f64(g_ctx, b.data(), a.data(), N, nb, 0));        // :77
    std::vector<short> res(pcm.size());
    for (size_t i = 0; i < pcm.size(); i++) res[i] = (short)(a[2 * i] / (double)N);   // :80
    fwrite(res.data(), sizeof(short), res.size(), out);
    fclose(in); fclose(out);
    printf("Processing End\n");
    return 0;
}

static int run_denoise(int mode, int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.raw out.raw\n", argv[0]); return 1; }
    FILE *in = open_or_die(argv[1], "rb"), *out = open_or_die(argv[2], "wb");
    std::vector<short> pcm = read_blocks(in, 512, 0);
    const long nb = (long)pcm.size() / 512;
    jdsp_denoise *h = nullptr;
    CK(jdsp_denoise_create(g_ctx, mode, &h));
    std::vector<short> res((size_t)(nb > 0 ? nb : 1) * 512);
    long n_out = 0;
    CK(jdsp_denoise_process(h, pcm.data(), nb, res.data(), nullptr, &n_out));
    fwrite(res.data(), sizeof(short), (size_t)n_out * 512, out);
    jdsp_denoise_destroy(h);
    fclose(in); fclose(out);
    printf("Processing End\n");
    return 0;
}

static int run_conv3d(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: jdsp_conv3d in.wav out.raw taps.f64\n"); return 1; }
    FILE *in = open_or_die(argv[1], "rb"), *out = open_or_die(argv[2], "wb"), *tf = open_or_die(argv[3], "rb");
    std::vector<double> taps;
    double v;
    while (fread(&v, sizeof(double), 1, tf) == 1) taps.push_back(v);
    fclose(tf);
    const int n_fft = 8192;                                               // FFT_PROCESSING_SIZE (:48)
    if (taps.empty() || (int)taps.size() > n_fft) { fprintf(stderr, "taps: 1..8192 doubles expected\n"); return 1; }
    jdsp_fastconv *h = nullptr;
    CK(jdsp_fastconv_create(g_ctx, taps.data(), (int)taps.size(), 1, n_fft, &h));
    const int block = jdsp_fastconv_block_len(h);                         // BLOCK_SIZE 1024 for 7169 taps
    std::vector<short> pcm = read_blocks(in, block, 44);
    const long nb = (long)pcm.size() / block;
    std::vector<short> res((size_t)(nb > 0 ? nb : 1) * block);
    long n_out = 0;
    CK(jdsp_fastconv_process(h, pcm.data(), nb, res.data(), nullptr, &n_out));
    fwrite(res.data(), sizeof(short), (size_t)n_out * block, out);
    jdsp_fastconv_destroy(h);
    fclose(in); fclose(out);
    printf("Processing End\n");
    return 0;
}

static int run_mfcc(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: jdsp_mfcc list.txt   (lines: in.wav out.mfc)\n"); return 1; }
    FILE *list = open_or_die(argv[1], "rb");
    jdsp_mfcc_cfg cfg;
    jdsp_mfcc_native_cfg(&cfg);
    jdsp_mfcc *h = nullptr;
    CK(jdsp_mfcc_create(g_ctx, &cfg, &h));
    // One run = one continuous stream: the keep buffer and the "skip the very first vector" flag
    // are static/never reset between files (MFCC:95,198), so file k>0 starts with the last 512
    // samples of file k-1's last block and emits 2B vectors; the first file emits 2B-1.
    std::vector<short> keep(512, 0);
    bool first = true;
    char a[255], b[255];
    while (fscanf(list, "%254s %254s", a, b) == 2) {                      // :72
        FILE *in = open_or_die(a, "rb"), *out = open_or_die(b, "wb");
        std::vector<short> pcm = read_blocks(in, 1024, 44);              // :83,:88
        const long nb = (long)pcm.size() / 1024;
        if (nb > 0) {
            std::vector<short> stream(keep);
            stream.insert(stream.end(), pcm.begin(), pcm.end());
            const long skip = first ? 1 : 0;                              // :95-97
            const long nf = 2 * nb - skip;
            std::vector<double> feats((size_t)(nf > 0 ? nf : 1) * 12);
            if (nf > 0) {
                CK(jdsp_mfcc_frames(h, stream.data() + 512 * skip, (long)stream.size() - 512 * skip, nullptr, nf, feats.data()));
                fwrite(feats.data(), sizeof(double), (size_t)nf * 12, out);   // :99
            }
            keep.assign(pcm.end() - 512, pcm.end());                      // :228
            first = false;
        }
        fclose(in); fclose(out);
    }
    jdsp_mfcc_destroy(h);
    fclose(list);
    printf("Processing End\n");
    return 0;
}

static int run_mvdr(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: jdsp_mvdr left.wav right.wav out.raw\n"); return 1; }
    FILE *fl = open_or_die(argv[1], "rb"), *fr = open_or_die(argv[2], "rb"), *out = open_or_die(argv[3], "wb");
    std::vector<short> l = read_blocks(fl, 512, 44), r = read_blocks(fr, 512, 44);      // :80-81,:85-93
    const long nb = (long)std::min(l.size(), r.size()) / 512;                           // the loop stops at the shorter file
    jdsp_mvdr *h = nullptr;
    CK(jdsp_mvdr_create(g_ctx, 0.0, &h));                                               // dAngle = 0 (:60)
    std::vector<short> res((size_t)(nb > 0 ? nb : 1) * 512);
    long n_out = 0;
    CK(jdsp_mvdr_process(h, l.data(), r.data(), nb, res.data(), nullptr, &n_out));
    fwrite(res.data(), sizeof(short), (size_t)n_out * 512, out);
    jdsp_mvdr_destroy(h);
    fclose(fl); fclose(fr); fclose(out);
    printf("Processing End\n");
    return 0;
}

static int run_pitch1(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: jdsp_pitch1 in.wav\n"); return 1; }
    FILE *in = open_or_die(argv[1], "rb");
    std::vector<short> pcm = read_blocks(in, 512, 44);                                  // :53,:57
    const long nb = (long)pcm.size() / 512;
    std::vector<int32_t> arg((size_t)(nb > 0 ? nb : 1));
    std::vector<float> rmax((size_t)(nb > 0 ? nb : 1));
    CK(jdsp_pitch_autocorr(g_ctx, pcm.data(), nb, nullptr, arg.data(), rmax.data(), nullptr));
    for (long b = 0; b < nb; b++)
        printf("Estimation arg %d , dMin %f pitch %f \n", arg[b], (double)rmax[b], 16000.0 / (double)arg[b]);   // :109
    fclose(in);
    printf("Processing End\n");
    return 0;
}

// ---- GMMAlgorithm_Test_Auto_ver2.cpp / Viterbi_version1.cpp ------------------------------------------
// Both mains read: argv[1] = a text file naming NUM_OF_CLASS class list files, each naming .mfc files of raw
// double[12] vectors (what jdsp_mfcc writes); argv[2] = NUM_OF_CLASS parameter records.  The reference's
// `while (!feof(f)) fscanf(f, "%s", name)` loops run once more after the last name when the file ends in a
// newline and then dereference a NULL FILE*; here a list simply ends at its last name.
static std::vector<std::string> read_names(const char *path)
{
    std::vector<std::string> names;
    FILE *f = open_or_die(path, "rb");
    char tok[255];
    while (fscanf(f, "%254s", tok) == 1) names.push_back(tok);
    fclose(f);
    return names;
}

struct MfcBatch {
    std::vector<double> feats;
    std::vector<int64_t> first{0};
    std::vector<int> list_of;                 // which class list (0-based) each utterance came from
};

static MfcBatch read_mfc_lists(const std::vector<std::string> &lists, size_t from, size_t count)
{
    MfcBatch b;
    for (size_t i = 0; i < count; i++) {
        for (const std::string &mfc : read_names(lists[from + i].c_str())) {
            FILE *f = open_or_die(mfc.c_str(), "rb");
            fseek(f, 0L, SEEK_END);
            const long n = ftell(f) / (long)sizeof(double) / 12;                        // GMMTest:98-101
            fseek(f, 0L, SEEK_SET);
            const size_t at = b.feats.size();
            b.feats.resize(at + (size_t)n * 12);
            if (n > 0 && fread(&b.feats[at], sizeof(double), (size_t)n * 12, f) != (size_t)n * 12) die("short .mfc read");
            fclose(f);
            b.first.push_back(b.first.back() + n);
            b.list_of.push_back((int)i);
        }
    }
    return b;
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int run_gmmtest(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: jdsp_gmmtest test_list.txt params.bin\n"); return 1; }
    for (int i = 1; i < 3; i++) printf("%d-th path %s \n", i, argv[i]);                 // GMMTest:62-63
    const int C = env_int("JDSP_NUM_OF_CLASS", 25);                                     // GMMTest:26
    const std::vector<std::string> lists = read_names(argv[1]);
    FILE *fp = open_or_die(argv[2], "rb");
    for (size_t from = 0; from + C <= lists.size(); from += C) {                        // the outer while of :72
        std::vector<jdsp_gmm_param> params(C);
        if (fread(params.data(), sizeof(jdsp_gmm_param), C, fp) != (size_t)C) break;    // :74-78
        jdsp_gmm *h = nullptr;
        CK(jdsp_gmm_create(g_ctx, params.data(), C, &h));
        MfcBatch b = read_mfc_lists(lists, from, C);
        const long n_utts = (long)b.list_of.size();
        std::vector<double> scores((size_t)(n_utts > 0 ? n_utts : 1) * C);
        std::vector<int> best((size_t)(n_utts > 0 ? n_utts : 1));
        CK(jdsp_gmm_score(h, b.feats.data(), b.first.data(), n_utts, scores.data(), best.data()));
        for (long u = 0; u < n_utts; u++) {
            for (int c = 0; c < C; c++) printf(" %d-th class probability %f \n", c + 1, scores[(size_t)u * C + c]);   // :125
            printf(" %d -th result %d \n", b.list_of[u] + 1, best[u] + 1);                                          // :127
        }
        jdsp_gmm_destroy(h);
    }
    fclose(fp);
    return 0;
}

static int run_viterbi(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: jdsp_viterbi test_list.txt params.bin\n"); return 1; }
    for (int i = 1; i < 3; i++) printf("%d-th path %s \n", i, argv[i]);                 // Viterbi:66-67
    const int C = env_int("JDSP_NUM_OF_CLASS", 1);                                      // Viterbi:26
    const std::vector<std::string> lists = read_names(argv[1]);
    FILE *fp = open_or_die(argv[2], "rb");
    for (size_t from = 0; from + C <= lists.size(); from += C) {
        std::vector<jdsp_hmm_param> params(C);
        if (fread(params.data(), sizeof(jdsp_hmm_param), C, fp) != (size_t)C) break;    // :78-81
        jdsp_hmm *h = nullptr;
        CK(jdsp_hmm_create(g_ctx, params.data(), C, &h));
        MfcBatch b = read_mfc_lists(lists, from, C);
        const long n_utts = (long)b.list_of.size(), nf = (long)b.first.back();
        std::vector<double> scores((size_t)(n_utts > 0 ? n_utts : 1) * C), trellis((size_t)C * 6 * (nf > 0 ? nf : 1));
        std::vector<int> best((size_t)(n_utts > 0 ? n_utts : 1)), path((size_t)C * (nf > 0 ? nf : 1));
        CK(jdsp_hmm_viterbi(h, b.feats.data(), b.first.data(), n_utts, scores.data(), best.data(), path.data(), trellis.data()));
        for (long u = 0; u < n_utts; u++) {
            const long a = (long)b.first[u], e = (long)b.first[u + 1];
            for (int c = 0; c < C; c++) {                                               // HMMRecognition's printing
                for (long i = e - 1; i > a; i--) {                                      // :209-222
                    const int st = path[(size_t)c * nf + i];
                    printf("max accumulated prob %f \n", trellis[((size_t)c * 6 + st) * nf + i]);
                }
                printf("decoding result ! \n");                                         // :227
                // :228-230 prints the doubles of dDecodingReslt through %d (undefined); the states are printed here
                for (long i = a; i + 1 < e; i++) printf("%d ,", path[(size_t)c * nf + i]);
                printf("\n");
                printf(" %d-th class probability %f \n", c + 1, scores[(size_t)u * C + c]);   // :127
            }
            printf(" %d -th result %d \n", b.list_of[u] + 1, best[u] + 1);                    // :129
        }
        jdsp_hmm_destroy(h);
    }
    fclose(fp);
    return 0;
}

int main(int argc, char **argv)
{
    std::string prog = argv[0];
    const size_t slash = prog.find_last_of('/');
    if (slash != std::string::npos) prog = prog.substr(slash + 1);
    const char *dev = getenv("JDSP_DEVICE");
    if (jdsp_create(dev ? atoi(dev) : 0, &g_ctx) != JDSP_OK) die("jdsp_create");
    int rc = 1;
    if (prog == "jdsp_fftalg") rc = run_fftalg(argc, argv);
    else if (prog == "jdsp_specsub") rc = run_denoise(JDSP_SPECSUB, argc, argv);
    else if (prog == "jdsp_wiener") rc = run_denoise(JDSP_WIENER, argc, argv);
    else if (prog == "jdsp_conv3d") rc = run_conv3d(argc, argv);
    else if (prog == "jdsp_mfcc") rc = run_mfcc(argc, argv);
    else if (prog == "jdsp_mvdr") rc = run_mvdr(argc, argv);
    else if (prog == "jdsp_pitch1") rc = run_pitch1(argc, argv);
    else if (prog == "jdsp_gmmtest") rc = run_gmmtest(argc, argv);
    else if (prog == "jdsp_viterbi") rc = run_viterbi(argc, argv);
    else fprintf(stderr, "unknown program name %s (expected jdsp_fftalg|jdsp_specsub|jdsp_wiener|jdsp_conv3d|jdsp_mfcc|jdsp_mvdr|jdsp_pitch1|jdsp_gmmtest|jdsp_viterbi)\n", prog.c_str());
    jdsp_destroy(g_ctx);
    return rc;
}
