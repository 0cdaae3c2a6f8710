/*
 * jdsp_oracle.c -- CPU restatement of the JeicybooDSP FFT-based spectral path.
 * TEST INFRASTRUCTURE ONLY (see jdsp_oracle.h for the rules and pinning status).
 *
 * Plain C, IEEE double, single thread, structured per frame like the reference
 * (window and transcendental per-bin path recomputed every frame), so that it
 * also serves as the "port" CPU baseline timed by bench.py.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no fused multiply-adds, so
 * the FFTAlgorithm_ver2 restatement stays bit-identical to the reference build).
 */
#include "jdsp_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* FFTAlgorithm_ver2.cpp:15 */
#define PI_FFTALG 3.14159265358
/* SpectralSubtraction_final.cpp:52, WienerFilter_final.cpp:41, MFCC...:26 */
#define PI_APPS 3.141592

/* (short)double as x86-64 gcc/msvc do it: truncate to int32 (cvttsd2si; NaN and
 * out-of-int32-range give INT_MIN), keep the low 16 bits.  The reference's
 * out-of-range casts are undefined behaviour in C; this pins them. */
static short cast_i16(double v)
{
    int t;
    if (!(v > -2147483649.0 && v < 2147483648.0)) t = (int)0x80000000u;
    else t = (int)v;
    return (short)(unsigned short)((unsigned)t & 0xFFFFu);
}

static int ilog2(int n) { int b = 0; while ((1 << b) < n) b++; return b; }

/* ------------------------------------------------------------------------- */
/* FFTAlgorithm_ver2.cpp:186-207                                              */
void orc_bitrev_table(int n_fft, int block_len, short *table)
{
    /* :188 -- bit count from BLOCK_LEN; (int)log2(512.0) == 9 exactly */
    int bits = (int)log2((double)block_len);
    for (int k = 0; k < n_fft; k++) {
        short walk = (short)k;   /* :192 iTemp */
        short rev = walk;        /* :193 */
        for (int i = 1; i < bits; i++) {          /* :195-200, all in 16-bit */
            walk = (short)(walk >> 1);
            /* psBit[k] <<= 1 on a short: done on the unsigned 16-bit pattern (a negative left
             * operand is undefined in C; every compiler the reference met wraps) */
            rev = (short)(unsigned short)(((unsigned)(unsigned short)rev) << 1);
            rev = (short)(rev | (walk & 1));
        }
        table[k] = (short)(rev & (n_fft - 1));    /* :202 */
    }
}

/* FFTAlgorithm_ver2.cpp:94-149.  Same butterflies and the same twiddle
 * expressions; twiddles are evaluated once per (stage, n) instead of once per
 * element -- cos()/sin() are pure, so the bits are the same. */
void orc_fft_process(const orc_cplx *in, orc_cplx *out, int n_fft, int forward, int block_len)
{
    short *rev = (short *)malloc(sizeof(short) * (size_t)n_fft);
    orc_cplx *tw = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)n_fft);
    orc_bitrev_table(n_fft, block_len, rev);
    for (int k = 0; k < n_fft; k++) out[k] = in[rev[k]];        /* :204-205 */

    for (int groups = n_fft / 2;; groups /= 2) {                 /* iNpoint */
        int span = n_fft / groups;      /* iN2: 2,4,8..  */
        int half = span / 2;            /* iN1           */
        int next = span * 2;            /* iN3           */
        for (int g = 0; g < groups; g++) {                       /* :111-122 */
            orc_cplx *p = out + (size_t)span * g;
            for (int m = 0; m < half; m++) {
                orc_cplx a = p[m], b = p[m + half];
                p[m].re = a.re + b.re;
                p[m].im = a.im + b.im;
                p[m + half].re = a.re - b.re;
                p[m + half].im = a.im - b.im;
            }
        }
        if (groups == 1) break;                                  /* :124 */
        for (int n = 0; n < span; n++) {                         /* :135-142 */
            if (forward) {
                tw[n].re = cos(-2 * PI_FFTALG * n / (double)next);
                tw[n].im = sin(-2 * PI_FFTALG * n / (double)next);
            } else {
                tw[n].re = cos(2 * PI_FFTALG * n / (double)next);
                tw[n].im = sin(2 * PI_FFTALG * n / (double)next);
            }
        }
        for (int k = 0; k < groups / 2; k++) {                   /* :128-145 */
            orc_cplx *p = out + (size_t)k * next + span;
            for (int n = 0; n < span; n++) {
                orc_cplx t = p[n];
                p[n].re = tw[n].re * t.re - tw[n].im * t.im;
                p[n].im = tw[n].re * t.im + tw[n].im * t.re;
            }
        }
    }
    free(tw);
    free(rev);
}

/* FFTAlgorithm_ver2.cpp:162-173 */
void orc_dft_process(const short *in, orc_cplx *out, int n_fft)
{
    for (int k = 0; k < n_fft; k++)
        for (int i = 0; i < n_fft; i++) {
            out[k].re += in[i] * cos(2 * PI_FFTALG * i * k / (double)n_fft);
            out[k].im += in[i] * -sin(2 * PI_FFTALG * i * k / (double)n_fft);
        }
}

/* FFTAlgorithm_ver2.cpp:175-184 */
void orc_idft_process(const orc_cplx *in, orc_cplx *out, int n_fft)
{
    for (int k = 0; k < n_fft; k++)
        for (int i = 0; i < n_fft; i++) {
            double c = cos(2 * PI_FFTALG * i * k / (double)n_fft);
            double s = sin(2 * PI_FFTALG * i * k / (double)n_fft);
            out[k].re += (in[i].re * c - in[i].im * s);
            out[k].im += (in[i].re * s + in[i].im * c);
        }
}

/* FFTAlgorithm_ver2.cpp:151-160 */
void orc_ifft_process(const orc_cplx *in, orc_cplx *out, int n_fft)
{
    for (int k = 0; k < n_fft; k++)
        for (int i = 0; i < n_fft; i++) {
            double c = cos(2 * PI_FFTALG * i * k / (double)n_fft);
            double s = sin(2 * PI_FFTALG * i * k / (double)n_fft);
            out[k].re += (in[i].re * c - in[i].im * s) * 1 / (double)n_fft;
            out[k].im += (in[i].re * s + in[i].im * c) * 1 / (double)n_fft;
        }
}

/* FFTAlgorithm_ver2.cpp:62-86 */
void orc_fft_roundtrip_i16(const short *pcm, int n_blocks, int n_fft, short *out)
{
    orc_cplx *a = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    orc_cplx *b = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    for (int blk = 0; blk < n_blocks; blk++) {
        const short *src = pcm + (size_t)blk * n_fft;
        for (int i = 0; i < n_fft; i++) { a[i].re = src[i]; a[i].im = 0.0; }   /* :68-70 */
        orc_fft_process(a, b, n_fft, 1, n_fft);                                 /* :75 */
        orc_fft_process(b, a, n_fft, 0, n_fft);                                 /* :77 */
        for (int i = 0; i < n_fft; i++)
            out[(size_t)blk * n_fft + i] = cast_i16(a[i].re / (double)n_fft);   /* :80 */
    }
    free(a);
    free(b);
}

/* ------------------------------------------------------------------------- */
/* FFTW3 contract: fftw_plan_dft_1d(n, in, out, sign, FFTW_ESTIMATE) +
 * fftw_execute (SS:229-230,244-245; WF:192-193,215-216; 3D:139-143,154;
 * MFCC:216-217).  FFTW is not in the reference tree; this is its documented
 * definition, out[k] = sum_j in[j] exp(sign*2*pi*i*j*k/n), evaluated with an
 * iterative radix-2 transform whose twiddles come straight from cos/sin of the
 * true pi.  A small per-size twiddle cache plays the role of the plan. */
#define ORC_MAX_LOG2 20
static orc_cplx *g_tw[ORC_MAX_LOG2 + 1];
static unsigned *g_rev[ORC_MAX_LOG2 + 1];

static void dft_plan(int lg)
{
    int n = 1 << lg;
    if (g_tw[lg]) return;
    orc_cplx *tw = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)(n / 2 + 1));
    unsigned *rv = (unsigned *)malloc(sizeof(unsigned) * (size_t)n);
    for (int k = 0; k < n / 2; k++) {
        double ang = -2.0 * M_PI * (double)k / (double)n;
        tw[k].re = cos(ang);
        tw[k].im = sin(ang);
    }
    for (int k = 0; k < n; k++) {
        unsigned r = 0;
        for (int b = 0; b < lg; b++) r |= ((unsigned)(k >> b) & 1u) << (lg - 1 - b);
        rv[k] = r;
    }
    g_rev[lg] = rv;
    g_tw[lg] = tw;
}

void orc_dft_c2c(const orc_cplx *in, orc_cplx *out, int n, int sign)
{
    int lg = ilog2(n);
    dft_plan(lg);
    const orc_cplx *tw = g_tw[lg];
    const unsigned *rv = g_rev[lg];
    for (int k = 0; k < n; k++) out[k] = in[rv[k]];
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int base = 0; base < n; base += len)
            for (int j = 0; j < half; j++) {
                double wr = tw[j * step].re;
                double wi = sign < 0 ? tw[j * step].im : -tw[j * step].im;
                orc_cplx u = out[base + j], v = out[base + j + half];
                double tr = v.re * wr - v.im * wi;
                double ti = v.re * wi + v.im * wr;
                out[base + j].re = u.re + tr;
                out[base + j].im = u.im + ti;
                out[base + j + half].re = u.re - tr;
                out[base + j + half].im = u.im - ti;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* SS:226 / WF:189 / MFCC:213 */
void orc_hamming(int n, double *w)
{
    for (int i = 0; i < n; i++) w[i] = (0.54 - 0.46 * cos(2 * PI_APPS * i / (n - 1)));
}

/* SS:218-230 (== WF:181-193): [previous hop, current hop] -> *window -> forward */
static void windowed_forward(const short *frame, int n, orc_cplx *buf, orc_cplx *spec)
{
    for (int i = 0; i < n; i++) { buf[i].re = frame[i]; buf[i].im = 0.0; }
    for (int i = 0; i < n; i++)                                   /* window recomputed per frame, as the reference does */
        buf[i].re *= (0.54 - 0.46 * cos(2 * PI_APPS * i / (n - 1)));
    orc_dft_c2c(buf, spec, n, -1);
}

void orc_stft(const short *pcm, long n_frames, int n, int hop, orc_cplx *spec)
{
    orc_cplx *buf = (orc_cplx *)malloc(sizeof(orc_cplx) * (size_t)n);
    for (long f = 0; f < n_frames; f++)
        windowed_forward(pcm + (size_t)f * hop, n, buf, spec + (size_t)f * n);
    free(buf);
}

/* SS:121-156 == WF:261-296 */
int orc_vad_block(const short *block, int block_len, double *energy, int *zcr)
{
    const int n = 2 * block_len;                 /* FFT_PROCESSING_SIZE = KEEP_LEN + BLOCK_LEN */
    short *s = (short *)calloc((size_t)n + 1, sizeof(short));   /* s[n] = 0: defined value for the :139 over-read */
    double e = 0.0;
    int z = 0;
    memcpy(s + block_len, block, sizeof(short) * (size_t)block_len);   /* keep half stays zero (:127, :154 dead) */
    for (int i = 0; i < n; i++) {
        s[i] = (short)(s[i] * (0.54 - 0.46 * cos(2 * PI_APPS * i / (n - 1))));   /* :131 truncates back to short */
        e += pow(s[i], 2.0);                                                    /* :135 */
        if (s[i] * s[i + 1] < 0) z++;            /* :139: s[i+1] is not windowed yet */
    }
    e /= n;                                      /* :143 */
    if (energy) *energy = e;
    if (zcr) *zcr = z;
    free(s);
    return (e > 700.0 || z < 200.0) ? 1 : 0;     /* :147 */
}

/* ------------------------------------------------------------------------- */
/* BLOCK_LEN = KEEP_LEN and FFT_PROCESSING_SIZE = 2 BLOCK_LEN are macros in the reference (SS:53-55: 512 / 512 /
 * 1024); here they are a run-time parameter of the stream (BASELINE config 3 words them 256 / 256 / 512). */
#define DN_BLOCK_MAX 512
#define DN_FFT_MAX 1024
struct orc_denoise {
    int mode;
    int blk, nfft;               /* BLOCK_LEN (= KEEP_LEN), FFT_PROCESSING_SIZE */
    /* main(): SS:68-72 */
    int run_len;                 /* iNumOfIteration */
    short stash[DN_BLOCK_MAX];   /* rgsTempBuffer */
    double noise[DN_FFT_MAX];    /* rgdEstimatedNS */
    /* EstimateNoiseSpectrum statics: SS:161,164 */
    double avg[DN_FFT_MAX];
    short est_keep[DN_BLOCK_MAX];
    /* SpectralSubtraction / WienerFiltering statics: SS:202,208-209 */
    int calls;
    short keep[DN_BLOCK_MAX];
    double ola[DN_FFT_MAX];
    int last_voice;
    orc_cplx a[DN_FFT_MAX], b[DN_FFT_MAX];
};

orc_denoise *orc_denoise_create2(int mode, int block_len)
{
    if (block_len < 2 || block_len > DN_BLOCK_MAX || (block_len & (block_len - 1))) return NULL;
    orc_denoise *s = (orc_denoise *)calloc(1, sizeof(orc_denoise));
    s->mode = mode;
    s->blk = block_len;
    s->nfft = 2 * block_len;
    return s;
}
orc_denoise *orc_denoise_create(int mode) { return orc_denoise_create2(mode, 512); }
void orc_denoise_destroy(orc_denoise *s) { free(s); }
const double *orc_denoise_noise(const orc_denoise *s) { return s->noise; }
int orc_denoise_last_voice(const orc_denoise *s) { return s->last_voice; }

/* SS:159-198 == WF:120-159 */
static void estimate_noise(orc_denoise *s, const short *in)
{
    const int B = s->blk, N = s->nfft;
    short frame[DN_FFT_MAX];
    if (s->run_len == 2) memcpy(s->est_keep, s->stash, sizeof(short) * B);     /* :165-167 */
    memcpy(frame, s->est_keep, sizeof(short) * B);
    memcpy(frame + B, in, sizeof(short) * B);
    windowed_forward(frame, N, s->a, s->b);                                    /* :168-180 */
    for (int i = 0; i < N; i++) {                                              /* :182-187 */
        s->avg[i] += sqrt(s->b[i].re * s->b[i].re + s->b[i].im * s->b[i].im);
        if (s->run_len >= 3) s->avg[i] /= 2.0;
    }
    if (s->run_len == 10)                                                      /* :189-193 */
        memcpy(s->noise, s->avg, sizeof(double) * N);
    memcpy(s->est_keep, in, sizeof(short) * B);                                /* :195 */
}

int orc_denoise_block(orc_denoise *s, const short *in, short *out, double *ola_out)
{
    const int B = s->blk, N = s->nfft;
    short frame[DN_FFT_MAX];
    /* main loop, SS:98-109 */
    s->last_voice = orc_vad_block(in, B, NULL, NULL);
    if (!s->last_voice) {
        s->run_len++;
        if (s->run_len == 1) memcpy(s->stash, in, sizeof(short) * B);
        else estimate_noise(s, in);
    } else {
        s->run_len = 0;
    }
    /* SS:201-264 / WF:162-235 */
    s->calls++;
    if (s->calls == 1) {                                        /* :211-216 */
        memcpy(s->keep, in, sizeof(short) * B);
        return 0;
    }
    memcpy(frame, s->keep, sizeof(short) * B);
    memcpy(frame + B, in, sizeof(short) * B);
    windowed_forward(frame, N, s->a, s->b);                     /* :218-230 */
    for (int i = 0; i < N; i++) {
        double re = s->b[i].re, im = s->b[i].im;
        double ang = atan2(im, re);                             /* SS:234 / WF:197 */
        double amp;
        if (s->mode == ORC_SPECSUB) {
            amp = sqrt(re * re + im * im) - s->noise[i];        /* SS:238, no clamp */
        } else {
            double p = re * re + im * im;                       /* WF:201 */
            double r = (pow(s->noise[i], 2.0) / p);             /* WF:204 (0/0 -> NaN stays NaN) */
            if (r >= 1.0) r = 1.0;                              /* WF:205-207 */
            amp = fabs(sqrt(p)) * (1.0 - r);                    /* WF:208 */
        }
        s->a[i].re = amp * cos(ang);                            /* SS:240-241 / WF:211-212 */
        s->a[i].im = amp * sin(ang);
    }
    orc_dft_c2c(s->a, s->b, N, +1);                             /* SS:244-245 */
    for (int i = 0; i < N; i++) s->ola[i] += 1. / N * s->b[i].re;   /* SS:248 */
    for (int i = 0; i < B; i++) {
        out[i] = cast_i16(s->ola[i]);                           /* SS:252 */
        if (ola_out) ola_out[i] = s->ola[i];
    }
    memmove(s->ola, s->ola + B, sizeof(double) * B);            /* SS:255-256 */
    memset(s->ola + B, 0, sizeof(double) * B);
    memcpy(s->keep, in, sizeof(short) * B);                     /* SS:257 */
    return s->calls >= 3;                                       /* SS:260-263 */
}

long orc_denoise_stream2(int mode, int block_len, const short *pcm, long n_blocks, short *out, double *ola_out)
{
    orc_denoise *s = orc_denoise_create2(mode, block_len);
    short blk[DN_BLOCK_MAX];
    double pre[DN_BLOCK_MAX];
    long n_out = 0;
    if (!s) return -1;
    for (long b = 0; b < n_blocks; b++) {
        if (orc_denoise_block(s, pcm + (size_t)b * block_len, blk, pre)) {
            memcpy(out + (size_t)n_out * block_len, blk, sizeof(short) * block_len);
            if (ola_out) memcpy(ola_out + (size_t)n_out * block_len, pre, sizeof(double) * block_len);
            n_out++;
        }
    }
    orc_denoise_destroy(s);
    return n_out;
}

long orc_denoise_stream(int mode, const short *pcm, long n_blocks, short *out, double *ola_out)
{
    return orc_denoise_stream2(mode, 512, pcm, n_blocks, out, ola_out);
}

/* ------------------------------------------------------------------------- */
/* Fast_Convolution_Based_3DAudio_Impl.cpp:82-84,102-177 */
long orc_fastconv_stream(const short *pcm, long n_blocks, const double *taps, int n_taps,
                         int n_fft, short *out, double *pre_cast)
{
    const int block = n_fft - n_taps + 1;                    /* 1024 native */
    const int save = n_taps - 1;                             /* SAVE_LENGTH 7168 */
    const int n_hist = (save + block - 1) / block;           /* MAX_QUEUE_SIZE 7 */
    orc_cplx *x = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    orc_cplx *X = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    orc_cplx *h = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    orc_cplx *H = (orc_cplx *)calloc((size_t)n_fft, sizeof(orc_cplx));
    long n_out = 0;
    for (int i = 0; i < n_taps; i++) h[i].re = taps[i];      /* :82-84 */
    for (long b = n_hist; b < n_blocks; b++) {               /* calls 1..n_hist return false (:119-123) */
        long end = (b + 1) * (long)block;                    /* one past the newest sample */
        for (int i = 0; i < n_fft; i++) {                    /* :125-137 */
            long pos = end - n_fft + i;
            /* blocks queued by the first n_hist calls are uninitialised heap (:120): zero here */
            x[i].re = (pos >= (long)n_hist * block) ? (double)pcm[pos] : 0.0;
            x[i].im = 0.0;
        }
        orc_dft_c2c(x, X, n_fft, -1);                        /* :142 */
        orc_dft_c2c(h, H, n_fft, -1);                        /* :143 -- recomputed every block, as the reference */
        for (int i = 0; i < n_fft; i++) {                    /* :149-152 */
            double re = X[i].re * H[i].re - X[i].im * H[i].im;
            double im = X[i].re * H[i].im + X[i].im * H[i].re;
            x[i].re = re;
            x[i].im = im;
        }
        orc_dft_c2c(x, X, n_fft, +1);                        /* :154 */
        for (int i = 0; i < block; i++) {                    /* :156-158 */
            double v = X[i + n_taps - 1].re * 1. / n_fft;
            out[(size_t)n_out * block + i] = cast_i16(v);
            if (pre_cast) pre_cast[(size_t)n_out * block + i] = v;
        }
        n_out++;
    }
    free(x); free(X); free(h); free(H);
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* MFCCFeatureExtraction_auto_version1.cpp:23-33 */
void orc_mfcc_native_cfg(orc_mfcc_cfg *c)
{
    c->win_len = 1024; c->hop = 512; c->n_fft = 1024; c->n_bins = 512;
    c->n_chan = 38; c->n_cep = 12; c->lifter = 22;
    c->half_rate = 22050.0; c->preemph = 0.96;
}

/* :118-152 */
void orc_mel_init(const orc_mfcc_cfg *c, double *mel, int *fi, double *fb)
{
    const int C = c->n_chan, NB = c->n_bins;
    double unit = 1127.0 * log(1 + (c->half_rate / 700.0)) / (C + 1);      /* :124 */
    for (int i = 1; i <= C + 1; i++) {                                     /* :126-129 */
        double m = unit * i;
        mel[i - 1] = 700 * (exp(m / 1127.0) - 1.0);
    }
    for (int i = 0, k = 0; i < NB; i++) {                                  /* :131-137 */
        if ((i / (double)(NB - 1)) * c->half_rate > mel[k]) {
            if (k < C) k++;
        }
        fi[i] = k;
    }
    for (int i = 0; i < NB; i++) {                                         /* :139-150 */
        int k = fi[i];
        double f = (i / (double)(NB - 1)) * c->half_rate;
        if (k == 0) fb[i] = (mel[k] - f) / (mel[k] - 0);
        else fb[i] = (mel[k] - f) / (mel[k] - mel[k - 1]);
        if (fb[i] < 0) fb[i] = 0;
    }
}

/* :154-174 MelFilterBank: triangular weighting in bin order, then ln */
void orc_mel_filterbank(const orc_mfcc_cfg *c, const int *fi, const double *fb, const double *mag, double *mel)
{
    const int C = c->n_chan;
    double *m = (double *)calloc((size_t)C, sizeof(double));               /* rgdMelFiltered = {0} (:156) */
    for (int i = 0; i < c->n_bins; i++) {                                  /* :157-168 */
        int k = fi[i];
        if (k == 0) {
            m[k] += (1 - fb[i]) * mag[i];
        } else {
            m[k - 1] += fb[i] * mag[i];
            if (k != C) m[k] += (1 - fb[i]) * mag[i];
        }
    }
    for (int i = 0; i < C; i++) mel[i] = log(m[i]);                        /* :170-172 */
    free(m);
}

/* :176-183 DCT: ACCUMULATES into cep (the caller zeroes it, :224) */
void orc_dct(const orc_mfcc_cfg *c, const double *mel, double *cep)
{
    const int C = c->n_chan;
    for (int i = 1; i <= c->n_cep; i++)
        for (int k = 1; k <= C; k++)
            cep[i - 1] += sqrt(2.0 / C) * mel[k - 1] * cos(PI_APPS * i * (k - 0.5) / (double)C);
}

/* :185-192 Liftering, in place */
void orc_liftering(const orc_mfcc_cfg *c, double *cep)
{
    for (int i = 1; i <= c->n_cep; i++)
        cep[i - 1] = cep[i - 1] * (1 + 0.5 * c->lifter * sin(PI_APPS * i / c->lifter));
}

void orc_mfcc_frame(const orc_mfcc_cfg *c, const int *fi, const double *fb,
                    const short *frame, double *cep)
{
    const int W = c->win_len, N = c->n_fft, C = c->n_chan;
    orc_cplx *x = (orc_cplx *)calloc((size_t)N, sizeof(orc_cplx));
    orc_cplx *X = (orc_cplx *)calloc((size_t)N, sizeof(orc_cplx));
    double *mag = (double *)calloc((size_t)N, sizeof(double));
    double *m = (double *)calloc((size_t)C, sizeof(double));
    for (int i = 1; i < W; i++)                                            /* :208-210, x[0] stays 0 */
        x[i].re = frame[i] - c->preemph * frame[i - 1];
    for (int i = 0; i < W; i++)                                            /* :212-214 */
        x[i].re *= (0.54 - 0.46 * cos(2 * PI_APPS * i / (W - 1)));
    orc_dft_c2c(x, X, N, -1);                                              /* :216-217 */
    for (int i = 0; i < N; i++)                                            /* :218-220 */
        mag[i] = sqrt(pow(X[i].re, 2) + pow(X[i].im, 2));
    orc_mel_filterbank(c, fi, fb, mag, m);                                 /* :223 */
    for (int i = 0; i < c->n_cep; i++) cep[i] = 0.0;                       /* :224 memset */
    orc_dct(c, m, cep);                                                    /* :225 */
    orc_liftering(c, cep);                                                 /* :226 */
    free(x); free(X); free(mag); free(m);
}

/* :86-104 + :203-205,228 */
long orc_mfcc_stream(const orc_mfcc_cfg *c, const short *pcm, long n_blocks, double *feats)
{
    const int hop = c->hop, W = c->win_len;
    if (W != 2 * hop) return -1;
    int *fi = (int *)calloc((size_t)c->n_bins, sizeof(int));
    double *fb = (double *)calloc((size_t)c->n_bins, sizeof(double));
    double *mel = (double *)calloc((size_t)c->n_chan + 1, sizeof(double));
    short *padded = (short *)calloc((size_t)hop + (size_t)n_blocks * W, sizeof(short));
    long n_out = 0;
    orc_mel_init(c, mel, fi, fb);                                          /* :85 */
    memcpy(padded + hop, pcm, sizeof(short) * (size_t)n_blocks * W);       /* keep buffer starts at zero (:198) */
    for (long f = 1; f < 2 * n_blocks; f++) {                              /* very first vector skipped (:95-97) */
        orc_mfcc_frame(c, fi, fb, padded + (size_t)f * hop, feats + (size_t)n_out * c->n_cep);
        n_out++;
    }
    free(fi); free(fb); free(mel); free(padded);
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* PitchEstimation_method1.cpp:69-116 */
void orc_pitch_stream(const short *pcm, long n_blocks, int *arg, double *rmax, double *autocorr)
{
    orc_cplx *x = (orc_cplx *)calloc(1024, sizeof(orc_cplx));
    orc_cplx *X = (orc_cplx *)calloc(1024, sizeof(orc_cplx));
    short keep[512] = {0};                                         /* :74 */
    double r[512];
    for (long b = 0; b < n_blocks; b++) {
        const short *in = pcm + (size_t)b * 512;
        for (int i = 0; i < 512; i++) { x[i].re = keep[i]; x[i].im = 0; }          /* :79-81 */
        for (int i = 0; i < 512; i++) { x[512 + i].re = in[i]; x[512 + i].im = 0; } /* :82-84 */
        orc_dft_c2c(x, X, 1024, -1);                                                /* :88 */
        for (int i = 0; i < 1024; i++) {                                            /* :90-93 */
            x[i].re = X[i].re * X[i].re + X[i].im * X[i].im;
            x[i].im = 0;
        }
        orc_dft_c2c(x, X, 1024, +1);                                                /* :94 */
        for (int i = 0; i < 512; i++) r[i] = X[i].re * 1. / 1024;                   /* :95-97 */
        double best = r[511];                                                       /* :100 */
        int at = 0;
        for (int i = 511; i > 100; i--)                                             /* :102-108 */
            if (r[i] >= best) { at = i; best = r[i]; }
        arg[b] = at;
        rmax[b] = best;
        if (autocorr) memcpy(autocorr + (size_t)b * 512, r, sizeof(r));
        memcpy(keep, in, sizeof(keep));                                             /* :112 */
    }
    free(x);
    free(X);
}

/* ------------------------------------------------------------------------- */
/* BeamForming_MVDR_ver1.cpp */
#define MV_N 1024
#define MV_BLOCK 512
#define MV_KEEP 511

/* EstimateSpatialCorrMtx :244-270: one [previous block, block] frame per channel, ADDED to R (row-major) */
void orc_mvdr_estimate(const short *temp_l, const short *temp_r, double *R4)
{
    orc_cplx *fl = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx)), *FL = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx));
    orc_cplx *fr = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx)), *FR = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx));
    for (int i = 0; i < MV_N; i++) { fl[i].re = temp_l[i]; fl[i].im = 0; fr[i].re = temp_r[i]; fr[i].im = 0; }
    orc_dft_c2c(fl, FL, MV_N, -1);
    orc_dft_c2c(fr, FR, MV_N, -1);
    for (int i = 0; i < MV_N; i++) {                                                     /* :263-268 */
        R4[0] += (pow(FL[i].re, 2.0) + pow(FL[i].im, 2.0)) / MV_N;
        R4[1] += (-FL[i].re * FR[i].im + FL[i].im * FR[i].re) / MV_N;
        R4[2] += (-FR[i].re * FL[i].im + FR[i].im * FL[i].re) / MV_N;
        R4[3] += (pow(FR[i].re, 2.0) + pow(FR[i].im, 2.0)) / MV_N;
    }
    free(fl); free(FL); free(fr); free(FR);
}

/* BF:207-242 in full, as its printf shows it (:232): frame = [zeros(KEEP_LEN = n/2 - 1), block(n/2), 0]; every sample
 * is windowed and truncated back to short (:217), the energy summed (:221), the zero crossings counted against the
 * NOT-yet-windowed next sample (:225; the last pair reads one short past the array: defined as 0 here, and the sample
 * in front of it, frame[n-1], is 0 anyway); only the energy decides (:233). */
int orc_mvdr_vad_block(const short *block, int n_fft, double *energy, int *zcr)
{
    const int n = n_fft, B = n / 2, K = n / 2 - 1;
    short *s = (short *)calloc((size_t)n + 1, sizeof(short));
    double e = 0.0;
    int z = 0;
    memcpy(s + K, block, sizeof(short) * (size_t)B);
    for (int i = 0; i < n; i++) {
        s[i] = (short)(s[i] * (0.54 - 0.46 * cos(2 * PI_APPS * i / (n - 1))));
        e += pow(s[i], 2.0);
        if (s[i] * s[i + 1] < 0) z++;
    }
    e /= n;
    if (energy) *energy = e;
    if (zcr) *zcr = z;
    free(s);
    return e > 700.0;
}

/* ProcessMVDR's statics (:130-131,:137) */
struct orc_mvdr {
    double keep_l[MV_KEEP], keep_r[MV_KEEP];
    int count;
};
orc_mvdr *orc_mvdr_create(void) { return (orc_mvdr *)calloc(1, sizeof(orc_mvdr)); }
void orc_mvdr_destroy(orc_mvdr *s) { free(s); }

/* ProcessMVDR :124-205 for one block with the CALLER's matrix R4 (row-major).  Returns 1 when the reference
 * returns true (from the second call on); out / pre_cast are written on every call like the reference's
 * rgsOutputBuffer. */
int orc_mvdr_process_block(orc_mvdr *st, const short *L, const short *Rr, double d_time, const double *R4,
                           short *out, double *pre_cast)
{
    orc_cplx *fl = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx)), *FL = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx));
    orc_cplx *fr = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx)), *FR = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx));
    orc_cplx *mg = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx)), *MG = (orc_cplx *)calloc(MV_N, sizeof(orc_cplx));
    st->count++;
    for (int i = 0; i < MV_KEEP; i++) { fl[i].re = st->keep_l[i]; fr[i].re = st->keep_r[i]; }       /* :136-137 */
    for (int i = 0; i < MV_BLOCK; i++) { fl[i + MV_KEEP].re = L[i]; fr[i + MV_KEEP].re = Rr[i]; }   /* :138-141 */
    orc_dft_c2c(fl, FL, MV_N, -1);
    orc_dft_c2c(fr, FR, MV_N, -1);
    {
        /* mxAutoCorr.inverse() for a 2x2: adjugate times 1/det (complex scalars, zero imaginary parts) */
        double complex a = R4[0], bb = R4[1], c = R4[2], d = R4[3];
        double complex invdet = 1.0 / (a * d - bb * c);
        double complex i00 = d * invdet, i01 = -bb * invdet, i10 = -c * invdet, i11 = a * invdet;
        for (int i = 0; i < MV_N; i++) {
            double ang = 2 * PI_APPS * i * (16000.0 / MV_N) * d_time;                /* :164-165 */
            double complex s0 = 1.0, s1 = cos(ang) + I * sin(ang);
            double complex w0 = i00 * s0 + i01 * s1, w1 = i10 * s0 + i11 * s1;       /* :170 */
            double complex den = conj(s0) * w0 + conj(s1) * w1;                      /* :171 */
            w0 /= den;
            w1 /= den;
            double lw0 = creal(w0), lw1 = -cimag(w0), rw0 = creal(w1), rw1 = -cimag(w1);   /* :175-178 */
            /* :180-183 -- the imaginary part uses the ALREADY OVERWRITTEN real part */
            FL[i].re = FL[i].re * lw0 - FL[i].im * lw1;
            FL[i].im = FL[i].re * lw1 + FL[i].im * lw0;
            FR[i].re = FR[i].re * rw0 - FR[i].im * rw1;
            FR[i].im = FR[i].re * rw1 + FR[i].im * rw0;
            mg[i].re = FL[i].re + FR[i].re;                                          /* :184-185 */
            mg[i].im = FL[i].im + FR[i].im;
        }
    }
    orc_dft_c2c(mg, MG, MV_N, +1);                                                   /* :189-190 */
    for (int i = 0; i < MV_BLOCK; i++) {
        double v = MG[i + MV_KEEP].re * 1. / MV_N;                                   /* :193 */
        if (out) out[i] = cast_i16(v);
        if (pre_cast) pre_cast[i] = v;
    }
    for (int i = 0; i < MV_KEEP; i++) { st->keep_l[i] = fl[MV_KEEP + i].re; st->keep_r[i] = fr[MV_KEEP + i].re; }   /* :195-196 */
    free(fl); free(FL); free(fr); free(FR); free(mg); free(MG);
    return st->count > 1;                                                            /* :201-204 */
}

long orc_mvdr_stream(const short *left, const short *right, long n_blocks, double d_time,
                     short *out, double *pre_cast, double *corr4, double *corr_trace)
{
    short temp_l[2 * MV_BLOCK] = {0}, temp_r[2 * MV_BLOCK] = {0};      /* :111 */
    double R[4] = {0, 0, 0, 0};                                         /* :113 */
    short ob[MV_BLOCK];
    double pb[MV_BLOCK];
    orc_mvdr *st = orc_mvdr_create();
    int iter = 0;
    long n_out = 0;
    for (long b = 0; b < n_blocks; b++) {
        const short *L = left + (size_t)b * MV_BLOCK, *Rr = right + (size_t)b * MV_BLOCK;
        if (!orc_mvdr_vad_block(L, MV_N, NULL, NULL)) {                                            /* :191-211 */
            iter++;
            if (iter > 1) {
                memcpy(temp_l + MV_BLOCK, L, sizeof(short) * MV_BLOCK);
                memcpy(temp_r + MV_BLOCK, Rr, sizeof(short) * MV_BLOCK);
                orc_mvdr_estimate(temp_l, temp_r, R);
            }
            memcpy(temp_l, L, sizeof(short) * MV_BLOCK);
            memcpy(temp_r, Rr, sizeof(short) * MV_BLOCK);
        } else {
            iter = 0;
        }
        if (corr_trace) memcpy(corr_trace + 4 * b, R, sizeof(R));
        if (orc_mvdr_process_block(st, L, Rr, d_time, R, ob, pb)) {
            memcpy(out + (size_t)n_out * MV_BLOCK, ob, sizeof(ob));
            if (pre_cast) memcpy(pre_cast + (size_t)n_out * MV_BLOCK, pb, sizeof(pb));
            n_out++;
        }
    }
    if (corr4) memcpy(corr4, R, sizeof(R));
    orc_mvdr_destroy(st);
    return n_out;
}

/* ------------------------------------------------------------------------- */
/* Generalised MVDR (see jdsp_oracle.h): no reference counterpart. */
#define MVN_MAX 8
static int solve_cplx(int n, double complex A[MVN_MAX][MVN_MAX], double complex *b)
{
    for (int p = 0; p < n; p++) {
        int best = p;
        for (int r = p + 1; r < n; r++)
            if (cabs(A[r][p]) > cabs(A[best][p])) best = r;
        if (best != p) {
            for (int c = 0; c < n; c++) { double complex t = A[p][c]; A[p][c] = A[best][c]; A[best][c] = t; }
            double complex t = b[p]; b[p] = b[best]; b[best] = t;
        }
        double complex piv = A[p][p];
        for (int c = 0; c < n; c++) A[p][c] /= piv;
        b[p] /= piv;
        for (int r = 0; r < n; r++) {
            if (r == p) continue;
            double complex f = A[r][p];
            for (int c = 0; c < n; c++) A[r][c] -= f * A[p][c];
            b[r] -= f * b[p];
        }
    }
    return 0;
}

long orc_mvdrn_stream2(const short *pcm, long chan_stride, int n_mics, long n_blocks, const double *delays,
                       double loading, int n_fft, short *out, double *pre_cast)
{
    const int N = n_fft, B = n_fft / 2, K = n_fft / 2 - 1;       /* FFT_PROCESSING_LEN, BLOCK_LEN, KEEP_LEN (:38-40) */
    const int M = n_mics, NB = N / 2 + 1;
    if (N != 1024 && N != 512) return -1;
    orc_cplx *x = (orc_cplx *)calloc(N, sizeof(orc_cplx));
    orc_cplx *X = (orc_cplx *)calloc((size_t)M * N, sizeof(orc_cplx));
    orc_cplx *Y = (orc_cplx *)calloc(N, sizeof(orc_cplx)), *y = (orc_cplx *)calloc(N, sizeof(orc_cplx));
    double complex *R = (double complex *)calloc((size_t)NB * M * M, sizeof(double complex));
    int iter = 0, count = 0;
    long n_out = 0;
    for (long b = 0; b < n_blocks; b++) {
        const short *c0 = pcm + (size_t)b * B;
        if (!orc_mvdr_vad_block(c0, N, NULL, NULL)) {
            iter++;
            if (iter > 1) {                       /* frame = [block b-1, block b] of every microphone */
                for (int m = 0; m < M; m++) {
                    const short *s = pcm + (size_t)m * chan_stride + (size_t)(b - 1) * B;
                    for (int i = 0; i < N; i++) { x[i].re = s[i]; x[i].im = 0; }
                    orc_dft_c2c(x, X + (size_t)m * N, N, -1);
                }
                for (int k = 0; k < NB; k++)
                    for (int r = 0; r < M; r++)
                        for (int c = 0; c < M; c++) {
                            double complex xr = X[(size_t)r * N + k].re + I * X[(size_t)r * N + k].im;
                            double complex xc = X[(size_t)c * N + k].re + I * X[(size_t)c * N + k].im;
                            R[((size_t)k * M + r) * M + c] += xr * conj(xc) / N;
                        }
            }
        } else {
            iter = 0;
        }
        count++;
        for (int m = 0; m < M; m++) {
            const short *s = pcm + (size_t)m * chan_stride + (size_t)b * B;
            memset(x, 0, sizeof(orc_cplx) * N);
            if (b > 0) for (int i = 0; i < K; i++) x[i].re = s[i - B];      /* first KEEP_LEN samples of block b-1 */
            for (int i = 0; i < B; i++) x[i + K].re = s[i];
            orc_dft_c2c(x, X + (size_t)m * N, N, -1);
        }
        for (int k = 0; k < NB; k++) {
            double complex A[MVN_MAX][MVN_MAX], cv[MVN_MAX], w[MVN_MAX];
            double tr = 0;
            for (int r = 0; r < M; r++) tr += creal(R[((size_t)k * M + r) * M + r]);
            for (int r = 0; r < M; r++) {
                for (int c = 0; c < M; c++) A[r][c] = R[((size_t)k * M + r) * M + c];
                A[r][r] += loading * tr / M;
                double ang = 2 * PI_APPS * k * (16000.0 / N) * (delays ? delays[r] : 0.0);
                cv[r] = cos(ang) + I * sin(ang);
                w[r] = cv[r];
            }
            solve_cplx(M, A, w);
            double complex den = 0;
            for (int r = 0; r < M; r++) den += conj(cv[r]) * w[r];
            double complex acc = 0;
            for (int r = 0; r < M; r++) {
                double complex xr = X[(size_t)r * N + k].re + I * X[(size_t)r * N + k].im;
                acc += conj(w[r] / den) * xr;
            }
            Y[k].re = creal(acc); Y[k].im = cimag(acc);
            if (k > 0 && k < N / 2) { Y[N - k].re = creal(acc); Y[N - k].im = -cimag(acc); }
        }
        orc_dft_c2c(Y, y, N, +1);
        if (count > 1) {
            for (int i = 0; i < B; i++) {
                double v = y[i + K].re * 1. / N;
                out[(size_t)n_out * B + i] = cast_i16(v);
                if (pre_cast) pre_cast[(size_t)n_out * B + i] = v;
            }
            n_out++;
        }
    }
    free(x); free(X); free(Y); free(y); free(R);
    return n_out;
}

long orc_mvdrn_stream(const short *pcm, long chan_stride, int n_mics, long n_blocks, const double *delays,
                      double loading, short *out, double *pre_cast)
{
    return orc_mvdrn_stream2(pcm, chan_stride, n_mics, n_blocks, delays, loading, 1024, out, pre_cast);
}


/* ======================================================================================
 * GMM scoring / HMM recursion
 * ====================================================================================== */

/* GMMAlgorithm_Test_Auto_ver2.cpp:216-235 (the live #if 1 branch) = Viterbi_version1.cpp:248-267. */
double orc_gmm_probability(const double *x, const double *mean, const double *cov, const double *eig)
{
    double y[4];
    for (int j = 0; j < 4; j++) {                       /* InputMatrx * EigenMatrx, :228 */
        double a = 0.0;
        for (int i = 0; i < 12; i++) a += x[i] * eig[i * 4 + j];
        y[j] = a;
    }
    double p = 1.0;
    for (int i = 0; i < 4; i++) {                       /* :230-233 */
        const double c = cov[i * 12 + i];
        const double d = y[i] - mean[i];
        p *= (1.0 / sqrt(2.0 * PI_APPS)) * (1.0 / sqrt(c)) * exp((-1 / 2.0) * (d * d) / c);
    }
    return p;
}

static double gmm_mixture(const double *x, const orc_gmm_param *g)
{
    double t = 0.0;
    for (int k = 0; k < 4; k++)                         /* GMMTest:155-157, Viterbi:183-185,:193-195 */
        t += g->alpa[k] * orc_gmm_probability(x, g->mean[k], &g->covariance[k][0][0], &g->eigenVector[k][0][0]);
    return t;
}

/* GMMTest:151-162 */
double orc_gmm_recognition(const double *feats, long n_frames, const orc_gmm_param *g)
{
    double acc = 0.0;
    for (long i = 0; i < n_frames; i++) acc += log(gmm_mixture(feats + 12 * i, g));
    return acc / (double)n_frames;
}

/* GMMTest:113-127 */
int orc_gmm_classify(const double *feats, long n_frames, const orc_gmm_param *classes, int n_classes,
                     double *scores)
{
    double best = 0.0;
    int arg = 0;
    for (int u = 0; u < n_classes; u++) {
        const double s = orc_gmm_recognition(feats, n_frames, &classes[u]);
        if (scores) scores[u] = s;
        if (u == 0) { best = s; arg = 0; }
        else if (best < s) { best = s; arg = u; }
    }
    return arg;
}

/* Viterbi:157-246 */
double orc_hmm_viterbi(const double *feats, long n_frames, const orc_hmm_param *h, int *path, double *trellis)
{
    enum { S = 6 };
    double prev[S], cur[S];
    double ret = 0.0;                                   /* dTempProb's initial value, :159 */
    if (path) memset(path, 0, sizeof(int) * (size_t)(n_frames > 0 ? n_frames : 0));
    /* the arg-max of :209-225 only reads column i of the trellis, so it is taken as each column is finished;
     * the reference walks i downwards and returns the value found last, i.e. at i = 1 */
    for (long i = 0; i < n_frames; i++) {
        const double *x = feats + 12 * i;
        if (i == 0) {
            for (int m = 0; m < S; m++)                 /* :182-187 */
                cur[m] = log(gmm_mixture(x, &h->gMMParam[m])) + log(1.0 / (double)S);
        } else {
            for (int m = 0; m < S; m++) {               /* :190-206 */
                const double b = gmm_mixture(x, &h->gMMParam[m]);
                for (int u = 0; u < S; u++) {
                    const double t = log(prev[u]) + log(h->transProb[u][m]) + log(b);
                    if (u == 0) cur[m] = t;
                    else if (cur[m] < t) cur[m] = t;
                }
            }
            double best = cur[0];                       /* :212-221 */
            int arg = 0;
            for (int m = 1; m < S; m++)
                if (cur[m] > best) { best = cur[m]; arg = m; }
            if (path) path[i] = arg;
            if (i == 1) ret = best;
        }
        for (int m = 0; m < S; m++) {
            if (trellis) trellis[(size_t)m * n_frames + i] = cur[m];
            prev[m] = cur[m];
        }
    }
    return ret;
}
