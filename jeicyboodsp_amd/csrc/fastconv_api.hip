// fastconv_api.hip -- C ABI of the overlap-save convolver (Fast_Convolution_Based_3DAudio_Impl.cpp).
#include "jdsp_internal.h"

using jdsp::fail;

#ifndef JDSP_STAMP
#define JDSP_STAMP 0
#endif
#if JDSP_STAMP
namespace jdsp { int read_wave_stamps(unsigned long long *host, int n); }
#endif
extern "C" {

int jdsp_fastconv_create(jdsp_ctx *ctx, const double *taps, int n_taps, int n_filters, int n_fft, jdsp_fastconv **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    if ((n_fft != 1024 && n_fft != 8192) || !taps || n_taps < 1 || n_taps > n_fft || n_filters < 1 || n_filters > 64)
        return fail(ctx, JDSP_EINVAL, "jdsp_fastconv_create: unsupported configuration (n_fft 1024 or 8192, 1 <= n_taps <= n_fft)");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    jdsp_fastconv *h = new (std::nothrow) jdsp_fastconv();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_fastconv_create");
    h->ctx = ctx;
    h->n_fft = n_fft;
    h->n_taps = n_taps;
    h->n_filters = n_filters;
    h->block = n_fft - n_taps + 1;                                   // BLOCK_SIZE 1024 = 8192 - 7169 + 1
    h->n_hist = (n_taps - 1 + h->block - 1) / h->block;              // MAX_QUEUE_SIZE 7
    const size_t n = (size_t)n_filters * n_fft;
    // H = FFT(h zero-padded to n_fft) (:82-84,:140,:143), once, in double on the device
    std::vector<double> hpad(2 * n, 0.0);
    for (int f = 0; f < n_filters; f++)
        for (int i = 0; i < n_taps; i++) hpad[2 * ((size_t)f * n_fft + i)] = taps[(size_t)f * n_taps + i];
    double *d_h = nullptr, *d_H = nullptr;
    hipError_t e = hipMalloc((void **)&d_h, n * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&d_H, n * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&h->H, n * sizeof(float2));
    const int hl = n_taps - 1 > 0 ? n_taps - 1 : 1;
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&h->hist[i], (size_t)hl * sizeof(short));
    if (e == hipSuccess) e = hipMemcpy(d_h, hpad.data(), n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_fft == 8192 && !ctx->conv_tw4096) {
        std::vector<float2> a(4096), b(4096);
        jdsp::fill_conv_twiddles(a.data(), b.data());
        e = hipMalloc((void **)&ctx->conv_tw4096, 4096 * sizeof(float2));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->conv_tw8192, 4096 * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(ctx->conv_tw4096, a.data(), 4096 * sizeof(float2), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ctx->conv_tw8192, b.data(), 4096 * sizeof(float2), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_create: alloc", e);
    if (!rc) rc = jdsp_fft_process_f64_dev(ctx, d_h, d_H, n_fft, n_filters, 1);
    if (!rc && jdsp::launch_spectrum_to_f32(ctx->stream, (const double2 *)d_H, h->H, (long)n,
                                            n_fft == 1024 ? 1.0f / 2048.0f : 1.0f))   // 1024: the kernel's 1/2 and 1/1024
        rc = fail(ctx, JDSP_EHIP, "spectrum_to_f32 launch", hipGetLastError());
    if (!rc && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_create: sync", e);
    if (d_h) (void)hipFree(d_h);
    if (d_H) (void)hipFree(d_H);
    // Uniformly partitioned form of the same convolution (fastconv_kernels.hip): spectra of the 512-tap
    // partitions, each zero-padded to 1024, bins 0..512.  JDSP_FASTCONV_PARTITIONED=0 keeps the 8192-point kernel.
    const char *env = getenv("JDSP_FASTCONV_PARTITIONED");
    if (!rc && n_fft == 8192 && h->block % 512 == 0 && !(env && env[0] == '0')) {
        const int P = (n_taps + 511) / 512;
        const size_t rows = (size_t)n_filters * P;
        std::vector<double> part(rows * 1024 * 2, 0.0);
        for (int f = 0; f < n_filters; f++)
            for (int i = 0; i < n_taps; i++)
                part[2 * (((size_t)f * P + i / 512) * 1024 + i % 512)] = taps[(size_t)f * n_taps + i];
        double *d_p = nullptr, *d_P = nullptr;
        e = hipMalloc((void **)&d_p, rows * 1024 * 16);
        if (e == hipSuccess) e = hipMalloc((void **)&d_P, rows * 1024 * 16);
        if (e == hipSuccess) e = hipMalloc((void **)&h->Hp, rows * jdsp::kUpolsRowPitch * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(d_p, part.data(), rows * 1024 * 16, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_create: partitions", e);
        if (!rc) rc = jdsp::ensure_stft1024_table_rect(ctx);
        if (!rc) rc = jdsp_fft_process_f64_dev(ctx, d_p, d_P, 1024, (long)rows, 1);
        if (!rc && jdsp::launch_spectrum_rows_to_f32(ctx->stream, (const double2 *)d_P, h->Hp, (long)rows))
            rc = fail(ctx, JDSP_EHIP, "spectrum_rows_to_f32 launch", hipGetLastError());
        if (!rc && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_create: sync", e);
        if (d_p) (void)hipFree(d_p);
        if (d_P) (void)hipFree(d_P);
        if (!rc) h->n_part = P;
    }
    if (!rc) rc = jdsp_fastconv_reset(h);
    if (rc) {
        jdsp_fastconv_destroy(h);
        return rc;
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_fastconv_destroy(jdsp_fastconv *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->H) (void)hipFree(h->H);
    if (h->Hp) (void)hipFree(h->Hp);
    if (h->staged) (void)hipFree(h->staged);
    if (h->X) (void)hipFree(h->X);
    for (int i = 0; i < 2; i++)
        if (h->hist[i]) (void)hipFree(h->hist[i]);
    delete h;
    return JDSP_OK;
}

int jdsp_fastconv_reset(jdsp_fastconv *h)
{
    if (!h) return JDSP_EINVAL;
    const int hl = h->n_taps - 1 > 0 ? h->n_taps - 1 : 1;
    for (int i = 0; i < 2; i++) JDSP_HIP(h->ctx, hipMemsetAsync(h->hist[i], 0, (size_t)hl * sizeof(short), h->ctx->stream));
    h->calls = 0;
    h->cur = 0;
    return JDSP_OK;
}

int jdsp_fastconv_set_position(jdsp_fastconv *h, long blocks_consumed)
{
    if (!h || blocks_consumed < 0) return JDSP_EINVAL;
    int rc = jdsp_fastconv_reset(h);                 // history = silence; the caller feeds the halo blocks itself
    if (rc) return rc;
    h->calls = blocks_consumed;
    return JDSP_OK;
}

int jdsp_fastconv_reserve(jdsp_fastconv *h, long n_blocks)
{
    if (!h || n_blocks < 0) return JDSP_EINVAL;
    if (!h->n_part) return JDSP_OK;                  // only the partitioned path has a workspace
    jdsp_ctx *ctx = h->ctx;
    const long samples = n_blocks * h->block;
    if (samples <= h->ws_samples) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h->staged) (void)hipFree(h->staged);
    if (h->X) (void)hipFree(h->X);
    h->staged = nullptr; h->X = nullptr; h->ws_samples = 0;
    const long total = 512L * h->n_part + samples;
    JDSP_HIP(ctx, hipMalloc((void **)&h->staged, (size_t)total * sizeof(short)));
    JDSP_HIP(ctx, hipMalloc((void **)&h->X, (size_t)(total / 512) * jdsp::kUpolsRowPitch * sizeof(float2)));
    h->ws_samples = samples;
    return JDSP_OK;
}

int jdsp_fastconv_block_len(const jdsp_fastconv *h) { return h ? h->block : 0; }
int jdsp_fastconv_hist_blocks(const jdsp_fastconv *h) { return h ? h->n_hist : 0; }

long jdsp_fastconv_blocks_out(const jdsp_fastconv *h, long n_blocks)
{
    if (!h || n_blocks < 0) return 0;
    const long first = h->calls >= h->n_hist ? 0 : h->n_hist - h->calls;      // :119-123
    return n_blocks > first ? n_blocks - first : 0;
}

int jdsp_fastconv_process_dev(jdsp_fastconv *h, const int16_t *pcm_dev, long n_blocks, int16_t *out_dev,
                              float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_fastconv_process: n_blocks < 0");
    const long n_out = jdsp_fastconv_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_dev || (n_out > 0 && !out_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_fastconv_process: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    jdsp::ConvStream s;
    s.pcm = pcm_dev;
    s.hist = h->hist[h->cur];
    s.n_samples = n_blocks * h->block;
    s.global0 = h->calls * h->block;
    s.valid_from = (long)h->n_hist * h->block;
    s.hist_len = h->n_taps - 1;
    const int first = (int)(n_blocks - n_out);
    if (h->n_part) {
        int rc = jdsp_fastconv_reserve(h, n_blocks);
        if (rc) return rc;
        if (jdsp::launch_fastconv_upols(ctx->stream, s, n_out, first, h->block, h->n_part, h->n_filters, h->Hp,
                                        ctx->stft1024_table_rect, h->staged, h->X, out_dev, precast_dev, n_out * h->block,
                                        h->hist[h->cur ^ 1]))
            return fail(ctx, JDSP_EHIP, "fastconv (partitioned) launch", hipGetLastError());
    } else if (jdsp::launch_fastconv(ctx->stream, h->n_fft, s, n_out, first, h->block, h->n_taps, h->n_filters, h->H,
                              ctx->stft1024_table, ctx->conv_tw4096, ctx->conv_tw8192, out_dev, precast_dev,
                              n_out * h->block, h->hist[h->cur ^ 1]))
        return fail(ctx, JDSP_EHIP, "fastconv launch", hipGetLastError());
    h->cur ^= 1;
    h->calls += n_blocks;
    return JDSP_OK;
}

int jdsp_fastconv_process(jdsp_fastconv *h, const int16_t *pcm_host, long n_blocks, int16_t *out_host,
                          float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_fastconv_process: n_blocks < 0");
    const long n_out = jdsp_fastconv_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_fastconv_process: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_b = (size_t)n_blocks * h->block * 2;
    const size_t out_n = (size_t)(n_out > 0 ? n_out : 1) * h->block * h->n_filters;
    int16_t *d_in = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_n * 2);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_n * 4);
    int rc = JDSP_OK;
    if (e != hipSuccess) rc = fail(ctx, JDSP_ENOMEM, "jdsp_fastconv_process: hipMalloc", e);
    if (!rc && (e = hipMemcpyAsync(d_in, pcm_host, in_b, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_process: H2D", e);
    if (!rc) rc = jdsp_fastconv_process_dev(h, d_in, n_blocks, d_out, d_pre, nullptr);
    const size_t got = (size_t)n_out * h->block * h->n_filters;
    if (!rc && got && (e = hipMemcpyAsync(out_host, d_out, got * 2, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_process: D2H", e);
    if (!rc && got && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, got * 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_process: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_fastconv_process: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    return rc;
}

#if JDSP_STAMP
/* diagnostic build only (tools/wave_timeline.py): (start, end, where) of the last convolver launch's first n waves */
int jdsp_debug_wave_stamps(unsigned long long *host, int n) { return jdsp::read_wave_stamps(host, n); }
#endif
}  // extern "C"
