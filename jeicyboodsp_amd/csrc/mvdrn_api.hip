// mvdrn_api.hip -- C ABI of the generalised (n_mics <= 8, per-bin covariance) MVDR beamformer.
#include "jdsp_internal.h"

using jdsp::fail;

static void mvdrn_free_ws(jdsp_mvdrn *h)
{
    void *p[] = {h->flags, h->events, h->ev_n, h->ver_base, h->snap_mask, h->spec, h->weights, h->chunk_ws};
    for (void *q : p)
        if (q) (void)hipFree(q);
    h->flags = nullptr;
    h->events = h->ev_n = h->ver_base = nullptr;
    h->snap_mask = nullptr;
    h->spec = h->weights = nullptr;
    h->chunk_ws = nullptr;
    h->chunk_cap = 0;
    h->cap_blocks = 0;
}

extern "C" {

int jdsp_mvdrn_create(jdsp_ctx *ctx, int n_mics, const double *delays_s, double loading, jdsp_mvdrn **out)
{
    return jdsp_mvdrn_create_cfg(ctx, n_mics, delays_s, loading, 1024, out);
}

int jdsp_mvdrn_block_len(const jdsp_mvdrn *h) { return h ? h->block : 0; }

int jdsp_mvdrn_create_cfg(jdsp_ctx *ctx, int n_mics, const double *delays_s, double loading, int n_fft, jdsp_mvdrn **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    if (n_mics < 2 || n_mics > 8 || !(loading >= 0)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_create: 2 <= n_mics <= 8, loading >= 0");
    if (n_fft != 1024 && n_fft != 512) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_create_cfg: n_fft must be 1024 or 512");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    jdsp_mvdrn *h = new (std::nothrow) jdsp_mvdrn();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_mvdrn_create");
    h->ctx = ctx;
    h->n_mics = n_mics;
    h->loading = loading;
    h->n_fft = n_fft;
    h->block = n_fft / 2;
    h->n_bins = n_fft / 2 + 1;
    const int nb = h->n_bins, keep = n_fft / 2 - 1;
    std::vector<double2> steer((size_t)513 * 8, make_double2(0.0, 0.0));
    for (int k = 0; k < nb; k++)
        for (int m = 0; m < n_mics; m++) {
            // the reference's steering phase (BeamForming_MVDR_ver1.cpp:164-165) per microphone delay
            const double ang = 2 * 3.141592 * k * (16000.0 / n_fft) * (delays_s ? delays_s[m] : 0.0);
            steer[(size_t)k * 8 + m] = make_double2(cos(ang), sin(ang));
        }
    double w[512] = {0};
    for (int i = 0; i < h->block; i++) w[i] = (0.54 - 0.46 * cos(2 * 3.141592 * (keep + i) / (n_fft - 1)));   // :217
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipMalloc((void **)&h->cov[i], sizeof(double2) * 513 * 64);
        if (e == hipSuccess) e = hipMalloc((void **)&h->prev[i], sizeof(short) * 512 * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&h->run_len[i], sizeof(int));
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->plan, sizeof(jdsp::DenoisePlan));
    if (e == hipSuccess) e = hipMalloc((void **)&h->steer, sizeof(double2) * steer.size());
    if (e == hipSuccess) e = hipMalloc((void **)&h->w_vad, sizeof(w));
    if (e == hipSuccess) e = hipMemcpy(h->steer, steer.data(), sizeof(double2) * steer.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->w_vad, w, sizeof(w), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jdsp_mvdrn_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_mvdrn_create: alloc", e);
    }
    rc = jdsp_mvdrn_reset(h);
    if (rc) {
        jdsp_mvdrn_destroy(h);
        return rc;
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_mvdrn_destroy(jdsp_mvdrn *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    mvdrn_free_ws(h);
    for (int i = 0; i < 2; i++) {
        if (h->cov[i]) (void)hipFree(h->cov[i]);
        if (h->prev[i]) (void)hipFree(h->prev[i]);
        if (h->run_len[i]) (void)hipFree(h->run_len[i]);
    }
    if (h->plan) (void)hipFree(h->plan);
    if (h->steer) (void)hipFree(h->steer);
    if (h->w_vad) (void)hipFree(h->w_vad);
    delete h;
    return JDSP_OK;
}

int jdsp_mvdrn_reset(jdsp_mvdrn *h)
{
    if (!h) return JDSP_EINVAL;
    hipStream_t s = h->ctx->stream;
    for (int i = 0; i < 2; i++) {
        JDSP_HIP(h->ctx, hipMemsetAsync(h->cov[i], 0, sizeof(double2) * 513 * 64, s));
        JDSP_HIP(h->ctx, hipMemsetAsync(h->prev[i], 0, sizeof(short) * 512 * 8, s));
        JDSP_HIP(h->ctx, hipMemsetAsync(h->run_len[i], 0, sizeof(int), s));
    }
    h->calls = 0;
    h->cur = 0;
    return JDSP_OK;
}

long jdsp_mvdrn_blocks_out(const jdsp_mvdrn *h, long n_blocks)
{
    if (!h || n_blocks < 0) return 0;
    const long first = h->calls >= 1 ? 0 : 1;
    return n_blocks > first ? n_blocks - first : 0;
}

static int mvdrn_reserve(jdsp_mvdrn *h, long n_blocks)
{
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks <= h->cap_blocks) return JDSP_OK;
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mvdrn_free_ws(h);
    const size_t n = (size_t)n_blocks;
    hipError_t e = hipMalloc((void **)&h->flags, n);
    if (e == hipSuccess) e = hipMalloc((void **)&h->events, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ev_n, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ver_base, (n / 64 + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->snap_mask, (n / 64 + 1) * sizeof(unsigned long long));
    // worst case: every block is an estimation frame -- one spectrum set and one weight set per block
    if (e == hipSuccess) e = hipMalloc((void **)&h->spec, n * h->n_mics * (size_t)h->n_bins * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&h->weights, (n + 1) * (size_t)h->n_bins * 8 * sizeof(float2));
    // the chunked covariance update's sums and entering matrices: a call of n blocks has at most min(n, kMvnChunks) chunks
    // (1 KB per bin and chunk: 134 MB at 128 chunks of 513 bins -- a per-block caller gets 1 MB)
    const int chunk_cap = (int)(n < (size_t)jdsp::kMvnChunks ? n : (size_t)jdsp::kMvnChunks);
    if (e == hipSuccess) e = hipMalloc((void **)&h->chunk_ws, sizeof(double2) * 2 * (size_t)chunk_cap * (size_t)h->n_bins * 64);
    if (e != hipSuccess) {
        mvdrn_free_ws(h);
        return fail(ctx, JDSP_ENOMEM, "jdsp_mvdrn: workspace", e);
    }
    h->cap_blocks = n_blocks;
    h->chunk_cap = chunk_cap;
    return JDSP_OK;
}

int jdsp_mvdrn_process_dev(jdsp_mvdrn *h, const int16_t *pcm_dev, long chan_stride, long n_blocks, int16_t *out_dev,
                           float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0 || chan_stride < n_blocks * h->block) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_process: bad sizes");
    const long n_out = jdsp_mvdrn_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_dev || (n_out > 0 && !out_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_process: NULL buffer");
    if (((uintptr_t)pcm_dev & 15u) || (chan_stride & 7))
        return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_process: pcm must be 16-byte aligned and chan_stride a multiple of 8");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mvdrn_reserve(h, n_blocks);
    if (rc) return rc;
    const int in = h->cur, ou = h->cur ^ 1;
    hipStream_t s = ctx->stream;
    if (h->n_fft == 512) {
        if (jdsp::launch_vad256(s, pcm_dev, n_blocks, h->w_vad, h->flags, nullptr, nullptr, 0) ||
            jdsp::launch_run_plan(s, h->flags, n_blocks, h->run_len[in], h->run_len[ou], 0, h->ver_base, h->snap_mask,
                                  h->events, h->ev_n, h->plan) ||
            jdsp::launch_mvdrn512(s, pcm_dev, chan_stride, h->n_mics, n_blocks, h->calls, h->prev[in], h->prev[ou], h->events,
                                  h->plan, h->ver_base, h->snap_mask, h->spec, h->cov[in], h->cov[ou], h->steer, h->loading,
                                  h->weights, ctx->stft1024_table, out_dev, precast_dev, h->chunk_ws, h->chunk_cap))
            return fail(ctx, JDSP_EHIP, "mvdrn512 launch", hipGetLastError());
        h->cur ^= 1;
        h->calls += n_blocks;
        return JDSP_OK;
    }
    if (jdsp::launch_vad(s, pcm_dev, n_blocks, h->w_vad, 0, h->flags, nullptr, nullptr) ||
        jdsp::launch_run_plan(s, h->flags, n_blocks, h->run_len[in], h->run_len[ou], 0, h->ver_base, h->snap_mask, h->events,
                              h->ev_n, h->plan) ||
        jdsp::launch_mvdrn(s, pcm_dev, chan_stride, h->n_mics, n_blocks, h->calls, h->prev[in], h->prev[ou], h->events,
                           h->plan, h->ver_base, h->snap_mask, h->spec, h->cov[in], h->cov[ou], h->steer, h->loading,
                           h->weights, ctx->stft1024_table, out_dev, precast_dev, h->chunk_ws, h->chunk_cap))
        return fail(ctx, JDSP_EHIP, "mvdrn launch", hipGetLastError());
    h->cur ^= 1;
    h->calls += n_blocks;
    return JDSP_OK;
}

int jdsp_mvdrn_process(jdsp_mvdrn *h, const int16_t *pcm_host, long chan_stride, long n_blocks, int16_t *out_host,
                       float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0 || chan_stride < n_blocks * h->block) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_process: bad sizes");
    const long n_out = jdsp_mvdrn_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdrn_process: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const long stride_dev = n_blocks * h->block;               // repack the planes tightly (a multiple of 8)
    const size_t in_b = (size_t)stride_dev * h->n_mics * 2, out_b = (size_t)(n_out > 0 ? n_out : 1) * h->block * 2;
    int16_t *d_in = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_b * 2);
    hipStream_t s = ctx->stream;
    int rc = JDSP_OK;
    for (int m = 0; m < h->n_mics && e == hipSuccess; m++)
        e = hipMemcpyAsync(d_in + (size_t)m * stride_dev, pcm_host + (size_t)m * chan_stride, (size_t)stride_dev * 2,
                           hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdrn_process: staging", e);
    if (!rc) rc = jdsp_mvdrn_process_dev(h, d_in, stride_dev, n_blocks, d_out, d_pre, nullptr);
    if (!rc && n_out > 0 && (e = hipMemcpyAsync(out_host, d_out, (size_t)n_out * h->block * 2, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdrn_process: D2H", e);
    if (!rc && n_out > 0 && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, (size_t)n_out * h->block * 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdrn_process: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdrn_process: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    return rc;
}

}  // extern "C"
