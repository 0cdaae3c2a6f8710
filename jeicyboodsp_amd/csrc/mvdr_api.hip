// mvdr_api.hip -- C ABI of the two-microphone MVDR beamformer (BeamForming_MVDR_ver1.cpp).
#include "jdsp_internal.h"

using jdsp::fail;

static void mvdr_free_ws(jdsp_mvdr *h)
{
    void *p[] = {h->flags, h->events, h->ev_n, h->ver_base, h->snap_mask, h->delta, h->rver, h->tile_sums, h->wtab};
    for (void *q : p)
        if (q) (void)hipFree(q);
    h->flags = nullptr;
    h->events = h->ev_n = h->ver_base = nullptr;
    h->snap_mask = nullptr;
    h->delta = h->rver = h->tile_sums = nullptr;
    h->wtab = nullptr;
    h->cap_blocks = 0;
}

extern "C" {

int jdsp_mvdr_create(jdsp_ctx *ctx, double d_time, jdsp_mvdr **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    jdsp_mvdr *h = new (std::nothrow) jdsp_mvdr();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_mvdr_create");
    h->ctx = ctx;
    h->d_time = d_time;
    std::vector<double2> steer(1024);
    for (int i = 0; i < 1024; i++) {                                           // :164-165
        const double ang = 2 * 3.141592 * i * (16000.0 / 1024) * d_time;
        steer[i] = make_double2(cos(ang), sin(ang));
    }
    double w[512];
    for (int i = 0; i < 512; i++) w[i] = (0.54 - 0.46 * cos(2 * 3.141592 * (511 + i) / (1024 - 1)));   // :217, frame offset 511
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&h->st[i], sizeof(jdsp::MvdrState));
    if (e == hipSuccess) e = hipMalloc((void **)&h->plan, sizeof(jdsp::DenoisePlan));
    if (e == hipSuccess) e = hipMalloc((void **)&h->steer, sizeof(double2) * 1024);
    if (e == hipSuccess) e = hipMalloc((void **)&h->w_vad, sizeof(w));
    if (e == hipSuccess) e = hipMalloc((void **)&h->sh_range, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->sh_zero_run, sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->sh_zero_run, 0, sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(h->steer, steer.data(), sizeof(double2) * 1024, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->w_vad, w, sizeof(w), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jdsp_mvdr_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_mvdr_create: alloc", e);
    }
    rc = jdsp_mvdr_reset(h);
    if (rc) {
        jdsp_mvdr_destroy(h);
        return rc;
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_mvdr_destroy(jdsp_mvdr *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    mvdr_free_ws(h);
    for (int i = 0; i < 2; i++)
        if (h->st[i]) (void)hipFree(h->st[i]);
    if (h->plan) (void)hipFree(h->plan);
    if (h->steer) (void)hipFree(h->steer);
    if (h->w_vad) (void)hipFree(h->w_vad);
    if (h->sh_range) (void)hipFree(h->sh_range);
    if (h->sh_zero_run) (void)hipFree(h->sh_zero_run);
    delete h;
    return JDSP_OK;
}

int jdsp_mvdr_reset(jdsp_mvdr *h)
{
    if (!h) return JDSP_EINVAL;
    for (int i = 0; i < 2; i++) JDSP_HIP(h->ctx, hipMemsetAsync(h->st[i], 0, sizeof(jdsp::MvdrState), h->ctx->stream));
    h->calls = 0;
    h->cur = 0;
    return JDSP_OK;
}

long jdsp_mvdr_blocks_out(const jdsp_mvdr *h, long n_blocks)
{
    if (!h || n_blocks < 0) return 0;
    const long first = h->calls >= 1 ? 0 : 1;                                  // :201-204
    return n_blocks > first ? n_blocks - first : 0;
}

static int mvdr_reserve(jdsp_mvdr *h, long n_blocks)
{
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks <= h->cap_blocks) return JDSP_OK;
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mvdr_free_ws(h);
    const size_t n = (size_t)n_blocks;
    hipError_t e = hipMalloc((void **)&h->flags, n);
    if (e == hipSuccess) e = hipMalloc((void **)&h->events, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ev_n, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ver_base, (n / 64 + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->snap_mask, (n / 64 + 1) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&h->delta, n * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->rver, (n + 1) * 4 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&h->tile_sums, (n / 1024 + 1) * 4 * sizeof(double));
    // a call has at most as many events as blocks: the weight table needs no more rows than that
    const size_t wrows = n + 1 < (size_t)jdsp::kMvdrTableVersions ? n + 1 : (size_t)jdsp::kMvdrTableVersions;
    if (e == hipSuccess) e = hipMalloc((void **)&h->wtab, sizeof(float4) * wrows * 1024);
    if (e != hipSuccess) {
        mvdr_free_ws(h);
        return fail(ctx, JDSP_ENOMEM, "jdsp_mvdr: workspace", e);
    }
    h->cap_blocks = n_blocks;
    return JDSP_OK;
}

int jdsp_mvdr_process_dev(jdsp_mvdr *h, const int16_t *left_dev, const int16_t *right_dev, long n_blocks,
                          int16_t *out_dev, float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_process: n_blocks < 0");
    const long n_out = jdsp_mvdr_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!left_dev || !right_dev || (n_out > 0 && !out_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_process: NULL buffer");
    if (((uintptr_t)left_dev & 15u) || ((uintptr_t)right_dev & 15u))
        return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_process: inputs must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mvdr_reserve(h, n_blocks);
    if (rc) return rc;
    const jdsp::MvdrState *st_in = h->st[h->cur];
    jdsp::MvdrState *st_out = h->st[h->cur ^ 1];
    hipStream_t s = ctx->stream;
    if (jdsp::launch_vad(s, left_dev, n_blocks, h->w_vad, 0, h->flags, nullptr, nullptr) ||
        jdsp::launch_run_plan(s, h->flags, n_blocks, &st_in->run_len, &st_out->run_len, 0, h->ver_base, h->snap_mask,
                              h->events, h->ev_n, h->plan) ||
        jdsp::launch_mvdr(s, left_dev, right_dev, n_blocks, h->calls, st_in, st_out, h->events, h->plan, h->ver_base,
                          h->snap_mask, h->delta, h->rver, h->steer, ctx->stft1024_table, out_dev, precast_dev, h->wtab, h->tile_sums))
        return fail(ctx, JDSP_EHIP, "mvdr launch", hipGetLastError());
    h->cur ^= 1;
    h->calls += n_blocks;
    return JDSP_OK;
}

int jdsp_mvdr_process(jdsp_mvdr *h, const int16_t *left_host, const int16_t *right_host, long n_blocks,
                      int16_t *out_host, float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_process: n_blocks < 0");
    const long n_out = jdsp_mvdr_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!left_host || !right_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_process: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_b = (size_t)n_blocks * 1024, out_b = (size_t)(n_out > 0 ? n_out : 1) * 1024;
    int16_t *d_l = nullptr, *d_r = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipError_t e = hipMalloc((void **)&d_l, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_b * 2);
    hipStream_t s = ctx->stream;
    int rc = JDSP_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(d_l, left_host, in_b, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r, right_host, in_b, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_process: staging", e);
    if (!rc) rc = jdsp_mvdr_process_dev(h, d_l, d_r, n_blocks, d_out, d_pre, nullptr);
    if (!rc && n_out > 0 && (e = hipMemcpyAsync(out_host, d_out, (size_t)n_out * 1024, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_process: D2H", e);
    if (!rc && n_out > 0 && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, (size_t)n_out * 2048, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_process: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_process: sync", e);
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    return rc;
}

/* ---- multi-GPU: one rank's share of one stereo stream ------------------------------------------ */
int jdsp_mvdr_shard_vad_dev(jdsp_mvdr *h, const int16_t *left_ext_dev, const int16_t *right_ext_dev, long ext0, long b0,
                            long b1, long n_total, uint8_t *flags_own_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (!(0 <= ext0 && ext0 <= b0 && b0 <= b1 && b1 <= n_total) || (b0 >= 1 ? ext0 != b0 - 1 : ext0 != 0))
        return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_shard_vad: need ext0 = max(b0-1, 0) <= b0 <= b1 <= n_total");
    if (b1 > b0 && (!left_ext_dev || !right_ext_dev || !flags_own_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_shard_vad: NULL buffer");
    if (((uintptr_t)left_ext_dev & 15u) || ((uintptr_t)right_ext_dev & 15u))
        return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_shard_vad: inputs must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mvdr_reserve(h, n_total);
    if (!rc) rc = jdsp_mvdr_reset(h);
    if (rc) return rc;
    h->sh_ext0 = ext0; h->sh_b0 = b0; h->sh_b1 = b1; h->sh_total = n_total;
    h->sh_left = left_ext_dev; h->sh_right = right_ext_dev;
    if (jdsp::launch_vad(ctx->stream, left_ext_dev + (b0 - ext0) * 512, b1 - b0, h->w_vad, 0, flags_own_dev, nullptr, nullptr))
        return fail(ctx, JDSP_EHIP, "vad launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_mvdr_shard_summary_dev(jdsp_mvdr *h, const uint8_t *flags_all_dev, double *sum4_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (!flags_all_dev || !sum4_dev) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_shard_summary: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (jdsp::launch_run_plan(s, flags_all_dev, h->sh_total, h->sh_zero_run, nullptr, 0, h->ver_base, h->snap_mask, h->events,
                              h->ev_n, h->plan) ||
        jdsp::launch_mvdr_shard_summary(s, h->sh_left, h->sh_right, h->sh_b1 - h->sh_ext0, h->sh_ext0, h->sh_b0, h->sh_b1,
                                        h->st[h->cur], h->events, h->plan, h->ver_base, h->snap_mask, ctx->stft1024_table,
                                        h->sh_range, h->delta, sum4_dev, h->tile_sums))
        return fail(ctx, JDSP_EHIP, "mvdr shard summary launch", hipGetLastError());
    return JDSP_OK;
}

long jdsp_mvdr_shard_blocks_out(const jdsp_mvdr *h)
{
    if (!h) return 0;
    const long lo = h->sh_b0 > 1 ? h->sh_b0 : 1;
    return h->sh_b1 > lo ? h->sh_b1 - lo : 0;
}

int jdsp_mvdr_shard_finish_dev(jdsp_mvdr *h, const double *sums_all_dev, int world, int rank, int16_t *out_dev,
                               float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    const long n_out = jdsp_mvdr_shard_blocks_out(h);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (!sums_all_dev || world < 1 || rank < 0 || rank >= world || (n_out > 0 && !out_dev))
        return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_shard_finish: bad argument");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (jdsp::launch_mvdr_shard_finish(ctx->stream, h->sh_left, h->sh_right, h->sh_b1 - h->sh_ext0, h->sh_ext0, h->sh_b0,
                                       h->sh_b1, h->st[h->cur], h->st[h->cur ^ 1], h->plan, h->ver_base, h->snap_mask,
                                       h->sh_range, h->delta, sums_all_dev, rank, h->rver, h->steer, ctx->stft1024_table,
                                       out_dev, precast_dev, h->tile_sums))
        return fail(ctx, JDSP_EHIP, "mvdr shard finish launch", hipGetLastError());
    return JDSP_OK;
}

/* ---- the program's two helper functions on their own (compat: EstimateSpatialCorrMtx, ProcessMVDR) ---------- */
int jdsp_mvdr_estimate_corr(jdsp_mvdr *h, const int16_t *left_frames_host, const int16_t *right_frames_host, long n_frames,
                            double *corr4_inout_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_frames < 0 || n_frames > (1L << 28)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_estimate_corr: n_frames");
    if (n_frames == 0) return JDSP_OK;
    if (!left_frames_host || !right_frames_host || !corr4_inout_host) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_estimate_corr: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mvdr_reserve(h, 2 * n_frames);
    if (rc) return rc;
    // frame i = blocks (2i, 2i+1) of a 2 n_frames-block stream; "event" i = block 2i+1 with its predecessor (:250-253)
    std::vector<int> ev((size_t)n_frames);
    for (long i = 0; i < n_frames; i++) ev[(size_t)i] = (int)(2 * i + 1);
    const jdsp::DenoisePlan plan = {(int)n_frames, 0, 0, 0};
    const size_t in_b = (size_t)n_frames * 2048;
    int16_t *d_l = nullptr, *d_r = nullptr;
    double *d_tot = nullptr;
    hipStream_t s = ctx->stream;
    hipError_t e = hipMalloc((void **)&d_l, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_tot, 4 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_l, left_frames_host, in_b, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r, right_frames_host, in_b, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h->events, ev.data(), sizeof(int) * (size_t)n_frames, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h->plan, &plan, sizeof(plan), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_estimate_corr: staging", e);
    double tot[4] = {0, 0, 0, 0};
    if (!rc && jdsp::launch_mvdr_corr_total(s, d_l, d_r, 2 * n_frames, h->st[h->cur], h->events, h->plan, ctx->stft1024_table,
                                            h->delta, d_tot, h->tile_sums))
        rc = fail(ctx, JDSP_EHIP, "mvdr corr launch", hipGetLastError());
    if (!rc && (e = hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_estimate_corr: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_estimate_corr: sync", e);
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    if (d_tot) (void)hipFree(d_tot);
    if (!rc)
        for (int c = 0; c < 4; c++) corr4_inout_host[c] += tot[c];               // rgdSpatialCorr[..] += (:263-268)
    return rc;
}

int jdsp_mvdr_apply(jdsp_mvdr *h, const int16_t *left_host, const int16_t *right_host, long n_blocks,
                    const double *corr4_host, int16_t *out_host, float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0 || !corr4_host) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_apply: bad argument");
    const long n_out = jdsp_mvdr_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!left_host || !right_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_mvdr_apply: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = mvdr_reserve(h, n_blocks);
    if (rc) return rc;
    const size_t in_b = (size_t)n_blocks * 1024, out_b = (size_t)(n_out > 0 ? n_out : 1) * 1024;
    int16_t *d_l = nullptr, *d_r = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipStream_t s = ctx->stream;
    jdsp::MvdrState *st_in = h->st[h->cur], *st_out = h->st[h->cur ^ 1];
    hipError_t e = hipMalloc((void **)&d_l, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_b * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(d_l, left_host, in_b, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r, right_host, in_b, hipMemcpyHostToDevice, s);
    // every block uses matrix version 0 = the caller's rgdSpatialCorr; the handle's own matrix and run length carry over
    if (e == hipSuccess) e = hipMemcpyAsync(h->rver, corr4_host, 4 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(h->ver_base, 0, ((size_t)n_blocks / 64 + 1) * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(h->snap_mask, 0, ((size_t)n_blocks / 64 + 1) * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemcpyAsync(st_out, st_in, sizeof(jdsp::MvdrState), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_apply: staging", e);
    if (!rc && jdsp::launch_mvdr_apply(s, d_l, d_r, n_blocks, h->calls, st_in, st_out, h->ver_base, h->snap_mask, h->rver,
                                       h->steer, ctx->stft1024_table, d_out, d_pre))
        rc = fail(ctx, JDSP_EHIP, "mvdr launch", hipGetLastError());
    if (!rc && n_out > 0 && (e = hipMemcpyAsync(out_host, d_out, (size_t)n_out * 1024, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_apply: D2H", e);
    if (!rc && n_out > 0 && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, (size_t)n_out * 2048, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_apply: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_mvdr_apply: sync", e);
    if (d_l) (void)hipFree(d_l);
    if (d_r) (void)hipFree(d_r);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    if (!rc) {
        h->cur ^= 1;
        h->calls += n_blocks;
    }
    return rc;
}

int jdsp_mvdr_corr(jdsp_mvdr *h, double *corr4_host)
{
    if (!h || !corr4_host) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    JDSP_HIP(ctx, hipMemcpyAsync(corr4_host, h->st[h->cur]->corr, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDSP_OK;
}

}  // extern "C"
