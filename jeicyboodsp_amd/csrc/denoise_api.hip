// denoise_api.hip -- C ABI of the spectral-subtraction / Wiener stream object.
#include "jdsp_internal.h"

using jdsp::fail;

static void free_workspace(jdsp_denoise *h)
{
    void *p[] = {h->flags, h->ev_n, h->ver_base, h->snap_mask, h->events, h->dbg_energy, h->dbg_zcr, h->rows,
                 h->acc.lat_alpha, h->acc.lat_chunk, h->acc.chunk_alpha, h->acc.chunk_beta, h->acc.a_start};
    for (void *q : p)
        if (q) (void)hipFree(q);
    h->flags = nullptr;
    h->ev_n = h->ver_base = h->events = nullptr;
    h->snap_mask = nullptr;
    h->dbg_energy = nullptr;
    h->dbg_zcr = nullptr;
    h->rows = nullptr;
    h->acc.lat_alpha = h->acc.chunk_alpha = h->acc.chunk_beta = h->acc.a_start = nullptr;
    h->acc.lat_chunk = nullptr;
    h->cap_rows = 0;
    h->cap_blocks = 0;
}

extern "C" {

int jdsp_denoise_create(jdsp_ctx *ctx, int mode, jdsp_denoise **out) { return jdsp_denoise_create_cfg(ctx, mode, 1024, 512, out); }

int jdsp_denoise_block_len(const jdsp_denoise *h) { return h ? h->block : 0; }

int jdsp_denoise_create_cfg(jdsp_ctx *ctx, int mode, int n_fft, int hop, jdsp_denoise **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    if (mode != JDSP_SPECSUB && mode != JDSP_WIENER) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_create: mode");
    if (!((n_fft == 1024 && hop == 512) || (n_fft == 512 && hop == 256)))
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_create_cfg: (n_fft, hop) must be (1024, 512) or (512, 256)");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    jdsp_denoise *h = new (std::nothrow) jdsp_denoise();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_denoise_create");
    h->ctx = ctx;
    h->mode = mode;
    h->n_fft = n_fft;
    h->block = hop;
    hipError_t e = hipSuccess;
    if (n_fft == 512) {
        // Hamming over 512 points exactly as the reference writes it (SS:226, PI 3.141592): the VAD's FP64 second half
        // (SS:131: the first half multiplies the never-updated, all-zero keep buffer) and the halved FP32 window
        double w_hi[256];
        float w_h[512];
        for (int i = 0; i < 512; i++) {
            const double w = (0.54 - 0.46 * cos(2 * 3.141592 * i / (512 - 1)));
            w_h[i] = (float)(0.5 * w);
            if (i >= 256) w_hi[i - 256] = w;
        }
        e = hipMalloc((void **)&h->w_hi256, sizeof(w_hi));
        if (e == hipSuccess) e = hipMalloc((void **)&h->win512h, sizeof(w_h));
        if (e == hipSuccess) e = hipMemcpy(h->w_hi256, w_hi, sizeof(w_hi), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->win512h, w_h, sizeof(w_h), hipMemcpyHostToDevice);
    }
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void **)&h->st[i], sizeof(jdsp::DenoiseState));
    if (e == hipSuccess) e = hipMalloc((void **)&h->plan, sizeof(jdsp::DenoisePlan));
    if (e == hipSuccess) e = hipMalloc((void **)&h->sh_range, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->sh_a_in, 1024 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->sh_zero_run, sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->sh_zero_run, 0, sizeof(int));
    if (e == hipSuccess && jdsp::ensure_vad_window(ctx)) e = hipErrorUnknown;
    h->w_hi = ctx->vad_w_hi;
    if (e != hipSuccess) {
        jdsp_denoise_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_denoise_create: alloc", e);
    }
    rc = jdsp_denoise_reset(h);
    if (rc) {
        jdsp_denoise_destroy(h);
        return rc;
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_denoise_destroy(jdsp_denoise *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    free_workspace(h);
    for (int i = 0; i < 2; i++)
        if (h->st[i]) (void)hipFree(h->st[i]);
    if (h->plan) (void)hipFree(h->plan);
    if (h->sh_range) (void)hipFree(h->sh_range);
    if (h->sh_a_in) (void)hipFree(h->sh_a_in);
    if (h->sh_zero_run) (void)hipFree(h->sh_zero_run);
    if (h->w_hi256) (void)hipFree(h->w_hi256);
    if (h->win512h) (void)hipFree(h->win512h);
    delete h;
    return JDSP_OK;
}

int jdsp_denoise_reset(jdsp_denoise *h)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    for (int i = 0; i < 2; i++) JDSP_HIP(ctx, hipMemsetAsync(h->st[i], 0, sizeof(jdsp::DenoiseState), ctx->stream));
    h->calls = 0;
    h->cur = 0;
    h->last_blocks = 0;
    return JDSP_OK;
}

int jdsp_denoise_set_option(jdsp_denoise *h, const char *name, long value)
{
    if (!h || !name) return JDSP_EINVAL;
    if (!strcmp(name, "vad_trace")) {
        if (value != 0 && value != 1) return fail(h->ctx, JDSP_EINVAL, "jdsp_denoise_set_option: vad_trace must be 0 or 1");
        h->opt_vad_trace = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "blocks_per_wave")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(h->ctx, JDSP_EINVAL, "jdsp_denoise_set_option: blocks_per_wave must be 0 (auto), 1, 2, 4 or 8");
        h->opt_k = (int)value;
        return JDSP_OK;
    }
    return fail(h->ctx, JDSP_EINVAL, "jdsp_denoise_set_option: unknown option");
}

long jdsp_denoise_blocks_out(const jdsp_denoise *h, long n_blocks)
{
    if (!h || n_blocks < 0) return 0;
    const long first_emit = h->calls >= 2 ? 0 : 2 - h->calls;       // SS:260-263
    return n_blocks > first_emit ? n_blocks - first_emit : 0;
}

static int reserve2(jdsp_denoise *h, long max_blocks);

int jdsp_denoise_reserve(jdsp_denoise *h, long max_blocks)
{
    if (!h || max_blocks < 0) return JDSP_EINVAL;
    return reserve2(h, max_blocks);
}

// max_blocks: blocks the run-length plan covers (a sharded run: the whole stream's)
static int reserve2(jdsp_denoise *h, long max_blocks)
{
    jdsp_ctx *ctx = h->ctx;
    if (max_blocks <= h->cap_blocks) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_workspace(h);
    const size_t n = (size_t)max_blocks;
    hipError_t e = hipMalloc((void **)&h->flags, n);
    if (e == hipSuccess) e = hipMalloc((void **)&h->ev_n, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->ver_base, (n / 64 + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->snap_mask, (n / 64 + 1) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&h->events, n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&h->dbg_energy, n * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc((void **)&h->dbg_zcr, n * sizeof(int));
    // worst case: every block feeds the noise average; an estimate can latch at most every 10th block.  (No magnitude
    // rows: noise_accum_kernel folds them into per-chunk maps in registers.)
    const size_t n_rows = n / 10 + 2;
    if (e == hipSuccess) e = hipMalloc((void **)&h->rows, n_rows * 1024 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->acc.lat_alpha, n_rows * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->acc.lat_chunk, n_rows * sizeof(int));
    const size_t n_chunks = n < (size_t)jdsp::kNoiseChunks ? (n > 0 ? n : 1) : (size_t)jdsp::kNoiseChunks;   // launch_noise_estimate's grid
    if (e == hipSuccess) e = hipMalloc((void **)&h->acc.chunk_alpha, n_chunks * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->acc.chunk_beta, n_chunks * 1024 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&h->acc.a_start, n_chunks * 1024 * sizeof(float));
    if (e != hipSuccess) {
        free_workspace(h);
        return fail(ctx, e == hipErrorOutOfMemory ? JDSP_ENOMEM : JDSP_EHIP, "jdsp_denoise_reserve", e);
    }
    h->cap_blocks = max_blocks;
    h->cap_rows = (long)n_rows;
    return JDSP_OK;
}

int jdsp_denoise_process_dev(jdsp_denoise *h, const int16_t *pcm_dev, long n_blocks, int16_t *out_dev,
                             float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_process: n_blocks < 0");
    const long n_out = jdsp_denoise_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_dev || (n_out > 0 && !out_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_process: NULL buffer");
    if (((uintptr_t)pcm_dev & 15u) || ((uintptr_t)out_dev & 15u))
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_process: buffers must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp_denoise_reserve(h, n_blocks);      // no-op once sized (call jdsp_denoise_reserve before graph capture)
    if (rc) return rc;
    const jdsp::DenoiseState *st_in = h->st[h->cur];
    jdsp::DenoiseState *st_out = h->st[h->cur ^ 1];
    hipStream_t s = ctx->stream;
    if (h->n_fft == 512) {
        if (jdsp::launch_vad256(s, pcm_dev, n_blocks, h->w_hi256, h->flags, h->opt_vad_trace ? h->dbg_energy : nullptr,
                                h->opt_vad_trace ? h->dbg_zcr : nullptr) ||
            jdsp::launch_denoise_plan(s, h->flags, n_blocks, st_in, st_out, h->ver_base, h->snap_mask, h->events, h->ev_n,
                                      h->plan) ||
            jdsp::launch_noise_estimate512(s, pcm_dev, n_blocks, st_in, st_out, h->events, h->ev_n, h->plan,
                                           h->ver_base, h->snap_mask, ctx->stft1024_table, h->win512h, h->acc, h->rows) ||
            jdsp::launch_denoise512(s, h->mode, ctx->n_cu, pcm_dev, n_blocks, h->calls, st_in, st_out, h->ver_base, h->snap_mask,
                                    h->rows, ctx->stft1024_table, h->win512h, out_dev, precast_dev))
            return fail(ctx, JDSP_EHIP, "denoise512 launch", hipGetLastError());
        h->cur ^= 1;
        h->calls += n_blocks;
        h->last_blocks = n_blocks;
        h->last_trace_valid = h->opt_vad_trace;
        return JDSP_OK;
    }
    if (jdsp::launch_vad(s, pcm_dev, n_blocks, h->w_hi, 1, h->flags, h->opt_vad_trace ? h->dbg_energy : nullptr,
                         h->opt_vad_trace ? h->dbg_zcr : nullptr) ||
        jdsp::launch_denoise_plan(s, h->flags, n_blocks, st_in, st_out, h->ver_base, h->snap_mask, h->events, h->ev_n,
                                  h->plan) ||
        jdsp::launch_noise_estimate(s, pcm_dev, n_blocks, st_in, st_out, h->events, h->ev_n, h->plan,
                                    h->ver_base, h->snap_mask, ctx->stft1024_table, h->acc, h->rows) ||
        jdsp::launch_denoise(s, h->mode, h->opt_k, ctx->n_cu, pcm_dev, n_blocks, h->calls, st_in, st_out, h->ver_base,
                             h->snap_mask, h->rows, ctx->stft1024_table, out_dev, precast_dev))
        return fail(ctx, JDSP_EHIP, "denoise launch", hipGetLastError());
    h->cur ^= 1;
    h->calls += n_blocks;
    h->last_blocks = n_blocks;
    h->last_trace_valid = h->opt_vad_trace;
    return JDSP_OK;
}

int jdsp_denoise_process(jdsp_denoise *h, const int16_t *pcm_host, long n_blocks, int16_t *out_host,
                         float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_process: n_blocks < 0");
    const long n_out = jdsp_denoise_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_process: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t blk = (size_t)h->block;
    const size_t in_b = (size_t)n_blocks * blk * sizeof(int16_t);
    const size_t out_b = (size_t)(n_out > 0 ? n_out : 1) * blk * sizeof(int16_t);
    int16_t *d_in = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_b * 2);
    int rc = JDSP_OK;
    if (e != hipSuccess) rc = fail(ctx, JDSP_ENOMEM, "jdsp_denoise_process: hipMalloc", e);
    if (!rc && (e = hipMemcpyAsync(d_in, pcm_host, in_b, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_process: H2D", e);
    if (!rc) rc = jdsp_denoise_process_dev(h, d_in, n_blocks, d_out, d_pre, nullptr);
    if (!rc && n_out > 0 &&
        (e = hipMemcpyAsync(out_host, d_out, (size_t)n_out * blk * 2, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_process: D2H", e);
    if (!rc && n_out > 0 && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, (size_t)n_out * blk * 4, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_process: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_process: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    return rc;
}

int jdsp_denoise_apply(jdsp_denoise *h, const int16_t *pcm_host, long n_blocks, const double *noise_host,
                       int16_t *out_host, float *precast_host, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_blocks < 0 || !noise_host) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_apply: bad argument");
    const long n_out = jdsp_denoise_blocks_out(h, n_blocks);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (n_blocks == 0) return JDSP_OK;
    if (!pcm_host || (n_out > 0 && !out_host)) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_apply: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp_denoise_reserve(h, n_blocks);
    if (rc) return rc;
    float row[1024];
    for (int i = 0; i < h->n_fft; i++) row[i] = (float)noise_host[i];          // pdEstimatedNoiseSpec[FFT_PROCESSING_SIZE]
    const size_t blk_b = (size_t)h->block * sizeof(int16_t);
    const size_t in_b = (size_t)n_blocks * blk_b, out_b = (size_t)(n_out > 0 ? n_out : 1) * blk_b;
    int16_t *d_in = nullptr, *d_out = nullptr;
    float *d_pre = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, in_b);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && precast_host) e = hipMalloc((void **)&d_pre, out_b * 2);
    hipStream_t s = ctx->stream;
    jdsp::DenoiseState *st_in = h->st[h->cur], *st_out = h->st[h->cur ^ 1];
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, pcm_host, in_b, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h->rows, row, sizeof(float) * (size_t)h->n_fft, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(h->ver_base, 0, ((size_t)n_blocks / 64 + 1) * sizeof(int), s);   // every block uses row 0
    if (e == hipSuccess) e = hipMemsetAsync(h->snap_mask, 0, ((size_t)n_blocks / 64 + 1) * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemcpyAsync(st_out, st_in, sizeof(jdsp::DenoiseState), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_apply: staging", e);
    if (!rc && (h->n_fft == 512
                    ? jdsp::launch_denoise512(s, h->mode, ctx->n_cu, d_in, n_blocks, h->calls, st_in, st_out, h->ver_base,
                                              h->snap_mask, h->rows, ctx->stft1024_table, h->win512h, d_out, d_pre)
                    : jdsp::launch_denoise(s, h->mode, h->opt_k, ctx->n_cu, d_in, n_blocks, h->calls, st_in, st_out, h->ver_base,
                                           h->snap_mask, h->rows, ctx->stft1024_table, d_out, d_pre))) {
        const hipError_t le = hipGetLastError();
        rc = fail(ctx, JDSP_EHIP, "denoise launch", le);
    }
    if (!rc && n_out > 0 && (e = hipMemcpyAsync(out_host, d_out, (size_t)n_out * blk_b, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_apply: D2H", e);
    if (!rc && n_out > 0 && precast_host &&
        (e = hipMemcpyAsync(precast_host, d_pre, (size_t)n_out * blk_b * 2, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_apply: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_denoise_apply: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_pre) (void)hipFree(d_pre);
    if (!rc) {
        h->cur ^= 1;
        h->calls += n_blocks;
        h->last_blocks = 0;
    }
    return rc;
}

int jdsp_vad_blocks_ex(jdsp_ctx *ctx, int variant, int block_len, const int16_t *pcm_host, long n_blocks,
                       uint8_t *voice_host, int64_t *energy_sum_host, int32_t *zcr_host)
{
    if (!ctx) return JDSP_EINVAL;
    if ((variant != JDSP_VAD_DENOISE && variant != JDSP_VAD_MVDR) || (block_len != 512 && block_len != 256))
        return fail(ctx, JDSP_EINVAL, "jdsp_vad_blocks_ex: variant 0 | 1, block_len 512 | 256");
    if (n_blocks < 0 || (n_blocks > 0 && !pcm_host)) return fail(ctx, JDSP_EINVAL, "jdsp_vad_blocks: bad argument");
    if (n_blocks == 0) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const double *w = nullptr;
    int rc = jdsp::ensure_vad_window_ex(ctx, variant, block_len, &w);
    if (rc) return rc;
    const size_t n = (size_t)n_blocks, in_bytes = n * (size_t)block_len * sizeof(int16_t);
    int16_t *d_in = nullptr;
    unsigned char *d_v = nullptr;
    long long *d_e = nullptr;
    int *d_z = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_v, n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_e, n * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_z, n * 4);
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, pcm_host, in_bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_vad_blocks: staging", e);
    const int use_zcr = variant == JDSP_VAD_DENOISE;                    // BF:233 tests the energy alone
    if (!rc && (block_len == 512 ? jdsp::launch_vad(s, d_in, n_blocks, w, use_zcr, d_v, d_e, d_z)
                                 : jdsp::launch_vad256(s, d_in, n_blocks, w, d_v, d_e, d_z, use_zcr))) {
        const hipError_t le = hipGetLastError();
        rc = fail(ctx, JDSP_EHIP, "vad launch", le);
    }
    if (!rc && voice_host && (e = hipMemcpyAsync(voice_host, d_v, n, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_vad_blocks: D2H", e);
    if (!rc && energy_sum_host && (e = hipMemcpyAsync(energy_sum_host, d_e, n * 8, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_vad_blocks: D2H", e);
    if (!rc && zcr_host && (e = hipMemcpyAsync(zcr_host, d_z, n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_vad_blocks: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_vad_blocks: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_v) (void)hipFree(d_v);
    if (d_e) (void)hipFree(d_e);
    if (d_z) (void)hipFree(d_z);
    return rc;
}

int jdsp_vad_blocks(jdsp_ctx *ctx, const int16_t *pcm_host, long n_blocks, uint8_t *voice_host,
                    int64_t *energy_sum_host, int32_t *zcr_host)
{
    return jdsp_vad_blocks_ex(ctx, JDSP_VAD_DENOISE, 512, pcm_host, n_blocks, voice_host, energy_sum_host, zcr_host);
}

/* ---- multi-GPU: one rank's share of a stream (include/jdsp.h "sharded denoise") ------------- */
int jdsp_denoise_shard_vad_dev(jdsp_denoise *h, const int16_t *pcm_ext_dev, long ext0, long b0, long b1, long n_total,
                               uint8_t *flags_own_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (!(0 <= ext0 && ext0 <= b0 && b0 <= b1 && b1 <= n_total) || (b0 >= 2 ? ext0 != b0 - 2 : ext0 != 0))
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_vad: need ext0 = max(b0-2, 0) <= b0 <= b1 <= n_total");
    if (b1 > b0 && (!pcm_ext_dev || !flags_own_dev)) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_vad: NULL buffer");
    if ((uintptr_t)pcm_ext_dev & 15u) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_vad: pcm must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = reserve2(h, n_total);
    if (rc) return rc;
    rc = jdsp_denoise_reset(h);                       // a sharded run is one fresh global stream
    if (rc) return rc;
    h->sh_ext0 = ext0; h->sh_b0 = b0; h->sh_b1 = b1; h->sh_total = n_total; h->sh_pcm = pcm_ext_dev;
    if (h->n_fft == 512 ? jdsp::launch_vad256(ctx->stream, pcm_ext_dev + (b0 - ext0) * 256, b1 - b0, h->w_hi256, flags_own_dev,
                                              nullptr, nullptr)
                        : jdsp::launch_vad(ctx->stream, pcm_ext_dev + (b0 - ext0) * 512, b1 - b0, h->w_hi, 1, flags_own_dev,
                                           nullptr, nullptr)) {
        const hipError_t le = hipGetLastError();
        return fail(ctx, JDSP_EHIP, "vad launch", le);
    }
    return JDSP_OK;
}

int jdsp_denoise_shard_summary_dev(jdsp_denoise *h, const uint8_t *flags_all_dev, float *summary_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (!h->sh_pcm && h->sh_b1 > h->sh_b0) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_summary: call shard_vad first");
    if (!flags_all_dev || !summary_dev) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_summary: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (jdsp::launch_run_plan(s, flags_all_dev, h->sh_total, h->sh_zero_run, nullptr, 10, h->ver_base, h->snap_mask,
                              h->events, h->ev_n, h->plan) ||
        (h->n_fft == 512
             ? jdsp::launch_shard_summary512(s, h->sh_pcm, h->sh_b1 - h->sh_ext0, h->sh_ext0, h->sh_b0, h->sh_b1, h->events,
                                             h->ev_n, h->plan, h->ver_base, h->snap_mask, ctx->stft1024_table, h->win512h,
                                             h->sh_range, h->acc, h->rows, summary_dev)
             : jdsp::launch_shard_summary(s, h->sh_pcm, h->sh_b1 - h->sh_ext0, h->sh_ext0, h->sh_b0, h->sh_b1, h->events,
                                          h->ev_n, h->plan, h->ver_base, h->snap_mask, ctx->stft1024_table, h->sh_range,
                                          h->acc, h->rows, summary_dev))) {
        const hipError_t le = hipGetLastError();
        return fail(ctx, JDSP_EHIP, "shard summary launch", le);
    }
    return JDSP_OK;
}

int jdsp_denoise_shard_rows_dev(jdsp_denoise *h, const float *summaries_all_dev, int world, int rank, float *last_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (!summaries_all_dev || !last_dev || world < 1 || rank < 0 || rank >= world)
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_rows: bad argument");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (h->n_fft == 512 ? jdsp::launch_shard_rows512(ctx->stream, summaries_all_dev, rank, h->sh_b0, h->sh_b1, h->plan,
                                                     h->sh_range, h->acc, h->sh_a_in, h->rows, last_dev)
                        : jdsp::launch_shard_rows(ctx->stream, summaries_all_dev, rank, h->sh_b0, h->sh_b1, h->plan,
                                                  h->sh_range, h->acc, h->sh_a_in, h->rows, last_dev)) {
        const hipError_t le = hipGetLastError();
        return fail(ctx, JDSP_EHIP, "shard rows launch", le);
    }
    return JDSP_OK;
}

long jdsp_denoise_shard_blocks_out(const jdsp_denoise *h)
{
    if (!h) return 0;
    const long lo = h->sh_b0 > 2 ? h->sh_b0 : 2;                      // SS:260-263: global blocks 0 and 1 emit nothing
    return h->sh_b1 > lo ? h->sh_b1 - lo : 0;
}

int jdsp_denoise_shard_finish_dev(jdsp_denoise *h, const float *last_all_dev, int world, int rank, int16_t *out_dev,
                                  float *precast_dev, long *n_out_blocks)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    const long n_out = jdsp_denoise_shard_blocks_out(h);
    if (n_out_blocks) *n_out_blocks = n_out;
    if (!last_all_dev || world < 1 || rank < 0 || rank >= world || (n_out > 0 && !out_dev))
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_finish: bad argument");
    if ((uintptr_t)out_dev & 15u) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_shard_finish: out must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (h->n_fft == 512 ? jdsp::launch_shard_row0_512(s, last_all_dev, rank, h->rows) : jdsp::launch_shard_row0(s, last_all_dev, rank, h->rows)) {
        const hipError_t le = hipGetLastError();
        return fail(ctx, JDSP_EHIP, "row0 launch", le);
    }
    if (n_out > 0) {
        jdsp::DenoiseShard sh;
        sh.ver_block_off = h->sh_ext0;
        sh.ver_row_off = h->sh_range + 2;
        const long lo = h->sh_b0 > 2 ? h->sh_b0 : 2;
        sh.emit_from = lo - h->sh_ext0;
        sh.emit_to = h->sh_b1 - h->sh_ext0;
        // fresh state: the two halo blocks in front of the shard rebuild the overlap tail
        if (h->n_fft == 512
                ? jdsp::launch_denoise512(s, h->mode, ctx->n_cu, h->sh_pcm, h->sh_b1 - h->sh_ext0, h->sh_ext0, h->st[h->cur],
                                          h->st[h->cur ^ 1], h->ver_base, h->snap_mask, h->rows, ctx->stft1024_table,
                                          h->win512h, out_dev, precast_dev, &sh)
                : jdsp::launch_denoise(s, h->mode, h->opt_k, ctx->n_cu, h->sh_pcm, h->sh_b1 - h->sh_ext0, h->sh_ext0,
                                       h->st[h->cur], h->st[h->cur ^ 1], h->ver_base, h->snap_mask, h->rows,
                                       ctx->stft1024_table, out_dev, precast_dev, &sh)) {
            const hipError_t le = hipGetLastError();
            return fail(ctx, JDSP_EHIP, "denoise launch", le);
        }
    }
    return JDSP_OK;
}

int jdsp_denoise_noise(jdsp_denoise *h, double *noise_host)
{
    if (!h || !noise_host) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    float tmp[1024];
    JDSP_HIP(ctx, hipMemcpyAsync(tmp, h->st[h->cur]->noise, sizeof(tmp), hipMemcpyDeviceToHost, ctx->stream));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < h->n_fft; i++) noise_host[i] = tmp[i];
    return JDSP_OK;
}

int jdsp_denoise_vad_trace(jdsp_denoise *h, long n, uint8_t *voice_host, int64_t *energy_sum_host, int32_t *zcr_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n < 0 || n > h->last_blocks) return fail(ctx, JDSP_EINVAL, "jdsp_denoise_vad_trace: n exceeds the last call");
    if ((energy_sum_host || zcr_host) && !h->last_trace_valid)
        return fail(ctx, JDSP_EINVAL, "jdsp_denoise_vad_trace: energies / ZCR are kept only with set_option(\"vad_trace\", 1) before the call");
    if (n == 0) return JDSP_OK;
    if (voice_host) JDSP_HIP(ctx, hipMemcpyAsync(voice_host, h->flags, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (energy_sum_host)
        JDSP_HIP(ctx, hipMemcpyAsync(energy_sum_host, h->dbg_energy, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (zcr_host) JDSP_HIP(ctx, hipMemcpyAsync(zcr_host, h->dbg_zcr, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDSP_OK;
}

}  // extern "C"
