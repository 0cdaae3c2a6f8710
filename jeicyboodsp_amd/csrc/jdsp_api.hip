// jdsp_api.hip -- the C ABI (include/jdsp.h) over the gfx950 kernels.
#include "jdsp_internal.h"

static std::string g_create_error;

namespace jdsp {

int fail(jdsp_ctx *ctx, int code, const char *what, hipError_t e)
{
    std::string msg = what ? what : "error";
    if (e != hipSuccess) {
        msg += ": ";
        msg += hipGetErrorString(e);
    }
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

int ensure_stft1024_table(jdsp_ctx *ctx)
{
    if (ctx->stft1024_table) return 0;
    const int n = stft1024_table_count();
    std::vector<float2> host((size_t)n);
    fill_stft1024_table(host.data(), 0);
    JDSP_HIP(ctx, hipMalloc((void **)&ctx->stft1024_table, sizeof(float2) * (size_t)n));
    JDSP_HIP(ctx, hipMemcpy(ctx->stft1024_table, host.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    return 0;
}

int ensure_stft1024_table_rect(jdsp_ctx *ctx)
{
    if (ctx->stft1024_table_rect) return 0;
    const int n = stft1024_table_count();
    std::vector<float2> host((size_t)n);
    fill_stft1024_table(host.data(), 2);
    JDSP_HIP(ctx, hipMalloc((void **)&ctx->stft1024_table_rect, sizeof(float2) * (size_t)n));
    JDSP_HIP(ctx, hipMemcpy(ctx->stft1024_table_rect, host.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    return 0;
}

int ensure_vad_window(jdsp_ctx *ctx)
{
    if (ctx->vad_w_hi) return 0;
    double w[512];
    for (int i = 0; i < 512; i++) w[i] = (0.54 - 0.46 * cos(2 * 3.141592 * (512 + i) / (1024 - 1)));   // SS:131
    JDSP_HIP(ctx, hipMalloc((void **)&ctx->vad_w_hi, sizeof(w)));
    JDSP_HIP(ctx, hipMemcpy(ctx->vad_w_hi, w, sizeof(w), hipMemcpyHostToDevice));
    return 0;
}

int ensure_vad_window_ex(jdsp_ctx *ctx, int variant, int block_len, const double **w_out)
{
    const int bi = block_len == 512 ? 0 : 1;
    if (!ctx->vad_w_ex[variant][bi]) {
        const int n = 2 * block_len, keep = variant == 0 ? block_len : block_len - 1;
        double w[512];
        for (int i = 0; i < block_len; i++) w[i] = (0.54 - 0.46 * cos(2 * 3.141592 * (keep + i) / (n - 1)));   // SS:131 / BF:217
        JDSP_HIP(ctx, hipMalloc((void **)&ctx->vad_w_ex[variant][bi], sizeof(double) * block_len));
        JDSP_HIP(ctx, hipMemcpy(ctx->vad_w_ex[variant][bi], w, sizeof(double) * block_len, hipMemcpyHostToDevice));
    }
    *w_out = ctx->vad_w_ex[variant][bi];
    return 0;
}

}  // namespace jdsp

using jdsp::fail;

extern "C" {

static int ensure_c2c_tw(jdsp_ctx *ctx, int n_fft, int lg);

int jdsp_abi_version(void) { return JDSP_ABI_VERSION; }

int jdsp_create(int device, jdsp_ctx **out)
{
    if (!out) return fail(nullptr, JDSP_EINVAL, "jdsp_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, JDSP_ENODEV, "jdsp_create: no HIP device (this library has no CPU fallback)", e);
    if (device < 0 || device >= count) return fail(nullptr, JDSP_ENODEV, "jdsp_create: device ordinal out of range");
    jdsp_ctx *ctx = new (std::nothrow) jdsp_ctx();
    if (!ctx) return fail(nullptr, JDSP_ENOMEM, "jdsp_create: out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, JDSP_EHIP, "jdsp_create: device query", e);
    }
    ctx->n_cu = prop.multiProcessorCount;
    ctx->hbm_bytes = prop.totalGlobalMem;
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = std::string("jdsp_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        delete ctx;
        return fail(nullptr, JDSP_ENODEV, m.c_str());
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, JDSP_EHIP, "jdsp_create: hipStreamCreate", e);
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return JDSP_OK;
}

int jdsp_destroy(jdsp_ctx *ctx)
{
    if (!ctx) return JDSP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stft1024_table) (void)hipFree(ctx->stft1024_table);
    if (ctx->stft_f64_table) (void)hipFree(ctx->stft_f64_table);
    for (auto &p : ctx->c2c_tw)
        if (p) (void)hipFree(p);
    if (ctx->vad_w_hi) (void)hipFree(ctx->vad_w_hi);
    for (auto &row : ctx->vad_w_ex)
        for (auto &p : row)
            if (p) (void)hipFree(p);
    if (ctx->win512) (void)hipFree(ctx->win512);
    if (ctx->win512_hann) (void)hipFree(ctx->win512_hann);
    if (ctx->stft1024_table_hann) (void)hipFree(ctx->stft1024_table_hann);
    if (ctx->stft1024_table_rect) (void)hipFree(ctx->stft1024_table_rect);
    for (auto &p : ctx->pipe_buf)
        if (p) (void)hipFree(p);
    for (auto &ev : ctx->pipe_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->pipe_in) (void)hipStreamDestroy(ctx->pipe_in);
    if (ctx->pipe_out) (void)hipStreamDestroy(ctx->pipe_out);
    if (ctx->conv_tw4096) (void)hipFree(ctx->conv_tw4096);
    if (ctx->conv_tw8192) (void)hipFree(ctx->conv_tw8192);
    if (ctx->switch_ev) (void)hipEventDestroy(ctx->switch_ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return JDSP_OK;
}

const char *jdsp_last_error(const jdsp_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

// Work already enqueued on the stream the handle is leaving (create/reset memsets, the previous call's
// state updates: overlap tail, run length, noise rows) must be visible to whatever is enqueued next on the
// stream it moves to: record an event on the old stream and make the new one wait for it.  Nothing is done
// when the stream does not change (the common case, and the only one inside a graph capture), nor when
// either stream is being captured -- an event recorded outside a capture cannot order captured work, so
// there the ordering stays the caller's (bench.py: wait_stream before and after the capture).
static int switch_stream(jdsp_ctx *ctx, hipStream_t next)
{
    if (next == ctx->stream) return JDSP_OK;
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
    const bool ok_a = hipStreamIsCapturing(ctx->stream, &a) == hipSuccess;
    const bool ok_b = hipStreamIsCapturing(next, &b) == hipSuccess;
    if (!ok_a || !ok_b) (void)hipGetLastError();
    if (ok_a && ok_b && a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone) {
        JDSP_HIP(ctx, hipSetDevice(ctx->device));
        if (!ctx->switch_ev) JDSP_HIP(ctx, hipEventCreateWithFlags(&ctx->switch_ev, hipEventDisableTiming));
        JDSP_HIP(ctx, hipEventRecord(ctx->switch_ev, ctx->stream));
        JDSP_HIP(ctx, hipStreamWaitEvent(next, ctx->switch_ev, 0));
    }
    ctx->stream = next;
    return JDSP_OK;
}

int jdsp_set_stream(jdsp_ctx *ctx, void *hip_stream)
{
    if (!ctx) return JDSP_EINVAL;
    return switch_stream(ctx, (hipStream_t)hip_stream);
}

int jdsp_use_own_stream(jdsp_ctx *ctx)
{
    if (!ctx) return JDSP_EINVAL;
    return switch_stream(ctx, ctx->own_stream);
}

int jdsp_set_option(jdsp_ctx *ctx, const char *name, long value)
{
    if (!ctx || !name) return JDSP_EINVAL;
    if (!strcmp(name, "stft.frames_per_wave")) {
        if (value < 0 || value > 4096) return fail(ctx, JDSP_EINVAL, "stft.frames_per_wave out of range");
        ctx->opt_stft_fpw = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "stft.read_pass")) {
        if (value < -1 || value > 1) return fail(ctx, JDSP_EINVAL, "stft.read_pass: -1 (auto), 0 or 1");
        ctx->opt_stft_read_pass = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "stft.read_pass_wg_per_cu")) {
        if (value < 0 || value > 64) return fail(ctx, JDSP_EINVAL, "stft.read_pass_wg_per_cu: 0 (default) .. 64");
        ctx->opt_stft_touch_wg = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "stft.f64_kernel")) {
        if (value != 0 && value != 1) return fail(ctx, JDSP_EINVAL, "stft.f64_kernel: 0 (four waves per SIMD) or 1 (round 2's kernel)");
        ctx->opt_stft_f64_kernel = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "stft.f64_frames_per_wave")) {
        if (value < 0 || value > 4096) return fail(ctx, JDSP_EINVAL, "stft.f64_frames_per_wave out of range");
        ctx->opt_stft_f64_fpw = (int)value;
        return JDSP_OK;
    }
    if (!strcmp(name, "stft.window")) {
        if (value != 0 && value != 1) return fail(ctx, JDSP_EINVAL, "stft.window: 0 (Hamming) or 1 (Hann)");
        ctx->opt_stft_window = (int)value;
        return JDSP_OK;
    }
    return fail(ctx, JDSP_EINVAL, "jdsp_set_option: unknown option");
}

int jdsp_synchronize(jdsp_ctx *ctx)
{
    if (!ctx) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDSP_OK;
}

int jdsp_device_info(jdsp_ctx *ctx, int *n_cu, size_t *hbm_bytes, char *name, size_t name_len)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_cu) *n_cu = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (name && name_len) snprintf(name, name_len, "%s", ctx->name);
    return JDSP_OK;
}

int jdsp_malloc(jdsp_ctx *ctx, size_t bytes, void **dev_ptr)
{
    if (!ctx || !dev_ptr) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dev_ptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return fail(ctx, JDSP_ENOMEM, "jdsp_malloc", e);
    if (e != hipSuccess) return fail(ctx, JDSP_EHIP, "jdsp_malloc", e);
    return JDSP_OK;
}

int jdsp_free(jdsp_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipFree(dev_ptr));
    return JDSP_OK;
}

int jdsp_host_alloc(jdsp_ctx *ctx, size_t bytes, void **host_ptr)
{
    if (!ctx || !host_ptr) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, JDSP_ENOMEM, "jdsp_host_alloc", e);
    return JDSP_OK;
}

int jdsp_host_free(jdsp_ctx *ctx, void *host_ptr)
{
    if (!ctx) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipHostFree(host_ptr));
    return JDSP_OK;
}

int jdsp_memcpy_h2d(jdsp_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    if (!ctx) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDSP_OK;
}

int jdsp_memcpy_d2h(jdsp_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!ctx) return JDSP_EINVAL;
    JDSP_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JDSP_OK;
}

/* ---- FFTAlgorithm_ver2 -------------------------------------------------------- */
static int ilog2_exact(int n)
{
    int b = 0;
    while ((1 << b) < n) b++;
    return (1 << b) == n ? b : -1;
}

int jdsp_bitrev_table(jdsp_ctx *ctx, int n_fft, int block_len, int16_t *table_host)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_fft < 1 || n_fft > 32768 || block_len < 1 || !table_host)
        return fail(ctx, JDSP_EINVAL, "jdsp_bitrev_table: bad argument");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const int bits = (int)log2((double)block_len);       // FFTAlgorithm_ver2.cpp:188
    short *d = nullptr;
    JDSP_HIP(ctx, hipMalloc((void **)&d, sizeof(short) * (size_t)n_fft));
    int rc = JDSP_OK;
    hipError_t e;
    if (jdsp::launch_bitrev_table(ctx->stream, d, n_fft, bits)) rc = fail(ctx, JDSP_EHIP, "bitrev launch", hipGetLastError());
    if (!rc && (e = hipMemcpyAsync(table_host, d, sizeof(short) * (size_t)n_fft, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_bitrev_table: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_bitrev_table: sync", e);
    (void)hipFree(d);
    return rc;
}

int jdsp_fft_process_f64_dev(jdsp_ctx *ctx, const double *in_dev, double *out_dev, int n_fft, long batch, int forward)
{
    if (!ctx) return JDSP_EINVAL;
    const int lg = ilog2_exact(n_fft);
    if (lg < 1 || lg > 13) return fail(ctx, JDSP_EINVAL, "jdsp_fft_process_f64: n_fft must be a power of two in [2, 8192]");
    if (batch < 0 || (batch > 0 && (!in_dev || !out_dev))) return fail(ctx, JDSP_EINVAL, "jdsp_fft_process_f64: bad buffer");
    if (batch == 0) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    {
        const int rc_tw = ensure_c2c_tw(ctx, n_fft, lg);
        if (rc_tw) return rc_tw;
    }
    if (jdsp::launch_fft_process_f64(ctx->stream, (const double2 *)in_dev, (double2 *)out_dev, n_fft, lg, batch, forward,
                                     ctx->c2c_tw[lg]))
        return fail(ctx, JDSP_EHIP, "fft_process_f64 launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_fft_process_f64(jdsp_ctx *ctx, const double *in_host, double *out_host, int n_fft, long batch, int forward)
{
    if (!ctx) return JDSP_EINVAL;
    if (batch < 0 || n_fft < 2) return fail(ctx, JDSP_EINVAL, "jdsp_fft_process_f64: bad argument");
    if (batch == 0) return JDSP_OK;
    if (!in_host || !out_host) return fail(ctx, JDSP_EINVAL, "jdsp_fft_process_f64: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = sizeof(double) * 2 * (size_t)n_fft * (size_t)batch;
    double *d_in = nullptr, *d_out = nullptr;
    JDSP_HIP(ctx, hipMalloc((void **)&d_in, bytes));
    hipError_t e = hipMalloc((void **)&d_out, bytes);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, JDSP_ENOMEM, "jdsp_fft_process_f64: hipMalloc", e);
    }
    int rc = JDSP_OK;
    if ((e = hipMemcpyAsync(d_in, in_host, bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_fft_process_f64: H2D", e);
    if (!rc) rc = jdsp_fft_process_f64_dev(ctx, d_in, d_out, n_fft, batch, forward);
    if (!rc && (e = hipMemcpyAsync(out_host, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_fft_process_f64: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_fft_process_f64: sync", e);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

int jdsp_dft_direct_f64_dev(jdsp_ctx *ctx, int kind, const void *in_dev, double *inout_dev, int n, long batch)
{
    if (!ctx) return JDSP_EINVAL;
    if (kind < JDSP_DFT_I16 || kind > JDSP_IDFT_OVER_N) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: kind");
    if (n < 1 || batch < 0 || batch > 65535) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: n >= 1, 0 <= batch <= 65535");
    if (batch == 0) return JDSP_OK;
    if (!in_dev || !inout_dev) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: NULL buffer");
    if ((uintptr_t)inout_dev & 15u || (kind != JDSP_DFT_I16 && ((uintptr_t)in_dev & 15u)))
        return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: complex buffers must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (jdsp::launch_dft_direct_f64(ctx->stream, kind, in_dev, (double2 *)inout_dev, n, batch))
        return fail(ctx, JDSP_EHIP, "dft_direct launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_dft_direct_f64(jdsp_ctx *ctx, int kind, const void *in_host, double *inout_host, int n, long batch)
{
    if (!ctx) return JDSP_EINVAL;
    if (kind < JDSP_DFT_I16 || kind > JDSP_IDFT_OVER_N) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: kind");
    if (n < 1 || batch < 0 || batch > 65535) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: n >= 1, 0 <= batch <= 65535");
    if (batch == 0) return JDSP_OK;
    if (!in_host || !inout_host) return fail(ctx, JDSP_EINVAL, "jdsp_dft_direct_f64: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t out_b = sizeof(double) * 2 * (size_t)n * (size_t)batch;
    const size_t in_b = kind == JDSP_DFT_I16 ? sizeof(int16_t) * (size_t)n * (size_t)batch : out_b;
    void *d_in = nullptr;
    double *d_out = nullptr;
    JDSP_HIP(ctx, hipMalloc(&d_in, in_b));
    hipError_t e = hipMalloc((void **)&d_out, out_b);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, JDSP_ENOMEM, "jdsp_dft_direct_f64: hipMalloc", e);
    }
    int rc = JDSP_OK;
    if ((e = hipMemcpyAsync(d_in, in_host, in_b, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_out, inout_host, out_b, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_dft_direct_f64: H2D", e);
    if (!rc) rc = jdsp_dft_direct_f64_dev(ctx, kind, d_in, d_out, n, batch);
    if (!rc && (e = hipMemcpyAsync(inout_host, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_dft_direct_f64: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_dft_direct_f64: sync", e);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

/* ---- PitchEstimation_method1 -------------------------------------------------- */
int jdsp_pitch_autocorr_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_blocks, const int16_t *prev_block_dev,
                            int32_t *arg_dev, float *rmax_dev, float *autocorr_dev)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_blocks < 0 || (n_blocks > 0 && (!pcm_dev || !arg_dev || !rmax_dev)))
        return fail(ctx, JDSP_EINVAL, "jdsp_pitch_autocorr: bad buffer");
    if (n_blocks == 0) return JDSP_OK;
    if (((uintptr_t)pcm_dev & 15u) || ((uintptr_t)prev_block_dev & 15u))
        return fail(ctx, JDSP_EINVAL, "jdsp_pitch_autocorr: pcm must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    if (jdsp::launch_pitch(ctx->stream, pcm_dev, n_blocks, prev_block_dev, ctx->stft1024_table, arg_dev, rmax_dev,
                           autocorr_dev))
        return fail(ctx, JDSP_EHIP, "pitch launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_pitch_autocorr(jdsp_ctx *ctx, const int16_t *pcm_host, long n_blocks, const int16_t *prev_block_host,
                        int32_t *arg_host, float *rmax_host, float *autocorr_host)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_blocks < 0 || (n_blocks > 0 && (!pcm_host || !arg_host || !rmax_host)))
        return fail(ctx, JDSP_EINVAL, "jdsp_pitch_autocorr: bad buffer");
    if (n_blocks == 0) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)n_blocks;
    int16_t *d_in = nullptr, *d_prev = nullptr;
    int32_t *d_arg = nullptr;
    float *d_max = nullptr, *d_ac = nullptr;
    hipError_t e = hipMalloc((void **)&d_in, n * 1024);
    if (e == hipSuccess) e = hipMalloc((void **)&d_arg, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_max, n * 4);
    if (e == hipSuccess && prev_block_host) e = hipMalloc((void **)&d_prev, 1024);
    if (e == hipSuccess && autocorr_host) e = hipMalloc((void **)&d_ac, n * 2048);
    hipStream_t s = ctx->stream;
    int rc = JDSP_OK;
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, pcm_host, n * 1024, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && prev_block_host) e = hipMemcpyAsync(d_prev, prev_block_host, 1024, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_pitch_autocorr: staging", e);
    if (!rc) rc = jdsp_pitch_autocorr_dev(ctx, d_in, n_blocks, d_prev, d_arg, d_max, d_ac);
    if (!rc && (e = hipMemcpyAsync(arg_host, d_arg, n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_pitch_autocorr: D2H", e);
    if (!rc && (e = hipMemcpyAsync(rmax_host, d_max, n * 4, hipMemcpyDeviceToHost, s)) != hipSuccess) rc = fail(ctx, JDSP_EHIP, "jdsp_pitch_autocorr: D2H", e);
    if (!rc && autocorr_host && (e = hipMemcpyAsync(autocorr_host, d_ac, n * 2048, hipMemcpyDeviceToHost, s)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_pitch_autocorr: D2H", e);
    if ((e = hipStreamSynchronize(s)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_pitch_autocorr: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_prev) (void)hipFree(d_prev);
    if (d_arg) (void)hipFree(d_arg);
    if (d_max) (void)hipFree(d_max);
    if (d_ac) (void)hipFree(d_ac);
    return rc;
}

/* ---- STFT ------------------------------------------------------------------ */
int jdsp_stft_i16_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames, int n_fft, int hop, jdsp_c32 *spec_dev)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_fft != 1024 && n_fft != 512) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_dev: n_fft must be 1024 or 512");
    if (hop < 1) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_dev: hop must be >= 1");
    if (n_frames < 0 || (n_frames > 0 && (!pcm_dev || !spec_dev)))
        return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_dev: bad buffer");
    if (n_frames == 0) return JDSP_OK;
    if ((uintptr_t)pcm_dev & 1u) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_dev: pcm must be 2-byte aligned");
    if ((uintptr_t)spec_dev & 15u) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_dev: spec must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    const int wk = ctx->opt_stft_window;
    if (wk == 1 && !ctx->stft1024_table_hann) {          // same tables, Hann in the window slots
        const int n = jdsp::stft1024_table_count();
        std::vector<float2> host((size_t)n);
        jdsp::fill_stft1024_table(host.data(), 1);
        JDSP_HIP(ctx, hipMalloc((void **)&ctx->stft1024_table_hann, sizeof(float2) * (size_t)n));
        JDSP_HIP(ctx, hipMemcpy(ctx->stft1024_table_hann, host.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    }
    const float2 *tab = wk == 1 ? ctx->stft1024_table_hann : ctx->stft1024_table;
    if (n_fft == 512) {
        float2 *&w512 = wk == 1 ? ctx->win512_hann : ctx->win512;
        if (!w512) {
            float2 w[256];
            jdsp::fill_win512(w, wk);
            JDSP_HIP(ctx, hipMalloc((void **)&w512, sizeof(w)));
            JDSP_HIP(ctx, hipMemcpy(w512, w, sizeof(w), hipMemcpyHostToDevice));
        }
        if (jdsp::launch_stft512(ctx->stream, ctx->n_cu, pcm_dev, n_frames, hop, (float2 *)spec_dev, tab, w512))
            return fail(ctx, JDSP_EHIP, "stft512 launch", hipGetLastError());
        return JDSP_OK;
    }
    if (jdsp::launch_stft1024(ctx->stream, ctx->n_cu, ctx->opt_stft_fpw, pcm_dev, n_frames, hop, (float2 *)spec_dev, tab,
                              ctx->opt_stft_read_pass, ctx->opt_stft_touch_wg))
        return fail(ctx, JDSP_EHIP, "stft1024 launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_stft_half_i16_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames, jdsp_c32 *spec_dev, long row_pitch)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_frames < 0 || (n_frames > 0 && (!pcm_dev || !spec_dev))) return fail(ctx, JDSP_EINVAL, "jdsp_stft_half_i16_dev: bad buffer");
    if (row_pitch < 513) return fail(ctx, JDSP_EINVAL, "jdsp_stft_half_i16_dev: row_pitch must be at least 513");
    if (n_frames == 0) return JDSP_OK;
    if (((uintptr_t)pcm_dev & 15u) || ((uintptr_t)spec_dev & 15u))
        return fail(ctx, JDSP_EINVAL, "jdsp_stft_half_i16_dev: buffers must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    if (jdsp::launch_stft1024_half(ctx->stream, pcm_dev, n_frames, (float2 *)spec_dev, row_pitch, ctx->stft1024_table))
        return fail(ctx, JDSP_EHIP, "stft half launch", hipGetLastError());
    return JDSP_OK;
}

static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();               // an ordinary (pageable) pointer is not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Pinned buffers: the batch goes through in chunks of kPipeFrames frames, double-buffered on the
// device, with the copy-in of chunk c+1 and the copy-out of chunk c-1 running beside the
// transform of chunk c (three streams, events for the hand-offs).
static int stft_pipelined(jdsp_ctx *ctx, const int16_t *pcm_host, long n_frames, int n_fft, int hop, jdsp_c32 *spec_host)
{
    const long kPipeFrames = 4096;
    hipError_t e = hipSuccess;
    if (!ctx->pipe_in) {
        e = hipStreamCreateWithFlags(&ctx->pipe_in, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->pipe_out, hipStreamNonBlocking);
        for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->pipe_ev[i], hipEventDisableTiming);
        if (e != hipSuccess) return fail(ctx, JDSP_EHIP, "stft pipeline: streams", e);
    }
    const size_t in_cap = sizeof(int16_t) * (size_t)(kPipeFrames * hop + n_fft), out_cap = sizeof(jdsp_c32) * (size_t)kPipeFrames * n_fft;
    for (int b = 0; b < 4; b++) {
        const size_t want = b < 2 ? in_cap : out_cap;
        if (ctx->pipe_cap[b] < want) {
            if (ctx->pipe_buf[b]) (void)hipFree(ctx->pipe_buf[b]);
            ctx->pipe_buf[b] = nullptr;
            ctx->pipe_cap[b] = 0;
            if ((e = hipMalloc(&ctx->pipe_buf[b], want)) != hipSuccess) return fail(ctx, JDSP_ENOMEM, "stft pipeline: buffers", e);
            ctx->pipe_cap[b] = want;
        }
    }
    hipEvent_t *ev_in = ctx->pipe_ev, *ev_comp = ctx->pipe_ev + 2, *ev_out = ctx->pipe_ev + 4;
    hipStream_t comp = ctx->stream;
    int rc = JDSP_OK;
    long c = 0;
    for (long f0 = 0; f0 < n_frames && !rc; f0 += kPipeFrames, c++) {
        const int b = (int)(c & 1);
        const long nf = n_frames - f0 < kPipeFrames ? n_frames - f0 : kPipeFrames;
        int16_t *d_in = (int16_t *)ctx->pipe_buf[b];
        jdsp_c32 *d_out = (jdsp_c32 *)ctx->pipe_buf[2 + b];
        if (c >= 2) e = hipStreamWaitEvent(ctx->pipe_in, ev_comp[b], 0);          // the transform that read d_in is done
        if (e == hipSuccess) e = hipMemcpyAsync(d_in, pcm_host + f0 * hop, sizeof(int16_t) * (size_t)((nf - 1) * hop + n_fft), hipMemcpyHostToDevice, ctx->pipe_in);
        if (e == hipSuccess) e = hipEventRecord(ev_in[b], ctx->pipe_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(comp, ev_in[b], 0);
        if (e == hipSuccess && c >= 2) e = hipStreamWaitEvent(comp, ev_out[b], 0);    // d_out has been copied out
        if (e != hipSuccess) { rc = fail(ctx, JDSP_EHIP, "stft pipeline: copy in", e); break; }
        rc = jdsp_stft_i16_dev(ctx, d_in, nf, n_fft, hop, d_out);
        if (rc) break;
        e = hipEventRecord(ev_comp[b], comp);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->pipe_out, ev_comp[b], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(spec_host + (size_t)f0 * n_fft, d_out, sizeof(jdsp_c32) * (size_t)nf * n_fft, hipMemcpyDeviceToHost, ctx->pipe_out);
        if (e == hipSuccess) e = hipEventRecord(ev_out[b], ctx->pipe_out);
        if (e != hipSuccess) rc = fail(ctx, JDSP_EHIP, "stft pipeline: copy out", e);
    }
    hipError_t e1 = hipStreamSynchronize(ctx->pipe_in), e2 = hipStreamSynchronize(comp), e3 = hipStreamSynchronize(ctx->pipe_out);
    if (!rc && (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)) rc = fail(ctx, JDSP_EHIP, "stft pipeline: sync", e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3));
    return rc;
}

static int ensure_c2c_tw(jdsp_ctx *ctx, int n_fft, int lg)
{
    if (ctx->c2c_tw[lg]) return JDSP_OK;
    std::vector<double2> host((size_t)n_fft / 2 + 1);
    jdsp::fill_c2c_twiddles(host.data(), n_fft);
    JDSP_HIP(ctx, hipMalloc((void **)&ctx->c2c_tw[lg], sizeof(double2) * host.size()));
    JDSP_HIP(ctx, hipMemcpy(ctx->c2c_tw[lg], host.data(), sizeof(double2) * host.size(), hipMemcpyHostToDevice));
    return JDSP_OK;
}

int jdsp_stft_i16_f64_dev(jdsp_ctx *ctx, const int16_t *pcm_dev, long n_frames, int n_fft, int hop, double *spec_dev)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_fft != 1024) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64_dev: n_fft must be 1024");
    if (hop < 1) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64_dev: hop must be >= 1");
    if (n_frames < 0 || (n_frames > 0 && (!pcm_dev || !spec_dev))) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64_dev: bad buffer");
    if (n_frames == 0) return JDSP_OK;
    if ((uintptr_t)pcm_dev & 1u) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64_dev: pcm must be 2-byte aligned");
    if ((uintptr_t)spec_dev & 15u) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64_dev: spec must be 16-byte aligned");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_c2c_tw(ctx, 512, 9);
    if (rc) return rc;
    if (!ctx->stft_f64_table) {
        std::vector<double> host(1024 + 2 * 512);
        jdsp::fill_stft1024_f64_table(host.data());
        JDSP_HIP(ctx, hipMalloc((void **)&ctx->stft_f64_table, sizeof(double) * host.size()));
        JDSP_HIP(ctx, hipMemcpy(ctx->stft_f64_table, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
    }
    // hop-512 batches out of HBM: the read pass of the FP32 path (stft_kernels.hip), slab by slab -- the transform then
    // finds its PCM in the Infinity Cache and the memory system sees a read stream, then a write stream
    const int rp = ctx->opt_stft_read_pass;
    const bool touch = hop == 512 && ((uintptr_t)pcm_dev & 15u) == 0 && (rp > 0 || (rp < 0 && n_frames >= 16384));
    const long slab = touch ? 65536 : n_frames;
    for (long f0 = 0; f0 < n_frames; f0 += slab) {
        const long nf = n_frames - f0 < slab ? n_frames - f0 : slab;
        if ((touch && jdsp::launch_pcm_read_pass(ctx->stream, ctx->n_cu, ctx->opt_stft_touch_wg, pcm_dev + f0 * hop, hop * (nf + 1))) ||
            jdsp::launch_stft1024_f64(ctx->stream, ctx->n_cu, pcm_dev + f0 * hop, nf, hop, ctx->stft_f64_table, ctx->c2c_tw[9],
                                      (double2 *)spec_dev + f0 * 1024, ctx->opt_stft_f64_kernel, ctx->opt_stft_f64_fpw)) {
            const hipError_t le = hipGetLastError();
            return fail(ctx, JDSP_EHIP, "stft f64 launch", le);
        }
    }
    return JDSP_OK;
}

int jdsp_stft_i16_f64(jdsp_ctx *ctx, const int16_t *pcm_host, long n_samples, int n_fft, int hop, double *spec_host,
                      long *n_frames_out)
{
    if (!ctx) return JDSP_EINVAL;
    if (n_fft != 1024 || hop < 1) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64: unsupported n_fft/hop");
    const long n_frames = n_samples >= n_fft ? (n_samples - n_fft) / hop + 1 : 0;
    if (n_frames_out) *n_frames_out = n_frames;
    if (n_frames == 0) return JDSP_OK;
    if (!pcm_host || !spec_host) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16_f64: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = sizeof(int16_t) * (size_t)((n_frames - 1) * hop + n_fft);
    const size_t out_bytes = 2 * sizeof(double) * (size_t)n_frames * (size_t)n_fft;
    int16_t *d_in = nullptr;
    double *d_out = nullptr;
    JDSP_HIP(ctx, hipMalloc((void **)&d_in, in_bytes));
    hipError_t e = hipMalloc((void **)&d_out, out_bytes);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, e == hipErrorOutOfMemory ? JDSP_ENOMEM : JDSP_EHIP, "jdsp_stft_i16_f64: hipMalloc", e);
    }
    int rc = JDSP_OK;
    if ((e = hipMemcpyAsync(d_in, pcm_host, in_bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16_f64: H2D", e);
    if (!rc) rc = jdsp_stft_i16_f64_dev(ctx, d_in, n_frames, n_fft, hop, d_out);
    if (!rc && (e = hipMemcpyAsync(spec_host, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16_f64: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16_f64: sync", e);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

int jdsp_stft_i16(jdsp_ctx *ctx, const int16_t *pcm_host, long n_samples, int n_fft, int hop, jdsp_c32 *spec_host,
                  long *n_frames_out)
{
    if (!ctx) return JDSP_EINVAL;
    if ((n_fft != 1024 && n_fft != 512) || hop < 1) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16: unsupported n_fft/hop");
    long n_frames = n_samples >= n_fft ? (n_samples - n_fft) / hop + 1 : 0;
    if (n_frames_out) *n_frames_out = n_frames;
    if (n_frames == 0) return JDSP_OK;
    if (!pcm_host || !spec_host) return fail(ctx, JDSP_EINVAL, "jdsp_stft_i16: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (n_frames > 4096 && (((long)hop * 4096 * 2) & 15) == 0 && is_pinned_host(pcm_host) && is_pinned_host(spec_host))
        return stft_pipelined(ctx, pcm_host, n_frames, n_fft, hop, spec_host);
    const size_t in_bytes = sizeof(int16_t) * (size_t)((n_frames - 1) * hop + n_fft);
    const size_t out_bytes = sizeof(jdsp_c32) * (size_t)n_frames * (size_t)n_fft;
    int16_t *d_in = nullptr;
    jdsp_c32 *d_out = nullptr;
    JDSP_HIP(ctx, hipMalloc((void **)&d_in, in_bytes));
    hipError_t e = hipMalloc((void **)&d_out, out_bytes);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return fail(ctx, e == hipErrorOutOfMemory ? JDSP_ENOMEM : JDSP_EHIP, "jdsp_stft_i16: hipMalloc", e);
    }
    int rc = JDSP_OK;
    if ((e = hipMemcpyAsync(d_in, pcm_host, in_bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16: H2D", e);
    if (!rc) rc = jdsp_stft_i16_dev(ctx, d_in, n_frames, n_fft, hop, d_out);
    if (!rc && (e = hipMemcpyAsync(spec_host, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_stft_i16: sync", e);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
