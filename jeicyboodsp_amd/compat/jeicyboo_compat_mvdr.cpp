// jeicyboo_compat_mvdr.cpp -- see jeicyboo_compat_mvdr.h.
#include "jeicyboo_compat_mvdr.h"

#include <cstdio>
#include <cstdlib>

namespace {

jdsp_ctx *g_ctx = nullptr;
jdsp_mvdr *g_h = nullptr;
double g_dtime = 0;

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "jeicyboo_compat_mvdr: %s: %s\n", what, jdsp_last_error(g_ctx));
    abort();
}
#define CK(call) do { if ((call) != JDSP_OK) die(#call); } while (0)

jdsp_ctx *context()
{
    if (!g_ctx) {
        const char *dev = getenv("JDSP_DEVICE");
        if (jdsp_create(dev ? atoi(dev) : 0, &g_ctx) != JDSP_OK) die("jdsp_create");
    }
    return g_ctx;
}

}  // namespace

void JeicybooMvdrReset(void)
{
    if (g_h) { jdsp_mvdr_destroy(g_h); g_h = nullptr; }
}

bool VoiceActivityDetection(short *block, int n)
{
    if (n != 512) { fprintf(stderr, "VoiceActivityDetection: iFrameCount must be 512\n"); abort(); }
    int64_t e = 0;
    CK(jdsp_vad_blocks_ex(context(), JDSP_VAD_MVDR, 512, block, 1, nullptr, &e, nullptr));   // frame offset KEEP_LEN 511 (:37)
    return e > 716800;                                    // dEnergy = sum / 1024 > THRESHOLD_OF_ENERGY 700 (:233)
}

static jdsp_mvdr *handle(double d_time)
{
    if (g_h && d_time != g_dtime) JeicybooMvdrReset();
    if (!g_h) {
        CK(jdsp_mvdr_create(context(), d_time, &g_h));
        g_dtime = d_time;
    }
    return g_h;
}

// :244-270 -- the frame's contribution is ADDED to the caller's rgdSpatialCorr, whatever the caller did before
void EstimateSpatialCorrMtx(short *temp_l, short *temp_r, int, double (*corr)[2], int n)
{
    if (n != 1024) { fprintf(stderr, "EstimateSpatialCorrMtx: iFrameCount must be 1024\n"); abort(); }
    CK(jdsp_mvdr_estimate_corr(handle(g_dtime), temp_l, temp_r, 1, &corr[0][0]));
}

// :124-205 -- weights from the CALLER's rgdSpatialCorr (read, never written); statics = the handle's keep buffers
bool ProcessMVDR(short *left, short *right, int n, short *out, double d_time, double (*corr)[2])
{
    if (n != 512) { fprintf(stderr, "ProcessMVDR: iBlockLen must be 512\n"); abort(); }
    long n_out = 0;
    CK(jdsp_mvdr_apply(handle(d_time), left, right, 1, &corr[0][0], out, nullptr, &n_out));
    return n_out > 0;                                     // :201-204: false for the first block
}
