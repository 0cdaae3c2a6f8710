// valu_rate.hip -- issue rate of the instruction kinds the transform kernels are made of (GPU box only).
// Every case: N_ITER iterations of 16 independent instructions per wave, `waves_per_simd` waves per SIMD on every CU;
// prints cycles per wave-instruction per SIMD (4.0 = full rate for a wave64 on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float cfv __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int N_ITER = 16384;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(float *out, float seed)
{
    cfv a[16];
    float s[16];
    int iv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = cfv{seed + i, seed - i}; s[i] = seed * (i + 1); iv[i] = (int)threadIdx.x * (i + 3); }
    const cfv b = {1.0001f, 0.9999f}, c = {0.5f, -0.25f};
    const int addr = ((64 - (int)threadIdx.x) & 63) << 2;
    const unsigned long long mask = __ballot(threadIdx.x & 1);
    for (int it = 0; it < N_ITER; it++) {
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA_SEL(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKADD_SEL(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
#define PKMUL_SEL(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[1,0]" : "+v"(a[i]) : "v"(b));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
#define RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(s[i]));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b.x));
#define CVT_SDWA(i) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(s[i]) : "v"(iv[i]));
#define BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(s[i]) : "v"(addr));
#define BPERM_NOWAIT(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(s[i]) : "v"(addr));
#define DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
#define MUL24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]));
#define MULF64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&a[i]) : "v"(*(const double *)&b));
#define CVTF64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[i]) : "v"(*(double *)&a[i]));
#define ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(s[(i + 1) & 15]));
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "s"(mask));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(s[i]), "v"(b.x) : "vcc");
#define ANDB(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]));
#define CVTI(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(iv[i]) : "v"(s[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b.x));
#define MADU24(i) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(iv[i]) : "v"(iv[(i + 1) & 15]));
        if (KIND == 0) { REP16(PKFMA) }
        if (KIND == 1) { REP16(PKFMA_SEL) }
        if (KIND == 2) { REP16(PKADD_SEL) }
        if (KIND == 3) { REP16(PKMUL_SEL) }
        if (KIND == 4) { REP16(FMA) }
        if (KIND == 5) { REP16(RSQ) }
        if (KIND == 6) { REP16(CNDMASK) }
        if (KIND == 7) { REP16(CVT_SDWA) }
        if (KIND == 8) { REP16(BPERM_NOWAIT) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (KIND == 9) { REP16(DPPMOV) }
        if (KIND == 10) { REP16(MUL24) }
        if (KIND == 11) { REP16(MULF64) }
        if (KIND == 12) { REP16(CVTF64) }
        if (KIND == 13) { REP16(ADDF) }
        if (KIND == 14) { REP16(MOV) }
        if (KIND == 15) { REP16(CND64) }
        if (KIND == 16) { REP16(CMP) }
        if (KIND == 17) { REP16(ANDB) }
        if (KIND == 18) { REP16(CVTI) }
        if (KIND == 19) { REP16(PKMUL) }
        if (KIND == 20) { REP16(PKADD) }
        if (KIND == 21) { REP16(MAXF) }
        if (KIND == 22) { REP16(MADU24) }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += a[i].x + a[i].y + s[i] + (float)iv[i];
    if (acc == 1234.5678f) out[threadIdx.x] = acc;
}

template <int KIND>
static float run(int waves_per_simd, int n_cu, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = n_cu * 4 * waves_per_simd;
    rate_kernel<KIND><<<grid, 64>>>(d_out, 1.0f);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        rate_kernel<KIND><<<grid, 64>>>(d_out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

typedef float (*runner)(int, int, float *);
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    float *d_out; CK(hipMalloc(&d_out, 4096));
    const char *names[] = {"v_pk_fma_f32", "v_pk_fma_f32 op_sel/neg", "v_pk_add_f32 op_sel/neg", "v_pk_mul_f32 op_sel/neg", "v_fma_f32", "v_rsq_f32",
                           "v_cndmask_b32 (vcc)", "v_cvt_f32_i32 sdwa", "ds_bpermute_b32", "v_mov_b32 dpp row_ror", "v_mul_i32_i24", "v_mul_f64", "v_cvt_i32_f64",
                           "v_add_f32", "v_mov_b32", "v_cndmask_b32_e64 (sgpr)", "v_cmp_lt_f32 vcc", "v_and_b32", "v_cvt_i32_f32", "v_pk_mul_f32", "v_pk_add_f32",
                           "v_max_f32", "v_mad_u32_u24"};
    runner runs[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>, run<14>, run<15>,
                     run<16>, run<17>, run<18>, run<19>, run<20>, run<21>, run<22>};
    for (int i = 0; i < 40; i++) run<4>(4, n_cu, d_out);              // spin the clocks up
    const float ref = run<4>(4, n_cu, d_out);                         // v_fma_f32, 4 waves per SIMD
    printf("%d CUs, nominal clock %.2f GHz.  Times relative to v_fma_f32 at 4 waves per SIMD (= 1.00; %.3f ms for %d x 16 instructions x 4 waves)\n", n_cu, ghz, ref, N_ITER);
    printf("%-28s %10s %10s %10s   (cost per wave64 instruction per SIMD, in v_fma_f32 issue slots)\n", "instruction", "1 wave", "2 waves", "4 waves");
    for (int k = 0; k < 23; k++) {
        printf("%-28s", names[k]);
        for (int w = 0; w < 3; w++) {
            const float t = runs[k](1 << w, n_cu, d_out);
            printf(" %10.2f", t / ref * 4.0 / (1 << w));
        }
        printf("\n");
    }
    return 0;
}
