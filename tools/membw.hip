// membw.hip -- what the memory system gives for the STFT's traffic shape (GPU box only):
// 512 MiB written + 64 MiB read per launch, in several store patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// A: grid-stride, every wave-instruction writes 1 KiB, consecutive waves consecutive KiB
template<bool NT> __global__ void fill_stride(f32x4* out, size_t n4, float v){
  size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  f32x4 x={v,v,v,v};
  for(;i<n4;i+=st){ if(NT) __builtin_nontemporal_store(x,out+i); else out[i]=x; }
}
// B: each wave owns `chunk` consecutive 8 KiB frames (the STFT shape): 8 x 1 KiB stores per frame
template<bool NT> __global__ __launch_bounds__(64) void fill_frames(f32x4* out, long nframes, int fpw, float v){
  long f0=(long)blockIdx.x*fpw, f1=f0+fpw; if(f1>nframes) f1=nframes;
  f32x4 x={v,v,v,v};
  for(long f=f0;f<f1;f++){ f32x4* d=out+f*512+threadIdx.x;
#pragma unroll
    for(int j=0;j<8;j++){ if(NT) __builtin_nontemporal_store(x,d+64*j); else d[64*j]=x; } }
}
// C: same as B plus the 1 KiB read per frame (one dwordx4 per lane), folded into the data
template<bool NT> __global__ __launch_bounds__(64) void copy_frames(const f32x4* in, f32x4* out, long nframes, int fpw){
  long f0=(long)blockIdx.x*fpw, f1=f0+fpw; if(f1>nframes) f1=nframes;
  for(long f=f0;f<f1;f++){ f32x4 x=in[f*64+threadIdx.x]; f32x4* d=out+f*512+threadIdx.x;
#pragma unroll
    for(int j=0;j<8;j++){ if(NT) __builtin_nontemporal_store(x,d+64*j); else d[64*j]=x; } }
}
// E: the reads alone (1 KiB per wave), result folded into one never-taken store
__global__ __launch_bounds__(64) void read_frames(const f32x4* in, f32x4* out, long nframes){
  long f=blockIdx.x; f32x4 x=in[f*64+threadIdx.x]; if(x.x==12345.678f) out[f]=x;
}
// D: 256-thread blocks, block owns consecutive frames, waves interleave frames
template<bool NT> __global__ __launch_bounds__(256) void fill_frames_wg(f32x4* out, long nframes, int fpb, float v){
  long f0=(long)blockIdx.x*fpb, f1=f0+fpb; if(f1>nframes) f1=nframes;
  int w=threadIdx.x>>6, l=threadIdx.x&63; f32x4 x={v,v,v,v};
  for(long f=f0+w;f<f1;f+=4){ f32x4* d=out+f*512+l;
#pragma unroll
    for(int j=0;j<8;j++){ if(NT) __builtin_nontemporal_store(x,d+64*j); else d[64*j]=x; } }
}
template<class F> float timeit(F f,int it=20){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize();
  std::vector<float> t; for(int r=0;r<5;r++){ hipEventRecord(a); for(int i=0;i<it;i++) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); t.push_back(ms/it*1e3f);} std::sort(t.begin(),t.end()); return t[2]; }
int main(){
  const long B=65536; size_t obytes=(size_t)B*8192, ibytes=(size_t)B*1024;
  f32x4 *out,*in; CK(hipMalloc(&out,obytes)); CK(hipMalloc(&in,ibytes)); CK(hipMemset(in,0,ibytes));
  size_t n4=obytes/16;
  auto rep=[&](const char* n,float us,double bytes){ printf("%-34s %8.1f us  %7.0f GB/s\n",n,us,bytes/us/1e3); };
  for(int g: {2048,4096,8192,16384}){ char nm[64];
    snprintf(nm,64,"A stride grid=%d x256",g); rep(nm,timeit([&]{fill_stride<false><<<g,256>>>(out,n4,1.f);}),obytes);
    snprintf(nm,64,"A stride NT grid=%d x256",g); rep(nm,timeit([&]{fill_stride<true><<<g,256>>>(out,n4,1.f);}),obytes); }
  for(int fpw: {1,4,8,16,32}){ char nm[64]; int grid=(B+fpw-1)/fpw;
    snprintf(nm,64,"B frames fpw=%d",fpw); rep(nm,timeit([&]{fill_frames<false><<<grid,64>>>(out,B,fpw,1.f);}),obytes);
    snprintf(nm,64,"B frames NT fpw=%d",fpw); rep(nm,timeit([&]{fill_frames<true><<<grid,64>>>(out,B,fpw,1.f);}),obytes);
    snprintf(nm,64,"C copy frames fpw=%d",fpw); rep(nm,timeit([&]{copy_frames<false><<<grid,64>>>(in,out,B,fpw);}),obytes+ibytes);
    snprintf(nm,64,"C copy frames NT fpw=%d",fpw); rep(nm,timeit([&]{copy_frames<true><<<grid,64>>>(in,out,B,fpw);}),obytes+ibytes); }
  for(int fpb: {16,64,256}){ char nm[64]; int grid=(B+fpb-1)/fpb;
    snprintf(nm,64,"D wg256 fpb=%d",fpb); rep(nm,timeit([&]{fill_frames_wg<false><<<grid,256>>>(out,B,fpb,1.f);}),obytes);
    snprintf(nm,64,"D wg256 NT fpb=%d",fpb); rep(nm,timeit([&]{fill_frames_wg<true><<<grid,256>>>(out,B,fpb,1.f);}),obytes); }
  // COLD input: the same copy shape rotating over 6 distinct 64 MiB inputs (384 MiB > the 256 MiB Infinity Cache),
  // so every launch's reads come from HBM; and the reads alone.
  {
    f32x4* ins[6]; for(int i=0;i<6;i++){ CK(hipMalloc(&ins[i],ibytes)); CK(hipMemset(ins[i],0,ibytes)); }
    int rot=0;
    for(int fpw: {1,2,4,8}){ char nm[64]; int grid=(B+fpw-1)/fpw;
      snprintf(nm,64,"C copy frames NT fpw=%d COLD in",fpw); rep(nm,timeit([&]{copy_frames<true><<<grid,64>>>(ins[(rot++)%6],out,B,fpw);},60),obytes+ibytes);
      snprintf(nm,64,"C copy frames NT fpw=%d warm in",fpw); rep(nm,timeit([&]{copy_frames<true><<<grid,64>>>(ins[0],out,B,fpw);},60),obytes+ibytes); }
    rep("E read-only 64MiB COLD (fpw=1)",timeit([&]{read_frames<<<B,64>>>(ins[(rot++)%6],out,B);},60),ibytes);
    rep("E read-only 64MiB warm (fpw=1)",timeit([&]{read_frames<<<B,64>>>(ins[0],out,B);},60),ibytes);
  }
  // plain float4 copy 512 MiB -> 512 MiB for calibration against the guide's 6.29 TB/s
  f32x4* src; CK(hipMalloc(&src,obytes)); CK(hipMemset(src,0,obytes));
  rep("hipMemcpyDtoD 512MiB (r+w)",timeit([&]{hipMemcpyAsync(out,src,obytes,hipMemcpyDeviceToDevice,0);}),2.0*obytes);
  rep("hipMemset 512MiB",timeit([&]{hipMemsetAsync(out,0,obytes,0);}),obytes);
  return 0;
}
