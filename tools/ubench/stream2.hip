// Scratch micro-benchmark: what does a READ-ONLY stream reach on this GPU, as a function of buffer
// size, blocks per CU, load flavour (plain / nontemporal / LDS-DMA) and unroll?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;

template<int U, int MODE>  // MODE 0 plain, 1 nontemporal
__global__ __launch_bounds__(256) void rd(const v4* __restrict__ p, long n4, float* out){
  float s=0;
  const long stride=(long)gridDim.x*blockDim.x;
  long i=(long)blockIdx.x*blockDim.x+threadIdx.x;
  for(; i+(U-1)*stride<n4; i+=U*stride){
    v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]= MODE==1 ? __builtin_nontemporal_load(p+i+u*stride) : p[i+u*stride];
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w;
  }
  for(; i<n4; i+=stride){ v4 v=p[i]; s+=v.x+v.y+v.z+v.w; }
  if(s==123.456f) out[0]=s;
}
// LDS-DMA: each wave streams 1 KiB pieces straight into LDS (no VGPR), then touches them
template<int SLOTS>
__global__ __launch_bounds__(256) void rd_lds(const float* __restrict__ p, long n, float* out){
  __shared__ __attribute__((aligned(16))) float buf[4][SLOTS][256];   // per wave SLOTS x 1 KiB
  const int wave=threadIdx.x>>6, lane=threadIdx.x&63;
  const long wid=(long)blockIdx.x*4+wave, nw=(long)gridDim.x*4;
  float s=0;
  for(long base=wid*SLOTS*256; base+SLOTS*256<=n; base+=nw*SLOTS*256){
    #pragma unroll
    for(int k=0;k<SLOTS;k++)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p+base+k*256+lane*4),
                                       (void __attribute__((address_space(3)))*)(&buf[wave][k][0]), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0)
    #pragma unroll
    for(int k=0;k<SLOTS;k++) s+=buf[wave][k][lane];
  }
  if(s==123.456f) out[0]=s;
}
template<typename F> float timeit(F f,int iters){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<2;i++) f();
  hipEventRecord(a); for(int i=0;i<iters;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/iters*1000.f;
}
int main(){
  float* out; CK(hipMalloc(&out,64));
  const long GB=1L<<30;
  float* big; CK(hipMalloc(&big,4*GB)); CK(hipMemset(big,0,4*GB));   // 4 GiB: far beyond the 256 MiB cache
  for(long mb : {58L, 256L, 1024L, 4096L}){
    long n=mb*1024*1024/4, n4=n/4;
    int reps = mb>=1024?5:40;
    // rotate over the 4 GiB so that a small buffer is never cache resident
    long nslots=(4*GB/4)/n; long slot=0;
    printf("--- %ld MiB per launch\n",mb);
    for(int bpc : {2,4,8,16}){
      int g=256*bpc;
      #define RUN(U,MODE,name) { float us=timeit([&]{ const v4* q=(const v4*)(big+(slot++%nslots)*n); hipLaunchKernelGGL((rd<U,MODE>),dim3(g),dim3(256),0,0,q,n4,out);},reps); \
        printf("%-10s U=%d blocks/CU=%2d: %8.1f us  %.2f TB/s\n",name,U,bpc,us,mb*1.048576/us); }
      RUN(4,0,"plain") RUN(8,0,"plain") RUN(8,1,"nontemp")
      { float us=timeit([&]{ const float* q=big+(slot++%nslots)*n; hipLaunchKernelGGL((rd_lds<8>),dim3(g),dim3(256),0,0,q,n,out);},reps);
        printf("%-10s S=8 blocks/CU=%2d: %8.1f us  %.2f TB/s\n","lds-dma",bpc,us,mb*1.048576/us); }
    }
  }
  return 0;
}
