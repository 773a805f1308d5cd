// Scratch micro-benchmarks: how fast can gfx950 stream rows in the patterns the MLP kernels use?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// pattern A: grid-stride float4 copy-less read (sum) -- the classic ceiling
__global__ void read_linear(const float4* __restrict__ p, long n4, float* out){
  float s=0; for(long i=(long)blockIdx.x*blockDim.x+threadIdx.x;i<n4;i+=(long)gridDim.x*blockDim.x){float4 v=p[i]; s+=v.x+v.y+v.z+v.w;}
  if(s==123.456f) out[0]=s;
}
// pattern B: wave w owns R rows; reads each row in 1KB pieces, U pieces in flight
template<int R,int U>
__global__ __launch_bounds__(512) void read_rows(const float* __restrict__ W,const float* __restrict__ V,int d_in,int d_out,float* out){
  int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  int j0=(blockIdx.x*(blockDim.x>>6)+wave)*R; if(j0>=d_out) return;
  float s=0;
  for(int k0=0;k0<d_in;k0+=256*U){
    float4 w[R][U],v[R][U];
    #pragma unroll
    for(int r=0;r<R;r++)
    #pragma unroll
    for(int u=0;u<U;u++){ int k=k0+u*256+lane*4; int kk=k<d_in?k:0; long row=(long)min(j0+r,d_out-1)*d_in;
      w[r][u]=*(const float4*)(W+row+kk); v[r][u]=*(const float4*)(V+row+kk);}
    #pragma unroll
    for(int r=0;r<R;r++)
    #pragma unroll
    for(int u=0;u<U;u++){ s+=w[r][u].x*v[r][u].x+w[r][u].y+v[r][u].z+w[r][u].w; }
  }
  if(s==123.456f) out[0]=s;
}
// pattern C: MFMA-operand layout: lane (idx=l&15, s=l>>4) reads float4 of row idx at k+4s: 16 rows x 64 B per load
template<int RG,int U,int SEG>
__global__ __launch_bounds__(512) void read_mfma(const float* __restrict__ W,const float* __restrict__ V,int d_in,int d_out,int kparts,float* out){
  int lane=threadIdx.x&63, wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  // SEG = lanes per row segment (4 -> 64B, 8 -> 128B, 16 -> 256B); rows per load = 64/SEG
  int idx=lane/SEG, s4=(lane%SEG)*4; int rows=64/SEG;
  int j0=blockIdx.x*RG*(rows/2);  // half rows W half rows V
  int klen=d_in/kparts; klen-=klen%(SEG*4); int kb=wave*klen, ke=(wave==kparts-1)?d_in-(d_in%(SEG*4)):kb+klen;
  float s=0;
  const float* p[RG];
  for(int g=0;g<RG;g++){ int row=min(j0+g*(rows/2)+(idx%(rows/2)),d_out-1); p[g]=((idx>=rows/2)?V:W)+(long)row*d_in+s4; }
  for(int k=kb;k<ke;k+=SEG*4*U){
    float4 a[RG][U];
    #pragma unroll
    for(int u=0;u<U;u++)
    #pragma unroll
    for(int g=0;g<RG;g++){ int kk=min(k+u*SEG*4,ke-SEG*4); a[g][u]=*(const float4*)(p[g]+kk);}
    #pragma unroll
    for(int u=0;u<U;u++)
    #pragma unroll
    for(int g=0;g<RG;g++) s+=a[g][u].x+a[g][u].y*a[g][u].z+a[g][u].w;
  }
  if(s==123.456f) out[0]=s;
}
template<typename F> float timeit(F f,int iters){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<3;i++) f();
  hipEventRecord(a); for(int i=0;i<iters;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/iters*1000.f;
}
int main(){
  const int d_in=2688,d_out=2688; long n=(long)d_in*d_out;
  // several distinct buffers to defeat the 256MB infinity cache when wanted
  const int NBUF=12; float *W[NBUF],*V[NBUF],*out; CK(hipMalloc(&out,64));
  for(int i=0;i<NBUF;i++){CK(hipMalloc(&W[i],n*4)); CK(hipMalloc(&V[i],n*4)); CK(hipMemset(W[i],0,n*4)); CK(hipMemset(V[i],0,n*4));}
  int it=0;
  for(int nb: {1,NBUF}){
    printf("--- %d buffer set(s) (%s)\n",nb,nb==1?"cache-resident":"HBM");
    float us=timeit([&]{int i=(it++)%nb; hipLaunchKernelGGL(read_linear,dim3(2048),dim3(256),0,0,(const float4*)W[i],n/4,out); hipLaunchKernelGGL(read_linear,dim3(2048),dim3(256),0,0,(const float4*)V[i],n/4,out);},50);
    printf("linear 2x: %.1f us -> %.2f TB/s\n",us,2*n*4/us/1e6);
    #define RUN(R,U,WAVES) { int rows_per_block=WAVES*R; dim3 g((d_out+rows_per_block-1)/rows_per_block), b(WAVES*64); \
      float us=timeit([&]{int i=(it++)%nb; hipLaunchKernelGGL((read_rows<R,U>),g,b,0,0,W[i],V[i],d_in,d_out,out);},50); \
      printf("rows R=%d U=%d waves=%d blocks=%d: %.1f us -> %.2f TB/s\n",R,U,WAVES,g.x,us,2*n*4/us/1e6);}
    #define RUNM(RG,U,SEG) { int rows=64/SEG; int rpb=RG*(rows/2); dim3 g((d_out+rpb-1)/rpb), b(512); \
      float us=timeit([&]{int i=(it++)%nb; hipLaunchKernelGGL((read_mfma<RG,U,SEG>),g,b,0,0,W[i],V[i],d_in,d_out,8,out);},50); \
      printf("mfma-layout RG=%d U=%d SEG=%d(%dB/row) blocks=%d: %.1f us -> %.2f TB/s\n",RG,U,SEG,SEG*16,g.x,us,2*n*4/us/1e6);}
    RUNM(2,4,4) RUNM(2,8,4) RUNM(1,8,4) RUNM(2,4,8) RUNM(4,4,8) RUNM(4,4,16) RUNM(8,2,16) RUNM(8,4,16)
    RUN(2,1,4) RUN(2,2,8) RUN(2,2,4) RUN(1,2,4) RUN(1,4,4) RUN(1,4,8) RUN(2,4,4) RUN(1,8,4) RUN(1,2,2) RUN(1,4,1)
  }
  return 0;
}
