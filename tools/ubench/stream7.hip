// Scratch micro-benchmark (round 4): does the ORDER in which the chip walks a 1 GiB read matter?
//   front : grid-stride (the chip reads one moving window, as rd<> of stream2.hip)
//   rows  : block b walks its own contiguous region of bytes / grid (the tangent-weight stream of the K-column product:
//           a block per pair of weight rows)
//   chunks: block b reads chunk b, b + grid, ... of `chunk` bytes (blocks in address order, each chunk contiguous)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;
template<int U> __global__ __launch_bounds__(256) void rd_front(const v4* __restrict__ p, long n4, float* out){
  float s=0; const long stride=(long)gridDim.x*blockDim.x; long i=(long)blockIdx.x*blockDim.x+threadIdx.x;
  for(; i+(U-1)*stride<n4; i+=U*stride){ v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]=p[i+u*stride];
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w; }
  if(s==123.456f) out[0]=s;
}
// block b: region [b*per, (b+1)*per) float4, walked 256*U float4 per trip
template<int U> __global__ __launch_bounds__(256) void rd_rows(const v4* __restrict__ p, long per4, float* out){
  float s=0; const v4* q=p+(long)blockIdx.x*per4;
  for(long i=threadIdx.x; i+(U-1)*256<per4; i+=U*256){ v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]=q[i+u*256];
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w; }
  if(s==123.456f) out[0]=s;
}
// one chunk of 256*U float4 per block, grid = n4 / (256 U) blocks in address order (non-persistent)
template<int U> __global__ __launch_bounds__(256) void rd_chunks(const v4* __restrict__ p, float* out){
  float s=0; const v4* q=p+(long)blockIdx.x*256*U+threadIdx.x; v4 v[U];
  #pragma unroll
  for(int u=0;u<U;u++) v[u]=q[u*256];
  #pragma unroll
  for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w;
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
  const long GB=1L<<30; float* big; CK(hipMalloc(&big,4*GB)); CK(hipMemset(big,0,4*GB));
  const long n=GB/4, n4=n/4; long slot=0;
  auto src=[&]{ return (const v4*)(big+(slot++%4)*n); };
  for(int bpc : {1,2,4,8}){ int g=256*bpc;
    float a=timeit([&]{hipLaunchKernelGGL((rd_front<4>),dim3(g),dim3(256),0,0,src(),n4,out);},8);
    float b=timeit([&]{hipLaunchKernelGGL((rd_rows<4>),dim3(g),dim3(256),0,0,src(),n4/g,out);},8);
    float c=timeit([&]{hipLaunchKernelGGL((rd_rows<8>),dim3(g),dim3(256),0,0,src(),n4/g,out);},8);
    printf("1 GiB, %d blocks/CU: front U=4 %7.1f us %5.2f TB/s | rows U=4 %7.1f us %5.2f TB/s | rows U=8 %7.1f us %5.2f TB/s\n",bpc,a,GB/a/1e6,b,GB/b/1e6,c,GB/c/1e6);
  }
  // rows with MORE blocks than resident slots (2688 x ... like one block per weight row): 2048, 4096, 16384 regions
  for(int g : {1344, 2688, 16384}){
    float b=timeit([&]{hipLaunchKernelGGL((rd_rows<4>),dim3(g),dim3(256),0,0,src(),n4/g,out);},8);
    printf("1 GiB, rows, %5d blocks (region %6.1f KB): %7.1f us %5.2f TB/s\n",g,GB/1024.0/g,b,GB/b/1e6);
  }
  { float c=timeit([&]{hipLaunchKernelGGL((rd_chunks<4>),dim3((unsigned)(n4/1024)),dim3(256),0,0,src(),out);},8);
    printf("1 GiB, chunks of 16 KB, one per block: %7.1f us %5.2f TB/s\n",c,GB/c/1e6);
    float d=timeit([&]{hipLaunchKernelGGL((rd_chunks<8>),dim3((unsigned)(n4/2048)),dim3(256),0,0,src(),out);},8);
    printf("1 GiB, chunks of 32 KB, one per block: %7.1f us %5.2f TB/s\n",d,GB/d/1e6);
    float e=timeit([&]{hipLaunchKernelGGL((rd_chunks<2>),dim3((unsigned)(n4/512)),dim3(256),0,0,src(),out);},8);
    printf("1 GiB, chunks of  8 KB, one per block: %7.1f us %5.2f TB/s\n",e,GB/e/1e6); }
  return 0;
}
