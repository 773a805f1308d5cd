// Scratch micro-benchmark: WRITE-only stream ceiling (plain / nontemporal stores), sizes and occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;
template<int MODE>
__global__ __launch_bounds__(256) void wr(v4* __restrict__ p, long n4, float x){
  const long stride=(long)gridDim.x*blockDim.x;
  const v4 v={x,x+1,x+2,x+3};
  for(long i=(long)blockIdx.x*blockDim.x+threadIdx.x;i<n4;i+=stride){
    if(MODE==1) __builtin_nontemporal_store(v,p+i); else p[i]=v;
  }
}
template<typename F> float timeit(F f,int iters){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<2;i++) f();
  hipEventRecord(a); for(int i=0;i<iters;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/iters*1000.f;
}
int main(){
  const long GB=1L<<30; float* big; CK(hipMalloc(&big,4*GB));
  for(long mb : {40L, 256L, 1024L}){
    long n=mb*1024*1024/4, n4=n/4; int reps= mb>=1024?5:40; long nslots=(4*GB/4)/n, slot=0;
    printf("--- %ld MiB per launch\n",mb);
    for(int bpc : {2,4,8,16}){
      int g=256*bpc;
      { float us=timeit([&]{ v4* q=(v4*)(big+(slot++%nslots)*n); hipLaunchKernelGGL((wr<0>),dim3(g),dim3(256),0,0,q,n4,1.f);},reps);
        printf("plain   blocks/CU=%2d: %8.1f us  %.2f TB/s\n",bpc,us,mb*1.048576/us); }
      { float us=timeit([&]{ v4* q=(v4*)(big+(slot++%nslots)*n); hipLaunchKernelGGL((wr<1>),dim3(g),dim3(256),0,0,q,n4,1.f);},reps);
        printf("nontemp blocks/CU=%2d: %8.1f us  %.2f TB/s\n",bpc,us,mb*1.048576/us); }
    }
  }
  return 0;
}
