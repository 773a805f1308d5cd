// Scratch micro-benchmark (round 2): the questions behind the C2 matvec chain design.
//  E1  cache-hot vs cache-cold reads of matvec-sized buffers (is a re-read of W served faster than HBM?)
//  E2  does a hot 40 MB W survive 40 MB of streamed V reads + 40 MB of streamed result writes
//      (plain vs non-temporal policy on the streams)?
//  E3  read-only and write-only streams in ONE launch (roles by block) vs two launches
//  E4  write-stream bandwidth at matvec sizes
//  E5  a chain of dependent small launches vs one launch of the same bytes
//  E6  120 MB in one launch: 80 MB read + 40 MB written by different blocks (floor of a fused matvec)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;

template<int U, int MODE>  // MODE 0 plain, 1 nontemporal
__device__ __forceinline__ float rd_body(const v4* __restrict__ p, long n4, long tid, long nthreads){
  float s=0;
  long i=tid;
  for(; i+(U-1)*nthreads<n4; i+=U*nthreads){
    v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]= MODE==1 ? __builtin_nontemporal_load(p+i+u*nthreads) : p[i+u*nthreads];
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w;
  }
  for(; i<n4; i+=nthreads){ v4 v=p[i]; s+=v.x+v.y+v.z+v.w; }
  return s;
}
template<int MODE>
__device__ __forceinline__ void wr_body(v4* __restrict__ p, long n4, long tid, long nthreads, float val){
  const v4 v={val,val+1,val+2,val+3};
  for(long i=tid;i<n4;i+=nthreads){ if(MODE==1) __builtin_nontemporal_store(v,p+i); else p[i]=v; }
}
template<int U,int MODE>
__global__ __launch_bounds__(256) void rd(const v4* __restrict__ p, long n4, float* out){
  float s=rd_body<U,MODE>(p,n4,(long)blockIdx.x*blockDim.x+threadIdx.x,(long)gridDim.x*blockDim.x);
  if(s==123.456f) out[0]=s;
}
template<int MODE>
__global__ __launch_bounds__(256) void wr(v4* __restrict__ p, long n4, float val){
  wr_body<MODE>(p,n4,(long)blockIdx.x*blockDim.x+threadIdx.x,(long)gridDim.x*blockDim.x,val);
}
// roles by block: blocks [0, rblocks) read pr, the others write pw
template<int RMODE,int WMODE>
__global__ __launch_bounds__(256) void rw(const v4* __restrict__ pr, long nr4, v4* __restrict__ pw, long nw4, int rblocks, float* out){
  if((int)blockIdx.x<rblocks){
    float s=rd_body<4,RMODE>(pr,nr4,(long)blockIdx.x*blockDim.x+threadIdx.x,(long)rblocks*blockDim.x);
    if(s==123.456f) out[0]=s;
  } else {
    wr_body<WMODE>(pw,nw4,(long)(blockIdx.x-rblocks)*blockDim.x+threadIdx.x,(long)(gridDim.x-rblocks)*blockDim.x,1.f);
  }
}
// interleaved roles: even blocks read, odd blocks write (both kinds on every CU / XCD)
template<int RMODE,int WMODE>
__global__ __launch_bounds__(256) void rw_il(const v4* __restrict__ pr, long nr4, v4* __restrict__ pw, long nw4, int rnum, int period, float* out){
  const int grp=blockIdx.x/period, r=blockIdx.x%period, ngrp=gridDim.x/period;
  if(r<rnum){
    const long b=(long)grp*rnum+r, nb=(long)ngrp*rnum;
    float s=rd_body<4,RMODE>(pr,nr4,b*blockDim.x+threadIdx.x,nb*blockDim.x);
    if(s==123.456f) out[0]=s;
  } else {
    const long b=(long)grp*(period-rnum)+(r-rnum), nb=(long)ngrp*(period-rnum);
    wr_body<WMODE>(pw,nw4,b*blockDim.x+threadIdx.x,nb*blockDim.x,1.f);
  }
}
template<typename F> float timeit(F f,int iters){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<3;i++) f();
  hipEventRecord(a); for(int i=0;i<iters;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/iters*1000.f;
}
int main(){
  float* out; CK(hipMalloc(&out,64));
  const long GB=1L<<30, MB=1L<<20;
  float* big; CK(hipMalloc(&big,4*GB)); CK(hipMemset(big,0,4*GB));
  long slot=0;
  auto cold=[&](long bytes)->float*{ long nslots=4*GB/bytes; return big+((slot++)%nslots)*(bytes/4); };
  const int R=40;
  printf("=== E1 hot vs cold reads (2 blocks/CU, U=4)\n");
  for(long mb : {11L,22L,28L,40L,58L,80L,120L,200L}){
    long bytes=mb*MB, n4=bytes/16;
    float* hotbuf=big;  // fixed address
    float c0=timeit([&]{ hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);},R);
    float c1=timeit([&]{ hipLaunchKernelGGL((rd<4,1>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);},R);
    float h0=timeit([&]{ hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)hotbuf,n4,out);},R);
    float h1=timeit([&]{ hipLaunchKernelGGL((rd<4,1>),dim3(512),dim3(256),0,0,(const v4*)hotbuf,n4,out);},R);
    float h4=timeit([&]{ hipLaunchKernelGGL((rd<4,0>),dim3(1024),dim3(256),0,0,(const v4*)hotbuf,n4,out);},R);
    printf("%4ld MiB: cold plain %6.1f us %5.2f TB/s | cold nt %6.1f us %5.2f | hot plain %6.1f us %5.2f | hot nt %6.1f us %5.2f | hot plain 4blk/CU %6.1f us %5.2f\n",
      mb,c0,bytes/c0*1e-6,c1,bytes/c1*1e-6,h0,bytes/h0*1e-6,h1,bytes/h1*1e-6,h4,bytes/h4*1e-6);
  }
  printf("=== E2 hot W (40 MiB, fixed) between a 40 MiB cold read stream and a 40 MiB cold write stream\n");
  {
    long bytes=40*MB, n4=bytes/16;
    float* W=big+ (3*GB)/4;  // fixed
    for(int vm=0; vm<2; ++vm) for(int om=0; om<2; ++om){
      float all=timeit([&]{
        hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)W,n4,out);
        if(vm) hipLaunchKernelGGL((rd<4,1>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);
        else   hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);
        if(om) hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);
        else   hipLaunchKernelGGL((wr<0>),dim3(512),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);
      },R);
      float noW=timeit([&]{
        if(vm) hipLaunchKernelGGL((rd<4,1>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);
        else   hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);
        if(om) hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);
        else   hipLaunchKernelGGL((wr<0>),dim3(512),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);
      },R);
      printf("V %s, O %s: W+V+O %6.1f us, V+O %6.1f us => W read %6.1f us (%5.2f TB/s)\n",vm?"nt   ":"plain",om?"nt   ":"plain",all,noW,all-noW,bytes/(all-noW)*1e-6);
    }
  }
  printf("=== E4 write streams (cold)\n");
  for(long mb : {11L,28L,40L,120L}){
    long bytes=mb*MB, n4=bytes/16;
    for(int bpc : {1,2,4,8}){
      float w0=timeit([&]{ hipLaunchKernelGGL((wr<0>),dim3(256*bpc),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);},R);
      float w1=timeit([&]{ hipLaunchKernelGGL((wr<1>),dim3(256*bpc),dim3(256),0,0,(v4*)cold(bytes),n4,1.f);},R);
      printf("%4ld MiB blocks/CU=%d: plain %6.1f us %5.2f TB/s | nt %6.1f us %5.2f TB/s\n",mb,bpc,w0,bytes/w0*1e-6,w1,bytes/w1*1e-6);
    }
  }
  printf("=== E3 read 28 MiB + write 28 MiB: two launches vs one launch with roles\n");
  {
    long bytes=28*MB, n4=bytes/16;
    float two=timeit([&]{
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),n4,out);
      hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(bytes),n4,1.f); },R);
    printf("two launches: %6.1f us\n",two);
    for(int g : {512,1024}){
      float a=timeit([&]{ hipLaunchKernelGGL((rw<0,1>),dim3(g),dim3(256),0,0,(const v4*)cold(bytes),n4,(v4*)cold(bytes),n4,g/2,out);},R);
      float b=timeit([&]{ hipLaunchKernelGGL((rw_il<0,1>),dim3(g),dim3(256),0,0,(const v4*)cold(bytes),n4,(v4*)cold(bytes),n4,1,2,out);},R);
      float c=timeit([&]{ hipLaunchKernelGGL((rw_il<0,0>),dim3(g),dim3(256),0,0,(const v4*)cold(bytes),n4,(v4*)cold(bytes),n4,1,2,out);},R);
      printf("one launch grid %d: split halves %6.1f us | interleaved (nt st) %6.1f us | interleaved (plain st) %6.1f us   (%5.2f TB/s best)\n",g,a,b,c,2*bytes/fminf(a,fminf(b,c))*1e-6);
    }
  }
  printf("=== E5 chain of dependent launches: 6 x 10 MiB vs 1 x 60 MiB; 4 x (22,58,28,11)\n");
  {
    long b10=10*MB;
    float six=timeit([&]{ for(int k=0;k<6;k++) hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(b10),b10/16,out);},R);
    float one=timeit([&]{ hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(6*b10),6*b10/16,out);},R);
    printf("6 x 10 MiB: %6.1f us ; 1 x 60 MiB: %6.1f us ; per extra launch %5.2f us\n",six,one,(six-one)/5);
    long sz[5]={22*MB,58*MB,28*MB,28*MB,11*MB};
    float chain=timeit([&]{
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(sz[0]),sz[0]/16,out);
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(sz[1]),sz[1]/16,out);
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(sz[2]),sz[2]/16,out);
      hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(sz[3]),sz[3]/16,1.f);
      hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(sz[4]),sz[4]/16,1.f);
    },R);
    printf("ideal 5-launch matvec chain (22r,58r,28r,28w,11w MiB of pure streams): %6.1f us\n",chain);
    float chain4=timeit([&]{
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(sz[0]),sz[0]/16,out);
      hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(sz[1]),sz[1]/16,out);
      hipLaunchKernelGGL((rw_il<0,1>),dim3(1024),dim3(256),0,0,(const v4*)cold(sz[2]),sz[2]/16,(v4*)cold(sz[3]),sz[3]/16,1,2,out);
      hipLaunchKernelGGL((wr<1>),dim3(512),dim3(256),0,0,(v4*)cold(sz[4]),sz[4]/16,1.f);
    },R);
    printf("ideal 4-launch chain (22r,58r,[28r+28w],11w): %6.1f us\n",chain4);
  }
  printf("=== E6 one launch: 80 MiB read + 40 MiB written (roles 2:1 interleaved)\n");
  {
    long rb=80*MB, wb=40*MB;
    for(int g : {768,1536,3072}){
      float a=timeit([&]{ hipLaunchKernelGGL((rw_il<0,1>),dim3(g),dim3(256),0,0,(const v4*)cold(rb),rb/16,(v4*)cold(wb),wb/16,2,3,out);},R);
      float b=timeit([&]{ hipLaunchKernelGGL((rw_il<1,1>),dim3(g),dim3(256),0,0,(const v4*)cold(rb),rb/16,(v4*)cold(wb),wb/16,2,3,out);},R);
      printf("grid %4d: plain ld / nt st %6.1f us (%5.2f TB/s) | nt ld / nt st %6.1f us (%5.2f TB/s)\n",g,a,(rb+wb)/a*1e-6,b,(rb+wb)/b*1e-6);
    }
    float r=timeit([&]{ hipLaunchKernelGGL((rd<4,0>),dim3(512),dim3(256),0,0,(const v4*)cold(120*MB),120*MB/16,out);},R);
    printf("120 MiB read only: %6.1f us (%5.2f TB/s)\n",r,120.0*MB/r*1e-6);
  }
  return 0;
}
