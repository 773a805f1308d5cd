// Scratch micro-benchmark (round 2): can W2 (27.6 MiB) survive in the eight 4 MiB L2s between the forward
// kernel (reads W2 AND V2, 55 MiB) and the backward data chain (re-reads W2), if both kernels give the same
// XCD the same slice of W2 and the V2 stream is non-temporal?
//   fwd(pol): block b (XCD b % 8) streams tile t(b) of W (plain loads) and of V (plain | nt loads)
//   small   : a few tiny kernels in between (the head)
//   bwd(map): block b re-reads tile t(b) of W  (affine: same t(b); shifted: t(b) + 1 -> another XCD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;

template<int VNT>
__global__ __launch_bounds__(256) void fwd(const v4* __restrict__ W, const v4* __restrict__ V, long tile4, float* out){
  const long base=(long)blockIdx.x*tile4;
  float s=0;
  for(long i=threadIdx.x;i<tile4;i+=4*256){
    v4 a[4],b[4];
    #pragma unroll
    for(int u=0;u<4;u++){ long j=i+u*256; if(j<tile4){ a[u]=W[base+j]; b[u]= VNT? __builtin_nontemporal_load(V+base+j) : V[base+j]; } else {a[u]=b[u]=v4{0,0,0,0};} }
    #pragma unroll
    for(int u=0;u<4;u++) s+=a[u].x+a[u].w+b[u].y+b[u].z;
  }
  if(s==123.456f) out[0]=s;
}
template<int NT>
__global__ __launch_bounds__(256) void bwd(const v4* __restrict__ W, long tile4, int shift, int nt, float* out){
  const long t=((long)blockIdx.x+shift)%nt;
  const long base=t*tile4;
  float s=0;
  for(long i=threadIdx.x;i<tile4;i+=4*256){
    v4 a[4];
    #pragma unroll
    for(int u=0;u<4;u++){ long j=i+u*256; a[u]= j<tile4 ? (NT? __builtin_nontemporal_load(W+base+j) : W[base+j]) : v4{0,0,0,0}; }
    #pragma unroll
    for(int u=0;u<4;u++) s+=a[u].x+a[u].w;
  }
  if(s==123.456f) out[0]=s;
}
__global__ void tiny(float* p, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) p[i]=p[i]*1.0001f+1.f; }
__global__ __launch_bounds__(256) void wr(v4* __restrict__ p, long n4){
  const v4 v={1,2,3,4};
  for(long i=(long)blockIdx.x*blockDim.x+threadIdx.x;i<n4;i+=(long)gridDim.x*blockDim.x) __builtin_nontemporal_store(v,p+i);
}
int main(){
  float* out; CK(hipMalloc(&out,1<<20));
  const long MB=1L<<20;
  const long wbytes=28901376;   // 2688 x 2688 x 4
  const int nt=504;             // tiles (as fwd_mfma_kernel: 42 row blocks x 12 K ranges)
  const long tile4=wbytes/16/nt;  // float4 per tile
  float *W,*V[8],*O; CK(hipMalloc(&W,wbytes+4096)); CK(hipMalloc(&O,40*MB));
  for(int k=0;k<8;k++){ CK(hipMalloc(&V[k],wbytes+4096)); CK(hipMemset(V[k],0,wbytes)); }
  CK(hipMemset(W,0,wbytes));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int vnt=0;vnt<2;vnt++) for(int shift=0;shift<2;shift++) for(int bnt=0;bnt<2;bnt++) for(int mid=0;mid<2;mid++){
    float tot=0; const int R=30;
    for(int r=0;r<R+3;r++){
      const v4* v=(const v4*)V[r%8];
      if(vnt) hipLaunchKernelGGL((fwd<1>),dim3(nt),dim3(256),0,0,(const v4*)W,v,tile4,out);
      else    hipLaunchKernelGGL((fwd<0>),dim3(nt),dim3(256),0,0,(const v4*)W,v,tile4,out);
      hipLaunchKernelGGL(tiny,dim3(88),dim3(256),0,0,out,88*256);
      hipLaunchKernelGGL(tiny,dim3(11),dim3(256),0,0,out,11*256);
      hipEventRecord(e0);
      if(bnt) hipLaunchKernelGGL((bwd<1>),dim3(nt),dim3(256),0,0,(const v4*)W,tile4,shift,nt,out);
      else    hipLaunchKernelGGL((bwd<0>),dim3(nt),dim3(256),0,0,(const v4*)W,tile4,shift,nt,out);
      hipEventRecord(e1);
      if(mid) hipLaunchKernelGGL(wr,dim3(512),dim3(256),0,0,(v4*)O,40*MB/16);   // the result stream of a matvec + the layer-1 streams evict?
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms,e0,e1); if(r>=3) tot+=ms;
    }
    printf("V %s | bwd map %s | bwd loads %s | 40 MiB written between matvecs %d : W2 re-read %5.1f us (%.2f TB/s)\n",
      vnt?"nt   ":"plain", shift?"shifted":"affine ", bnt?"nt   ":"plain", mid, tot/R*1000, wbytes/(tot/R*1e-3)*1e-12);
  }
  return 0;
}
