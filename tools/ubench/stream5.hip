// Scratch micro-benchmark (round 2): does the ACCESS PATTERN of the MFMA-fed weight loads cost bandwidth?
// The matvec kernels feed v_mfma_f32_16x16x4_f32 straight from global memory: one wave-level
// global_load_dwordx4 covers 16 rows x 64 B.  Compare, at the sizes of the C2 layers and with the
// same blocks / waves / loads in flight:
//   lin    : one wave instruction = 1 KiB contiguous
//   tile16 : one wave instruction = 16 rows x 64 B (row stride = d_in floats), K-steps of 16 floats
//   tile4  : 4 rows x 256 B
// Layout of the "matrix": rows x d_in floats, row-major.  A block of 4 waves owns 64 rows (16 per wave
// for tile16) and a K range, like fwd_mfma_kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
typedef float __attribute__((ext_vector_type(4))) v4;

// grid = (rows/64, ksplit); each wave: rows [blk*64 + wave*16, +16), K range [ky*kpb, +kpb)
// tile16: one instruction = 16 rows x 64 B; U K-steps (16 floats each) in flight per iteration
template<int U>
__global__ __launch_bounds__(256) void rd_tile16(const float* __restrict__ p, int d_in, int kpb, float* out){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  const int row0=(blockIdx.x*4+wave)*16, k0=blockIdx.y*kpb;
  const float* base=p+(long)(row0+(lane&15))*d_in+k0+(lane>>4)*4;
  float s=0;
  for(int k=0;k<kpb;k+=16*U){ v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]=*(const v4*)(base+k+u*16);
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w; }
  if(s==123.456f) out[0]=s;
}
// tile4: one instruction = 4 rows x 256 B; the 16 rows of the wave = 4 instructions per 64-float K step
__global__ __launch_bounds__(256) void rd_tile4(const float* __restrict__ p, int d_in, int kpb, float* out){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  const int row0=(blockIdx.x*4+wave)*16, k0=blockIdx.y*kpb;
  const float* base=p+(long)(row0+(lane>>4))*d_in+k0+(lane&15)*4;
  float s=0;
  for(int k=0;k<kpb;k+=64){ v4 v[4];
    #pragma unroll
    for(int i=0;i<4;i++) v[i]=*(const v4*)(base+(long)i*4*d_in+k);
    #pragma unroll
    for(int i=0;i<4;i++) s+=v[i].x+v[i].y+v[i].z+v[i].w; }
  if(s==123.456f) out[0]=s;
}
// tile1: one instruction = 1 row x 1 KiB; 4 rows in flight per iteration
__global__ __launch_bounds__(256) void rd_tile1(const float* __restrict__ p, int d_in, int kpb, float* out){
  const int lane=threadIdx.x&63, wave=threadIdx.x>>6;
  const int row0=(blockIdx.x*4+wave)*16, k0=blockIdx.y*kpb;
  const float* base=p+(long)row0*d_in+k0+lane*4;
  float s=0;
  for(int k=0;k<kpb;k+=256)
    for(int r=0;r<16;r+=4){ v4 v[4];
      #pragma unroll
      for(int i=0;i<4;i++) v[i]=*(const v4*)(base+(long)(r+i)*d_in+k);
      #pragma unroll
      for(int i=0;i<4;i++) s+=v[i].x+v[i].y+v[i].z+v[i].w; }
  if(s==123.456f) out[0]=s;
}
template<int U>
__global__ __launch_bounds__(256) void rd_lin(const v4* __restrict__ p, long n4, float* out){
  float s=0; const long stride=(long)gridDim.x*blockDim.x; long i=(long)blockIdx.x*blockDim.x+threadIdx.x;
  for(; i+(U-1)*stride<n4; i+=U*stride){ v4 v[U];
    #pragma unroll
    for(int u=0;u<U;u++) v[u]=p[i+u*stride];
    #pragma unroll
    for(int u=0;u<U;u++) s+=v[u].x+v[u].y+v[u].z+v[u].w; }
  if(s==123.456f) out[0]=s;
}
template<typename F> float timeit(F f,int iters){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  for(int i=0;i<3;i++) f();
  hipEventRecord(a); for(int i=0;i<iters;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/iters*1000.f;
}
int main(){
  float* out; CK(hipMalloc(&out,64));
  const long GB=1L<<30;
  float* big; CK(hipMalloc(&big,4*GB)); CK(hipMemset(big,0,4*GB));
  long slot=0;
  auto cold=[&](long bytes)->float*{ long nslots=4*GB/bytes; return big+((slot++)%nslots)*(bytes/4); };
  const int R=40;
  struct Shape{int rows,d_in,ksplit;};
  // rows = 2 x 2688 (W and V rows): the bytes of a forward layer
  for(Shape sh : {Shape{5376,1024,4}, Shape{5376,1024,2}, Shape{5376,2560,10}, Shape{5376,2560,5}, Shape{5376,2688,12}, Shape{5376,2688,6}}){
    long bytes=(long)sh.rows*sh.d_in*4; int kpb=sh.d_in/sh.ksplit;
    dim3 grid(sh.rows/64, sh.ksplit);
    float lin=timeit([&]{ hipLaunchKernelGGL((rd_lin<4>),dim3(512),dim3(256),0,0,(const v4*)cold(bytes),bytes/16,out);},R);
    float a2=timeit([&]{ hipLaunchKernelGGL((rd_tile16<2>),grid,dim3(256),0,0,cold(bytes),sh.d_in,kpb,out);},R);
    float a4=kpb%64==0?timeit([&]{ hipLaunchKernelGGL((rd_tile16<4>),grid,dim3(256),0,0,cold(bytes),sh.d_in,kpb,out);},R):0.f;
    float a8=kpb%128==0?timeit([&]{ hipLaunchKernelGGL((rd_tile16<8>),grid,dim3(256),0,0,cold(bytes),sh.d_in,kpb,out);},R):0.f;
    float t4=kpb%64==0?timeit([&]{ hipLaunchKernelGGL(rd_tile4,grid,dim3(256),0,0,cold(bytes),sh.d_in,kpb,out);},R):0.f;
    float t1=kpb%256==0?timeit([&]{ hipLaunchKernelGGL(rd_tile1,grid,dim3(256),0,0,cold(bytes),sh.d_in,kpb,out);},R):0.f;
    printf("rows %d x d_in %d (%.1f MB), grid %dx%d kpb %d: linear %5.1f us %.2f TB/s | 16x64B U2 %5.1f U4 %5.1f U8 %5.1f us (best %.2f TB/s) | 4x256B %5.1f us | 1x1KiB %5.1f us\n",
      sh.rows,sh.d_in,bytes/1e6,grid.x,grid.y,kpb,lin,bytes/lin*1e-6,a2,a4,a8,bytes/fminf(a2,fminf(a4>0?a4:1e9f,a8>0?a8:1e9f))*1e-6,t4,t1);
  }
  return 0;
}
