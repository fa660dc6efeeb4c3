// Hardware-semantics probe for gfx950: pins the facts the kernels rely on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

static inline uint16_t f2bf(float f){uint32_t u; std::memcpy(&u,&f,4); uint32_t r=u+0x7fff+((u>>16)&1); return (uint16_t)(r>>16);}
static inline float bf2f(uint16_t h){uint32_t u=((uint32_t)h)<<16; float f; std::memcpy(&f,&u,4); return f;}

// 1. MFMA 16x16x32: A[16][32] row-major (k contiguous), B given as Bt[16][32] (n rows, k contiguous)
__global__ void k_mfma(const uint16_t* A,const uint16_t* Bt,float* D){
  int l=threadIdx.x;
  bf16x8 a=*(const bf16x8*)(A+(l&15)*32+(l>>4)*8);
  bf16x8 b=*(const bf16x8*)(Bt+(l&15)*32+(l>>4)*8);
  f32x4 c={0,0,0,0};
  c=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,c,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)*4+r)*16+(l&15)]=c[r];   // row=(l>>4)*4+r (A row), col=l&15 (B row)
}
// 2. ds_read_b64_tr_b16 semantics
__global__ void k_tr(uint16_t* out, int stride_elems){
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for(int i=threadIdx.x;i<4096;i+=64) lds[i]=(uint16_t)i;
  __syncthreads();
  int l=threadIdx.x;
  // lane i of each 16-group points at row (i>>2) [stride], col (i&3)*4; groups offset by 4 rows
  int g=l>>4,i=l&15;
  unsigned addr=(unsigned)(uintptr_t)(&lds[((g*4)+(i>>2))*stride_elems+(i&3)*4]);
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)":"=v"(v):"v"(addr):"memory");
  out[l*4+0]=(uint16_t)(v&0xffff); out[l*4+1]=(uint16_t)((v>>16)&0xffff);
  out[l*4+2]=(uint16_t)((v>>32)&0xffff); out[l*4+3]=(uint16_t)((v>>48)&0xffff);
}
// 3. cvt_pk
__global__ void k_cvt(const float* in,uint32_t* out,int n){
  int i=blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  float lo=in[2*i],hi=in[2*i+1]; uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2":"=v"(r):"v"(lo),"v"(hi));
  out[i]=r;
}
// 4. copy bandwidth
__global__ void k_copy(const float4* __restrict__ a,float4* __restrict__ b,size_t n){
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st) b[i]=a[i];
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s arch %s CUs %d clock %d MHz mem %.1f GB\n",p.name,p.gcnArchName,p.multiProcessorCount,p.clockRate/1000,p.totalGlobalMem/1e9);
  // 1
  {std::vector<uint16_t> A(512),B(512); std::vector<float> Af(512),Bf(512);
   for(int i=0;i<512;i++){float a=(float)((i*37)%23-11)/8.f, b=(float)((i*53)%19-9)/4.f; A[i]=f2bf(a);B[i]=f2bf(b);Af[i]=bf2f(A[i]);Bf[i]=bf2f(B[i]);}
   uint16_t *dA,*dB; float* dD; CK(hipMalloc(&dA,1024));CK(hipMalloc(&dB,1024));CK(hipMalloc(&dD,1024));
   CK(hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),1024,hipMemcpyHostToDevice));
   k_mfma<<<1,64>>>(dA,dB,dD); std::vector<float> D(256); CK(hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost));
   double maxe=0; for(int i=0;i<16;i++)for(int j=0;j<16;j++){double s=0;for(int k=0;k<32;k++)s+=Af[i*32+k]*Bf[j*32+k]; maxe=fmax(maxe,fabs(s-D[i*16+j]));}
   printf("MFMA16x16x32 D[row=(l>>4)*4+r = A-row][col=l&15 = B-row] maxerr=%g %s\n",maxe,maxe<1e-3?"OK":"MISMATCH");}
  // 2
  for(int stride: {16,40}){
   uint16_t* d; CK(hipMalloc(&d,512)); k_tr<<<1,64>>>(d,stride); std::vector<uint16_t> o(256); CK(hipMemcpy(o.data(),d,512,hipMemcpyDeviceToHost));
   printf("tr_b16 stride=%d: lane i in group g points at row g*4+(i>>2), col (i&3)*4. results (row,col) per elem:\n",stride);
   for(int l=0;l<64;l++){ if(l<20||l>=60||(l%16)<2){printf(" lane%2d:",l); for(int j=0;j<4;j++){int v=o[l*4+j]; printf(" (%d,%d)",v/stride,v%stride);} printf("\n");}}
   // check hypothesis: lane l elem j == (row g*4+j, col i)
   int ok=1; for(int l=0;l<64;l++)for(int j=0;j<4;j++){int g=l>>4,i=l&15; if(o[l*4+j]!=(g*4+j)*stride+i) ok=0;}
   printf("hypothesis elem j = M[g*4+j][i]: %s\n",ok?"CONFIRMED":"REJECTED");}
  // 3
  {int n=1024; std::vector<float> in(2*n); for(int i=0;i<2*n;i++){in[i]=(float)sin(i*0.37)*powf(2.f,(i%40)-20);} in[0]=1.00390625f; in[1]=1.01171875f; in[2]=-1.00390625f;
   float* di; uint32_t* dout; CK(hipMalloc(&di,8*n));CK(hipMalloc(&dout,4*n)); CK(hipMemcpy(di,in.data(),8*n,hipMemcpyHostToDevice));
   k_cvt<<<(n+255)/256,256>>>(di,dout,n); std::vector<uint32_t> o(n); CK(hipMemcpy(o.data(),dout,4*n,hipMemcpyDeviceToHost));
   int bad=0; for(int i=0;i<n;i++){uint32_t e=f2bf(in[2*i])|((uint32_t)f2bf(in[2*i+1])<<16); if(e!=o[i]) bad++;}
   printf("v_cvt_pk_bf16_f32 (lo=src0, hi=src1, RNE) mismatches: %d/%d\n",bad,n);}
  // 4
  {size_t bytes=(size_t)2<<30; float4 *a,*b; CK(hipMalloc(&a,bytes));CK(hipMalloc(&b,bytes)); CK(hipMemset(a,1,bytes));
   hipEvent_t e0,e1; CK(hipEventCreate(&e0));CK(hipEventCreate(&e1));
   for(int grid: {2048,8192,65536}){
    k_copy<<<grid,256>>>(a,b,bytes/16); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for(int i=0;i<5;i++) k_copy<<<grid,256>>>(a,b,bytes/16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("copy 2GiB grid=%d: %.3f ms/iter -> %.2f TB/s (r+w)\n",grid,ms/5,2.0*bytes/(ms/5*1e-3)/1e12);}
  }
  return 0;
}
