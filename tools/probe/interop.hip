#include <hip/hip_runtime.h>
__global__ void k_axpy(const float* x,float* y,float a,int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]+=a*x[i];}
extern "C" int probe_axpy(const void* x,void* y,float a,int n,void* stream){
  hipLaunchKernelGGL(k_axpy,dim3((n+255)/256),dim3(256),0,(hipStream_t)stream,(const float*)x,(float*)y,a,n);
  return (int)hipGetLastError();
}
