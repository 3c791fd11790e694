// overlap experiment: (S,3,N) path stores + a tunable amount of dependent FP64 work per surface
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template<int ITER, bool STORE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(int S, int64_t N, const double* __restrict__ x0, const double* __restrict__ k0,
    double* __restrict__ xh, double* __restrict__ ko, uint8_t* __restrict__ v1, uint8_t* __restrict__ v2, double a, double b) {
  const int64_t i = ((int64_t)blockIdx.x*256 + threadIdx.x)*2;
  if (i >= N) return;
  d2 x[3], kk[3];
  for (int c=0;c<3;++c){ x[c] = *(const d2*)(x0 + c*N + i); kk[c] = *(const d2*)(k0 + c*N + i); }
  for (int s=0;s<S;++s){
#pragma unroll
    for (int j=0;j<ITER;++j){
#pragma unroll
      for (int c=0;c<3;++c){ x[c] = x[c]*a + b; kk[c] = kk[c]*a + b; }
    }
    if (STORE || s==S-1){
      const int so = STORE ? s : 0;
      for (int c=0;c<3;++c){
        const int64_t o = ((int64_t)so*3+c)*N + i;
        *(d2*)(xh+o) = x[c]; *(d2*)(ko+o) = kk[c];
      }
      const int64_t o = (int64_t)so*N + i;
      uint16_t m = (x[0].x>0?1:0) | (x[0].y>0?0x100:0);
      *(uint16_t*)(v1+o) = m; *(uint16_t*)(v2+o) = m;
    }
  }
}

template<int ITER, bool STORE, int WAVES>
float run(int S, int64_t N, double* x0, double* k0, double* xh, double* ko, uint8_t* v1, uint8_t* v2){
  hipEvent_t a,b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  unsigned grid = (unsigned)((N/2 + 255)/256);
  for (int it=0; it<20; ++it) hipLaunchKernelGGL((k<ITER,STORE,WAVES>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1,v2,0.999,1e-3);
  CHECK(hipEventRecord(a,0));
  for (int it=0; it<50; ++it) hipLaunchKernelGGL((k<ITER,STORE,WAVES>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1,v2,0.999,1e-3);
  CHECK(hipEventRecord(b,0)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms,a,b)); return ms/50;
}
#define ROW(IT) { float tc = run<IT,false,1>(S,N,x0,k0,xh,ko,v1,v2); float ts = run<IT,true,1>(S,N,x0,k0,xh,ko,v1,v2); \
   printf("fp64 instr/ray/surface %4d : compute-only %.3f ms   compute+stores %.3f ms\n", IT*6, tc, ts); }
int main(){
  const int S=12; const int64_t N = 9994752;
  double *x0,*k0,*xh,*ko; uint8_t *v1,*v2;
  CHECK(hipMalloc(&x0, 3*N*8)); CHECK(hipMalloc(&k0, 3*N*8));
  CHECK(hipMalloc(&xh, (size_t)S*3*N*8)); CHECK(hipMalloc(&ko, (size_t)S*3*N*8));
  CHECK(hipMalloc(&v1, (size_t)S*N)); CHECK(hipMalloc(&v2, (size_t)S*N));
  CHECK(hipMemset(x0, 0, 3*N*8)); CHECK(hipMemset(k0, 0, 3*N*8));
  ROW(0) ROW(5) ROW(10) ROW(15) ROW(20) ROW(30) ROW(40)
  return 0;
}
