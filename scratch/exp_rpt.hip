// store-pattern experiment: rays per thread (2 = one dwordx4 per row, 4 = two adjacent dwordx4, 4s = two
// dwordx4 one wave-width apart) for the compute-free (S,3,N) path writer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template<int VAR>
__global__ __launch_bounds__(256) void k_store(int S, int64_t N, const double* __restrict__ x0, const double* __restrict__ k0,
    double* __restrict__ xh, double* __restrict__ ko, uint8_t* __restrict__ v1) {
  // VAR 0: 2 rays/thread.  VAR 1: 4 adjacent rays/thread.  VAR 2: 2+2 rays, second pair 128 rays further
  const int R = (VAR == 0) ? 1 : 2;
  int64_t i0, i1;
  if (VAR == 0) { i0 = ((int64_t)blockIdx.x*256 + threadIdx.x)*2; i1 = i0; }
  else if (VAR == 1) { i0 = ((int64_t)blockIdx.x*256 + threadIdx.x)*4; i1 = i0 + 2; }
  else { const int64_t base = (int64_t)blockIdx.x*1024; const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
         i0 = base + w*256 + l*2; i1 = i0 + 128; }
  if (i0 >= N) return;
  d2 x[2][3], k[2][3];
  const int64_t ii[2] = {i0, i1};
  for (int r=0;r<R;++r) for (int c=0;c<3;++c){ x[r][c] = *(const d2*)(x0 + c*N + ii[r]); k[r][c] = *(const d2*)(k0 + c*N + ii[r]); }
  for (int s=0;s<S;++s){
    for (int r=0;r<R;++r) for (int c=0;c<3;++c){ x[r][c] += k[r][c]; }
    for (int c=0;c<3;++c) for (int r=0;r<R;++r){
      const int64_t o = ((int64_t)s*3+c)*N + ii[r];
      *(d2*)(xh+o) = x[r][c]; *(d2*)(ko+o) = k[r][c];
    }
    for (int r=0;r<R;++r){
      const int64_t o = (int64_t)s*N + ii[r];
      *(uint16_t*)(v1+o) = (uint16_t)((x[r][0].x>0?1:0) | (x[r][0].y>0?0x200:0));
    }
  }
}

template<int VAR>
float run(int S, int64_t N, double* x0, double* k0, double* xh, double* ko, uint8_t* v1){
  hipEvent_t a,b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int rpt = (VAR == 0) ? 2 : 4;
  unsigned grid = (unsigned)((N/rpt + 255)/256);
  for (int it=0; it<20; ++it) hipLaunchKernelGGL((k_store<VAR>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1);
  CHECK(hipEventRecord(a,0));
  for (int it=0; it<50; ++it) hipLaunchKernelGGL((k_store<VAR>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1);
  CHECK(hipEventRecord(b,0)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms,a,b)); return ms/50;
}

int main(){
  const int S=12; const int64_t N = 9994476 / 1024 * 1024;
  double *x0,*k0,*xh,*ko; uint8_t *v1;
  CHECK(hipMalloc(&x0, 3*N*8)); CHECK(hipMalloc(&k0, 3*N*8));
  CHECK(hipMalloc(&xh, (size_t)S*3*N*8)); CHECK(hipMalloc(&ko, (size_t)S*3*N*8)); CHECK(hipMalloc(&v1, (size_t)S*N));
  CHECK(hipMemset(x0, 0, 3*N*8)); CHECK(hipMemset(k0, 0, 3*N*8));
  double gb = (double)N*(48 + 49.0*S)/1e9;
  for (int rep=0; rep<3; ++rep){
    float t;
    t = run<0>(S,N,x0,k0,xh,ko,v1); printf("2 rays/thread            : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<1>(S,N,x0,k0,xh,ko,v1); printf("4 adjacent rays/thread   : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<2>(S,N,x0,k0,xh,ko,v1); printf("2+2 rays, 1 KiB apart    : %.3f ms  %.2f TB/s\n", t, gb/t);
  }
  return 0;
}
