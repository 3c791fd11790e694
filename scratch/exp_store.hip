// store-pattern experiment: how fast can the (S,3,N) path layout be written with no compute?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

// LAYOUT 0: (S,3,N) rows.  LAYOUT 1: tiled [tile][S][3][T] with T=512 rays per tile (one block)
template<int LAYOUT, bool NT, bool MASKS, int RPT>
__global__ __launch_bounds__(256) void k_store(int S, int64_t N, const double* __restrict__ x0, const double* __restrict__ k0,
    double* __restrict__ xh, double* __restrict__ ko, uint8_t* __restrict__ v1, uint8_t* __restrict__ v2) {
  const int64_t i = ((int64_t)blockIdx.x*256 + threadIdx.x)*2;
  if (i >= N) return;
  d2 x[3], k[3];
  for (int c=0;c<3;++c){ x[c] = *(const d2*)(x0 + c*N + i); k[c] = *(const d2*)(k0 + c*N + i); }
  for (int s=0;s<S;++s){
    for (int c=0;c<3;++c){ x[c] += k[c]; }
    for (int c=0;c<3;++c){
      int64_t o;
      if (LAYOUT==0) o = ((int64_t)s*3+c)*N + i;
      else { const int64_t tile = i / 512, r = i % 512; o = ((tile*S + s)*3 + c)*512 + r; }
      if (NT){ __builtin_nontemporal_store(x[c], (d2*)(xh+o)); __builtin_nontemporal_store(k[c], (d2*)(ko+o)); }
      else { *(d2*)(xh+o) = x[c]; *(d2*)(ko+o) = k[c]; }
    }
    if (MASKS){
      int64_t o;
      if (LAYOUT==0) o = (int64_t)s*N + i; else { const int64_t tile=i/512, r=i%512; o=(tile*S+s)*512 + r; }
      uint16_t m = (x[0].x>0?1:0) | (x[0].y>0?0x100:0);
      *(uint16_t*)(v1+o) = m; *(uint16_t*)(v2+o) = m;
    }
  }
}

template<int LAYOUT, bool NT, bool MASKS>
float run(int S, int64_t N, double* x0, double* k0, double* xh, double* ko, uint8_t* v1, uint8_t* v2){
  hipEvent_t a,b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  unsigned grid = (unsigned)((N/2 + 255)/256);
  for (int it=0; it<20; ++it) hipLaunchKernelGGL((k_store<LAYOUT,NT,MASKS,2>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1,v2);
  CHECK(hipEventRecord(a,0));
  for (int it=0; it<50; ++it) hipLaunchKernelGGL((k_store<LAYOUT,NT,MASKS,2>), dim3(grid), dim3(256), 0, 0, S,N,x0,k0,xh,ko,v1,v2);
  CHECK(hipEventRecord(b,0)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms,a,b)); return ms/50;
}

int main(){
  const int S=12; const int64_t N = 9994476 / 512 * 512;   // multiple of the tile
  double *x0,*k0,*xh,*ko; uint8_t *v1,*v2;
  CHECK(hipMalloc(&x0, 3*N*8)); CHECK(hipMalloc(&k0, 3*N*8));
  CHECK(hipMalloc(&xh, (size_t)S*3*N*8)); CHECK(hipMalloc(&ko, (size_t)S*3*N*8));
  CHECK(hipMalloc(&v1, (size_t)S*N)); CHECK(hipMalloc(&v2, (size_t)S*N));
  CHECK(hipMemset(x0, 0, 3*N*8)); CHECK(hipMemset(k0, 0, 3*N*8));
  double gb = (double)N*(48 + 50.0*S)/1e9;
  for (int rep=0; rep<2; ++rep){
    float t;
    t = run<0,false,true>(S,N,x0,k0,xh,ko,v1,v2); printf("rows   plain masks : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<0,true,true>(S,N,x0,k0,xh,ko,v1,v2);  printf("rows   NT    masks : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<0,false,false>(S,N,x0,k0,xh,ko,v1,v2); printf("rows   plain nomask: %.3f ms  %.2f TB/s\n", t, (gb-2.0*S*N/1e9)/t);
    t = run<0,true,false>(S,N,x0,k0,xh,ko,v1,v2); printf("rows   NT    nomask: %.3f ms  %.2f TB/s\n", t, (gb-2.0*S*N/1e9)/t);
    t = run<1,false,true>(S,N,x0,k0,xh,ko,v1,v2); printf("tiled  plain masks : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<1,true,true>(S,N,x0,k0,xh,ko,v1,v2);  printf("tiled  NT    masks : %.3f ms  %.2f TB/s\n", t, gb/t);
    t = run<1,false,false>(S,N,x0,k0,xh,ko,v1,v2); printf("tiled  plain nomask: %.3f ms  %.2f TB/s\n", t, (gb-2.0*S*N/1e9)/t);
  }
  return 0;
}
