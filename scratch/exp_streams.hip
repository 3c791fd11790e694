// how does HBM write bandwidth depend on the number of concurrent output streams?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

// each thread owns 2 consecutive elements and writes them to R rows of length N (row pitch = pitch)
template<bool PERSIST>
__global__ __launch_bounds__(256) void k_rows(int R, int64_t N, int64_t pitch, double* __restrict__ out, double v) {
  if (!PERSIST) {
    const int64_t i = ((int64_t)blockIdx.x*256 + threadIdx.x)*2;
    if (i >= N) return;
    d2 val = {v + i, v - i};
    for (int r=0;r<R;++r){ *(d2*)(out + (int64_t)r*pitch + i) = val; val += 1.0; }
  } else {
    for (int64_t i = ((int64_t)blockIdx.x*256 + threadIdx.x)*2; i < N; i += (int64_t)gridDim.x*512) {
      d2 val = {v + i, v - i};
      for (int r=0;r<R;++r){ *(d2*)(out + (int64_t)r*pitch + i) = val; val += 1.0; }
    }
  }
}
// fill-like: one stream, grid-stride, 4 x 16 B per thread per iteration
__global__ __launch_bounds__(256) void k_fill(int64_t n, double* __restrict__ out, double v){
  for (int64_t i = ((int64_t)blockIdx.x*256 + threadIdx.x)*2; i < n; i += (int64_t)gridDim.x*512) {
    d2 val = {v, v}; *(d2*)(out+i) = val;
  }
}
float timeit(void (*launch)(void*), void* ctx){
  hipEvent_t a,b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int it=0; it<10; ++it) launch(ctx);
  CHECK(hipEventRecord(a,0));
  for (int it=0; it<30; ++it) launch(ctx);
  CHECK(hipEventRecord(b,0)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms,a,b)); return ms/30;
}
struct Ctx { int R; int64_t N, pitch; double* out; bool persist; int grid; };
void l_rows(void* p){ Ctx* c=(Ctx*)p;
  if (c->persist) hipLaunchKernelGGL(k_rows<true>, dim3(c->grid), dim3(256), 0, 0, c->R, c->N, c->pitch, c->out, 1.0);
  else hipLaunchKernelGGL(k_rows<false>, dim3((unsigned)((c->N/2+255)/256)), dim3(256), 0, 0, c->R, c->N, c->pitch, c->out, 1.0); }
void l_fill(void* p){ Ctx* c=(Ctx*)p; hipLaunchKernelGGL(k_fill, dim3(c->grid), dim3(256), 0, 0, c->N, c->out, 1.0); }

int main(){
  const int64_t total = (int64_t)72 * 9994476;   // doubles: same bytes as 12 surfaces x 6 rows
  double* out; CHECK(hipMalloc(&out, (total + 72*600000)*8));
  Ctx c; c.out = out;
  c.N = total; c.grid = 256*8; printf("fill-like 1 stream grid-stride      : %.3f ms %.2f TB/s\n", timeit(l_fill,&c), total*8/1e9/timeit(l_fill,&c));
  {
    int R = 72; c.R=R; c.N = 9994476; c.persist=false;
    struct { const char* name; int64_t pitch; } v[] = {
      {"natural N          ", c.N},
      {"round 16 (128 B)   ", (c.N+15)/16*16},
      {"round 32 (256 B)   ", (c.N+31)/32*32},
      {"round 512 (4 KB)   ", (c.N+511)/512*512},
      {"4 KB + 128 B       ", (c.N+511)/512*512 + 16},
      {"4 KB + 256 B       ", (c.N+511)/512*512 + 32},
      {"4 KB + 512 B       ", (c.N+511)/512*512 + 64},
      {"4 KB + 1 KB        ", (c.N+511)/512*512 + 128},
      {"64 KB + 256 B      ", (c.N+8191)/8192*8192 + 32},
      {"2 MB               ", (c.N+262143)/262144*262144},
      {"2 MB + 256 B       ", (c.N+262143)/262144*262144 + 32},
    };
    for (auto& q : v){ c.pitch = q.pitch; float t = timeit(l_rows,&c);
      printf("R=72 pitch %s: %.3f ms %.2f TB/s\n", q.name, t, (double)R*c.N*8/1e9/t); }
  }
  return 0;
}
