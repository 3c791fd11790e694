// TEST INFRASTRUCTURE: self-test of the stand-in runtime (hip/hip_runtime.h) -- a vote inside a divergent region (over
// the active lanes only, like the Newton loop's `__all(done)` behind `if (i < N)`), a shuffle reduction where the wave
// has reconverged, a block barrier, a partial last block.  tests/test_hostemu.py compiles and runs it.
#include "hip/hip_runtime.h"
__global__ void kern(double *out, int N) {
    int i = blockIdx.x * 128 + threadIdx.x;
    double v = 0.0;
    if (i < N) {
        // lanes need different numbers of iterations; the loop ends when all ACTIVE lanes are done
        int it = 0; bool done = false;
        for (;;) { ++it; if (it >= 1 + (i % 5)) done = true; if (__all(done)) break; }
        v = (double)it;    // = 5 for every active lane of a wave that has a lane with i % 5 == 4
    }
    __shared__ double sh[2];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1];
}
int main() {
    double out[3];
    hipLaunchKernelGGL(kern, dim3(3), dim3(128), 0, 0, out, 300);
    printf("%g %g %g\n", out[0], out[1], out[2]);
    return (out[0] == 640 && out[1] == 640 && out[2] == 220) ? 0 : 1;
}
