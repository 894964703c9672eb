// What two dependent launches per ICP iteration cost on this stack before they do anything:
// back-to-back launches of (almost) empty kernels on one stream, microseconds per launch.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 profiles/launch_floor.hip -o /tmp/lf && /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_noop(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 12345) p[1] = 1; }
__global__ void k_touch(int *p) { if (threadIdx.x == 0) p[blockIdx.x & 1023] += 1; }   // dirty lines to write back
int main() {
    int *d; hipMalloc(&d, 4096 * sizeof(int)); hipMemset(d, 0, 4096 * sizeof(int));
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 4000;
    for (int grid : {1, 470, 1875}) for (int touch = 0; touch < 2; ++touch) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(a, s);
            for (int i = 0; i < N; ++i) {
                if (touch) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, d);
                else hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, s, d);
            }
            hipEventRecord(b, s); hipEventSynchronize(b);
        }
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("grid %4d x 256 threads, %s: %.2f us per launch\n", grid, touch ? "one store per workgroup" : "no memory traffic", 1e3 * ms / N);
    }
    return 0;
}
