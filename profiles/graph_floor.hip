// Does a hipGraph shorten the launch-bound ICP loop?  The loop is a chain of dependent kernels
// (k_icp -> k_fin -> k_icp ...) whose parameters never change inside a call (the state lives on the
// device), so 2 x K launches could be replayed as one graph.  This probe measures what a dependent
// kernel node costs in a graph against the same chain enqueued on a stream (launch_floor.hip):
// microseconds per kernel, empty kernels and kernels that dirty a line each.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 profiles/graph_floor.hip -o /tmp/gf && /tmp/gf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_noop(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 12345) p[1] = 1; }
__global__ void k_touch(int *p) { if (threadIdx.x == 0) p[blockIdx.x & 1023] += 1; }
// a dependent pair like k_icp -> k_fin: many workgroups each leave a word, one workgroup adds them
__global__ void k_many(int *p) { if (threadIdx.x == 0) p[1024 + (blockIdx.x & 1023)] = p[0] + 1; }
__global__ void k_one(int *p) { if (threadIdx.x < 64) { int v = p[1024 + threadIdx.x]; if (threadIdx.x == 0) p[0] = v; } }
int main() {
    int *d; CK(hipMalloc(&d, 4096 * sizeof(int))); CK(hipMemset(d, 0, 4096 * sizeof(int)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int K = 32;            // kernels per graph (16 iterations of two kernels)
    const int R = 125;           // graph launches timed  -> 4000 kernels
    for (int grid : {1, 470, 1875}) for (int kind = 0; kind < 3; ++kind) {
        auto enqueue = [&](int i) {
            if (kind == 0) hipLaunchKernelGGL(k_noop, dim3(grid), dim3(256), 0, s, d);
            else if (kind == 1) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, d);
            else if (i & 1) hipLaunchKernelGGL(k_one, dim3(1), dim3(1024), 0, s, d);
            else hipLaunchKernelGGL(k_many, dim3(grid), dim3(256), 0, s, d);
        };
        const char *name = kind == 0 ? "no memory traffic" : kind == 1 ? "one store per workgroup" : "many -> one pair";
        // plain stream
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < K * R; ++i) enqueue(i);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        }
        CK(hipEventElapsedTime(&ms, a, b));
        const double us_stream = 1e3 * ms / (K * R);
        // captured graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < K; ++i) enqueue(i);
        CK(hipStreamEndCapture(s, &g));
        hipEvent_t c0, c1; CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
        CK(hipEventRecord(c0, s));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(a, s));
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        }
        CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %4d, %-24s: stream %.2f us per kernel, graph of %d nodes %.2f us per kernel\n",
               grid, name, us_stream, K, 1e3 * ms / (K * R));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
