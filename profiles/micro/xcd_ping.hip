// Round trip of a word between two waves: same XCD or not, agent scope (sc1: the fabric) or "XCD scope" (sc0 only:
// through the L1, served by the XCD's own L2).  What a per-XCD solving wave could save the one-launch loop's pose hop.
//   hipcc --offload-arch=gfx950 -O2 xcd_ping.hip -o xcd_ping && ./xcd_ping
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned long long ld_sc0(const unsigned long long *p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc0(unsigned long long *p, unsigned long long v) {
    asm volatile("global_store_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// block `a` pings, block `b` answers; everybody else leaves.  mode 0: agent scope, 1: sc0
__global__ void k_ping(unsigned long long *slots, int a, int b, int mode, int rounds, unsigned long long *out, unsigned *xcc) {
    if (threadIdx.x != 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if ((int)blockIdx.x == a) xcc[0] = id;
    if ((int)blockIdx.x == b) xcc[1] = id;
    unsigned long long *ping = slots, *pong = slots + 32;      // (separate 256-B lines)
    if ((int)blockIdx.x == a) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int k = 1; k <= rounds; ++k) {
            if (mode) st_sc0(ping, k); else st_agent(ping, k);
            unsigned long long guard = 0;
            while ((mode ? ld_sc0(pong) : ld_agent(pong)) != (unsigned long long)k && ++guard < 20000ull) { }
        }
        out[0] = __builtin_amdgcn_s_memtime() - t0;
    } else if ((int)blockIdx.x == b) {
        for (int k = 1; k <= rounds; ++k) {
            unsigned long long guard = 0;
            while ((mode ? ld_sc0(ping) : ld_agent(ping)) != (unsigned long long)k && ++guard < 20000ull) { }
            if (mode) st_sc0(pong, k); else st_agent(pong, k);
        }
    }
}

int main() {
    unsigned long long *slots, *out;
    unsigned *xcc;
    hipMalloc(&slots, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, 64);
    const int rounds = 300;
    struct { int a, b; const char *what; } pairs[] = {{0, 8, "blocks 0 and 8 (same XCD if b % 8)"}, {0, 1, "blocks 0 and 1"}, {0, 4, "blocks 0 and 4"}, {0, 16, "blocks 0 and 16"}};
    for (auto &pr : pairs)
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(slots, 0, 4096);
            hipLaunchKernelGGL(k_ping, dim3(64), dim3(64), 0, 0, slots, pr.a, pr.b, mode, rounds, out, xcc);
            hipDeviceSynchronize();
            unsigned long long cyc; unsigned x[2];
            hipMemcpy(&cyc, out, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
            printf("%-38s XCC %u / %u  %-12s round trip %7.0f shader cycles = %.2f us at 2.1 GHz\n", pr.what, x[0] & 15, x[1] & 15,
                   mode ? "sc0 (XCD)" : "agent scope", (double)cyc / rounds, (double)cyc / rounds / 2100.0); fflush(stdout);
        }
    return 0;
}
