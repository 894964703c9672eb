// Hand-written HIP kernels for gfx950 (CDNA4, wave64) — SAGE-ICP registration hot path.
//
//   k_nn   semantic nearest-neighbour search over the 27-voxel neighbourhood of a GPU-resident
//          open-addressed voxel hash.   Replaces VoxelHashMap::GetCorrespondences' per-point
//          lambda (reference core/VoxelHashMap.cpp:51-96) and its acceptance test (:109-115),
//          with TransformPoints (core/Registration.cpp:103-111,133) fused in: the cumulative
//          pose is applied to the pristine frame instead of re-writing `source` every iteration.
//   k_gn   robust-weighted point-to-point Gauss-Newton accumulation (Registration.cpp:62-90) as
//          16 closed-form fp64 sums + count, wave-shuffle -> LDS -> one partial per workgroup.
//   k_fin  fixed-order reduction of the workgroup partials, 6x6 LDL^T solve, SE3 exp, pose
//          composition and the convergence test (Registration.cpp:92-93,135-137), all on device so
//          the host never round-trips inside the ICP loop.
//   k_tf   TransformPoints for the stand-alone API entry (Registration.cpp:103-111).
//
// Roofline: HBM-bound integer/byte + fp64 compare work (~0.1 flop/B) — no MFMA (a 6x6 outer
// product sum is not a dense contraction).  One wavefront owns one query: lanes 0..26 probe the
// 27 neighbour voxels in parallel (one 16-B slot load per probe step), then the wave walks the
// occupied voxels in reference order (x outer, y, z inner), 64 lanes loading one voxel block's
// points as one contiguous, coalesced run of 32-B records.
//
// Built with -ffp-contract=off: distances are the plain IEEE sequence
// dx*dx + (dy*dy + dz*dz) the CPU evaluates, so the argmin is index-exact against the oracle.

#include <hip/hip_runtime.h>

#include <cfloat>

#include "kernels.h"
#include "se3_math.h"
#include "sageicp_types.h"

namespace sageicp {

__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}

// ------------------------------------------------------------------------------------ k_nn
template <bool APPLY_POSE>
__global__ __launch_bounds__(256) void k_nn(NnParams P) {
    if (APPLY_POSE && P.st->done) return;

    const int lane = threadIdx.x & 63;
    // Workgroup b runs on XCD b % 8 (observed dispatch order; used for L2 affinity only):
    // give each XCD one contiguous eighth of the (spatially coherent) query range.
    const unsigned G = gridDim.x;  // multiple of 8
    const unsigned L = (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3);
    const unsigned wave = __builtin_amdgcn_readfirstlane(L * 4u + (threadIdx.x >> 6));
    const unsigned total_waves = G * 4u;
    const int chunk = (P.n + static_cast<int>(total_waves) - 1) / static_cast<int>(total_waves);
    const int q0 = static_cast<int>(wave) * chunk;
    const int q1 = min(q0 + chunk, P.n);

    double R[9], t[3];
    if (APPLY_POSE) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = P.st->R[i];
        t[0] = P.st->T[4]; t[1] = P.st->T[5]; t[2] = P.st->T[6];
    }

    // neighbour offset of this lane: x outer, y, z inner (VoxelHashMap.cpp:57-63)
    const int ox = lane / 9 - 1, oy = (lane / 3) % 3 - 1, oz = lane % 3 - 1;

    unsigned wave_candidates = 0;   // wave-uniform: sum of C_q over this wave's queries

    for (int q = q0; q < q1; ++q) {
        const Point4 fq = P.frame[q];
        double px, py, pz;
        if (APPLY_POSE) {
            px = R[0] * fq.x + R[1] * fq.y + R[2] * fq.z + t[0];
            py = R[3] * fq.x + R[4] * fq.y + R[5] * fq.z + t[1];
            pz = R[6] * fq.x + R[7] * fq.y + R[8] * fq.z + t[2];
        } else {
            px = fq.x; py = fq.y; pz = fq.z;
        }
        const double pl = fq.l;
        const int pli = static_cast<int>(pl);

        // static_cast<int>(p / voxel_size): exact fp64 divide, trunc toward zero.  One divide
        // sequence serves the three axes (lane 0/1/2), results broadcast by readlane.
        const double c = (lane == 0) ? px : ((lane == 1) ? py : pz);
        const int kc = static_cast<int>(c / P.voxel_size);
        const int kx = __builtin_amdgcn_readlane(kc, 0);
        const int ky = __builtin_amdgcn_readlane(kc, 1);
        const int kz = __builtin_amdgcn_readlane(kc, 2);

        // 27 parallel hash probes
        uint32_t blk = kEmptySlot;
        if (lane < 27) {
            const int vx = kx + ox, vy = ky + oy, vz = kz + oz;
            uint32_t s = voxel_hash(vx, vy, vz) & P.mask;
            for (;;) {
                const Slot e = P.table[s];
                if (e.blk == kEmptySlot) break;
                if (e.x == vx && e.y == vy && e.z == vz) { blk = e.blk; break; }
                s = (s + 1) & P.mask;
            }
        }
        unsigned long long occupied = __ballot(blk != kEmptySlot);

        double best = DBL_MAX;      // scaled squared distance (closest_distance2)
        double best_raw = 0.0;      // unscaled squared distance of that candidate
        int best_idx = -1;
        unsigned best_key = 0xFFFFFFFFu;  // (voxel order << 8) | slot : first-minimum tie-break

        while (occupied) {  // wave-uniform walk in reference enumeration order
            const int v = __builtin_ctzll(occupied);
            occupied &= occupied - 1;
            const uint32_t vb = rl_u32(blk, v);
            const uint32_t count = vb & 255u;
            const uint32_t base = (vb >> 8) * static_cast<uint32_t>(P.cap);
            wave_candidates += count;
            for (uint32_t s0 = 0; s0 < count; s0 += 64) {
                const uint32_t slot = s0 + lane;
                if (slot < count) {
                    const Point4 nb = P.pts[base + slot];
                    const double dx = nb.x - px, dy = nb.y - py, dz = nb.z - pz;
                    const double raw = dx * dx + (dy * dy + dz * dz);
                    double d = raw;
                    // same label, or either side unlabelled (VoxelHashMap.cpp:87-88)
                    if (static_cast<int>(nb.l) == pli || static_cast<int>(nb.l * pl) == 0)
                        d = d * P.sem_th;
                    if (d < best) {  // strict <: first minimum wins within the lane
                        best = d;
                        best_raw = raw;
                        best_idx = static_cast<int>(base + slot);
                        best_key = (static_cast<unsigned>(v) << 8) | slot;
                    }
                }
            }
        }

        // cross-lane argmin, lexicographic on (distance, enumeration key)
        double wbest = best;
        unsigned wkey = best_key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(wbest, off, 64);
            const unsigned ok = __shfl_xor(wkey, off, 64);
            if (ob < wbest || (ob == wbest && ok < wkey)) { wbest = ob; wkey = ok; }
        }
        if (wkey == 0xFFFFFFFFu) {
            if (lane == 0) P.nn_idx[q] = -1;   // no candidate at all -> rejected
        } else if (best_key == wkey) {
            // acceptance on the UNscaled Euclidean distance (VoxelHashMap.cpp:111)
            P.nn_idx[q] = (sqrt(best_raw) < P.max_dist) ? best_idx : -1;
        }
    }
    if (P.cand_counter && lane == 0 && wave_candidates)
        atomicAdd(P.cand_counter, static_cast<unsigned long long>(wave_candidates));
}

// ------------------------------------------------------------------------------------ k_gn
__global__ __launch_bounds__(256) void k_gn(GnParams P) {
    if (P.apply_pose && P.st->done) return;
    __shared__ double lds[4][kNumSums];

    double acc[kCount + 1];
#pragma unroll
    for (int i = 0; i <= kCount; ++i) acc[i] = 0.0;

    double R[9], t[3];
    if (P.apply_pose) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = P.st->R[i];
        t[0] = P.st->T[4]; t[1] = P.st->T[5]; t[2] = P.st->T[6];
    }
    const double k = P.kernel;
    const double k2 = k * k;

    for (int q = blockIdx.x * 256 + threadIdx.x; q < P.n; q += gridDim.x * 256) {
        Point4 g;
        if (P.tgt_pairs) {
            g = P.tgt_pairs[q];
        } else {
            const int idx = P.nn_idx[q];
            if (idx < 0) continue;
            g = P.pts[idx];
        }
        const Point4 fq = P.frame[q];
        double sx, sy, sz;
        if (P.apply_pose) {
            sx = R[0] * fq.x + R[1] * fq.y + R[2] * fq.z + t[0];
            sy = R[3] * fq.x + R[4] * fq.y + R[5] * fq.z + t[1];
            sz = R[6] * fq.x + R[7] * fq.y + R[8] * fq.z + t[2];
        } else {
            sx = fq.x; sy = fq.y; sz = fq.z;
        }
        const double rx = sx - g.x, ry = sy - g.y, rz = sz - g.z;
        const double r2 = rx * rx + (ry * ry + rz * rz);
        const double den = k + r2;
        const double w = k2 / (den * den);   // square(th) / square(th + residual2)
        const double wsx = w * sx, wsy = w * sy, wsz = w * sz;
        acc[kW] += w;
        acc[kWsx] += wsx; acc[kWsy] += wsy; acc[kWsz] += wsz;
        acc[kWxx] += wsx * sx; acc[kWxy] += wsx * sy; acc[kWxz] += wsx * sz;
        acc[kWyy] += wsy * sy; acc[kWyz] += wsy * sz; acc[kWzz] += wsz * sz;
        acc[kWrx] += w * rx; acc[kWry] += w * ry; acc[kWrz] += w * rz;
        acc[kWcx] += w * (sy * rz - sz * ry);
        acc[kWcy] += w * (sz * rx - sx * rz);
        acc[kWcz] += w * (sx * ry - sy * rx);
        acc[kCount] += 1.0;
    }

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i <= kCount; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) lds[wv][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kNumSums) {
        double v = 0.0;
        if (threadIdx.x <= kCount)
            v = ((lds[0][threadIdx.x] + lds[1][threadIdx.x]) + lds[2][threadIdx.x]) +
                lds[3][threadIdx.x];
        P.partials[blockIdx.x * kNumSums + threadIdx.x] = v;
    }
}

// ------------------------------------------------------------------------------------ k_fin
// mode 0: reduce partials + solve      (single GPU)
// mode 1: reduce partials -> st->sums  (multi GPU, before the RCCL all-reduce)
// mode 2: solve from st->sums          (multi GPU, after the all-reduce)
__global__ __launch_bounds__(256) void k_fin(IcpState *st, const double *partials, int nparts,
                                             int mode, int standalone) {
    if (!standalone && st->done) return;
    __shared__ double slice[8][32];
    __shared__ double S[kNumSums];

    if (mode != 2) {
        const int comp = threadIdx.x & 31, sl = threadIdx.x >> 5;
        double v = 0.0;
        if (comp < kNumSums)
            for (int b = sl; b < nparts; b += 8) v += partials[b * kNumSums + comp];
        slice[sl][comp] = v;
        __syncthreads();
        if (threadIdx.x < kNumSums) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += slice[i][threadIdx.x];   // fixed order
            S[threadIdx.x] = s;
            st->sums[threadIdx.x] = s;
        }
        __syncthreads();
        if (mode == 1) return;
    } else {
        if (threadIdx.x < kNumSums) S[threadIdx.x] = st->sums[threadIdx.x];
        __syncthreads();
    }

    if (threadIdx.x != 0) return;
    double JTJ[36], JTr[6], neg[6], x[6], est[7];
    assemble_normal_equations(S, JTJ, JTr);
    for (int i = 0; i < 6; ++i) neg[i] = -JTr[i];
    ldlt_solve6(JTJ, neg, x);
    se3_exp(x, est);

    double Tn[7];
    se3_mul(est, st->T, Tn);
    for (int i = 0; i < 7; ++i) st->T[i] = Tn[i];
    quat_to_mat(Tn, st->R);
    se3_mul(est, st->T_icp, Tn);
    for (int i = 0; i < 7; ++i) st->T_icp[i] = Tn[i];

    double lg[6];
    se3_log(est, lg);
    double nrm = 0.0;
    for (int i = 0; i < 6; ++i) nrm += lg[i] * lg[i];
    nrm = sqrt(nrm);
    st->last_step_norm = nrm;
    const int it = st->iter;
    if (it < kHistory) st->n_corr[it] = static_cast<uint32_t>(S[kCount]);
    st->iter = it + 1;
    if (nrm < kEstimationThreshold) {
        st->converged = 1;
        st->done = 1;
    } else if (it + 1 >= kMaxIterations) {
        st->done = 1;
    }
}

// ------------------------------------------------------------------------------------ k_tf
__global__ __launch_bounds__(256) void k_tf(Point4 *pts, int n, const IcpState *st) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Point4 p = pts[i];
    const double *R = st->R;
    const double x = R[0] * p.x + R[1] * p.y + R[2] * p.z + st->T[4];
    const double y = R[3] * p.x + R[4] * p.y + R[5] * p.z + st->T[5];
    const double z = R[6] * p.x + R[7] * p.y + R[8] * p.z + st->T[6];
    p.x = x; p.y = y; p.z = z;
    pts[i] = p;
}

// ------------------------------------------------------------------------------------ launchers
int nn_grid_for(int n) {
    // one wave per query chunk; 2048 workgroups x 4 waves = every wave slot of the chip
    // (256 CUs x 32 waves) once; smaller inputs shrink the grid in multiples of 8 (XCD remap).
    long waves = n;
    long blocks = (waves + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    blocks = ((blocks + 7) / 8) * 8;
    return static_cast<int>(blocks);
}

int gn_grid_for(int n) {
    long blocks = (static_cast<long>(n) + 255) / 256;
    if (blocks > kMaxGnBlocks) blocks = kMaxGnBlocks;
    if (blocks < 1) blocks = 1;
    return static_cast<int>(blocks);
}

void launch_nn(const NnParams &p, bool apply_pose, hipStream_t s) {
    if (p.n <= 0) return;
    const int grid = nn_grid_for(p.n);
    if (apply_pose)
        hipLaunchKernelGGL(k_nn<true>, dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(k_nn<false>, dim3(grid), dim3(256), 0, s, p);
}

int launch_gn(const GnParams &p, hipStream_t s) {
    const int grid = gn_grid_for(p.n);
    hipLaunchKernelGGL(k_gn, dim3(grid), dim3(256), 0, s, p);
    return grid;
}

void launch_fin(IcpState *st, const double *partials, int nparts, int mode, int standalone,
                hipStream_t s) {
    hipLaunchKernelGGL(k_fin, dim3(1), dim3(256), 0, s, st, partials, nparts, mode, standalone);
}

void launch_tf(Point4 *pts, int n, const IcpState *st, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tf, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, st);
}

}  // namespace sageicp
