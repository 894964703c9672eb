// fp64 SE(3) / 6x6 solve helpers used on the device (k_fin) and by the host side of the C ABI.
// Pose layout: T[7] = {qx, qy, qz, qw, tx, ty, tz} == Sophus::SE3d::data().
//
// These replace the third-party arithmetic at the reference's call sites
//   Registration.cpp:92  JTJ.ldlt().solve(-JTr)      -> ldlt_solve6
//   Registration.cpp:93  Sophus::SE3d::exp(x)        -> se3_exp
//   Registration.cpp:135 estimation * T_icp          -> se3_mul
//   Registration.cpp:137 estimation.log().norm()     -> se3_log
//   Registration.cpp:107 T * v3point                 -> rotation matrix form (quat_to_mat)
#pragma once

#include <cmath>

#include "sageicp_types.h"

namespace sageicp {

SAGE_HD inline void quat_to_mat(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double xx = x * x, yy = y * y, zz = z * z;
    const double xy = x * y, xz = x * z, yz = y * z;
    const double wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0 - 2.0 * (yy + zz); R[1] = 2.0 * (xy - wz);       R[2] = 2.0 * (xz + wy);
    R[3] = 2.0 * (xy + wz);       R[4] = 1.0 - 2.0 * (xx + zz); R[5] = 2.0 * (yz - wx);
    R[6] = 2.0 * (xz - wy);       R[7] = 2.0 * (yz + wx);       R[8] = 1.0 - 2.0 * (xx + yy);
}

SAGE_HD inline void mat_apply(const double R[9], const double t[3], const double p[3],
                              double o[3]) {
    o[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    o[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    o[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
}

// O = A * B (group product), quaternion renormalised.
SAGE_HD inline void se3_mul(const double A[7], const double B[7], double O[7]) {
    const double ax = A[0], ay = A[1], az = A[2], aw = A[3];
    const double bx = B[0], by = B[1], bz = B[2], bw = B[3];
    double x = aw * bx + ax * bw + ay * bz - az * by;
    double y = aw * by - ax * bz + ay * bw + az * bx;
    double z = aw * bz + ax * by - ay * bx + az * bw;
    double w = aw * bw - ax * bx - ay * by - az * bz;
    const double inv = 1.0 / sqrt(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    double R[9], t[3];
    quat_to_mat(A, R);
    mat_apply(R, A + 4, B + 4, t);
    O[0] = x; O[1] = y; O[2] = z; O[3] = w;
    O[4] = t[0]; O[5] = t[1]; O[6] = t[2];
}

SAGE_HD inline void se3_inv(const double A[7], double O[7]) {
    const double qi[4] = {-A[0], -A[1], -A[2], A[3]};
    double R[9];
    quat_to_mat(qi, R);
    const double z[3] = {0.0, 0.0, 0.0};
    double t[3];
    mat_apply(R, z, A + 4, t);
    O[0] = qi[0]; O[1] = qi[1]; O[2] = qi[2]; O[3] = qi[3];
    O[4] = -t[0]; O[5] = -t[1]; O[6] = -t[2];
}

// Lane policy of the few expensive scalar operations (fp64 divide, sincos) inside the 6x6 solve
// and the SE3 exponential.  SerialLanes evaluates them one after another (host code, and any
// single-lane caller); kernels.hip supplies a wavefront policy that spreads independent
// divisions / sincos arguments over lanes of the (otherwise idle) wave that finishes an ICP
// iteration.  Operands and operation order per element are the same, so results are bit-identical.
struct SerialLanes {
    static SAGE_HD void divide6(const double (&n)[6], const double (&d)[6], double (&q)[6]) {
        for (int i = 0; i < 6; ++i) q[i] = n[i] / d[i];
    }
    static SAGE_HD void sincos2(double a0, double a1, double &s0, double &c0, double &s1, double &c1) {
        s0 = sin(a0); c0 = cos(a0);
        s1 = sin(a1); c1 = cos(a1);
    }
};

// exp: tangent a = (upsilon, omega), translation first.
template <class Lanes>
SAGE_HD inline void se3_exp_t(const double a[6], double T[7]) {
    const double wx = a[3], wy = a[4], wz = a[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    const double th = sqrt(th2);
    double imag, real, A, B;   // q = (imag*w, real); V = I + A*W + B*W^2
    if (th < 1e-10) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
        A = 0.5 - th2 / 24.0;
        B = 1.0 / 6.0 - th2 / 120.0;
    } else {
        const double half = 0.5 * th;
        double sh, ch, sf, cf;
        Lanes::sincos2(half, th, sh, ch, sf, cf);
        const double num[6] = {sh, 1.0 - cf, th - sf, 0.0, 0.0, 0.0};
        const double den[6] = {th, th2, th2 * th, 1.0, 1.0, 1.0};
        double quo[6];
        Lanes::divide6(num, den, quo);
        imag = quo[0];
        real = ch;
        A = quo[1];
        B = quo[2];
    }
    T[0] = imag * wx; T[1] = imag * wy; T[2] = imag * wz; T[3] = real;
    // V*u = u + A (w x u) + B (w x (w x u))
    const double ux = a[0], uy = a[1], uz = a[2];
    const double cx = wy * uz - wz * uy, cy = wz * ux - wx * uz, cz = wx * uy - wy * ux;
    const double ccx = wy * cz - wz * cy, ccy = wz * cx - wx * cz, ccz = wx * cy - wy * cx;
    T[4] = ux + A * cx + B * ccx;
    T[5] = uy + A * cy + B * ccy;
    T[6] = uz + A * cz + B * ccz;
}
SAGE_HD inline void se3_exp(const double a[6], double T[7]) { se3_exp_t<SerialLanes>(a, T); }

SAGE_HD inline void se3_log(const double T[7], double a[6]) {
    const double n2 = T[0] * T[0] + T[1] * T[1] + T[2] * T[2];
    const double qw = T[3];
    double k, th;   // omega = k * q.vec
    if (n2 < 1e-20) {
        k = 2.0 / qw - (2.0 / 3.0) * n2 / (qw * qw * qw);
        th = 2.0 * n2 / qw;
    } else {
        const double n = sqrt(n2);
        const double at = (qw < 0.0) ? atan2(-n, -qw) : atan2(n, qw);
        k = 2.0 * at / n;
        th = k * n;
    }
    const double wx = k * T[0], wy = k * T[1], wz = k * T[2];
    double c;   // V^-1 = I - 0.5 W + c W^2
    if (fabs(th) < 1e-10) {
        c = 1.0 / 12.0;
    } else {
        const double half = 0.5 * th;
        c = (1.0 - th * cos(half) / (2.0 * sin(half))) / (th * th);
    }
    const double tx = T[4], ty = T[5], tz = T[6];
    const double cx = wy * tz - wz * ty, cy = wz * tx - wx * tz, cz = wx * ty - wy * tx;
    const double ccx = wy * cz - wz * cy, ccy = wz * cx - wx * cz, ccz = wx * cy - wy * cx;
    a[0] = tx - 0.5 * cx + c * ccx;
    a[1] = ty - 0.5 * cy + c * ccy;
    a[2] = tz - 0.5 * cz + c * ccz;
    a[3] = wx; a[4] = wy; a[5] = wz;
}

// Symmetric 6x6 solve A x = b by LDL^T with diagonal pivoting; zero pivots are
// pseudo-inverted (x = 0 for A = 0), matching Eigen::LDLT::solve's behaviour on the
// rank-deficient systems ICP can produce (no correspondences, planar scenes).
// A is row-major, only the lower triangle is read.
//
// Every loop has a compile-time trip count and every array index is a compile-time constant
// after unrolling (the run-time pivot index is matched by an if-cascade), so on the device the
// 6x6 lives in registers: a dynamically indexed A[][] goes to scratch memory and made the
// one-lane solve in k_fin several times slower.
SAGE_HD inline void ldlt_sym_swap(double (&A)[6][6], const int k, const int p) {
    // symmetric interchange of rows/columns k < p on the lower triangle
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if (j < k) { const double s = A[k][j]; A[k][j] = A[p][j]; A[p][j] = s; }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i > p) { const double s = A[i][k]; A[i][k] = A[i][p]; A[i][p] = s; }
    { const double s = A[k][k]; A[k][k] = A[p][p]; A[p][p] = s; }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i > k && i < p) { const double s = A[i][k]; A[i][k] = A[p][i]; A[p][i] = s; }
}

template <class Lanes>
SAGE_HD inline void ldlt_solve6_t(const double *Ain, const double *b, double *x) {
    double A[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) A[i][j] = Ain[i * 6 + j];
    int tr[6] = {0, 1, 2, 3, 4, 5};
    bool zero = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (!zero) {
            int piv = k;
            double big = fabs(A[k][k]);
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i > k) {
                    const double v = fabs(A[i][i]);
                    if (v > big) { big = v; piv = i; }
                }
            tr[k] = piv;
#pragma unroll
            for (int p = 0; p < 6; ++p)
                if (p > k && piv == p) ldlt_sym_swap(A, k, p);
            double tmp[6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < k) tmp[j] = A[j][j] * A[k][j];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < k) A[k][k] -= A[k][j] * tmp[j];
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i > k) {
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        if (j < k) A[i][k] -= A[i][j] * tmp[j];
                }
            const double akk = A[k][k];
            if (fabs(akk) > 0.0) {
                if (k < 5) {                       // A[i][k] /= akk for i > k, one lane each
                    double num[6], den[6], quo[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        num[i] = (i > k) ? A[i][k] : 0.0;
                        den[i] = (i > k) ? akk : 1.0;
                    }
                    Lanes::divide6(num, den, quo);
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        if (i > k) A[i][k] = quo[i];
                }
            } else if (k == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) { tr[j] = j; A[j][j] = 0.0; }
                zero = true;
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = b[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {          // y = P b
#pragma unroll
        for (int p = 0; p < 6; ++p)
            if (p > k && tr[k] == p) { const double s = y[k]; y[k] = y[p]; y[p] = s; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)            // L^-1
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < i) y[i] -= A[i][j] * y[j];
    {                                      // D^+
        double den[6], quo[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            den[i] = (fabs(A[i][i]) > 2.2250738585072014e-308) ? A[i][i] : 1.0;
        Lanes::divide6(y, den, quo);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            y[i] = (fabs(A[i][i]) > 2.2250738585072014e-308) ? quo[i] : 0.0;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)           // L^-T
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j > i) y[i] -= A[j][i] * y[j];
#pragma unroll
    for (int k = 5; k >= 0; --k) {         // P^T
#pragma unroll
        for (int p = 0; p < 6; ++p)
            if (p > k && tr[k] == p) { const double s = y[k]; y[k] = y[p]; y[p] = s; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}
SAGE_HD inline void ldlt_solve6(const double *Ain, const double *b, double *x) {
    ldlt_solve6_t<SerialLanes>(Ain, b, x);
}

// Assemble JTJ (row-major 6x6) and JTr from the 16 closed-form sums (sageicp_types.h Sum).
SAGE_HD inline void assemble_normal_equations(const double *S, double *JTJ, double *JTr) {
    for (int i = 0; i < 36; ++i) JTJ[i] = 0.0;
    const double w = S[kW], sx = S[kWsx], sy = S[kWsy], sz = S[kWsz];
    JTJ[0 * 6 + 0] = w; JTJ[1 * 6 + 1] = w; JTJ[2 * 6 + 2] = w;
    // top-right block: -hat(Sws)
    JTJ[0 * 6 + 4] = sz;  JTJ[0 * 6 + 5] = -sy;
    JTJ[1 * 6 + 3] = -sz; JTJ[1 * 6 + 5] = sx;
    JTJ[2 * 6 + 3] = sy;  JTJ[2 * 6 + 4] = -sx;
    // bottom-left block: hat(Sws) = transpose of the block above
    JTJ[4 * 6 + 0] = sz;  JTJ[5 * 6 + 0] = -sy;
    JTJ[3 * 6 + 1] = -sz; JTJ[5 * 6 + 1] = sx;
    JTJ[3 * 6 + 2] = sy;  JTJ[4 * 6 + 2] = -sx;
    // bottom-right: Sw(|s|^2 I - s s^T)
    const double xx = S[kWxx], xy = S[kWxy], xz = S[kWxz], yy = S[kWyy], yz = S[kWyz], zz = S[kWzz];
    JTJ[3 * 6 + 3] = yy + zz; JTJ[3 * 6 + 4] = -xy;     JTJ[3 * 6 + 5] = -xz;
    JTJ[4 * 6 + 3] = -xy;     JTJ[4 * 6 + 4] = xx + zz; JTJ[4 * 6 + 5] = -yz;
    JTJ[5 * 6 + 3] = -xz;     JTJ[5 * 6 + 4] = -yz;     JTJ[5 * 6 + 5] = xx + yy;
    JTr[0] = S[kWrx]; JTr[1] = S[kWry]; JTr[2] = S[kWrz];
    JTr[3] = S[kWcx]; JTr[4] = S[kWcy]; JTr[5] = S[kWcz];
}

}  // namespace sageicp
