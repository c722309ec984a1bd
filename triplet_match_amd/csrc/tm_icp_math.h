/* tm_icp_math.h — the rigid fit of icp_ (reference include/impl/scene.hpp:385-393: Eigen::umeyama(scn, mdl,
 * false)) from the 15 fixed-point correlation sums of tm_icp_accumulate, for host and device: Horn's closed
 * form (unit quaternion = dominant eigenvector of the 4x4 N matrix, cyclic Jacobi sweeps in double). The same
 * operations in the same order as search::detail::rigid_fit (include/triplet_match/search.hpp), so that the
 * device-side ICP loop (tm_icp_refine) and the host loop give the same poses bit for bit. */
#ifndef TM_ICP_MATH_H_
#define TM_ICP_MATH_H_

#include <cmath>
#include <cstdint>

#include "tm_math.h"

namespace tmx {

TM_HD void rigid_fit_sums(const long long* v, uint32_t count, float* T16) {
    const double inv = 1.0 / 4294967296.0, n = (double)count;
    double cs[3], cm[3], S[3][3];
    for (int c = 0; c < 3; ++c) {
        cs[c] = (double)v[c] * inv / n;
        cm[c] = (double)v[3 + c] * inv / n;
    }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) S[a][b] = (double)v[6 + 3 * a + b] * inv - n * cs[a] * cm[b];
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) off += N[p][q] * N[p][q];
        if (off < 1e-30) break;
        for (int p = 0; p < 4; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (::fabs(N[p][q]) < 1e-300) continue;
                const double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (::fabs(theta) + ::sqrt(theta * theta + 1.0));
                const double c = 1.0 / ::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double a = N[k][p], b = N[k][q];
                    N[k][p] = c * a - s * b;
                    N[k][q] = s * a + c * b;
                }
                for (int k = 0; k < 4; ++k) {
                    const double a = N[p][k], b = N[q][k];
                    N[p][k] = c * a - s * b;
                    N[q][k] = s * a + c * b;
                }
                for (int k = 0; k < 4; ++k) {
                    const double a = V[k][p], b = V[k][q];
                    V[k][p] = c * a - s * b;
                    V[k][q] = s * a + c * b;
                }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k)
        if (N[k][k] > N[best][best]) best = k;
    double q0 = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
    const double nq = ::sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
    q0 /= nq, qx /= nq, qy /= nq, qz /= nq;
    const double R[3][3] = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * q0), 2 * (qx * qz + qy * q0)},
                            {2 * (qx * qy + qz * q0), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * q0)},
                            {2 * (qx * qz - qy * q0), 2 * (qy * qz + qx * q0), 1 - 2 * (qx * qx + qy * qy)}};
    for (int i = 0; i < 16; ++i) T16[i] = 0.f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T16[c * 4 + r] = (float)R[r][c];
        T16[12 + r] = (float)(cm[r] - (R[r][0] * cs[0] + R[r][1] * cs[1] + R[r][2] * cs[2]));
    }
    T16[15] = 1.f;
}

}  // namespace tmx

#endif /* TM_ICP_MATH_H_ */
