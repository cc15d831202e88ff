/* nmf_oracle.c -- CPU restatement of the NMF / PPC factorisation that produces H (second half).
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path never links this.
 *
 * Follows (M = /root/reference/src/main/java/es/udc/fi/dc/irlab/nmf):
 *   M/AbstractNMFDriver.java:118-146          every iteration computes H2 and W2 from the SAME old (H, W), then swaps
 *   M/hcomputation/ComputeHJob.java:88-96     X = A^T W (score > 0 only: VectorByItemHDFSMapper.java:37-40), C = W^T W
 *                                             (CrossProductMapper), Y_j = C h_j (CHMapper), job 4 = the reducer below
 *   M/hcomputation/HComputationReducer.java:57-75        h_j <- h_j .* X_j ./ (Y_j + eps)
 *   M/ppc/hcomputation/PPCHComputationReducer.java:61-96 d = h.Y, e = h.X, X += d, Y += e, infinities clamped to
 *                                             Double.MAX_VALUE, h <- h .* X ./ (Y + eps); L1-normalised when
 *                                             iteration % normalizationFrequency == 0
 *   M/wcomputation/ComputeWJob.java:88-96 + WComputationMapper.java:100-118   w_i <- w_i .* (A h)_i ./ ((H^T H) w_i + eps), clamped
 *   M/MatrixComputationJob.java:41            eps = 1e-12
 * Pinned by the reference's vectors (tests/golden/factorization_test_data.json): NMFTestData W_init/H_init -> W_one/H_one,
 * W_ten/H_ten; PPCTestData likewise (their ten iterations never reach a normalisation) and the 5 x 7 toy h0p -> h1p.  The
 * normalisation branch itself is restated from the source only. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static double clampinf(double v) { return isinf(v) ? (v > 0 ? DBL_MAX : -DBL_MAX) : v; }

/* ratings: COO with 1-based ids; H: n_users x k, W: n_items x k (row r <-> id r + 1), updated in place */
int nmfo_run(int32_t n_users, int32_t n_items, int32_t k, int32_t iterations, int32_t ppc, int32_t norm_freq, int64_t nnz,
             const int32_t* user, const int32_t* item, const float* score, double* H, double* W) {
    const double eps = 1e-12;
    if (n_users <= 0 || n_items <= 0 || k <= 0) return -1;
    double* XH = malloc(sizeof(double) * (size_t)n_users * k);
    double* XW = malloc(sizeof(double) * (size_t)n_items * k);
    double* H2 = malloc(sizeof(double) * (size_t)n_users * k);
    double* W2 = malloc(sizeof(double) * (size_t)n_items * k);
    double* C = malloc(sizeof(double) * (size_t)k * k);
    double* Y = malloc(sizeof(double) * (size_t)k);
    char* seen_u = calloc((size_t)n_users, 1);
    char* seen_i = calloc((size_t)n_items, 1);
    int rc = 0;
    for (int64_t t = 0; t < nnz; t++) {
        if (!(score[t] > 0)) continue;
        if (user[t] < 1 || user[t] > n_users || item[t] < 1 || item[t] > n_items) { rc = -2; goto done; }
        seen_u[user[t] - 1] = 1;
        seen_i[item[t] - 1] = 1;
    }
    for (int32_t j = 0; j < n_users; j++) if (!seen_u[j]) { rc = -3; goto done; }   /* "User %d has not rated any item" */
    for (int32_t i = 0; i < n_items; i++) if (!seen_i[i]) { rc = -4; goto done; }   /* "Item %d has not been rated by anybody" */
    for (int32_t it = 1; it <= iterations; it++) {
        memset(XH, 0, sizeof(double) * (size_t)n_users * k);
        memset(XW, 0, sizeof(double) * (size_t)n_items * k);
        for (int64_t t = 0; t < nnz; t++) {
            if (!(score[t] > 0)) continue;
            const double a = (double)score[t];
            const double* w = W + (size_t)(item[t] - 1) * k;
            const double* h = H + (size_t)(user[t] - 1) * k;
            double* xh = XH + (size_t)(user[t] - 1) * k;
            double* xw = XW + (size_t)(item[t] - 1) * k;
            for (int32_t c = 0; c < k; c++) { xh[c] += a * w[c]; xw[c] += a * h[c]; }
        }
        /* H2 */
        memset(C, 0, sizeof(double) * (size_t)k * k);
        for (int32_t i = 0; i < n_items; i++)
            for (int32_t a = 0; a < k; a++)
                for (int32_t b = 0; b < k; b++) C[(size_t)a * k + b] += W[(size_t)i * k + a] * W[(size_t)i * k + b];
        for (int32_t j = 0; j < n_users; j++) {
            const double* h = H + (size_t)j * k;
            double* x = XH + (size_t)j * k;
            for (int32_t c = 0; c < k; c++) {
                double y = 0.0;
                for (int32_t a = 0; a < k; a++) y += C[(size_t)c * k + a] * h[a];
                Y[c] = y;
            }
            if (ppc) {
                double d = 0.0, e = 0.0;
                for (int32_t c = 0; c < k; c++) { d += h[c] * Y[c]; e += h[c] * x[c]; }
                for (int32_t c = 0; c < k; c++) { x[c] = clampinf(x[c] + d); Y[c] = clampinf(Y[c] + e); }
            }
            double l1 = 0.0;
            for (int32_t c = 0; c < k; c++) { H2[(size_t)j * k + c] = h[c] * (x[c] / (Y[c] + eps)); l1 += fabs(H2[(size_t)j * k + c]); }
            if (ppc && norm_freq != 0 && it % norm_freq == 0)
                for (int32_t c = 0; c < k; c++) H2[(size_t)j * k + c] /= l1;
        }
        /* W2 */
        memset(C, 0, sizeof(double) * (size_t)k * k);
        for (int32_t j = 0; j < n_users; j++)
            for (int32_t a = 0; a < k; a++)
                for (int32_t b = 0; b < k; b++) C[(size_t)a * k + b] += H[(size_t)j * k + a] * H[(size_t)j * k + b];
        for (int32_t i = 0; i < n_items; i++) {
            const double* w = W + (size_t)i * k;
            for (int32_t c = 0; c < k; c++) {
                double y = 0.0;
                for (int32_t a = 0; a < k; a++) y += C[(size_t)c * k + a] * w[a];
                W2[(size_t)i * k + c] = w[c] * (clampinf(XW[(size_t)i * k + c]) / (clampinf(y) + eps));
            }
        }
        memcpy(H, H2, sizeof(double) * (size_t)n_users * k);
        memcpy(W, W2, sizeof(double) * (size_t)n_items * k);
    }
done:
    free(XH); free(XW); free(H2); free(W2); free(C); free(Y); free(seen_u); free(seen_i);
    return rc;
}
