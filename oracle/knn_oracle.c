/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of simple_knn._C.distCUDA2, called by the reference at
 *   custom/threestudio-dreammesh4d/geometry/gaussian_base.py:435-438
 * (imported at geometry/sugar.py:11, geometry/dynamic_sugar.py:10).
 * The CUDA source (DSaurus/simple-knn, un-pinned, requirements.txt:50,
 * README.md:36) is ABSENT from /root/reference.  Published semantics:
 *   out[i] = mean of the squared Euclidean distances from point i to its 3
 *            nearest OTHER points (self excluded by index, duplicates count);
 * upstream finds them with a Morton-order box-pruned search that is exact, so
 * the result is defined up to float rounding of d^2 = dx*dx + dy*dy + dz*dz.
 *
 * PARITY UNPINNED (no reference tests / vectors for this path).  Pinned here
 * by: brute force (dm4d_oracle_dist2_brute) == grid search
 * (dm4d_oracle_dist2_knn3) bit-for-bit, and numpy brute force in tests/.
 *
 * Arithmetic contract shared with the HIP kernel: d2 = (dx*dx + dy*dy) + dz*dz
 * without contraction; result = ((b0 + b1) + b2) / 3.0f with b0<=b1<=b2.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float d2f(const float *p, const float *q)
{
    float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return (dx * dx + dy * dy) + dz * dz;
}
static inline void push3(float *b, float d)
{
    if (d < b[2]) {
        if (d < b[1]) {
            b[2] = b[1];
            if (d < b[0]) { b[1] = b[0]; b[0] = d; } else b[1] = d;
        } else b[2] = d;
    }
}

void dm4d_oracle_dist2_brute(int N, const float *pts, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            push3(b, d2f(pts + 3 * i, pts + 3 * j));
        }
        out[i] = ((b[0] + b[1]) + b[2]) / 3.0f;
    }
}

/* exact 3-NN through a uniform grid (ring expansion until the 3rd best is closer than the next ring) */
void dm4d_oracle_dist2_knn3(int N, const float *pts, float *out)
{
    if (N <= 0) return;
    if (N < 64) { dm4d_oracle_dist2_brute(N, pts, out); return; }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = pts[3 * i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    double ext = 0;
    for (int k = 0; k < 3; ++k) if (mx[k] - mn[k] > ext) ext = mx[k] - mn[k];
    if (!(ext > 0) || !isfinite(ext)) { dm4d_oracle_dist2_brute(N, pts, out); return; }
    int G = (int)cbrt((double)N / 2.0);
    if (G < 1) G = 1;
    if (G > 256) G = 256;
    double cell = ext / G * 1.0000001;
    int *cidx = (int *)malloc(sizeof(int) * (size_t)N);
    int *cstart = (int *)calloc((size_t)G * G * G + 1, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; ++i) {
        int c[3];
        for (int k = 0; k < 3; ++k) {
            int v = (int)((pts[3 * i + k] - mn[k]) / cell);
            c[k] = v < 0 ? 0 : (v >= G ? G - 1 : v);
        }
        cidx[i] = (c[2] * G + c[1]) * G + c[0];
        cstart[cidx[i] + 1]++;
    }
    for (int c = 0; c < G * G * G; ++c) cstart[c + 1] += cstart[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)G * G * G);
    memcpy(fill, cstart, sizeof(int) * (size_t)G * G * G);
    for (int i = 0; i < N; ++i) order[fill[cidx[i]]++] = i;
    free(fill);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < N; ++i) {
        const float *p = pts + 3 * i;
        int c0 = cidx[i] % G, c1 = (cidx[i] / G) % G, c2 = cidx[i] / (G * G);
        float b[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int ring = 0; ring <= G; ++ring) {
            /* everything outside rings < ring is at least (ring-1)*cell... use a safe bound */
            if (ring >= 1) {
                double lim = (double)(ring - 1) * cell;
                if (b[2] < FLT_MAX && (double)b[2] < lim * lim * 0.999999) break;
            }
            for (int z = c2 - ring; z <= c2 + ring; ++z) {
                if (z < 0 || z >= G) continue;
                for (int y = c1 - ring; y <= c1 + ring; ++y) {
                    if (y < 0 || y >= G) continue;
                    for (int x = c0 - ring; x <= c0 + ring; ++x) {
                        if (x < 0 || x >= G) continue;
                        int cheb = abs(x - c0);
                        if (abs(y - c1) > cheb) cheb = abs(y - c1);
                        if (abs(z - c2) > cheb) cheb = abs(z - c2);
                        if (cheb != ring) continue;
                        int c = (z * G + y) * G + x;
                        for (int e = cstart[c]; e < cstart[c + 1]; ++e) {
                            int j = order[e];
                            if (j == i) continue;
                            push3(b, d2f(p, pts + 3 * j));
                        }
                    }
                }
            }
        }
        out[i] = ((b[0] + b[1]) + b[2]) / 3.0f;
    }
    free(cidx);
    free(cstart);
    free(order);
}
