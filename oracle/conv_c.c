/* CPU oracle, TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; PARITY UNPINNED).
 *
 * Plain-C restatement of the Keras layers VxmDense is built from
 * (call sites train_synthmorph.py:296, 3d_reg.py:305):
 *   Conv3D(nf, 3, padding='same', strides=1) (+ bias, optional LeakyReLU),
 *   channels-last, Keras kernel layout [3][3][3][Cin][Cout], cross-correlation,
 *   zero padding.  Accumulates in double so it can serve as the "true value".
 * Also the trilinear clamp-to-edge warp (Appendix A2/A3) for full-size
 * baselines where the NumPy version is too slow.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC conv_c.c -o liboracle_c.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_conv3d_k3_same(const float* x, const float* w, const float* bias, float* out,
                           int B, int X, int Y, int Z, int Cin, int Cout,
                           int leaky, float alpha)
{
    const long rows = (long)B * X * Y;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * Cout);
#pragma omp for schedule(dynamic, 4)
        for (long r = 0; r < rows; ++r) {
            const int b = (int)(r / ((long)X * Y));
            const int xi = (int)((r / Y) % X);
            const int yi = (int)(r % Y);
            for (int zi = 0; zi < Z; ++zi) {
                for (int co = 0; co < Cout; ++co) acc[co] = bias ? (double)bias[co] : 0.0;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = xi + dx;
                    if (xx < 0 || xx >= X) continue;
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = yi + dy;
                        if (yy < 0 || yy >= Y) continue;
                        for (int dz = -1; dz <= 1; ++dz) {
                            const int zz = zi + dz;
                            if (zz < 0 || zz >= Z) continue;
                            const float* xp = x + ((((long)b * X + xx) * Y + yy) * Z + zz) * Cin;
                            const float* wp = w + (long)(((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1)) * Cin * Cout;
                            for (int ci = 0; ci < Cin; ++ci) {
                                const double xv = xp[ci];
                                const float* wr = wp + (long)ci * Cout;
                                for (int co = 0; co < Cout; ++co) acc[co] += xv * (double)wr[co];
                            }
                        }
                    }
                }
                float* op = out + ((((long)b * X + xi) * Y + yi) * Z + zi) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    float v = (float)acc[co];
                    if (leaky && v < 0.f) v *= alpha;
                    op[co] = v;
                }
            }
        }
        free(acc);
    }
}

/* float-accumulating variant used by the CPU baseline timing (faster, SIMD over Cout). */
void oracle_conv3d_k3_same_f32acc(const float* x, const float* w, const float* bias, float* out,
                                  int B, int X, int Y, int Z, int Cin, int Cout,
                                  int leaky, float alpha)
{
    const long rows = (long)B * X * Y;
#pragma omp parallel
    {
        float* acc = (float*)malloc(sizeof(float) * Cout);
#pragma omp for schedule(dynamic, 4)
        for (long r = 0; r < rows; ++r) {
            const int b = (int)(r / ((long)X * Y));
            const int xi = (int)((r / Y) % X);
            const int yi = (int)(r % Y);
            for (int zi = 0; zi < Z; ++zi) {
                for (int co = 0; co < Cout; ++co) acc[co] = bias ? bias[co] : 0.f;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = xi + dx;
                    if (xx < 0 || xx >= X) continue;
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = yi + dy;
                        if (yy < 0 || yy >= Y) continue;
                        for (int dz = -1; dz <= 1; ++dz) {
                            const int zz = zi + dz;
                            if (zz < 0 || zz >= Z) continue;
                            const float* xp = x + ((((long)b * X + xx) * Y + yy) * Z + zz) * Cin;
                            const float* wp = w + (long)(((dx + 1) * 3 + (dy + 1)) * 3 + (dz + 1)) * Cin * Cout;
                            for (int ci = 0; ci < Cin; ++ci) {
                                const float xv = xp[ci];
                                const float* wr = wp + (long)ci * Cout;
#pragma omp simd
                                for (int co = 0; co < Cout; ++co) acc[co] += xv * wr[co];
                            }
                        }
                    }
                }
                float* op = out + ((((long)b * X + xi) * Y + yi) * Z + zi) * Cout;
                for (int co = 0; co < Cout; ++co) {
                    float v = acc[co];
                    if (leaky && v < 0.f) v *= alpha;
                    op[co] = v;
                }
            }
        }
        free(acc);
    }
}

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* out[b,x,y,z,c] = vol[b, (x,y,z)+flow[b,x,y,z,:], c], linear, clamp-to-edge
 * (Appendix A3 weights: w_corner0 = loc1 - clipped, w_corner1 = 1 - w_corner0). */
void oracle_warp3d_linear(const float* vol, const float* flow, float* out,
                          int B, int X, int Y, int Z, int C)
{
    const long rows = (long)B * X * Y;
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int b = (int)(r / ((long)X * Y));
        const int xi = (int)((r / Y) % X);
        const int yi = (int)(r % Y);
        for (int zi = 0; zi < Z; ++zi) {
            const long vox = (((long)b * X + xi) * Y + yi) * Z + zi;
            const float* f = flow + vox * 3;
            const float loc[3] = {(float)xi + f[0], (float)yi + f[1], (float)zi + f[2]};
            const int mx[3] = {X - 1, Y - 1, Z - 1};
            int i0[3], i1[3];
            float w0[3], w1[3];
            for (int d = 0; d < 3; ++d) {
                const float fl = floorf(loc[d]);
                const float cl = clampf(loc[d], 0.f, (float)mx[d]);
                const float l0 = clampf(fl, 0.f, (float)mx[d]);
                const float l1 = clampf(l0 + 1.f, 0.f, (float)mx[d]);
                i0[d] = (int)l0;
                i1[d] = (int)l1;
                w0[d] = l1 - cl;
                w1[d] = 1.f - w0[d];
            }
            float* op = out + vox * C;
            for (int c = 0; c < C; ++c) op[c] = 0.f;
            for (int cx = 0; cx < 2; ++cx)
                for (int cy = 0; cy < 2; ++cy)
                    for (int cz = 0; cz < 2; ++cz) {
                        const float w = ((cx ? w1[0] : w0[0]) * (cy ? w1[1] : w0[1])) * (cz ? w1[2] : w0[2]);
                        const long src = ((((long)b * X + (cx ? i1[0] : i0[0])) * Y + (cy ? i1[1] : i0[1])) * Z +
                                          (cz ? i1[2] : i0[2])) * C;
                        for (int c = 0; c < C; ++c) op[c] += w * vol[src + c];
                    }
        }
    }
}
