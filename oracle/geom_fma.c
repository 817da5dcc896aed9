/* Plain-C statement of the per-view float64 geometry.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates /root/reference/tools/projection_2d_to_3d.py :424-425 (inv(pose) @ cloud),
 * :37-48 (K @ pts / z, round half to even, int64) and :51-70 (bounds, depth != 0,
 * |z - depth| < thresh) with the accumulation order written out: every dot product is a
 * k-ascending fma chain starting from +0.0, which is what OpenBLAS dgemm computes for these
 * 4x4 and 3x3 products (checked against NumPy in tests/test_oracle_geometry.py and against
 * the reference's outputs in tests/golden/proj_helpers.npz).  Unlike the NumPy statement this
 * one cannot change with the BLAS kernel picked for the host CPU.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>

static inline double dot4(const double *r, double x, double y, double z)
{
    double acc = 0.0;
    acc = fma(r[0], x, acc);
    acc = fma(r[1], y, acc);
    acc = fma(r[2], z, acc);
    acc = fma(r[3], 1.0, acc);
    return acc;
}

static inline double dot3(const double *r, double x, double y, double z)
{
    double acc = 0.0;
    acc = fma(r[0], x, acc);
    acc = fma(r[1], y, acc);
    acc = fma(r[2], z, acc);
    return acc;
}

/* xyz: n x 3 (row major); inv_pose: 4x4 row major; k33: 3x3 row major; depth: h x w float32.
 * pts_cam: n x 3 out; pix: n x 2 out as (x, y) with the x86 cast convention (NaN, inf and
 * out-of-range -> INT64_MIN); vis: n out (0/1). */
void bff_ref_view(const double *xyz, int64_t n, const double *inv_pose, const double *k33,
                  const float *depth, int h, int w, double thresh,
                  double *pts_cam, int64_t *pix, uint8_t *vis)
{
    for (int64_t i = 0; i < n; ++i) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const double cx = dot4(inv_pose + 0, x, y, z);
        const double cy = dot4(inv_pose + 4, x, y, z);
        const double cz = dot4(inv_pose + 8, x, y, z);
        pts_cam[3 * i] = cx; pts_cam[3 * i + 1] = cy; pts_cam[3 * i + 2] = cz;
        const double u = rint(dot3(k33 + 0, cx, cy, cz) / cz);
        const double v = rint(dot3(k33 + 3, cx, cy, cz) / cz);
        const int ub = (u >= -9223372036854775808.0 && u < 9223372036854775808.0);
        const int vb = (v >= -9223372036854775808.0 && v < 9223372036854775808.0);
        const int64_t px = ub ? (int64_t)u : INT64_MIN;
        const int64_t py = vb ? (int64_t)v : INT64_MIN;
        pix[2 * i] = px; pix[2 * i + 1] = py;
        uint8_t ok = 0;
        if (px >= 0 && px < w && py >= 0 && py < h) {
            const float d = depth[py * (int64_t)w + px];
            ok = (d != 0.0f) && (fabs(cz - (double)d) < thresh);
        }
        vis[i] = ok;
    }
}
