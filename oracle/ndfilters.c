/* CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the SciPy ndimage arithmetic that skimage.blob_log runs
 * (the only native code on the reference's hot path; it lives in the third-party
 * wheel scipy 1.15.3, `scipy/ndimage/src/ni_filters.c`, which is NOT under
 * /root/reference -- restated here from its published algorithm and pinned
 * bit-for-bit against the installed scipy by tests/test_oracle_c.py):
 *
 *   mmo_correlate1d_f64     NI_Correlate1D, symmetric-kernel branch, mode "reflect"
 *                           (SCI/ndimage/_filters.py:126-182 calls it):
 *                             out = in[c]*w[c];  for k = R..1: out += (in[c-k] + in[c+k]) * w[c-k]
 *                           accumulated in double, no FMA contraction (build with
 *                           -ffp-contract=off), result stored in the array dtype.
 *   mmo_gaussian_laplace    SCI/ndimage/_filters.py:644-707 via generic_laplace
 *                           :554-602 and gaussian_filter :327-430: for each axis a,
 *                           three sequential 1-D passes in axis order 0,1,2 with
 *                           the order-2 kernel on axis a; out = t0; out += t1; out += t2.
 *   mmo_peak_mask4d         skimage peak.py:28-50 on the (z,y,x,sigma) cube:
 *                           3^4 max with zero padding, equality, "> threshold",
 *                           all-equal cube has no peaks.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* scipy "reflect" = half-sample symmetric: d c b a | a b c d | d c b a */
static inline int64_t reflect_index(int64_t i, int64_t n)
{
    if (n == 1) return 0;
    int64_t period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

/* One 1-D pass along `axis` of a C-contiguous 3-D array.  `w` has 2R+1 weights
 * (already in correlation order; symmetric).  in/out may not alias. */
int mmo_correlate1d_f64(const double *in, double *out, const int64_t dims[3], int axis,
                        const double *w, int R)
{
    int64_t n = dims[axis];
    int64_t stride = 1;
    for (int a = 2; a > axis; --a) stride *= dims[a];
    int64_t outer = 1, inner = stride;
    for (int a = 0; a < axis; ++a) outer *= dims[a];
    double *line = (double *)malloc(sizeof(double) * (size_t)(n + 2 * R));
    if (!line) return -1;
    const double *fw = w + R; /* centre */
    for (int64_t o = 0; o < outer; ++o) {
        for (int64_t i = 0; i < inner; ++i) {
            const double *src = in + o * n * stride + i;
            double *dst = out + o * n * stride + i;
            for (int64_t j = -R; j < n + R; ++j)
                line[j + R] = src[reflect_index(j, n) * stride];
            const double *il = line + R;
            for (int64_t l = 0; l < n; ++l) {
                double acc = il[0] * fw[0];
                for (int jj = -R; jj < 0; ++jj)
                    acc += (il[jj] + il[-jj]) * fw[jj];
                dst[l * stride] = acc;
                ++il;
            }
        }
    }
    free(line);
    return 0;
}

/* float32 storage variant: values are widened to double per line, accumulated in
 * double and rounded to float on store (what NI_Correlate1D does for float32 arrays). */
int mmo_correlate1d_f32(const float *in, float *out, const int64_t dims[3], int axis,
                        const double *w, int R)
{
    int64_t n = dims[axis];
    int64_t stride = 1;
    for (int a = 2; a > axis; --a) stride *= dims[a];
    int64_t outer = 1, inner = stride;
    for (int a = 0; a < axis; ++a) outer *= dims[a];
    double *line = (double *)malloc(sizeof(double) * (size_t)(n + 2 * R));
    if (!line) return -1;
    const double *fw = w + R;
    for (int64_t o = 0; o < outer; ++o) {
        for (int64_t i = 0; i < inner; ++i) {
            const float *src = in + o * n * stride + i;
            float *dst = out + o * n * stride + i;
            for (int64_t j = -R; j < n + R; ++j)
                line[j + R] = (double)src[reflect_index(j, n) * stride];
            const double *il = line + R;
            for (int64_t l = 0; l < n; ++l) {
                double acc = il[0] * fw[0];
                for (int jj = -R; jj < 0; ++jj)
                    acc += (il[jj] + il[-jj]) * fw[jj];
                dst[l * stride] = (float)acc;
                ++il;
            }
        }
    }
    free(line);
    return 0;
}

/* gaussian_laplace for one sigma.  w0 = order-0 weights, w2 = order-2 weights, both
 * 2R+1 long (computed by the caller exactly as scipy's _gaussian_kernel1d does).
 * out = sum over a of  pass2( pass1( pass0(in) ) )  with w2 on axis a. */
int mmo_gaussian_laplace_f64(const double *in, double *out, const int64_t dims[3],
                             const double *w0, const double *w2, int R)
{
    int64_t nvox = dims[0] * dims[1] * dims[2];
    double *t1 = (double *)malloc(sizeof(double) * (size_t)nvox);
    double *t2 = (double *)malloc(sizeof(double) * (size_t)nvox);
    if (!t1 || !t2) { free(t1); free(t2); return -1; }
    for (int a = 0; a < 3; ++a) {
        /* gaussian_filter: axis 0 reads the input, later axes run "in place" on the
         * output array (scipy buffers each line, so that is well defined). */
        mmo_correlate1d_f64(in, t1, dims, 0, a == 0 ? w2 : w0, R);
        mmo_correlate1d_f64(t1, t2, dims, 1, a == 1 ? w2 : w0, R);
        if (a == 0) {
            mmo_correlate1d_f64(t2, out, dims, 2, w0, R);
        } else {
            mmo_correlate1d_f64(t2, t1, dims, 2, a == 2 ? w2 : w0, R);
            for (int64_t i = 0; i < nvox; ++i) out[i] += t1[i];
        }
    }
    free(t1);
    free(t2);
    return 0;
}

int mmo_gaussian_laplace_f32(const float *in, float *out, const int64_t dims[3],
                             const double *w0, const double *w2, int R)
{
    int64_t nvox = dims[0] * dims[1] * dims[2];
    float *t1 = (float *)malloc(sizeof(float) * (size_t)nvox);
    float *t2 = (float *)malloc(sizeof(float) * (size_t)nvox);
    if (!t1 || !t2) { free(t1); free(t2); return -1; }
    for (int a = 0; a < 3; ++a) {
        mmo_correlate1d_f32(in, t1, dims, 0, a == 0 ? w2 : w0, R);
        mmo_correlate1d_f32(t1, t2, dims, 1, a == 1 ? w2 : w0, R);
        if (a == 0) {
            mmo_correlate1d_f32(t2, out, dims, 2, w0, R);
        } else {
            mmo_correlate1d_f32(t2, t1, dims, 2, a == 2 ? w2 : w0, R);
            for (int64_t i = 0; i < nvox; ++i) out[i] += t1[i];
        }
    }
    free(t1);
    free(t2);
    return 0;
}

/* cube is (z, y, x, s) C-contiguous.  mask gets 0/1.  Returns the number of peaks. */
int64_t mmo_peak_mask4d_f64(const double *cube, const int64_t dims[3], int ns, double thr,
                            uint8_t *mask)
{
    const int64_t nz = dims[0], ny = dims[1], nx = dims[2];
    const int64_t sz = ny * nx * ns, sy = nx * ns, sx = ns;
    int64_t total = nz * ny * nx * ns, n_eq = 0, n_peaks = 0;
    for (int64_t z = 0; z < nz; ++z)
     for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x)
       for (int s = 0; s < ns; ++s) {
           double m = -INFINITY;
           for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
             for (int dx = -1; dx <= 1; ++dx)
              for (int ds = -1; ds <= 1; ++ds) {
                  int64_t zz = z + dz, yy = y + dy, xx = x + dx;
                  int ss = s + ds;
                  double v = 0.0; /* mode='constant', cval 0 */
                  if (zz >= 0 && zz < nz && yy >= 0 && yy < ny && xx >= 0 && xx < nx &&
                      ss >= 0 && ss < ns)
                      v = cube[zz * sz + yy * sy + xx * sx + ss];
                  if (v > m) m = v;
              }
           int64_t idx = z * sz + y * sy + x * sx + s;
           uint8_t eq = cube[idx] == m;
           n_eq += eq;
           mask[idx] = eq && cube[idx] > thr;
           n_peaks += mask[idx];
       }
    if (n_eq == total && total > 1) { /* trivial image: no peaks */
        memset(mask, 0, (size_t)total);
        return 0;
    }
    return n_peaks;
}
