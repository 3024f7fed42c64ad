/*
 * fsq_register_oracle.c - CPU restatement of phase_correlate.phase_correlate / _dftups
 * (reference phase_correlate.py:11-196).  TEST INFRASTRUCTURE (see fsq_oracle.c header).
 *
 * The reference uses numpy.fft (pocketfft, un-vendored third party).  Parity here is pinned by
 * outputs of the reference itself (tests/golden/registration.npz): shifts must agree exactly on the
 * 1/upsample grid, error/diffphase to rounding level (different FFT factorizations round
 * differently), as SURVEY.md 8c prescribes.  The transforms below are plain separable DFTs with
 * exactly-reduced twiddle indices: O(HW(H+W)), fine for test-sized inputs.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include "fsq_oracle.h"

typedef double complex cplx;
#define TWO_PI 6.283185307179586

static void dft_rows(const cplx *in, cplx *out, int H, int W, int sign)
{   /* transform along the last axis */
    cplx *tw = (cplx *)malloc(sizeof(cplx) * W);
    for (int k = 0; k < W; k++) tw[k] = cexp(sign * I * TWO_PI * k / W);
#pragma omp parallel for
    for (int h = 0; h < H; h++)
        for (int k = 0; k < W; k++) {
            cplx s = 0;
            for (int j = 0; j < W; j++) s += in[(size_t)h * W + j] * tw[(int)(((long)j * k) % W)];
            out[(size_t)h * W + k] = s;
        }
    free(tw);
}

static void dft_cols(const cplx *in, cplx *out, int H, int W, int sign)
{
    cplx *tw = (cplx *)malloc(sizeof(cplx) * H);
    for (int k = 0; k < H; k++) tw[k] = cexp(sign * I * TWO_PI * k / H);
#pragma omp parallel for
    for (int k = 0; k < H; k++)
        for (int w = 0; w < W; w++) {
            cplx s = 0;
            for (int j = 0; j < H; j++) s += in[(size_t)j * W + w] * tw[(int)(((long)j * k) % H)];
            out[(size_t)k * W + w] = s;
        }
    free(tw);
}

static void fft2(const cplx *in, cplx *out, int H, int W, int inverse)
{
    cplx *tmp = (cplx *)malloc(sizeof(cplx) * (size_t)H * W);
    dft_rows(in, tmp, H, W, inverse ? 1 : -1);
    dft_cols(tmp, out, H, W, inverse ? 1 : -1);
    if (inverse) for (size_t i = 0; i < (size_t)H * W; i++) out[i] /= ((double)H * W);
    free(tmp);
}

/* numpy argmax/max on complex: lexicographic (real, then imag), first occurrence */
static size_t cargmax(const cplx *a, size_t n)
{
    size_t b = 0;
    for (size_t i = 1; i < n; i++)
        if (creal(a[i]) > creal(a[b]) || (creal(a[i]) == creal(a[b]) && cimag(a[i]) > cimag(a[b]))) b = i;
    return b;
}

/* phase_correlate._dftups (phase_correlate.py:137-196) */
static void dftups(const cplx *data, int rows, int cols, int ur, int uc, int uf, double roff, double coff, cplx *out)
{
    cplx *ck = (cplx *)malloc(sizeof(cplx) * (size_t)cols * uc);
    cplx *rk = (cplx *)malloc(sizeof(cplx) * (size_t)ur * rows);
    cplx *t = (cplx *)malloc(sizeof(cplx) * (size_t)ur * cols);
    for (int c = 0; c < cols; c++) {
        double f = (double)((c + cols / 2) % cols) - floor(cols / 2.0);     /* ifftshift(arange) - floor(n/2) */
        for (int u = 0; u < uc; u++) ck[(size_t)c * uc + u] = cexp((-I * 2 * M_PI / (cols * uf)) * (f * (u - coff)));
    }
    for (int u = 0; u < ur; u++)
        for (int r = 0; r < rows; r++) {
            double f = (double)((r + rows / 2) % rows) - floor(rows / 2.0);
            rk[(size_t)u * rows + r] = cexp((-I * 2 * M_PI / (rows * uf)) * ((u - roff) * f));
        }
    for (int u = 0; u < ur; u++)
        for (int c = 0; c < cols; c++) {
            cplx s = 0;
            for (int r = 0; r < rows; r++) s += rk[(size_t)u * rows + r] * data[(size_t)r * cols + c];
            t[(size_t)u * cols + c] = s;
        }
    for (int u = 0; u < ur; u++)
        for (int v = 0; v < uc; v++) {
            cplx s = 0;
            for (int c = 0; c < cols; c++) s += t[(size_t)u * cols + c] * ck[(size_t)c * uc + v];
            out[(size_t)u * uc + v] = s;
        }
    free(ck); free(rk); free(t);
}

/* phase_correlate.phase_correlate (phase_correlate.py:11-134); out4 = row_shift, col_shift, error, diffphase */
int fsq_o_phase_correlate(const double *ref, const double *reg, int rows, int cols, int uf, double *out4)
{
    if (rows <= 0 || cols <= 0 || uf < 1) return FSQ_O_EINVAL;
    size_t N = (size_t)rows * cols;
    cplx *a = (cplx *)malloc(sizeof(cplx) * N), *F = (cplx *)malloc(sizeof(cplx) * N);
    cplx *G = (cplx *)malloc(sizeof(cplx) * N), *cc = (cplx *)malloc(sizeof(cplx) * N);
    if (!a || !F || !G || !cc) return FSQ_O_ENOMEM;
    for (size_t i = 0; i < N; i++) a[i] = ref[i];
    fft2(a, F, rows, cols, 0);
    for (size_t i = 0; i < N; i++) a[i] = reg[i];
    fft2(a, G, rows, cols, 0);
    for (size_t i = 0; i < N; i++) a[i] = F[i] * conj(G[i]);                 /* :71 */
    fft2(a, cc, rows, cols, 1);
    size_t am = cargmax(cc, N);
    double row_max = (double)(am / cols), col_max = (double)(am % cols);
    double mid_row = trunc(rows / 2.0), mid_col = trunc(cols / 2.0);        /* numpy.fix :75-76 */
    double row_shift = row_max > mid_row ? row_max - rows : row_max;
    double col_shift = col_max > mid_col ? col_max - cols : col_max;
    double error, diffphase;
    if (uf == 1) {                                                           /* :85-92 */
        double rf = 0, rg = 0;
        for (size_t i = 0; i < N; i++) { rf += creal(F[i] * conj(F[i])); rg += creal(G[i] * conj(G[i])); }
        rf /= (double)N; rg /= (double)N;
        cplx cm = cc[am];
        error = sqrt(fabs(creal(1.0 - cm * conj(cm) / (rg * rf))));
        diffphase = atan2(cimag(cm), creal(cm));
    } else {
        row_shift = nearbyint(row_shift * uf) / uf;                          /* :96-97 */
        col_shift = nearbyint(col_shift * uf) / uf;
        int up = (int)ceil(uf * 1.5);
        double dftshift = trunc(up / 2.0);
        cplx *u = (cplx *)malloc(sizeof(cplx) * (size_t)up * up);
        for (size_t i = 0; i < N; i++) a[i] = G[i] * conj(F[i]);
        dftups(a, rows, cols, up, up, uf, dftshift - row_shift * uf, dftshift - col_shift * uf, u);
        double norm = mid_row * mid_col * (double)uf * uf;
        for (int i = 0; i < up * up; i++) u[i] = conj(u[i]) / norm;
        size_t um = cargmax(u, (size_t)up * up);
        double rm = (double)(um / up) - dftshift, cm_ = (double)(um % up) - dftshift;
        row_shift = row_shift + rm / uf;
        col_shift = col_shift + cm_ / uf;
        cplx cmax = u[um], rg00, rf00;
        for (size_t i = 0; i < N; i++) a[i] = F[i] * conj(F[i]);
        dftups(a, rows, cols, 1, 1, uf, 0, 0, &rg00);
        for (size_t i = 0; i < N; i++) a[i] = G[i] * conj(G[i]);
        dftups(a, rows, cols, 1, 1, uf, 0, 0, &rf00);
        rg00 /= norm; rf00 /= norm;
        cplx e = 1.0 - cmax * conj(cmax) / (rg00 * rf00);
        error = sqrt(cabs(e));
        diffphase = atan2(cimag(cmax), creal(cmax));
        free(u);
        if (mid_row == 1) row_shift = 0;                                     /* :125-128 */
        if (mid_col == 1) col_shift = 0;
    }
    out4[0] = row_shift; out4[1] = col_shift; out4[2] = error; out4[3] = diffphase;
    free(a); free(F); free(G); free(cc);
    return 0;
}
