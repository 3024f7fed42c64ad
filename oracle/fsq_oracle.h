/* fsq_oracle.h - C interface of the CPU oracle (TEST INFRASTRUCTURE, see fsq_oracle.c). */
#ifndef FSQ_ORACLE_H
#define FSQ_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FSQ_O_EINVAL  (-1)   /* reference raises ValueError */
#define FSQ_O_ENOMEM  (-2)
#define FSQ_O_ERANGE  (-3)   /* outside the restatement's exactness domain */
#define FSQ_O_EASSERT (-4)   /* reference's assert at pflib.py:518 would fire */

#define FSQ_O_MODE_REF      0   /* bug-faithful: qrsolv solution aliases diag(R) */
#define FSQ_O_MODE_TEXTBOOK 1   /* same code with the diagonal copied (MINPACK behaviour) */

typedef struct {
    double p[7];      /* mpfit parameter order: H(eight), A(mplitude), p2, p3, s4, s5, theta[deg] */
    double fnorm;     /* mpfit.fnorm (summed squared residuals) */
    int32_t status, niter, nfev, pad;
} FsqOFit;

typedef struct {      /* one fitted candidate in pflib's tuple order (pflib.py:475) */
    double h0, w0, H, A, sigma_h, sigma_w, theta, rmse, r2, s_n;
    int32_t h, w;     /* candidate pixel */
} FsqORow;

int fsq_o_candidates(const uint16_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                     double c_std, int32_t *hw_out, int cap, int64_t *cm_out, double *thr_out);
int fsq_o_fit_roi(const int64_t *roi25, int mode, FsqOFit *out);
void fsq_o_model(const double *p7, double *g25);
void fsq_o_qrfac(double *a, int m, int n, int pivot, int *ipvt, double *rdiag, double *acnorm);
double fsq_o_illumina_s_n(const int64_t *roi25);
void fsq_o_set_force_norm_recompute(int on);
void fsq_o_fit_metrics(const int64_t *roi25, const double *p7, int h, int w, FsqORow *row);
int fsq_o_consolidate(const FsqORow *rows, int n, int H, int W, double r2_thr, int radius, int py2,
                      int32_t *keep_idx, int32_t *key_hw);
/* whole field: candidates -> fits (n_threads OpenMP threads) -> metrics -> consolidation.
 * rows_out[cap] receives ALL fitted candidates in raster order, fits_out[cap] their solver traces
 * (may be NULL); keep_idx/key_hw[cap] the consolidated table.  *n_cand, *n_keep are set. */
int fsq_o_find_peptides(const uint16_t *img, int H, int W, int med_size, const int64_t *K, int ksz,
                        double c_std, double r2_thr, int radius, int mode, int n_threads,
                        FsqORow *rows_out, FsqOFit *fits_out, int32_t *keep_idx, int32_t *key_hw,
                        int cap, int32_t *n_cand, int32_t *n_keep);
/* fit n ROIs (uint16[n][25]) with n_threads threads: CPU-baseline leg of bench.py */
int fsq_o_fit_rois_u16(const uint16_t *rois, int n, int mode, int n_threads, FsqOFit *out);
/* phase_correlate.phase_correlate: out4 = (row_shift, col_shift, error, diffphase) */
int fsq_o_phase_correlate(const double *ref, const double *reg, int rows, int cols, int upsample, double *out4);
double fsq_o_mexican_hat(const uint16_t *img, int H, int W, int h, int w, int brim, int radius);
double fsq_o_enorm(const double *x, int n, int inc);
double fsq_o_pairwise_sum(const double *a, long n);
double fsq_o_numpy_sum(const double *a, long n);
#ifdef __cplusplus
}
#endif
#endif
