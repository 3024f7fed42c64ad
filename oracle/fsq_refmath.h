/* fsq_refmath.h - the elementary functions the reference reaches through numpy (TEST INFRASTRUCTURE).
 * With FSQ_ORACLE_LIBM they are the host libm's (what the reference used in the build container,
 * glibc 2.35); otherwise the oracle's own restatement of those algorithms (fsq_refmath.c), which
 * is checked bit-for-bit against libm in tests/ so the oracle gives identical results on any host. */
#ifndef FSQ_REFMATH_H
#define FSQ_REFMATH_H
#include <math.h>
#define FSQ_PI 3.141592653589793
#ifdef FSQ_ORACLE_LIBM
static inline double fsq_ref_exp(double x) { return exp(x); }
/* through volatile pointers: gcc otherwise fuses sin+cos into sincos(), whose results differ
 * from sin()/cos() in ~0.14% of arguments (measured, glibc 2.35) */
static double (*volatile fsq_p_sin)(double) = sin;
static double (*volatile fsq_p_cos)(double) = cos;
static inline double fsq_ref_sin(double x) { return fsq_p_sin(x); }
static inline double fsq_ref_cos(double x) { return fsq_p_cos(x); }
/* through a volatile pointer as well: gcc folds pow(x, 2.0) into x*x, which is NOT what glibc's
 * pow returns for ~0.08% of arguments */
static double (*volatile fsq_p_pow)(double, double) = pow;
static inline double fsq_ref_pow(double x, double y) { return fsq_p_pow(x, y); }
static inline double fsq_ref_pow2(double x) { return fsq_p_pow(x, 2.0); }
#else
double fsq_ref_exp(double x);
double fsq_ref_sin(double x);
double fsq_ref_cos(double x);
double fsq_ref_pow(double x, double y);
double fsq_ref_pow2(double x);
#endif
#endif
