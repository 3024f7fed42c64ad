// Host-side check of csrc/fsq_x87.h (the integer restatement of OpenBLAS' x87 dnrm2 used by the tracking kernel)
// against the same chain in `long double`.  Built and run by tests/test_tracking.py.
#include "../fluorosequencingimageanalysis_amd/csrc/fsq_x87.h"
#include <cmath>
#include <cstdio>
#include <random>
static double ref(double dh, double dw)
{
    volatile long double a = (long double)dh * dh, b = (long double)dw * dw, s = a + b, r = sqrtl(s);
    return (double)r;
}
int main()
{
    std::mt19937_64 g(1);
    long bad = 0, n = 0, plain = 0;
    auto chk = [&](double a, double b) {
        const double x = fsq_dnrm2_2(a, b), y = ref(a, b);
        n++;
        if (x != y) bad++;
        if (y != std::sqrt(a * a + b * b)) plain++;
    };
    std::uniform_real_distribution<double> U(-3, 3), V(-1000, 1000);
    std::uniform_int_distribution<int> I(-60, 60), K(-300, 300);
    for (long i = 0; i < 500000; i++) {
        chk(I(g) / 20.0, I(g) / 20.0); chk(U(g), U(g)); chk(V(g), U(g) * 1e-3);
        chk(std::ldexp(U(g), K(g)), std::ldexp(U(g), K(g)));
    }
    chk(0, 0); chk(0, 1.5); chk(3, 4); chk(-0.0, 2); chk(1e150, 1e150);
    for (int i = 0; i < 64; i++) { const double a = std::ldexp(1.0, i) - 1; chk(a, 0); chk(a, a); chk(a, 1); }
    std::printf("n=%ld bad=%ld differs_from_plain_double=%ld\n", n, bad, plain);
    return bad != 0;
}
