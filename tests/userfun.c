/* The integrand of ttcross_amd/fortran/test_crs_user.f90 in C, with the reference's callback interface
 * fun(m, ind, n, par) (lib/dmrgg.f90:18): f = cos(sum x_i) / (1 + sum x_i^2), x_i = par[ind_i - 1].
 * Test infrastructure: handed to the ORACLE (ttxo_problem.user) and, through ttx_set_integrand_host, to the engine. */
#include <math.h>
#include <stdint.h>
double ttx_test_userfun(const int32_t *m, const int32_t *ind, const int32_t *n, const double *par)
{
    double s1 = 0.0, s2 = 0.0;
    (void)n;
    for (int i = 0; i < *m; i++) { const double x = par[ind[i] - 1]; s1 = s1 + x; s2 = s2 + x * x; }
    return cos(s1) / (1.0 + s2);
}

/* degenerate integrands for the robustness tests: NaN everywhere / NaN on part of the domain */
double ttx_test_userfun_nan(const int32_t *m, const int32_t *ind, const int32_t *n, const double *par)
{
    (void)m; (void)ind; (void)n; (void)par;
    return NAN;
}
double ttx_test_userfun_partnan(const int32_t *m, const int32_t *ind, const int32_t *n, const double *par)
{
    (void)n;
    if (ind[0] == 2 || ind[*m - 1] == 1) return NAN;
    return ttx_test_userfun(m, ind, n, par);
}
