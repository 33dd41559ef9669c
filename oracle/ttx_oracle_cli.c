/*
 * ttx_oracle_cli.c -- command-line twin of the reference drivers, on top of the TEST ORACLE.
 * (test infrastructure; see ttx_oracle.h).  Usage mirrors test_crs_ising.f90:25-29 / test_crs_mvn.f90:25-28:
 *     ttx_oracle ising KIND INDEX N RANK PIV [NPROC]
 *     ttx_oracle stdnorm D N RANK PIV [NPROC]
 *     ttx_oracle mvn D N RANK PIV [NPROC]
 * Prints the per-sweep lines of lib/dmrgg.f90:971-1008 and the footer of test_crs_ising.f90:156-169.
 */
#include "ttx_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Ising C_m values (Bailey, Borwein, Crandall, "Integrals of the Ising class"; the same published table
 * the reference driver embeds at test_crs_ising.f90:71-100), rounded to double */
static double ising_tru(char kind, int m)
{
    if (kind == 'c') switch (m) {
        case 2: return 1.0;
        case 3: return 0.78130241289648629687;
        case 4: return 0.70119986017642999982;
        case 5: return 0.66575980019993742832;
        case 6: return 0.64863420903100707526;
        case 8: return 0.63548402675916322614;
        case 16: return 0.63050394617323726351;
        case 32: return 0.63047350420733980638;
        case 64: return 0.63047350337438679649;
        case 128: case 256: case 512: case 1024: return 0.63047350337438679612;
    }
    if (kind == 'd') switch (m) {
        case 2: return 1.0 / 3;
        case 5: return 0.0024846057623403154800;
        case 6: return 0.00048914170018803477510;
    }
    if (kind == 'e') switch (m) {
        case 5: return 0.0034936537117295217407;
        case 6: return 0.00068783287182640943700;
    }
    return 0.0;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s ising KIND INDEX N RANK PIV [NPROC] | stdnorm|mvn D N RANK PIV [NPROC]\n", argv[0]); return 2; }
    int a = 2;
    char kind;
    if (!strcmp(argv[1], "ising")) kind = (char)(argv[a++][0] | 0x20);
    else if (!strcmp(argv[1], "stdnorm")) kind = 's';
    else if (!strcmp(argv[1], "mvn")) kind = 'm';
    else { fprintf(stderr, "unknown driver %s\n", argv[1]); return 2; }
    if (argc < a + 4) { fprintf(stderr, "too few arguments\n"); return 2; }
    int m = atoi(argv[a++]), n = atoi(argv[a++]), r = atoi(argv[a++]), piv = atoi(argv[a++]);
    int nproc = (argc > a) ? atoi(argv[a++]) : 1;
    if (n % 2 == 0) n++;                                   /* test_crs_ising.f90:40 */
    int d = (kind == 's' || kind == 'm') ? m : m - 1;
    double *par = (double *)calloc(2 * (size_t)n + 1, sizeof(double));
    double *qw = (double *)calloc((size_t)d * n, sizeof(double));
    double *aux = NULL;
    double tru, acc; int rescale;
    ttxo_driver_setup(kind, m, n, par, qw, &tru, &acc, &rescale);
    if (kind == 'c' || kind == 'd' || kind == 'e') tru = ising_tru(kind, m);
    int32_t *nn = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
    for (int i = 0; i < d; i++) nn[i] = n;
    ttxo_problem pb;
    memset(&pb, 0, sizeof pb);
    pb.d = d; pb.n = nn; pb.par = par; pb.npar = 2 * n + 1; pb.quadw = qw;
    pb.fun_id = (kind == 's') ? TTXO_FUN_STDNORM : (kind == 'm') ? TTXO_FUN_MVN : TTXO_FUN_ISING;
    if (kind == 'm') { aux = (double *)calloc((size_t)d + (size_t)d * d + 1, sizeof(double)); ttxo_mvn_init(d, 0.0, 1.0, aux); pb.aux = aux; pb.naux = d + d * d + 1; }
    pb.accuracy = acc; pb.maxrank = r; pb.piv = piv; pb.tru = tru; pb.has_tru = (tru != 0.0);
    pb.nproc = nproc; pb.verbose = 1;
    ttxo_result res;
    if (ttxo_dmrgg(&pb, &res)) return 1;
    printf("...with%12lld evaluations completed in %12.4E sec.\n", (long long)res.neval, res.seconds);
    printf("computed value: %.16e%s\n", res.value, rescale ? "  / 5**(m-1)" : "");
    if (tru != 0.0) {
        printf("analytic value: %.16e\n", tru);
        printf("correct digits:%7.2f\n", -log(fabs(1.0 - res.value / tru)) / log(10.0));
    }
    ttxo_free_result(&res);
    free(par); free(qw); free(nn); free(aux);
    return 0;
}
