/*
 * CPU oracle, plain C: interpolation-point selection with EXACTLY the arithmetic order of the HIP
 * kernels (pyscf_isdf_amd/csrc/select_ip.hip), so that pivot lists and Cholesky rows can be compared
 * bit for bit on identical AO input.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Pivot rule and stopping tolerance: pyscf/lib/scipy_helper.py:71-110 (argmax of the residual
 * diagonal, L[k,k] = sqrt(D[k]), D -= L[:,k]^2, tol = n*eps*max diag when tol < 0) applied to the
 * implicit Gram matrix A(r,r') = (sum_mu ao[mu,r] ao[mu,r'])^2, plus the deterministic tie rule
 * (lowest index with d >= (1 - tie_rtol) * max).
 *
 * Arithmetic contract (one IEEE fma chain per quantity, ascending index):
 *   d0[i]  = s*s,  s = fma(ao[mu,i], ao[mu,i], s)  over mu
 *   s0     = fma(ao[mu,i], ao[mu,p], s0)           over mu
 *   col    = s0*s0;  col = fma(-L[t,i], L[t,p], col) over t < j
 *   row    = col / sqrt(d[p]);  d[i] = fma(-row, row, d[i]), clamped at 0
 * Build with -ffp-contract=off so the compiler adds no fusions of its own.
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

long oracle_select_ip_cplx(const double* ao, int nao, int nh, long m, long ld, int k, double tol,
                           double tie_rtol, long* piv, double* L, long ldL);

long oracle_select_ip(const double* ao, int nao, long m, long ld, int k, double tol,
                      double tie_rtol, long* piv, double* L, long ldL)
{
        return oracle_select_ip_cplx(ao, nao, 0, m, ld, k, tol, tie_rtol, piv, L, ldL);
}

/* nh > 0: complex mode (rows [0,nh) = Re u, [nh,2nh) = Im u): Gram entry = (sum X_i.pv)^2 + (sum X_i.pvr)^2,
 * pvr = [Im u_p; -Re u_p]; the two sums are separate ascending fma chains, col = fma(s1, s1, s0*s0). */
long oracle_select_ip_cplx(const double* ao, int nao, int nh, long m, long ld, int k, double tol,
                           double tie_rtol, long* piv, double* L, long ldL)
{
        double* d = (double*)malloc(sizeof(double) * m);
        double* pv = (double*)malloc(sizeof(double) * nao);
        double* pvr = (double*)malloc(sizeof(double) * nao);
        double* pl = (double*)malloc(sizeof(double) * (k > 0 ? k : 1));
        long i, rank = 0;
        int j, mu, t;
#pragma omp parallel for private(mu)
        for (i = 0; i < m; i++) {
                double s = 0.0;
                for (mu = 0; mu < nao; mu++) {
                        double v = ao[(long)mu * ld + i];
                        s = fma(v, v, s);
                }
                d[i] = s * s;
        }
        for (j = 0; j < k; j++) {
                double dmax = 0.0;
                for (i = 0; i < m; i++) if (d[i] > dmax) dmax = d[i];
                if (j == 0 && tol < 0) tol = (double)m * DBL_EPSILON * dmax;
                if (!(dmax > tol)) break;
                double thr = dmax * (1.0 - tie_rtol);
                long p = -1;
                for (i = 0; i < m; i++) if (d[i] >= thr && d[i] > 0.0) { p = i; break; }
                if (p < 0) break;
                piv[j] = p;
                double dp = sqrt(d[p]);
                for (mu = 0; mu < nao; mu++) pv[mu] = ao[(long)mu * ld + p];
                for (mu = 0; mu < nao; mu++) pvr[mu] = (nh > 0) ? ((mu < nh) ? pv[nh + mu] : -pv[mu - nh]) : 0.0;
                for (t = 0; t < j; t++) pl[t] = L[(long)t * ldL + p];
#pragma omp parallel for private(mu, t)
                for (i = 0; i < m; i++) {
                        double row, dnew;
                        double dold = d[i];
                        if (i == p) {
                                row = dp; dnew = -1.0;
                        } else if (dold < 0.0) {
                                row = 0.0; dnew = -1.0;
                        } else {
                                double s0 = 0.0, s1 = 0.0;
                                for (mu = 0; mu < nao; mu++) s0 = fma(ao[(long)mu * ld + i], pv[mu], s0);
                                if (nh > 0)
                                        for (mu = 0; mu < nao; mu++) s1 = fma(ao[(long)mu * ld + i], pvr[mu], s1);
                                double col = fma(s1, s1, s0 * s0);
                                for (t = 0; t < j; t++) col = fma(-L[(long)t * ldL + i], pl[t], col);
                                row = col / dp;
                                dnew = fma(-row, row, dold);
                                if (dnew < 0.0) dnew = 0.0;
                        }
                        L[(long)j * ldL + i] = row;
                        d[i] = dnew;
                }
                rank++;
        }
        free(d); free(pv); free(pvr); free(pl);
        return rank;
}
