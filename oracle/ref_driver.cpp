// TEST INFRASTRUCTURE ONLY: a C-callable driver over the reference's own CPU SuiteSparseQR,
// compiled by oracle/build_ref.sh into oracle/_ref/libspqr_ref.so together with the reference's
// sources (taken where they lie, unchanged).  Nothing of the product links or loads this.
//
// It runs the route a caller of the reference runs (SuiteSparseQR.cpp:59-260 ->
// spqr_1factor.cpp:624-641 -> spqr_analyze + spqr_factorize) and hands out what that route
// computed: the symbolic object's arrays (the pattern of R as SPQR holds it: Super/Rp/Rj,
// the tree, the row form S), R as a sparse matrix, the column permutation, the rank, the dead
// columns, and the reference's own timers and flop count.
#include "spqr.hpp"
#include <cstring>
#include <cstdlib>

struct RefQR
{
    cholmod_common cc ;
    SuiteSparseQR_factorization <double> *QR ;
    cholmod_sparse *R ;
    Long *E ;
    Long rank_R ;
    double t_analyze, t_factorize, flops, flops_bound, tol_used ;
    Long istat [10] ;
} ;

static cholmod_sparse *wrap (Long m, Long n, const Long *Ap, const Long *Ai, const double *Ax,
    cholmod_sparse *A)
{
    memset (A, 0, sizeof (cholmod_sparse)) ;
    A->nrow = m ; A->ncol = n ; A->nzmax = Ap [n] ;
    A->p = (void *) Ap ; A->i = (void *) Ai ; A->x = (void *) Ax ;
    A->itype = CHOLMOD_LONG ; A->xtype = CHOLMOD_REAL ; A->dtype = CHOLMOD_DOUBLE ;
    A->sorted = 0 ; A->packed = 1 ; A->stype = 0 ;
    return (A) ;
}

extern "C" {

// what: bit 0 = keep the factorization object (symbolic + numeric blocks);
//       bit 1 = also run [R,E] = qr(A) for R as a sparse matrix (a second factorization).
void *ref_open (Long m, Long n, const Long *Ap, const Long *Ai, const double *Ax,
    int ordering, double tol, int what)
{
    RefQR *h = (RefQR *) calloc (1, sizeof (RefQR)) ;
    if (!h) return (NULL) ;
    cholmod_l_start (&h->cc) ;
    cholmod_sparse Amat ;
    cholmod_sparse *A = wrap (m, n, Ap, Ai, Ax, &Amat) ;
    if (what & 1)
    {
        h->QR = spqr_1factor <double> (ordering, tol, 0, FALSE, A, 0, NULL, NULL, NULL, &h->cc) ;
        if (!h->QR) { cholmod_l_finish (&h->cc) ; free (h) ; return (NULL) ; }
        h->t_analyze = h->cc.SPQR_analyze_time ;
        h->t_factorize = h->cc.SPQR_factorize_time ;
        h->flops = h->cc.SPQR_flopcount ;
        h->flops_bound = h->cc.SPQR_flopcount_bound ;
        h->tol_used = h->cc.SPQR_tol_used ;
        for (int k = 0 ; k < 10 ; k++) h->istat [k] = h->cc.SPQR_istat [k] ;
    }
    if (what & 2)
    {
        h->rank_R = SuiteSparseQR <double> (ordering, tol, n, A, &h->R, &h->E, &h->cc) ;
        if (h->rank_R < 0 || !h->R)
        {
            if (h->QR) spqr_freefac (&h->QR, &h->cc) ;
            cholmod_l_finish (&h->cc) ; free (h) ; return (NULL) ;
        }
        if (!(what & 1))
        {
            h->t_analyze = h->cc.SPQR_analyze_time ;
            h->t_factorize = h->cc.SPQR_factorize_time ;
            h->flops = h->cc.SPQR_flopcount ;
            h->flops_bound = h->cc.SPQR_flopcount_bound ;
            h->tol_used = h->cc.SPQR_tol_used ;
            for (int k = 0 ; k < 10 ; k++) h->istat [k] = h->cc.SPQR_istat [k] ;
        }
    }
    return (h) ;
}

void ref_close (void *hv)
{
    RefQR *h = (RefQR *) hv ;
    if (!h) return ;
    if (h->QR) spqr_freefac (&h->QR, &h->cc) ;
    if (h->E) cholmod_l_free (h->R->ncol, sizeof (Long), h->E, &h->cc) ;
    if (h->R) cholmod_l_free_sparse (&h->R, &h->cc) ;
    cholmod_l_finish (&h->cc) ;
    free (h) ;
}

// out[0..15]: nf, m, n, anz, rjsize, rank, maxfrank, n1cols, n1rows, nnz(R sparse), rank from qr,
//             has E, has Rdead, ntasks, maxstack, istat[0]
void ref_info (void *hv, Long *out)
{
    RefQR *h = (RefQR *) hv ;
    for (int k = 0 ; k < 16 ; k++) out [k] = -1 ;
    if (h->QR)
    {
        spqr_symbolic *s = h->QR->QRsym ;
        spqr_numeric <double> *q = h->QR->QRnum ;
        out [0] = s->nf ; out [1] = s->m ; out [2] = s->n ; out [3] = s->anz ;
        out [4] = s->Rp [s->nf] ; out [5] = h->QR->rank ; out [6] = q->maxfrank ;
        out [7] = h->QR->n1cols ; out [8] = h->QR->n1rows ;
        out [12] = (q->Rdead != NULL) ; out [13] = q->ntasks ; out [14] = q->maxstack ;
    }
    if (h->R)
    {
        out [9] = ((Long *) h->R->p) [h->R->ncol] ; out [10] = h->rank_R ; out [11] = (h->E != NULL) ;
    }
    out [15] = h->istat [0] ;
}

// out[0..7]: analyze time, factorize time, flop count, flop bound, tol used, norm_E_fro
void ref_dinfo (void *hv, double *out)
{
    RefQR *h = (RefQR *) hv ;
    out [0] = h->t_analyze ; out [1] = h->t_factorize ; out [2] = h->flops ;
    out [3] = h->flops_bound ; out [4] = h->tol_used ;
    out [5] = h->QR ? h->QR->QRnum->norm_E_fro : h->cc.SPQR_norm_E_fro ;
    out [6] = 0 ; out [7] = 0 ;
}

// copies the named integer array of the symbolic object; returns its length, -1 if unknown
Long ref_get (void *hv, const char *name, Long *out)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->QR) return (-1) ;
    spqr_symbolic *s = h->QR->QRsym ;
    Long nf = s->nf, m = s->m, n = s->n ;
    const Long *src = NULL ; Long len = -1 ;
    if      (!strcmp (name, "Super"))  { src = s->Super ;  len = nf+1 ; }
    else if (!strcmp (name, "Rp"))     { src = s->Rp ;     len = nf+1 ; }
    else if (!strcmp (name, "Rj"))     { src = s->Rj ;     len = s->Rp [nf] ; }
    else if (!strcmp (name, "Parent")) { src = s->Parent ; len = nf+1 ; }
    else if (!strcmp (name, "Childp")) { src = s->Childp ; len = nf+2 ; }
    else if (!strcmp (name, "Child"))  { src = s->Child ;  len = nf+1 ; }
    else if (!strcmp (name, "Post"))   { src = s->Post ;   len = nf+1 ; }
    else if (!strcmp (name, "Fm"))     { src = s->Fm ;     len = nf+1 ; }
    else if (!strcmp (name, "Cm"))     { src = s->Cm ;     len = nf+1 ; }
    else if (!strcmp (name, "Sp"))     { src = s->Sp ;     len = m+1 ; }
    else if (!strcmp (name, "Sj"))     { src = s->Sj ;     len = s->anz ; }
    else if (!strcmp (name, "Sleft"))  { src = s->Sleft ;  len = n+2 ; }
    else if (!strcmp (name, "PLinv"))  { src = s->PLinv ;  len = m ; }
    else if (!strcmp (name, "Qfill"))  { src = s->Qfill ;  len = s->Qfill ? n : 0 ; }
    else if (!strcmp (name, "Q1fill")) { src = h->QR->Q1fill ; len = h->QR->Q1fill ? h->QR->nacols : 0 ; }
    else if (!strcmp (name, "P1inv"))  { src = h->QR->P1inv ;  len = h->QR->P1inv ? h->QR->narows : 0 ; }
    else if (!strcmp (name, "E"))      { src = h->E ;      len = h->E ? n : 0 ; }
    if (len > 0 && out && src) memcpy (out, src, len * sizeof (Long)) ;
    return (len) ;
}

Long ref_get_rdead (void *hv, char *out)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->QR || !h->QR->QRnum->Rdead) return (0) ;
    memcpy (out, h->QR->QRnum->Rdead, h->QR->QRsym->n) ;
    return (h->QR->QRsym->n) ;
}

// E of [R,E] = qr(A) when only that was run
Long ref_get_E (void *hv, Long *out)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->R || !h->E) return (0) ;
    memcpy (out, h->E, h->R->ncol * sizeof (Long)) ;
    return (h->R->ncol) ;
}

// R as compressed columns (nrow = what info[..] says through Rp's length n+1)
Long ref_get_R (void *hv, Long *Rp, Long *Ri, double *Rx)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->R) return (-1) ;
    Long n = h->R->ncol, nz = ((Long *) h->R->p) [n] ;
    memcpy (Rp, h->R->p, (n+1) * sizeof (Long)) ;
    memcpy (Ri, h->R->i, nz * sizeof (Long)) ;
    memcpy (Rx, h->R->x, nz * sizeof (double)) ;
    return (h->R->nrow) ;
}

// the packed block of front f as the numeric object holds it (column-packed upper trapezoid):
// returns the number of doubles between this block and the next one on the stack, or -1
Long ref_rblock_ptrs (void *hv, Long *offsets)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->QR) return (-1) ;
    spqr_numeric <double> *q = h->QR->QRnum ;
    if (q->ns != 1) return (-1) ;
    for (Long f = 0 ; f < q->nf ; f++)
        offsets [f] = q->Rblock [f] ? (Long) (q->Rblock [f] - q->Stacks [0]) : -1 ;
    return (q->Stack_size [0]) ;
}

Long ref_stack_copy (void *hv, Long from, Long count, double *out)
{
    RefQR *h = (RefQR *) hv ;
    if (!h->QR || h->QR->QRnum->ns != 1) return (-1) ;
    memcpy (out, h->QR->QRnum->Stacks [0] + from, count * sizeof (double)) ;
    return (count) ;
}

// threads MKL will use for the BLAS/LAPACK calls of spqr_front (what "cores" means for the CPU baseline)
extern "C" int MKL_Get_Max_Threads (void) ;
int ref_threads (void) { return (MKL_Get_Max_Threads ( )) ; }

// A'A ordering by the reference's METIS binding alone (what spqr_1colamd.cpp:561-608 calls for
// ordering 6 when there are no singletons): cholmod_l_metis on A' with postorder, as CHOLMOD's
// analyze does it.  Returns 0 on success.
int ref_metis_order (Long m, Long n, const Long *Ap, const Long *Ai, Long *perm)
{
    cholmod_common cc ;
    cholmod_l_start (&cc) ;
    cholmod_sparse Amat ;
    cholmod_sparse *A = wrap (m, n, Ap, Ai, NULL, &Amat) ;
    A->xtype = CHOLMOD_PATTERN ;
    cholmod_sparse *AT = cholmod_l_transpose (A, 0, &cc) ;
    int ok = AT && cholmod_l_metis (AT, NULL, 0, TRUE, perm, &cc) ;
    cholmod_l_free_sparse (&AT, &cc) ;
    cholmod_l_finish (&cc) ;
    return (ok ? 0 : 1) ;
}

}
