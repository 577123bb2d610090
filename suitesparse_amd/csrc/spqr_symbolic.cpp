// Stand-alone symbolic front end for the GPU QR path.
//
// On a machine with SuiteSparse the arrays below come from spqr_analyze
// (SPQR/Source/spqr_analyze.cpp:75-1130, which obtains the supernodes from
// CHOLMOD).  This file produces the same arrays - same names, same meaning,
// same conventions (placeholder front nf, Parent[nf] = -1, Sleft of size n+2,
// front-local Sj, Fm/Cm without rank detection) - from A and a column
// permutation alone, so that the numeric GPU path can run and be tested where
// only this repository exists.  Two modes.  relax < 0: the arrays SPQR itself would
// hold for the column order Quser (spqr_analyze called with SPQR_ORDERING_GIVEN, the
// call spqr_1factor.cpp:624-641 makes after spqr_1colamd has ordered the columns):
// no further postordering of the columns, fundamental supernodes and CHOLMOD's relaxed
// amalgamation with its default parameters (CHOLMOD/Supernodal/cholmod_super_symbolic.c:
// 590-760, nrelax 4/16/48, zrelax 0.8/0.1/0.05), the frontal tree postordered by front
// width as spqr_analyze.cpp:561-580 does through cholmod_postorder; Super/Rp/Rj/Parent/
// Post/Fm/Cm are then identical to an SPQR run's (tests/test_reference_parity_cpu.py
// compares them with the reference compiled from its own sources).  relax >= 0: the
// columns are postordered here, the partition is the fundamental one, optionally relaxed
// by merging a front into its parent when that adds at most relax percent explicit
// zeros; front counts then differ from an SPQR run while every array stays
// self-consistent.

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <atomic>
#include <algorithm>
#include <new>

#include "../../include/spqrgpu_amd.h"
#include "spqr_symbolic.h"

typedef int64_t Long ;

// -----------------------------------------------------------------------------
// column elimination tree of A(:,Q) (the etree of (AQ)'(AQ)), Liu's algorithm
// with path compression, run on the rows' "previous column" links
// -----------------------------------------------------------------------------

static void column_etree (Long m, Long n, const Long *Ap, const Long *Ai, const Long *Q, Long *parent)
{
    std::vector<Long> anc (n, -1), prev (m, -1) ;
    for (Long k = 0 ; k < n ; k++)
    {
        parent [k] = -1 ;
        const Long col = Q ? Q [k] : k ;
        for (Long p = Ap [col] ; p < Ap [col+1] ; p++)
        {
            const Long row = Ai [p] ;
            Long i = prev [row] ;
            while (i != -1 && i < k)
            {
                Long next = anc [i] ;
                anc [i] = k ;
                if (next == -1) parent [i] = k ;
                i = next ;
            }
            prev [row] = k ;
        }
    }
}

// postorder of a forest given by parent[]; children visited in increasing order
static void tree_postorder (Long n, const Long *parent, Long *post)
{
    std::vector<Long> head (n, -1), next (n, -1), stack ;
    for (Long j = n-1 ; j >= 0 ; j--)
    {
        if (parent [j] != -1) { next [j] = head [parent [j]] ; head [parent [j]] = j ; }
    }
    Long k = 0 ;
    for (Long root = 0 ; root < n ; root++)
    {
        if (parent [root] != -1) continue ;
        stack.push_back (root) ;
        while (!stack.empty ( ))
        {
            Long j = stack.back ( ) ;
            Long c = head [j] ;
            if (c == -1) { post [k++] = j ; stack.pop_back ( ) ; }
            else { head [j] = next [c] ; stack.push_back (c) ; }
        }
    }
}

// Postorder of a forest with the children of a node visited in increasing order of weight,
// ties in increasing node number; weights are clamped to [0, n-1] as CHOLMOD does
// (CHOLMOD/Cholesky/cholmod_postorder.c:176-262): bucket lists by weight, filled in node order
// and emptied from the heaviest bucket down, each node pushed on the front of its parent's list.
static void weighted_postorder (Long n, const Long *parent, const Long *weight, Long *post)
{
    std::vector<Long> head (n, -1), next (n, -1), whead (n, -1), stack ;
    for (Long j = 0 ; j < n ; j++)
    {
        const Long p = parent [j] ;
        if (p < 0 || p >= n) continue ;
        Long w = std::min (std::max (weight [j], (Long) 0), n - 1) ;
        next [j] = whead [w] ;
        whead [w] = j ;
    }
    for (Long w = n-1 ; w >= 0 ; w--)
    {
        for (Long j = whead [w], nj ; j != -1 ; j = nj)
        {
            nj = next [j] ;
            next [j] = head [parent [j]] ;
            head [parent [j]] = j ;
        }
    }
    Long k = 0 ;
    for (Long root = 0 ; root < n ; root++)
    {
        if (parent [root] != -1) continue ;
        stack.push_back (root) ;
        while (!stack.empty ( ))
        {
            const Long j = stack.back ( ), c = head [j] ;
            if (c == -1) { post [k++] = j ; stack.pop_back ( ) ; }
            else { head [j] = next [c] ; stack.push_back (c) ; }
        }
    }
}

// -----------------------------------------------------------------------------
// the analysis
// -----------------------------------------------------------------------------

extern "C" spqr_amd_symbolic *spqr_amd_analyze
(
    int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *Quser,
    int relax, size_t gpuMemorySize, int *status
)
{
    return spqr_amd_analyze_rd (m, n, Ap, Ai, Quser, relax, gpuMemorySize, 0, status) ;
}

extern "C" spqr_amd_symbolic *spqr_amd_analyze_rd
(
    int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *Quser,
    int relax, size_t gpuMemorySize, int do_rank_detection, int *status
)
{
    int dummy ;
    if (!status) status = &dummy ;
    *status = 0 ;
    if (m < 0 || n < 0 || !Ap || (Ap [n] > 0 && !Ai)) { *status = 3 ; return NULL ; }
    spqr_amd_symbolic *S = NULL ;
    try
    {
        if (Quser)
        {
            // a permutation of 0..n-1, or nothing is done
            std::vector<char> seen (n, 0) ;
            for (Long k = 0 ; k < n ; k++)
            {
                const Long c = Quser [k] ;
                if (c < 0 || c >= n || seen [c]) { *status = 3 ; return NULL ; }
                seen [c] = 1 ;
            }
        }
        S = new spqr_amd_symbolic ( ) ;
        { static std::atomic<uint64_t> next_uid (1) ; S->uid = next_uid.fetch_add (1) ; }
        S->m = m ; S->n = n ; S->anz = Ap [n] ;
        S->do_rank_detection = do_rank_detection ? 1 : 0 ;
        const Long anz = Ap [n] ;

        // ---- etree of A(:,Quser), postorder, composed permutation ----------------------
        std::vector<Long> parent0 (n), post (n) ;
        column_etree (m, n, Ap, Ai, Quser, parent0.data ( )) ;
        tree_postorder (n, parent0.data ( ), post.data ( )) ;
        const bool cholmod_mode = (relax < 0) ;
        std::vector<Long> ipost (n), parent (n) ;
        for (Long k = 0 ; k < n ; k++) ipost [post [k]] = k ;
        S->Qfill.resize (n) ;
        if (cholmod_mode)
        {
            // the columns stay where the caller put them; post is used for the counts only
            for (Long k = 0 ; k < n ; k++) S->Qfill [k] = Quser ? Quser [k] : k ;
            parent = parent0 ;
        }
        else
        {
            for (Long k = 0 ; k < n ; k++)
            {
                S->Qfill [k] = Quser ? Quser [post [k]] : post [k] ;
                Long p = parent0 [post [k]] ;
                parent [k] = (p == -1) ? -1 : ipost [p] ;
            }
            for (Long k = 0 ; k < n ; k++) post [k] = k ;      // the columns are in postorder now
        }
        std::vector<Long> Qinv (n) ;
        for (Long k = 0 ; k < n ; k++) Qinv [S->Qfill [k]] = k ;

        // ---- S = A(P,Q) in row form, rows sorted by leftmost column -------------------
        // (spqr_stranspose1: SPQR/Source/spqr_stranspose1.cpp)
        std::vector<Long> leftmost (m, n), rowcount (m, 0) ;
        for (Long k = 0 ; k < n ; k++)
        {
            const Long col = S->Qfill [k] ;
            for (Long p = Ap [col] ; p < Ap [col+1] ; p++)
            {
                const Long i = Ai [p] ;
                if (i < 0 || i >= m) { *status = 3 ; delete S ; return NULL ; }
                if (leftmost [i] == n) leftmost [i] = k ;
                rowcount [i]++ ;
            }
        }
        S->Sleft.assign (n+2, 0) ;
        // Sleft[k] = first row of S whose leftmost column is k; Sleft[n] = # nonempty rows,
        // Sleft[n+1] = m (empty rows of A go last)
        for (Long i = 0 ; i < m ; i++) S->Sleft [leftmost [i] + 1]++ ;
        for (Long k = 0 ; k <= n ; k++) S->Sleft [k+1] += S->Sleft [k] ;
        S->PLinv.resize (m) ;
        {
            std::vector<Long> nextrow (S->Sleft.begin ( ), S->Sleft.begin ( ) + n + 1) ;
            for (Long i = 0 ; i < m ; i++) S->PLinv [i] = nextrow [leftmost [i]]++ ;
        }
        S->Sp.assign (m+1, 0) ;
        for (Long i = 0 ; i < m ; i++) S->Sp [S->PLinv [i] + 1] = rowcount [i] ;
        for (Long i = 0 ; i < m ; i++) S->Sp [i+1] += S->Sp [i] ;
        S->Sj.resize (anz) ;
        {
            std::vector<Long> w (S->Sp.begin ( ), S->Sp.begin ( ) + m) ;
            for (Long k = 0 ; k < n ; k++)
            {
                const Long col = S->Qfill [k] ;
                for (Long p = Ap [col] ; p < Ap [col+1] ; p++) S->Sj [w [S->PLinv [Ai [p]]]++] = k ;
            }
        }

        // ---- column counts of R (rows of R = columns of chol((AQ)'(AQ))) -----------------
        // skeleton/least-common-ancestor counting (Gilbert, Ng, Peyton), each row of S
        // visited at its leftmost column
        std::vector<Long> cc (n, 0) ;
        {
            std::vector<Long> first (n, -1), maxfirst (n, -1), prevleaf (n, -1), anc (n) ;
            for (Long k = 0 ; k < n ; k++)
            {
                // first[j] = postorder rank of the first descendant of j
                const Long j = post [k] ;
                if (first [j] == -1) { first [j] = k ; cc [j] = 1 ; }
                Long p = parent [j] ;
                if (p != -1 && first [p] == -1) first [p] = first [j] ;
                anc [j] = j ;
            }
            for (Long k = 0 ; k < n ; k++)
            {
                const Long j = post [k] ;
                if (parent [j] != -1) cc [parent [j]]-- ;
                for (Long row = S->Sleft [j] ; row < S->Sleft [j+1] ; row++)
                {
                    for (Long p = S->Sp [row] ; p < S->Sp [row+1] ; p++)
                    {
                        const Long i = S->Sj [p] ;
                        if (i <= j || first [j] <= maxfirst [i]) continue ;
                        maxfirst [i] = first [j] ;
                        const Long jprev = prevleaf [i] ;
                        prevleaf [i] = j ;
                        cc [j]++ ;
                        if (jprev != -1)
                        {
                            Long q = jprev ;
                            while (q != anc [q]) q = anc [q] ;
                            for (Long s = jprev ; s != q ; ) { Long sp = anc [s] ; anc [s] = q ; s = sp ; }
                            cc [q]-- ;
                        }
                    }
                }
                if (parent [j] != -1) anc [j] = parent [j] ;
            }
            for (Long k = 0 ; k < n ; k++) { const Long j = post [k] ; if (parent [j] != -1) cc [parent [j]] += cc [j] ; }
        }
        S->ColCount = cc ;

        // ---- fundamental supernodes --------------------------------------------------
        std::vector<Long> nchild (n, 0) ;
        for (Long j = 0 ; j < n ; j++) if (parent [j] != -1) nchild [parent [j]]++ ;
        std::vector<Long> super ;
        for (Long j = 0 ; j < n ; j++)
        {
            bool join = (j > 0 && parent [j-1] == j && nchild [j] == 1 && cc [j] == cc [j-1] - 1) ;
            if (!join) super.push_back (j) ;
        }
        super.push_back (n) ;
        Long nf = (Long) super.size ( ) - 1 ;

        // front tree
        std::vector<Long> col2front (n) ;
        for (Long f = 0 ; f < nf ; f++) for (Long j = super [f] ; j < super [f+1] ; j++) col2front [j] = f ;
        std::vector<Long> fparent (nf) ;
        for (Long f = 0 ; f < nf ; f++)
        {
            Long p = parent [super [f+1] - 1] ;
            fparent [f] = (p == -1) ? nf : col2front [p] ;
        }

        // ---- CHOLMOD's relaxed amalgamation (cholmod_super_symbolic.c:676-790): supernode s+1 is
        // merged into s when s+1 is the current parent of s and either the merged node has at most
        // nrelax0 columns, or no zeros are added, or the fraction z of explicit zeros in the merged
        // node passes the size-dependent thresholds ------------------------------------------
        if (cholmod_mode && nf > 1)
        {
            const Long nrelax0 = 4, nrelax1 = 16, nrelax2 = 48 ;
            const double zrelax0 = 0.8, zrelax1 = 0.1, zrelax2 = 0.05 ;
            std::vector<Long> merged (nf, -1), nscol (nf), zeros (nf, 0), snz (nf) ;
            std::vector<Long> sparent (nf) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                nscol [f] = super [f+1] - super [f] ;
                snz [f] = cc [super [f]] ;
                sparent [f] = (fparent [f] >= nf) ? -1 : fparent [f] ;
            }
            for (Long s = nf-2 ; s >= 0 ; s--)
            {
                if (sparent [s] == -1) continue ;
                Long ss, snext ;
                for (ss = sparent [s] ; merged [ss] != -1 ; ss = merged [ss]) ;
                const Long cur = ss ;
                for (ss = sparent [s] ; merged [ss] != -1 ; ss = snext) { snext = merged [ss] ; merged [ss] = cur ; }
                if (cur != s+1) continue ;
                const Long nscol0 = nscol [s], nscol1 = nscol [s+1], ns = nscol0 + nscol1 ;
                Long totzeros = zeros [s+1] ;
                const double lnz1 = (double) snz [s+1] ;
                bool merge ;
                if (ns <= nrelax0)
                {
                    merge = true ;
                }
                else
                {
                    const double lnz0 = (double) snz [s] ;
                    const double xnewzeros = (double) nscol0 * (lnz1 + (double) nscol0 - lnz0) ;
                    const Long newzeros = nscol0 * (snz [s+1] + nscol0 - snz [s]) ;
                    if (xnewzeros == 0)
                    {
                        merge = true ;
                    }
                    else
                    {
                        const double xtotzeros = (double) totzeros + xnewzeros ;
                        const double xns = (double) ns ;
                        const double xtotsize = (xns * (xns+1) / 2) + xns * (lnz1 - (double) nscol1) ;
                        const double z = xtotzeros / xtotsize ;
                        totzeros += newzeros ;
                        merge = ((ns <= nrelax1 && z < zrelax0) || (ns <= nrelax2 && z < zrelax1) || (z < zrelax2))
                            && (xtotsize < (double) INT64_MAX / sizeof (double)) ;
                    }
                }
                if (merge)
                {
                    zeros [s] = totzeros ;
                    merged [s+1] = s ;
                    snz [s] = nscol0 + snz [s+1] ;
                    nscol [s] += nscol [s+1] ;
                }
            }
            std::vector<Long> newsuper ;
            for (Long f = 0 ; f < nf ; f++) if (merged [f] == -1) newsuper.push_back (super [f]) ;
            newsuper.push_back (n) ;
            const Long nf2 = (Long) newsuper.size ( ) - 1 ;
            super.swap (newsuper) ;
            nf = nf2 ;
            for (Long f = 0 ; f < nf ; f++) for (Long j = super [f] ; j < super [f+1] ; j++) col2front [j] = f ;
            fparent.assign (nf, nf) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const Long p = parent [super [f+1] - 1] ;
                fparent [f] = (p == -1) ? nf : col2front [p] ;
            }
        }

        // ---- relaxed amalgamation: merge a front into its parent when it is the parent's
        // last child in column order (so the pivot ranges stay contiguous) and the merged
        // front gains at most `relax` percent explicit zeros ----------------------------
        if (!cholmod_mode && relax > 0 && nf > 1)
        {
            // sizes: fp[f], fn[f] = cc of first column.  Merging child c (last columns end at
            // super[p]) into parent p gives fp = fpc + fpp, fn = fpc + fnp (needs struct(c) minus
            // pivots to be inside struct(p), true for any etree child), zeros added =
            // fpc * (fpc + fnp) - (entries of the child's R block).
            std::vector<Long> fp (nf), fnn (nf), rnz (nf) ;
            std::vector<char> dead (nf, 0) ;
            const Long big_pivots = 256 ;
            for (Long f = 0 ; f < nf ; f++)
            {
                fp [f] = super [f+1] - super [f] ;
                fnn [f] = cc [super [f]] ;
                rnz [f] = fp [f] * fnn [f] - fp [f] * (fp [f] - 1) / 2 ;
            }
            // process children-first; a child is adjacent to its parent iff super[c+1] == super[p]
            // where c is the previous live front
            std::vector<Long> firstcol (super.begin ( ), super.end ( ) - 1) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const Long p = fparent [f] ;
                if (p >= nf) continue ;
                if (super [f+1] != firstcol [p]) continue ;            // not the adjacent child
                const Long mfp = fp [f] + fp [p] ;
                const Long mfn = fp [f] + fnn [p] ;
                const Long mrnz = mfp * mfn - mfp * (mfp - 1) / 2 ;
                const Long old = rnz [f] + rnz [p] ;
                const bool small = (mfp <= 4) ;
                // Two fronts that are each wide enough to fill the device gain nothing from becoming
                // one, and the tree loses a branch that could have run beside its siblings (on another
                // GPU, or in the same launches): they stay apart whatever the zero count says.
                const bool both_big = (fp [f] >= big_pivots && fp [p] >= big_pivots) ;
                if (both_big) continue ;
                if (small || (double) (mrnz - old) <= (double) relax * 0.01 * (double) mrnz)
                {
                    dead [f] = 1 ;
                    firstcol [p] = firstcol [f] ;
                    fp [p] = mfp ; fnn [p] = mfn ; rnz [p] = mrnz ;
                    // f's children become p's children
                    // (resolved below through the dead chain)
                }
            }
            std::vector<Long> newsuper, newid (nf, -1) ;
            for (Long f = 0 ; f < nf ; f++) if (!dead [f]) { newid [f] = (Long) newsuper.size ( ) ; newsuper.push_back (firstcol [f]) ; }
            newsuper.push_back (n) ;
            const Long nf2 = (Long) newsuper.size ( ) - 1 ;
            std::vector<Long> np (nf2) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                if (dead [f]) continue ;
                Long p = fparent [f] ;
                while (p < nf && dead [p]) p = fparent [p] ;
                np [newid [f]] = (p >= nf) ? nf2 : newid [p] ;
            }
            super.swap (newsuper) ;
            fparent.swap (np) ;
            nf = nf2 ;
            for (Long f = 0 ; f < nf ; f++) for (Long j = super [f] ; j < super [f+1] ; j++) col2front [j] = f ;
        }

        S->nf = nf ;
        S->Super = super ;
        S->Parent.assign (nf+1, -1) ;
        for (Long f = 0 ; f < nf ; f++) S->Parent [f] = fparent [f] ;
        S->Parent [nf] = -1 ;
        S->Childp.assign (nf+2, 0) ;
        for (Long f = 0 ; f < nf ; f++) S->Childp [fparent [f] + 1]++ ;
        for (Long f = 0 ; f <= nf ; f++) S->Childp [f+1] += S->Childp [f] ;
        S->Child.assign (nf+1, 0) ;
        {
            std::vector<Long> w (S->Childp.begin ( ), S->Childp.begin ( ) + nf + 1) ;
            for (Long f = 0 ; f < nf ; f++) S->Child [w [fparent [f]]++] = f ;
        }
        S->Post.resize (nf+1) ;
        for (Long f = 0 ; f <= nf ; f++) S->Post [f] = f ;     // fronts are numbered in postorder
        // (in the CHOLMOD mode Post and the child lists are redone below, once the widths are known)

        // ---- front patterns Rp, Rj ------------------------------------------------------
        S->Rp.assign (nf+1, 0) ;
        S->Rj.clear ( ) ;
        {
            std::vector<Long> mark (n, -1), cols ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const Long col1 = super [f], col2 = super [f+1] ;
                cols.clear ( ) ;
                for (Long j = col1 ; j < col2 ; j++) mark [j] = f ;
                for (Long row = S->Sleft [col1] ; row < S->Sleft [col2] ; row++)
                    for (Long p = S->Sp [row] ; p < S->Sp [row+1] ; p++)
                    {
                        const Long j = S->Sj [p] ;
                        if (mark [j] != f) { mark [j] = f ; cols.push_back (j) ; }
                    }
                for (Long cp = S->Childp [f] ; cp < S->Childp [f+1] ; cp++)
                {
                    const Long c = S->Child [cp] ;
                    const Long fpc = super [c+1] - super [c] ;
                    for (Long p = S->Rp [c] + fpc ; p < S->Rp [c+1] ; p++)
                    {
                        const Long j = S->Rj [p] ;
                        if (mark [j] != f) { mark [j] = f ; cols.push_back (j) ; }
                    }
                }
                std::sort (cols.begin ( ), cols.end ( )) ;
                for (Long j = col1 ; j < col2 ; j++) S->Rj.push_back (j) ;
                S->Rj.insert (S->Rj.end ( ), cols.begin ( ), cols.end ( )) ;
                S->Rp [f+1] = (Long) S->Rj.size ( ) ;
            }
        }
        S->rjsize = (Long) S->Rj.size ( ) ;

        if (cholmod_mode)
        {
            // the frontal tree postordered with the children of a front in increasing order of
            // their width, the placeholder root last; child lists in that order
            // (spqr_analyze.cpp:561-611)
            std::vector<Long> weight (nf+1) ;
            for (Long f = 0 ; f < nf ; f++) weight [f] = S->Rp [f+1] - S->Rp [f] ;
            weight [nf] = 1 ;
            weighted_postorder (nf+1, S->Parent.data ( ), weight.data ( ), S->Post.data ( )) ;
            std::vector<Long> w (S->Childp.begin ( ), S->Childp.begin ( ) + nf + 1) ;
            for (Long kf = 0 ; kf < nf ; kf++)
            {
                const Long c = S->Post [kf] ;
                S->Child [w [S->Parent [c]]++] = c ;
            }
        }

        // ---- Fm, Cm, map offsets, front-local Sj, flop count ----------------------------
        // (spqr_analyze.cpp:646-902 with do_rank_detection = false)
        S->Fm.assign (nf+1, 0) ; S->Cm.assign (nf+1, 0) ;
        S->RjmapOffsets.assign (nf+1, 0) ; S->RimapOffsets.assign (nf+1, 0) ;
        S->RjmapSize = 0 ; S->RimapSize = 0 ;
        S->maxfn = 0 ; S->flopcount = 0 ; S->rnz = 0 ;
        {
            std::vector<Long> Fmap (n, -1), Stair ;
            for (Long kf = 0 ; kf < nf ; kf++)
            {
                const Long f = S->Post [kf] ;
                const Long col1 = super [f], col2 = super [f+1] ;
                const Long p1 = S->Rp [f], p2 = S->Rp [f+1] ;
                const Long fp = col2 - col1, fn = p2 - p1 ;
                S->maxfn = std::max (S->maxfn, fn) ;
                for (Long p = p1 ; p < p2 ; p++) Fmap [S->Rj [p]] = p - p1 ;
                Stair.assign (fn, 0) ;
                for (Long j = 0 ; j < fp ; j++) Stair [j] = S->Sleft [col1+j+1] - S->Sleft [col1+j] ;
                for (Long cp = S->Childp [f] ; cp < S->Childp [f+1] ; cp++)
                {
                    const Long c = S->Child [cp] ;
                    const Long fpc = super [c+1] - super [c] ;
                    const Long cm = S->Cm [c] ;
                    for (Long ci = 0 ; ci < cm ; ci++) Stair [Fmap [S->Rj [S->Rp [c] + fpc + ci]]]++ ;
                }
                Long fm = 0 ;
                for (Long j = 0 ; j < fn ; j++) { fm += Stair [j] ; Stair [j] = fm ; }
                S->Fm [f] = fm ;
                const Long rm = std::min (fm, fp) ;
                const Long cn = fn - fp ;
                // with rank detection a front may pass up as many rows as it has (pivot columns can
                // die and leave their rows to the contribution block): spqr_analyze.cpp:826-842
                S->Cm [f] = do_rank_detection ? std::min (fm, cn) : std::min (std::max (fm - rm, (Long) 0), cn) ;
                S->rnz += rm*(rm+1)/2 + rm*(fn-rm) ;
                S->RjmapOffsets [f] = S->RjmapSize ; S->RjmapSize += cn ;
                S->RimapOffsets [f] = S->RimapSize ; S->RimapSize += S->Cm [f] ;
                for (Long row = S->Sleft [col1] ; row < S->Sleft [col2] ; row++)
                    for (Long p = S->Sp [row] ; p < S->Sp [row+1] ; p++) S->Sj [p] = Fmap [S->Sj [p]] ;
                for (Long j = 0 ; j < fn ; j++)
                {
                    Long t = std::min (std::max (j+1, Stair [j]), fm) ;
                    if (t > j)
                    {
                        double h = (double) (t - j) ;
                        S->flopcount += 3*h + 4*h*(double) (fn-j-1) ;
                    }
                }
            }
        }

        // ---- staging --------------------------------------------------------------------
        S->Stagingp.assign (nf+2, 0) ; S->StageMap.assign (nf+1, 0) ;
        S->FSize.assign (nf+1, 0) ; S->RSize.assign (nf+1, 0) ; S->SSize.assign (nf+1, 0) ;
        S->FOffsets.assign (nf+1, 0) ; S->ROffsets.assign (nf+1, 0) ; S->SOffsets.assign (nf+1, 0) ;
        S->numStages = 0 ;
        S->gpuMemorySize = gpuMemorySize ;
        int feasible = spqrgpu_amd_compute_front_staging (nf, S->Parent.data ( ), S->Childp.data ( ),
            S->Child.data ( ), S->Fm.data ( ), S->Cm.data ( ), S->Rp.data ( ), S->Sp.data ( ), S->Sleft.data ( ),
            S->Super.data ( ), S->Post.data ( ), S->RimapSize, S->RjmapSize, gpuMemorySize,
            &S->numStages, S->Stagingp.data ( ), S->StageMap.data ( ),
            (size_t *) S->FSize.data ( ), (size_t *) S->RSize.data ( ), (size_t *) S->SSize.data ( ),
            S->FOffsets.data ( ), S->ROffsets.data ( ), S->SOffsets.data ( )) ;
        if (!feasible) *status = 2 ;
        return S ;
    }
    catch (std::bad_alloc &)
    {
        delete S ;
        *status = 1 ;
        return NULL ;
    }
}

// The symbolic object from the arrays an SPQR build already has: what spqrgpu_kernel reads out of
// QRsym and QRsym->QRgpu (SPQR/SPQRGPU/spqrgpu_kernel.cpp:85-143).  Everything is copied.  Sj must
// hold front-local column indices, as spqr_analyze leaves it for the GPU path
// (SPQR/Source/spqr_analyze.cpp:858-882).  The staging arrays are taken as given (numStages,
// Stagingp, StageMap decide which fronts share a stage); the size and offset arrays of the
// reference's arenas are not needed, this driver places the fronts of a stage itself.
extern "C" spqr_amd_symbolic *spqr_amd_symbolic_from_arrays
(
    int64_t m, int64_t n, int64_t anz, int64_t nf,
    const int64_t *Sp, const int64_t *Sj, const int64_t *Sleft,
    const int64_t *Super, const int64_t *Rp, const int64_t *Rj,
    const int64_t *Parent, const int64_t *Childp, const int64_t *Child, const int64_t *Post,
    const int64_t *Fm, const int64_t *Cm,
    const int64_t *RimapOffsets, int64_t RimapSize, const int64_t *RjmapOffsets, int64_t RjmapSize,
    int64_t numStages, const int64_t *Stagingp, const int64_t *StageMap,
    size_t gpuMemorySize, int *status
)
{
    int dummy ;
    if (!status) status = &dummy ;
    *status = 0 ;
    if (m < 0 || n < 0 || anz < 0 || nf < 0 || !Sp || !Sleft || !Super || !Rp || !Parent || !Childp || !Child
        || !Post || !Fm || !Cm || !RimapOffsets || !RjmapOffsets || !Stagingp || !StageMap
        || (anz > 0 && !Sj) || (nf > 0 && Rp [nf] > 0 && !Rj) || numStages < 0)
    {
        *status = 3 ;
        return NULL ;
    }
    // the checks the driver's indexing relies on
    if (Sp [m] != anz || Super [nf] > n) { *status = 3 ; return NULL ; }
    for (Long f = 0 ; f < nf ; f++)
    {
        const Long fp = Super [f+1] - Super [f], fn = Rp [f+1] - Rp [f] ;
        if (fp < 0 || fn < fp || Fm [f] < 0 || Cm [f] < 0 || Cm [f] > fn - fp || Cm [f] > Fm [f]
            || Parent [f] <= f || Parent [f] > nf || StageMap [f] < 0 || StageMap [f] > numStages
            || Post [f] < 0 || Post [f] >= nf)
        {
            *status = 3 ;
            return NULL ;
        }
    }
    for (Long k = 0 ; k < anz ; k++) if (Sj [k] < 0) { *status = 3 ; return NULL ; }
    spqr_amd_symbolic *S = NULL ;
    try
    {
        S = new spqr_amd_symbolic ( ) ;
        { static std::atomic<uint64_t> next_uid ((uint64_t) 1 << 40) ; S->uid = next_uid.fetch_add (1) ; }
        S->m = m ; S->n = n ; S->anz = anz ; S->nf = nf ;
        S->Sp.assign (Sp, Sp + m + 1) ;
        S->Sj.assign (Sj, Sj + anz) ;
        S->Sleft.assign (Sleft, Sleft + n + 2) ;
        S->Super.assign (Super, Super + nf + 1) ;
        S->Rp.assign (Rp, Rp + nf + 1) ;
        S->rjsize = Rp [nf] ;
        S->Rj.assign (Rj, Rj + S->rjsize) ;
        S->Parent.assign (Parent, Parent + nf + 1) ;
        S->Childp.assign (Childp, Childp + nf + 2) ;
        S->Child.assign (Child, Child + nf + 1) ;
        S->Post.assign (Post, Post + nf) ;
        S->Post.push_back (nf) ;
        S->Fm.assign (Fm, Fm + nf) ; S->Fm.push_back (0) ;
        S->Cm.assign (Cm, Cm + nf) ; S->Cm.push_back (0) ;
        S->RimapOffsets.assign (RimapOffsets, RimapOffsets + nf) ; S->RimapOffsets.push_back (RimapSize) ;
        S->RjmapOffsets.assign (RjmapOffsets, RjmapOffsets + nf) ; S->RjmapOffsets.push_back (RjmapSize) ;
        S->RimapSize = RimapSize ; S->RjmapSize = RjmapSize ;
        S->numStages = numStages ;
        S->Stagingp.assign (Stagingp, Stagingp + numStages + 1) ;
        S->Stagingp.resize ((size_t) nf + 2, nf) ;
        S->StageMap.assign (StageMap, StageMap + nf) ; S->StageMap.push_back (0) ;
        S->gpuMemorySize = gpuMemorySize ;
        for (Long f = 0 ; f < nf ; f++)
        {
            const Long fn = Rp [f+1] - Rp [f], rm = std::min (Fm [f], Super [f+1] - Super [f]) ;
            S->maxfn = std::max (S->maxfn, fn) ;
            S->rnz += rm*(rm+1)/2 + rm*(fn-rm) ;
        }
        return S ;
    }
    catch (std::bad_alloc &)
    {
        delete S ;
        *status = 1 ;
        return NULL ;
    }
}

extern "C" void spqr_amd_symbolic_free (spqr_amd_symbolic *sym)
{
    delete sym ;
}

extern "C" const int64_t *spqr_amd_symbolic_array (const spqr_amd_symbolic *S, const char *name, int64_t *len)
{
    if (!S || !name) return NULL ;
    const std::vector<Long> *v = NULL ;
    #define PICK(x) if (!strcmp (name, #x)) v = &S->x ;
    PICK (Qfill) PICK (PLinv) PICK (Sleft) PICK (Sp) PICK (Sj) PICK (Super) PICK (Rp) PICK (Rj)
    PICK (Parent) PICK (Childp) PICK (Child) PICK (Post) PICK (Fm) PICK (Cm) PICK (RjmapOffsets)
    PICK (RimapOffsets) PICK (Stagingp) PICK (StageMap) PICK (FSize) PICK (RSize) PICK (SSize)
    PICK (FOffsets) PICK (ROffsets) PICK (SOffsets) PICK (ColCount)
    #undef PICK
    if (!v) return NULL ;
    if (len) *len = (int64_t) v->size ( ) ;
    return v->data ( ) ;
}

extern "C" double spqr_amd_symbolic_scalar (const spqr_amd_symbolic *S, const char *name)
{
    if (!S || !name) return -1 ;
    #define PICK(x) if (!strcmp (name, #x)) return (double) S->x ;
    PICK (m) PICK (n) PICK (anz) PICK (nf) PICK (maxfn) PICK (rjsize) PICK (RimapSize) PICK (RjmapSize)
    PICK (numStages) PICK (flopcount) PICK (rnz)
    #undef PICK
    return -1 ;
}
