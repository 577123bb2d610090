/* CPU oracle for the multifrontal QR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's algorithm, used as the checker by
 * tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The
 * product (suitesparse_amd/) never links, loads or calls anything here.
 *
 * Pinning: the reference's device kernels need nvcc and its CPU SPQR needs a
 * CUDA header, BLAS/LAPACK and METIS that are not part of the reference tree,
 * so neither is built here.  The oracle is pinned against (1) the known-answer
 * checks of the reference's own tests: ||triu(R)'triu(R) - A'A|| / ||A'A|| <=
 * 1e-12 on the demo geometries with srand(1), rand()/1e8 entries
 * (GPUQREngine/Demo/gpuqrengine_demo.cpp:21-28,94-112,283,305-308) and the
 * qrtest bars ||R'R-(AP)'(AP)||_1/||A'A||_1 <= 1e-10, residual <= 1e-10
 * (SPQR/Tcov/qrtest.cpp:1218-1257,2750-2763); (2) uniqueness of R up to row
 * signs, checked against LAPACK (numpy) in tests/test_oracle.py; (3) the
 * staging outcomes of the reference's own function at gpuMemorySize 1e10, 0 and 1 (one stage; an
 * empty stage plus one front per stage; infeasible).
 * Integer outputs (staging, assembly maps, S scatter, pack assembly) are
 * compared bit for bit with the product.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int64_t Long ;

typedef struct { Long findex ; double value ; } oracle_sentry ;   /* GPUQREngine_SEntry.hpp:13-23 */

/* ssgpu_maxQueueSize, GPUQREngine/Source/Scheduler/ssgpu_maxQueueSize.cpp:6-27 */
size_t oracle_max_queue_size (size_t gpuMemorySize)
{
    size_t q ;
    size_t tileSizeBytes = 32 * 32 * sizeof (double) ;
    if (gpuMemorySize <= 1) q = 50000 ;
    else q = 2 * (gpuMemorySize / tileSizeBytes) / 64 ;
    if (q < 50000) q = 50000 ;
    return q ;
}

/* GPUQREngine_FindStaircase, GPUQREngine/Source/GPUQREngine_ExpertDense.cpp:196-217 */
void oracle_find_staircase (const double *F, Long fm, Long fn, Long *Stair)
{
    Long lastStair = 0 ;
    for (Long j = 0 ; j < fn ; j++)
    {
        Long i ;
        for (i = fm-1 ; i > lastStair && F [i*fn+j] == 0.0 ; i--) ;
        Stair [j] = lastStair = i ;
    }
}

/* Householder QR of one row-major front, column by column, with the device
 * kernel's reflector convention (GPUQREngine/Include/Kernel/Factorize/
 * factorize_vt.cu:272-293: sigma, s, v1, tau; z = -tau v'A, A += v z, :305-372)
 * and the front semantics of the engine: every column k < min(fm,fn) consumes
 * row k, the whole front is reduced (all fn columns), entries below the
 * diagonal end up zero (factorize_vt.cu:492-496).  Stair (may be NULL) bounds
 * the rows touched per column as in BucketList.cpp:107-128: rows above
 * max(min(Stair[j],fm-1), j) inclusive. */
void oracle_front_qr (double *F, Long fm, Long fn, const Long *Stair)
{
    Long kmax = (fm < fn) ? fm : fn ;
    double *v = (double *) malloc ((size_t) (fm > 0 ? fm : 1) * sizeof (double)) ;
    double *z = (double *) malloc ((size_t) (fn > 0 ? fn : 1) * sizeof (double)) ;
    Long e = 0 ;
    for (Long k = 0 ; k < kmax ; k++)
    {
        Long t = fm ;
        if (Stair != NULL)
        {
            t = Stair [k] + 1 ;
            if (t > fm) t = fm ;
        }
        if (t < k+1) t = k+1 ;
        if (t < e) t = e ;          /* fill-in never shrinks the active rows */
        e = t ;
        double sigma = 0 ;
        for (Long i = k+1 ; i < e ; i++) sigma += F [i*fn+k] * F [i*fn+k] ;
        double x1 = F [k*fn+k] ;
        double s, v1, tau ;
        if (sigma <= 1e-90)
        {
            s = x1 ; v1 = 0 ; tau = 0 ;
        }
        else
        {
            s = sqrt (x1*x1 + sigma) ;
            v1 = (x1 <= 0) ? (x1 - s) : (-sigma / (x1 + s)) ;
            tau = -1 / (s * v1) ;
        }
        v [k] = v1 ;
        for (Long i = k+1 ; i < e ; i++) v [i] = F [i*fn+k] ;
        F [k*fn+k] = s ;
        for (Long i = k+1 ; i < e ; i++) F [i*fn+k] = 0 ;
        if (tau != 0)
        {
            /* z = -tau v'A and A += v z, walking the row-major front row by row;
             * each z [j] still sums over i in increasing order */
            for (Long j = k+1 ; j < fn ; j++) z [j] = 0 ;
            for (Long i = k ; i < e ; i++)
            {
                const double vi = v [i] ;
                const double *Fi = F + i*fn ;
                for (Long j = k+1 ; j < fn ; j++) z [j] += vi * Fi [j] ;
            }
            for (Long j = k+1 ; j < fn ; j++) z [j] *= -tau ;
            for (Long i = k ; i < e ; i++)
            {
                const double vi = v [i] ;
                double *Fi = F + i*fn ;
                for (Long j = k+1 ; j < fn ; j++) Fi [j] += vi * z [j] ;
            }
        }
    }
    free (z) ;
    free (v) ;
}

/* sassemble, GPUQREngine/Include/Kernel/Assemble/sAssemble.cu:5-19 */
void oracle_sassemble (double *F, const oracle_sentry *S, Long pstart, Long pend)
{
    for (Long p = pstart ; p < pend ; p++) F [S [p].findex] = S [p].value ;
}

/* packassemble, GPUQREngine/Include/Kernel/Assemble/packAssemble.cu:39-56:
 * P[pn*Rimap[ci] + Rjmap[cj]] = C[cld*ci + cj] for cj >= ci */
void oracle_packassemble (const double *C, Long cld, Long cm, Long cn,
    double *P, Long pn, const int *Rimap, const int *Rjmap)
{
    for (Long ci = 0 ; ci < cm ; ci++)
        for (Long cj = ci ; cj < cn ; cj++)
            P [pn * Rimap [ci] + Rjmap [cj]] = C [cld * ci + cj] ;
}

/* spqrgpu_buildAssemblyMaps, SPQR/SPQRGPU/spqrgpu_buildAssemblyMaps.cpp:8-190.
 * Sj holds front-local column indices (spqr_analyze.cpp:869-880). */
void oracle_build_assembly_maps
(
    Long nf, Long n, Long *Fmap, const Long *Post, const Long *Super, const Long *Rp, const Long *Rj,
    const Long *Sleft, const Long *Sp, const Long *Sj, const double *Sx,
    const Long *Fm, const Long *Cm, const Long *Childp, const Long *Child,
    Long *CompleteStair, int *CompleteRjmap, const Long *RjmapOffsets,
    int *CompleteRimap, const Long *RimapOffsets, oracle_sentry *cpuS
)
{
    Long sindex = 0 ;
    (void) n ; (void) Fm ;
    for (Long pf = 0 ; pf < nf ; pf++)
    {
        Long f = Post [pf] ;
        Long pstart = Rp [f], pend = Rp [f+1] ;
        for (Long p = pstart ; p < pend ; p++) Fmap [Rj [p]] = p - pstart ;
        Long *Stair = CompleteStair + Rp [f] ;
        Long col1 = Super [f], col2 = Super [f+1] ;
        Long fp = col2 - col1 ;
        Long fn = Rp [f+1] - Rp [f] ;
        for (Long j = 0 ; j < fp ; j++) Stair [j] = Sleft [j+col1+1] - Sleft [j+col1] ;
        for (Long j = fp ; j < fn ; j++) Stair [j] = 0 ;
        for (Long p = Childp [f] ; p < Childp [f+1] ; p++)
        {
            Long c = Child [p] ;
            int *Rjmap = CompleteRjmap + RjmapOffsets [c] ;
            Long pc = Rp [c] ;
            Long fnc = Rp [c+1] - pc ;
            Long fpc = Super [c+1] - Super [c] ;
            Long cm = Cm [c] ;
            Long cn = fnc - fpc ;
            for (Long pp = 0 ; pp < cn ; pp++) Rjmap [pp] = (int) Fmap [Rj [pc+fpc+pp]] ;
            pc += fpc ;
            for (Long ci = 0 ; ci < cm ; ci++) Stair [Fmap [Rj [pc+ci]]]++ ;
        }
        Long fm = 0 ;
        for (Long j = 0 ; j < fn ; j++)
        {
            Long t = fm ;
            fm += Stair [j] ;
            Stair [j] = t ;
        }
        Long Scount = Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]] ;
        if (Scount > 0)
        {
            for (Long k = 0 ; k < fp ; k++)
            {
                Long leftcol = k + col1 ;
                for (Long row = Sleft [leftcol] ; row < Sleft [leftcol+1] ; row++)
                {
                    Long i = Stair [k]++ ;
                    for (Long p = Sp [row] ; p < Sp [row+1] ; p++)
                    {
                        cpuS [sindex].findex = i*fn + Sj [p] ;
                        cpuS [sindex].value = Sx [p] ;
                        sindex++ ;
                    }
                }
            }
        }
        for (Long p = Childp [f] ; p < Childp [f+1] ; p++)
        {
            Long c = Child [p] ;
            int *Rimap = CompleteRimap + RimapOffsets [c] ;
            int *Rjmap = CompleteRjmap + RjmapOffsets [c] ;
            Long cm = Cm [c] ;
            for (Long ci = 0 ; ci < cm ; ci++)
            {
                Long j = Rjmap [ci] ;
                Long i = Stair [j]++ ;
                Rimap [ci] = (int) i ;
            }
        }
    }
}

/* spqrgpu_computeFrontStaging, SPQR/SPQRGPU/spqrgpu_computeFrontStaging.cpp:75-249.
 * sizeof(TaskDescriptor) = 104, TILESIZE = 32.  Returns 1 if feasible. */
int oracle_compute_front_staging
(
    Long numFronts, const Long *Parent, const Long *Childp, const Long *Child,
    const Long *Fm, const Long *Cm, const Long *Rp, const Long *Sp, const Long *Sleft,
    const Long *Super, const Long *Post, Long RimapSize, Long RjmapSize, size_t gpuMemorySize,
    Long *numStages, Long *Stagingp, Long *StageMap,
    size_t *FSize, size_t *RSize, size_t *SSize, Long *FOffsets, Long *ROffsets, Long *SOffsets
)
{
    (void) Parent ;
    if (gpuMemorySize > 1)
    {
        size_t s = 2 * oracle_max_queue_size (gpuMemorySize) * 104 ;
        gpuMemorySize = (gpuMemorySize > s) ? (gpuMemorySize - s) : 0 ;
    }
    if (gpuMemorySize > 1)
    {
        size_t s = RimapSize * sizeof (int) ;
        gpuMemorySize = (gpuMemorySize > s) ? (gpuMemorySize - s) : 0 ;
    }
    if (gpuMemorySize > 1)
    {
        size_t s = RjmapSize * sizeof (int) ;
        gpuMemorySize = (gpuMemorySize > s) ? (gpuMemorySize - s) : 0 ;
    }
    if (gpuMemorySize > 1)
    {
        size_t s = 1024 * 1024 ;
        gpuMemorySize = (gpuMemorySize > s) ? (gpuMemorySize - s) : 0 ;
    }
    size_t ReqMem = 0, RMem = 0, SMem = 0, VTMem = 0 ;
    Long stage = 0 ;
    Stagingp [0] = 0 ;
    for (Long p = 0 ; p < numFronts ; p++)
    {
        Long f = Post [p] ;
        Long fm = Fm [f] ;
        Long fn = Rp [f+1] - Rp [f] ;
        Long fp = Super [f+1] - Super [f] ;
        Long frank = (fm < fp) ? fm : fp ;
        Long cm = Cm [f] ;
        size_t frontMem = fm * fn ;
        size_t rMem = (frank + cm) * fn ;
        size_t sMem = 2 * (Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
        size_t vtMem = ((fm + 31) / 32) * 33 * 32 ;
        size_t childMemOld = 0, childMemNew = 0 ;
        for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
        {
            Long c = Child [cp] ;
            Long ccn = (Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]) ;
            Long ccm = Cm [c] ;
            if (StageMap [c] < stage) childMemOld += ccm * ccn ;
            else childMemNew += ccm * ccn ;
        }
        if ((ReqMem + (frontMem + childMemOld + sMem + vtMem)) * sizeof (double) < gpuMemorySize)
        {
            FOffsets [f] = ReqMem ; ROffsets [f] = RMem ; SOffsets [f] = SMem / 2 ;
            ReqMem += frontMem + childMemOld ; RMem += rMem ; SMem += sMem ; VTMem += vtMem ;
        }
        else if (gpuMemorySize == 0 ||
            ((frontMem + childMemOld + childMemNew + sMem + vtMem) * sizeof (double) < gpuMemorySize))
        {
            FSize [stage] = ReqMem ; RSize [stage] = RMem ; SSize [stage] = SMem / 2 ;
            Stagingp [++stage] = p ;
            ReqMem = 0 ; RMem = 0 ; SMem = 0 ; VTMem = 0 ;
            FOffsets [f] = ReqMem ; ROffsets [f] = RMem ; SOffsets [f] = SMem / 2 ;
            ReqMem += frontMem + childMemOld + childMemNew ; RMem += rMem ; SMem += sMem ; VTMem += vtMem ;
        }
        else
        {
            *numStages = 0 ;
            return 0 ;
        }
        StageMap [f] = stage ;
    }
    FSize [stage] = ReqMem ; RSize [stage] = RMem ; SSize [stage] = SMem / 2 ;
    Stagingp [++stage] = numFronts ;
    if (Stagingp [stage] == Stagingp [stage-1]) stage-- ;
    *numStages = stage ;
    (void) VTMem ;
    return 1 ;
}

/* Whole-tree numeric factorization on the CPU in the engine's data layout
 * (row-major fronts, S scatter, pack assembly, front QR), all fronts in one
 * stage: the differential twin of spqrgpu_kernel + GPUQREngine for one stage
 * (SPQR/SPQRGPU/spqrgpu_kernel.cpp:486-555, Scheduler FSM
 * GPUQREngine_FrontState.hpp:16-28).  Farena holds every front at FOffsets[f]
 * and must be zero on entry; on return front f holds R in its first
 * min(fm,fp) rows and C below. */
void oracle_multifrontal
(
    Long nf, const Long *Post, const Long *Parent, const Long *Super, const Long *Rp,
    const Long *Fm, const Long *Cm, const Long *Sleft, const Long *Sp,
    const Long *Stair, const int *Rjmap, const Long *RjmapOffsets,
    const int *Rimap, const Long *RimapOffsets, const oracle_sentry *cpuS,
    const Long *FOffsets, const Long *SOffsets, double *Farena
)
{
    for (Long k = 0 ; k < nf ; k++)
    {
        Long f = Post [k] ;
        Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
        double *F = Farena + FOffsets [f] ;
        Long Scount = Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]] ;
        if (Scount > 0) oracle_sassemble (F, cpuS + SOffsets [f], 0, Scount) ;
        /* children were assembled into F when they finished (below) */
        /* the engine treats Stair[j] as an inclusive row bound (BucketList.cpp:107-128) */
        oracle_front_qr (F, fm, fn, Stair + Rp [f]) ;
        Long pid = Parent [f] ;
        if (pid >= 0 && pid < nf && Cm [f] > 0 && fn - fp > 0)
        {
            Long frank = (fm < fp) ? fm : fp ;
            Long pn = Rp [pid+1] - Rp [pid] ;
            oracle_packassemble (F + frank*fn + fp, fn, Cm [f], fn - fp,
                Farena + FOffsets [pid], pn, Rimap + RimapOffsets [f], Rjmap + RjmapOffsets [f]) ;
        }
    }
}

/* A batch of independent fronts, one thread per front at a time: the CPU
 * baseline leg of the benchmark.  F holds the fronts back to back. */
void oracle_front_qr_batch (double *F, const Long *fm, const Long *fn, Long nfronts, int nthreads)
{
    Long *off = (Long *) malloc ((size_t) (nfronts + 1) * sizeof (Long)) ;
    off [0] = 0 ;
    for (Long k = 0 ; k < nfronts ; k++) off [k+1] = off [k] + fm [k] * fn [k] ;
    #pragma omp parallel for schedule(dynamic,1) num_threads(nthreads)
    for (Long k = 0 ; k < nfronts ; k++) oracle_front_qr (F + off [k], fm [k], fn [k], NULL) ;
    free (off) ;
}

/* ---------------------------------------------------------------------------
 * rank detection: the CPU path's dead-column rule
 * ------------------------------------------------------------------------- */

/* Householder QR of a front with SPQR's rank detection, restating
 * SPQR/Source/spqr_front.cpp:356-479 one reflection at a time (its fchunk <= 1
 * path; the blocked path applies the same reflections later): g is the row of
 * the next pivot; column k is reduced over rows g..fm-1; if k is one of the
 * first npiv (pivot) columns and the resulting |R(g,k)| = norm (F (g:fm-1,k)) is
 * <= tol, the column is dead: it is zeroed from row g down, Rdead [k] = 1 and g
 * does NOT advance, so the next live column pivots on the same row (R comes out
 * "squeezed", spqr_front.cpp:424-479).  Pivot columns that are reached after the
 * rows have run out (g == fm) are dead as well (spqr_front.cpp:334-354).  The
 * reflection itself follows the device convention above (R(g,k) >= 0), which
 * differs from LAPACK's dlarfg only in the sign of a row of R.  Row-major
 * fm-by-fn front; returns the number of live pivot columns (the rank of the
 * front's R block); non-pivot columns are reduced too, so that the contribution
 * block below row `rank` comes out upper trapezoidal.  Dead-column 2-norms are
 * accumulated into *wssq as a plain sum of squares (spqr_front keeps a scaled
 * sum, :434-452, for the same quantity). */
Long oracle_front_qr_tol (double *F, Long fm, Long fn, Long npiv, double tol, char *Rdead, double *wssq)
{
    double *v = (double *) malloc ((size_t) (fm > 0 ? fm : 1) * sizeof (double)) ;
    double *z = (double *) malloc ((size_t) (fn > 0 ? fn : 1) * sizeof (double)) ;
    Long g = 0, rank = 0 ;
    if (npiv > fn) npiv = fn ;
    for (Long k = 0 ; k < fn ; k++)
    {
        if (k == npiv) rank = g ;
        if (g >= fm)
        {
            /* hit the wall: no rows left */
            if (k < npiv) Rdead [k] = 1 ;
            continue ;
        }
        double sigma = 0 ;
        for (Long i = g+1 ; i < fm ; i++) sigma += F [i*fn+k] * F [i*fn+k] ;
        double x1 = F [g*fn+k] ;
        double s, v1, tau ;
        if (sigma <= 1e-90)
        {
            s = x1 ; v1 = 0 ; tau = 0 ;
        }
        else
        {
            s = sqrt (x1*x1 + sigma) ;
            v1 = (x1 <= 0) ? (x1 - s) : (-sigma / (x1 + s)) ;
            tau = -1 / (s * v1) ;
        }
        if (k < npiv && tol >= 0 && fabs (s) <= tol)
        {
            /* dead column: flag it, drop what is left of it, keep the pivot row for the next one */
            if (wssq) *wssq += s*s ;
            for (Long i = g ; i < fm ; i++) F [i*fn+k] = 0 ;
            Rdead [k] = 1 ;
            continue ;
        }
        v [g] = v1 ;
        for (Long i = g+1 ; i < fm ; i++) v [i] = F [i*fn+k] ;
        F [g*fn+k] = s ;
        for (Long i = g+1 ; i < fm ; i++) F [i*fn+k] = 0 ;
        if (tau != 0)
        {
            for (Long j = k+1 ; j < fn ; j++) z [j] = 0 ;
            for (Long i = g ; i < fm ; i++)
            {
                const double vi = v [i] ;
                const double *Fi = F + i*fn ;
                for (Long j = k+1 ; j < fn ; j++) z [j] += vi * Fi [j] ;
            }
            for (Long j = k+1 ; j < fn ; j++) z [j] *= -tau ;
            for (Long i = g ; i < fm ; i++)
            {
                const double vi = v [i] ;
                double *Fi = F + i*fn ;
                for (Long j = k+1 ; j < fn ; j++) Fi [j] += vi * z [j] ;
            }
        }
        g++ ;
    }
    if (npiv >= fn) rank = g ;
    free (z) ;
    free (v) ;
    return rank ;
}

/* The whole tree with rank detection, as spqr_kernel drives spqr_front
 * (SPQR/Source/spqr_kernel.cpp:134-321): per front, assemble, factorize with tol
 * over the front's pivot columns, hand the rows below the front's rank (at most
 * cn of them: the contribution block, spqr_cpack) to the parent.  The symbolic
 * object must come from an analysis with rank detection allowed, i.e. Cm [f] =
 * min (fm, cn) and Fm sized for it (SPQR/Source/spqr_analyze.cpp:826-842), so that
 * a parent has a row for every row a child may pass up; rows a child does not
 * use stay zero.  Rdead has one entry per column of S (pivot k of front f is
 * column Super [f] + k); Rank [f] gets the rank of front f's R block.  Farena
 * must be zero on entry; on return front f holds R in its first Rank [f] rows. */
void oracle_multifrontal_tol
(
    Long nf, const Long *Post, const Long *Parent, const Long *Super, const Long *Rp,
    const Long *Fm, const Long *Cm, const Long *Sleft, const Long *Sp,
    const int *Rjmap, const Long *RjmapOffsets,
    const int *Rimap, const Long *RimapOffsets, const oracle_sentry *cpuS,
    const Long *FOffsets, const Long *SOffsets, double *Farena,
    double tol, char *Rdead, Long *Rank, double *norm_E_fro
)
{
    double wssq = 0 ;
    for (Long k = 0 ; k < nf ; k++)
    {
        Long f = Post [k] ;
        Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
        double *F = Farena + FOffsets [f] ;
        Long Scount = Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]] ;
        if (Scount > 0) oracle_sassemble (F, cpuS + SOffsets [f], 0, Scount) ;
        Long rank = oracle_front_qr_tol (F, fm, fn, fp, tol, Rdead + Super [f], &wssq) ;
        Rank [f] = rank ;
        Long pid = Parent [f] ;
        Long cn = fn - fp ;
        Long cm = fm - rank ;
        if (cm > cn) cm = cn ;
        if (cm > Cm [f]) cm = Cm [f] ;      /* cannot happen with a rank-detection analysis */
        if (pid >= 0 && pid < nf && cm > 0 && cn > 0)
        {
            Long pn = Rp [pid+1] - Rp [pid] ;
            oracle_packassemble (F + rank*fn + fp, fn, cm, cn,
                Farena + FOffsets [pid], pn, Rimap + RimapOffsets [f], Rjmap + RjmapOffsets [f]) ;
        }
    }
    if (norm_E_fro) *norm_E_fro = sqrt (wssq) ;
}

/* --------------------------------------------------------------------------
 * the reference's per-task flop counts, restated from
 * GPUQREngine/Source/TaskDescriptor/TaskDescriptor_flops.cpp:18-53 with its
 * constant TILESIZE = 32 (GPUQREngine_Common.hpp:39)
 * -------------------------------------------------------------------------- */

Long oracle_flops_factorize_vt (int numTiles)
{
    Long m = 32 * (Long) numTiles, n = 32, v = 32 ;
    return 2 * (m-1) + v * (6 + 4*m*n + 5*n + 2*m) + ((-4*n -2*m -5) * (v*(v+1)/2))
        + 3 * (v*(v+1)*(2*v+1)/6) + (- 2*(m-v+1)*(n-v) - 2*(m-v-1)) ;
}

Long oracle_flops_factorize (int m_, int n_)
{
    Long m = m_, n = n_, v = (m < n) ? m : n ;
    return 2 * (m-1) + v * (6 + 4*m*n + 5*n + 2*m) + ((-4*n -2*m -4) * v*(v+1)/2)
        + 2 * v*(v+1)*(2*v+1)/6 + (- 2*(m-v+1)*(n-v) - 2*(m-v-1)) ;
}

Long oracle_flops_apply (int numTiles, int n_)
{
    Long m = 32 * (Long) numTiles, k = 32, n = n_ ;
    return k*n*(4*m - k + 3) ;
}
