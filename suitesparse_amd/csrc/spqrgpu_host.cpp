// SPQR <-> engine glue on the host: staging, assembly maps, and the per-stage
// numeric driver.  Everything here is integer bookkeeping and data movement;
// all floating-point work happens in the engine's HIP kernels.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <new>

#include "../../include/spqrgpu_amd.h"
#include "spqr_symbolic.h"

typedef int64_t Long ;

// -----------------------------------------------------------------------------
// staging
// -----------------------------------------------------------------------------

// Partition the postordered fronts into stages that fit the device memory.
// Restates spqrgpu_computeFrontStaging.cpp:75-249.  The constants are the
// reference's even where this engine has no use for them (two task queues of
// 104-byte descriptors, 33x32 VT tiles per 32 rows, 1 MB slack), because the
// stage boundaries must come out identical.
extern "C" int spqrgpu_amd_compute_front_staging
(
    int64_t numFronts, const int64_t *Parent, const int64_t *Childp, const int64_t *Child,
    const int64_t *Fm, const int64_t *Cm, const int64_t *Rp, const int64_t *Sp, const int64_t *Sleft,
    const int64_t *Super, const int64_t *Post, int64_t RimapSize, int64_t RjmapSize,
    size_t mem,
    int64_t *numStages, int64_t *Stagingp, int64_t *StageMap,
    size_t *FSize, size_t *RSize, size_t *SSize,
    int64_t *FOffsets, int64_t *ROffsets, int64_t *SOffsets
)
{
    (void) Parent ;
    const size_t task_descriptor_bytes = 104 ;
    const size_t reserve [4] = {
        2 * gpuqr_max_queue_size (mem) * task_descriptor_bytes,
        (size_t) RimapSize * sizeof (int),
        (size_t) RjmapSize * sizeof (int),
        (size_t) 1024 * 1024 } ;
    // 0 and 1 are test values ("one front per stage", "nothing fits") and pass through
    for (int k = 0 ; k < 4 ; k++)
    {
        if (mem > 1) mem = (mem > reserve [k]) ? (mem - reserve [k]) : 0 ;
    }

    size_t fsum = 0, rsum = 0, ssum = 0 ;      // running sizes of the open stage, in doubles
    Long stage = 0 ;
    Stagingp [0] = 0 ;
    for (Long p = 0 ; p < numFronts ; p++)
    {
        const Long f = Post [p] ;
        const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
        const Long frank = std::min (fm, fp) ;
        const size_t front = (size_t) (fm * fn) ;
        const size_t rc = (size_t) ((frank + Cm [f]) * fn) ;
        const size_t sent = (size_t) (2 * (Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]])) ;
        const size_t vt = (size_t) (((fm + 31) / 32) * 33 * 32) ;
        size_t kids_old = 0, kids_new = 0 ;
        for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
        {
            const Long c = Child [cp] ;
            const size_t cb = (size_t) (Cm [c] * ((Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]))) ;
            if (StageMap [c] < stage) kids_old += cb ; else kids_new += cb ;
        }
        const bool fits_here = (fsum + (front + kids_old + sent + vt)) * sizeof (double) < mem ;
        if (!fits_here)
        {
            const bool fits_alone = (mem == 0) ||
                ((front + kids_old + kids_new + sent + vt) * sizeof (double) < mem) ;
            if (!fits_alone)
            {
                *numStages = 0 ;
                return 0 ;
            }
            // close the stage (possibly empty) and open the next one with this front
            FSize [stage] = fsum ; RSize [stage] = rsum ; SSize [stage] = ssum / 2 ;
            Stagingp [++stage] = p ;
            fsum = rsum = ssum = 0 ;
            kids_old += kids_new ;          // all children now come from earlier stages
        }
        FOffsets [f] = (Long) fsum ; ROffsets [f] = (Long) rsum ; SOffsets [f] = (Long) (ssum / 2) ;
        fsum += front + kids_old ; rsum += rc ; ssum += sent ;
        StageMap [f] = stage ;
    }
    FSize [stage] = fsum ; RSize [stage] = rsum ; SSize [stage] = ssum / 2 ;
    Stagingp [++stage] = numFronts ;
    if (Stagingp [stage] == Stagingp [stage-1]) stage-- ;
    *numStages = stage ;
    return 1 ;
}

extern "C" int64_t spqrgpu_amd_numfronts_in_stage
(
    int64_t stage, const int64_t *Stagingp, const int64_t *StageMap, const int64_t *Post,
    const int64_t *Child, const int64_t *Childp
)
{
    Long leftover = 0 ;
    for (Long p = Stagingp [stage] ; p < Stagingp [stage+1] ; p++)
    {
        const Long f = Post [p] ;
        for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++) if (StageMap [Child [cp]] < stage) leftover++ ;
    }
    return leftover ;
}

// -----------------------------------------------------------------------------
// assembly maps
// -----------------------------------------------------------------------------

// Per front, in postorder: the staircase, the column map of every child's
// contribution block (Rjmap), the packed S entries with their position in the
// front, and the row each child row lands in (Rimap).  Rows of a front are laid
// out column by column of the staircase: first the rows of S whose leftmost
// column is that pivot, then the rows of the children's C blocks whose leading
// column is that column, children in Child order.
// Restates spqrgpu_buildAssemblyMaps.cpp:8-190.
extern "C" void spqrgpu_amd_build_assembly_maps
(
    int64_t numFronts, int64_t n, int64_t *Fmap, const int64_t *Post, const int64_t *Super,
    const int64_t *Rp, const int64_t *Rj, const int64_t *Sleft, const int64_t *Sp, const int64_t *Sj,
    const double *Sx, const int64_t *Fm, const int64_t *Cm, const int64_t *Childp, const int64_t *Child,
    int64_t *CompleteStair, int32_t *CompleteRjmap, const int64_t *RjmapOffsets,
    int32_t *CompleteRimap, const int64_t *RimapOffsets, gpuqr_sentry *cpuS
)
{
    (void) n ; (void) Fm ;
    Long snext = 0 ;
    for (Long k = 0 ; k < numFronts ; k++)
    {
        const Long f = Post [k] ;
        const Long col1 = Super [f], fp = Super [f+1] - col1, fn = Rp [f+1] - Rp [f] ;
        Long *Stair = CompleteStair + Rp [f] ;
        for (Long j = 0 ; j < fn ; j++) Fmap [Rj [Rp [f] + j]] = j ;

        // row counts per staircase column
        for (Long j = 0 ; j < fn ; j++) Stair [j] = (j < fp) ? (Sleft [col1+j+1] - Sleft [col1+j]) : 0 ;
        for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
        {
            const Long c = Child [cp] ;
            const Long fpc = Super [c+1] - Super [c] ;
            const Long *ccols = Rj + Rp [c] + fpc ;         // columns of the child's C block
            const Long cn = Rp [c+1] - Rp [c] - fpc ;
            int32_t *Rjmap = CompleteRjmap + RjmapOffsets [c] ;
            for (Long j = 0 ; j < cn ; j++) Rjmap [j] = (int32_t) Fmap [ccols [j]] ;
            for (Long i = 0 ; i < Cm [c] ; i++) Stair [Rjmap [i]]++ ;   // row i of C leads with its column i
        }
        // counts -> first row of each staircase column
        Long rows = 0 ;
        for (Long j = 0 ; j < fn ; j++) { const Long cnt = Stair [j] ; Stair [j] = rows ; rows += cnt ; }

        // S rows, pivot column by pivot column
        for (Long j = 0 ; j < fp ; j++)
        {
            for (Long row = Sleft [col1+j] ; row < Sleft [col1+j+1] ; row++)
            {
                const Long i = Stair [j]++ ;
                for (Long p = Sp [row] ; p < Sp [row+1] ; p++)
                {
                    cpuS [snext].findex = i * fn + Sj [p] ;
                    cpuS [snext].value = Sx [p] ;
                    snext++ ;
                }
            }
        }
        // children rows
        for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
        {
            const Long c = Child [cp] ;
            int32_t *Rimap = CompleteRimap + RimapOffsets [c] ;
            const int32_t *Rjmap = CompleteRjmap + RjmapOffsets [c] ;
            for (Long i = 0 ; i < Cm [c] ; i++) Rimap [i] = (int32_t) (Stair [Rjmap [i]]++) ;
        }
    }
}

// -----------------------------------------------------------------------------
// numeric driver
// -----------------------------------------------------------------------------

#include <thread>
#include <atomic>

// Run body(k) for k in [0,n) on a few host threads (dynamic, one index at a time).
template <class Body> static void parallel_for (Long n, Body body)
{
    unsigned hw = std::thread::hardware_concurrency ( ) ;
    const char *env = getenv ("OMP_NUM_THREADS") ;
    long want = env ? atol (env) : 16 ;
    if (want < 1) want = 1 ;
    unsigned nt = (unsigned) std::min<long> (want, hw ? hw : 1) ;
    if (nt > 16) nt = 16 ;
    if (n < 64 || nt <= 1) { for (Long k = 0 ; k < n ; k++) body (k) ; return ; }
    std::atomic<Long> next (0) ;
    std::vector<std::thread> pool ;
    auto work = [&] ( ) { for (Long k = next.fetch_add (16) ; k < n ; k = next.fetch_add (16))
        for (Long q = k ; q < std::min (k + 16, n) ; q++) body (q) ; } ;
    for (unsigned t = 1 ; t < nt ; t++) pool.emplace_back (work) ;
    work ( ) ;
    for (size_t t = 0 ; t < pool.size ( ) ; t++) pool [t].join ( ) ;
}

// The first frank rows of a reduced front (row-major, unpacked) -> SPQR's column-packed upper
// trapezoid (spqrgpu_kernel.cpp:417-439), walking the source row by row so that the reads
// from the pinned pull buffer are contiguous.  Column j holds min(j+1, frank) entries.
static void pack_r_block (const double *cpuF, Long fn, Long frank, double *out)
{
    const Long tri = frank * (frank + 1) / 2 ;
    for (Long i = 0 ; i < frank ; i++)
    {
        const double *row = cpuF + fn * i ;
        for (Long j = i ; j < frank ; j++) out [j * (j + 1) / 2 + i] = row [j] ;
        double *rect = out + tri + i ;
        for (Long j = frank ; j < fn ; j++) rect [(j - frank) * frank] = row [j] ;
    }
}

static inline Long r_block_size (Long fn, Long frank)
{
    return frank * (frank + 1) / 2 + frank * (fn - frank) ;
}

#define HIPCHK(call) do { if ((call) != hipSuccess) { code = 2 ; goto done ; } } while (0)

#include <chrono>
static double now_ms (void)
{
    return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now ( ).time_since_epoch ( )).count ( ) ;
}
// GPUQR_VERBOSE=1 prints where the host time of the numeric driver goes
#define TICK(label) do { if (verbose) { double t_ = now_ms ( ) ; fprintf (stderr, "  [spqrgpu] %-28s %9.2f ms\n", label, t_ - tlast) ; tlast = t_ ; } } while (0)

extern "C" void spqr_amd_numeric_free (spqr_amd_numeric *num)
{
    if (!num) return ;
    free (num->Rstack) ; free (num->Rblock) ; free (num->Rdead) ; free (num->Cstack) ; free (num->Cblock) ;
    free (num) ;
}

// One call = the whole numeric factorization of A = Q R on the device, stage
// by stage.  Mirrors spqrgpu_kernel.cpp:63-782: build S and the maps, then per
// stage zero the front arena, ship S, describe the fronts (plus push-only
// fronts for contribution blocks that waited in host "limbo"), run the
// engine, pack R column-wise onto the stack and park the C blocks of fronts
// whose parent lives in a later stage.
extern "C" int spqrgpu_amd_factorize
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    spqr_amd_numeric **num_out
)
{
    if (!S || !Ap || !Ax || !num_out) return 3 ;
    *num_out = NULL ;
    int code = 0 ;
    const bool verbose = getenv ("GPUQR_VERBOSE") != NULL ;
    double tlast = now_ms ( ) ;
    const Long nf = S->nf, n = S->n, m = S->m ;
    const Long *Super = S->Super.data ( ), *Rp = S->Rp.data ( ), *Post = S->Post.data ( ) ;
    const Long *Parent = S->Parent.data ( ), *Childp = S->Childp.data ( ), *Child = S->Child.data ( ) ;
    const Long *Fm = S->Fm.data ( ), *Cm = S->Cm.data ( ), *Sleft = S->Sleft.data ( ), *Sp = S->Sp.data ( ) ;
    const Long *StageMap = S->StageMap.data ( ), *Stagingp = S->Stagingp.data ( ) ;

    spqr_amd_numeric *num = (spqr_amd_numeric *) calloc (1, sizeof (spqr_amd_numeric)) ;
    if (!num) return 1 ;
    double *dF = NULL ; gpuqr_sentry *dS = NULL ; int32_t *dRimap = NULL, *dRjmap = NULL ;
    double *hR [2] = {NULL, NULL} ; gpuqr_sentry *hS = NULL ;
    gpuqr_front *fronts = NULL ;
    std::vector<double *> limbo ;

    try
    {
        if (S->numStages <= 0 && nf > 0) { code = 2 ; goto done ; }
        num->nf = nf ;
        num->rsize = S->rnz ;
        num->Rstack = (double *) malloc (std::max ((Long) 1, S->rnz) * sizeof (double)) ;
        num->Rblock = (Long *) calloc (std::max ((Long) 1, nf), sizeof (Long)) ;
        num->Rdead = (char *) calloc (std::max ((Long) 1, n), 1) ;
        if (!num->Rstack || !num->Rblock || !num->Rdead) { code = 1 ; goto done ; }

        // ---- Sx: values of S in row order (spqr_stranspose2) ------------------------------
        std::vector<double> Sx ((size_t) S->anz) ;
        {
            std::vector<Long> w (Sp, Sp + m) ;
            for (Long k = 0 ; k < n ; k++)
            {
                const Long col = S->Qfill [k] ;
                for (Long p = Ap [col] ; p < Ap [col+1] ; p++) Sx [w [S->PLinv [Ai [p]]]++] = Ax [p] ;
            }
        }

        TICK ("alloc + Sx") ;
        // ---- maps and packed S for the whole problem ---------------------------------------
        const Long RimapSize = std::max ((Long) 1, S->RimapSize), RjmapSize = std::max ((Long) 1, S->RjmapSize) ;
        std::vector<Long> Stair ((size_t) std::max ((Long) 1, Rp [nf])), Fmap ((size_t) std::max ((Long) 1, n)) ;
        std::vector<int32_t> Rimap ((size_t) RimapSize), Rjmap ((size_t) RjmapSize) ;
        if (hipHostMalloc ((void **) &hS, std::max ((Long) 1, S->anz) * sizeof (gpuqr_sentry), hipHostMallocDefault) != hipSuccess) { hS = NULL ; code = 1 ; goto done ; }
        spqrgpu_amd_build_assembly_maps (nf, n, Fmap.data ( ), Post, Super, Rp, S->Rj.data ( ), Sleft, Sp,
            S->Sj.data ( ), Sx.data ( ), Fm, Cm, Childp, Child, Stair.data ( ), Rjmap.data ( ),
            S->RjmapOffsets.data ( ), Rimap.data ( ), S->RimapOffsets.data ( ), hS) ;
        if (hipMalloc ((void **) &dRimap, RimapSize * sizeof (int32_t)) != hipSuccess) { dRimap = NULL ; code = 1 ; goto done ; }
        if (hipMalloc ((void **) &dRjmap, RjmapSize * sizeof (int32_t)) != hipSuccess) { dRjmap = NULL ; code = 1 ; goto done ; }
        HIPCHK (hipMemcpy (dRimap, Rimap.data ( ), RimapSize * sizeof (int32_t), hipMemcpyHostToDevice)) ;
        HIPCHK (hipMemcpy (dRjmap, Rjmap.data ( ), RjmapSize * sizeof (int32_t), hipMemcpyHostToDevice)) ;

        TICK ("assembly maps + upload") ;
        // ---- workspaces sized for the largest stage -----------------------------------------
        size_t maxF = 1, maxR = 1, maxS = 1 ; Long maxfronts = 1 ;
        for (Long st = 0 ; st < S->numStages ; st++)
        {
            maxF = std::max (maxF, (size_t) S->FSize [st]) ;
            maxR = std::max (maxR, (size_t) S->RSize [st]) ;
            maxS = std::max (maxS, (size_t) S->SSize [st]) ;
            maxfronts = std::max (maxfronts, Stagingp [st+1] - Stagingp [st]
                + spqrgpu_amd_numfronts_in_stage (st, Stagingp, StageMap, Post, Child, Childp)) ;
        }
        if (hipMalloc ((void **) &dF, maxF * sizeof (double)) != hipSuccess) { dF = NULL ; code = 1 ; goto done ; }
        if (hipMalloc ((void **) &dS, maxS * sizeof (gpuqr_sentry)) != hipSuccess) { dS = NULL ; code = 1 ; goto done ; }
        if (hipHostMalloc ((void **) &hR [0], maxR * sizeof (double), hipHostMallocDefault) != hipSuccess) { hR [0] = NULL ; code = 1 ; goto done ; }
        fronts = (gpuqr_front *) malloc ((size_t) maxfronts * sizeof (gpuqr_front)) ;
        if (!fronts) { code = 1 ; goto done ; }
        limbo.assign ((size_t) std::max ((Long) 1, nf), (double *) NULL) ;

        std::vector<Long> InvPost ((size_t) nf + 1) ;
        for (Long k = 0 ; k < nf ; k++) InvPost [Post [k]] = k ;
        InvPost [nf] = nf ;

        TICK ("workspace allocation") ;
        Long SOffset = 0 ;
        Long rhead = 0 ;            // next free slot of Rstack
        for (Long stage = 0 ; stage < S->numStages ; stage++)
        {
            const Long sStart = Stagingp [stage], sEnd = Stagingp [stage+1] ;
            const Long numFronts = sEnd - sStart ;
            Long next_child = numFronts ;
            HIPCHK (hipMemset (dF, 0, std::max ((size_t) 1, (size_t) S->FSize [stage]) * sizeof (double))) ;
            if (S->SSize [stage] > 0)
                HIPCHK (hipMemcpy (dS, hS + SOffset, (size_t) S->SSize [stage] * sizeof (gpuqr_sentry), hipMemcpyHostToDevice)) ;
            SOffset += S->SSize [stage] ;

            for (Long p = sStart ; p < sEnd ; p++)
            {
                const Long f = Post [p] ;
                const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                const Long frank = std::min (fm, fp) ;
                const Long rel = p - sStart ;
                const Long pid = Parent [f] ;
                Long prel = -1 ;
                if (pid < nf && StageMap [pid] == stage) prel = InvPost [pid] - sStart ;

                gpuqr_front &fr = fronts [rel] ;
                memset (&fr, 0, sizeof (fr)) ;
                fr.fids = rel ; fr.pids = prel ; fr.fidg = f ; fr.pidg = pid ;
                fr.fm = fm ; fr.fn = fn ; fr.fp = fp ; fr.rank = frank ;
                fr.ldn = fn ;
                fr.Stair = Stair.data ( ) + Rp [f] ;
                fr.gpuF = dF + S->FOffsets [f] ;
                fr.cpuR = hR [0] + S->ROffsets [f] ;
                fr.is_sparse = 1 ;
                fr.nc = (int32_t) (Childp [f+1] - Childp [f]) ;
                fr.Scount = std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
                fr.gpuS = dS + S->SOffsets [f] ;
                fr.cn = (int32_t) (fn - fp) ;
                fr.cm = (int32_t) Cm [f] ;
                if (prel != -1)
                {
                    fr.pn = (int32_t) (Rp [pid+1] - Rp [pid]) ;
                    fr.gpuRjmap = dRjmap + S->RjmapOffsets [f] ;
                    fr.gpuRimap = dRimap + S->RimapOffsets [f] ;
                    fr.gpuC = fr.gpuF + frank * fn + fp ;
                }
                else if (pid < nf)
                {
                    fr.is_staged = 1 ;      // parent is factorized in a later stage
                }

                // children factorized in earlier stages: their C blocks come back from limbo
                // and sit right behind the front in the arena (the staging reserved the room)
                Long child_off = S->FOffsets [f] + fm * fn ;
                for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
                {
                    const Long c = Child [cp] ;
                    if (StageMap [c] == stage) continue ;
                    const Long ccn = (Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]) ;
                    const Long ccm = Cm [c] ;
                    gpuqr_front &ch = fronts [next_child] ;
                    memset (&ch, 0, sizeof (ch)) ;
                    ch.fids = next_child ; ch.pids = rel ; ch.fidg = c ; ch.pidg = f ;
                    ch.fm = ccm ; ch.fn = ccn ; ch.fp = 0 ; ch.rank = 0 ; ch.ldn = ccn ;
                    ch.is_sparse = 1 ; ch.push_only = 1 ;
                    ch.pn = (int32_t) fn ; ch.cm = (int32_t) ccm ; ch.cn = (int32_t) ccn ;
                    ch.gpuRimap = dRimap + S->RimapOffsets [c] ;
                    ch.gpuRjmap = dRjmap + S->RjmapOffsets [c] ;
                    ch.gpuF = ch.gpuC = dF + child_off ;
                    if (ccm * ccn > 0)
                    {
                        if (!limbo [c]) { code = 2 ; goto done ; }
                        HIPCHK (hipMemcpy (ch.gpuF, limbo [c], (size_t) (ccm * ccn) * sizeof (double), hipMemcpyHostToDevice)) ;
                    }
                    free (limbo [c]) ; limbo [c] = NULL ;
                    child_off += ccm * ccn ;
                    next_child++ ;
                }
            }

            TICK ("stage setup (memset, S, fronts)") ;
            // ---- the device work ---------------------------------------------------------------
            gpuqr_stats stats ;
            int ecode = gpuqr_factor_sparse (S->gpuMemorySize, fronts, next_child, Parent, Childp, Child, &stats) ;
            TICK ("engine") ;
            if (ecode != GPUQR_SUCCESS) { code = (ecode == GPUQR_OUTOFMEMORY) ? 1 : 2 ; goto done ; }
            num->gpuKernelTime += stats.kernelTime ;
            num->gpuFlops += stats.flopsActual ;
            num->gpuNumKernelLaunches += stats.numLaunches ;
            num->numStagesRun++ ;

            // ---- R onto the stack (column-packed), staged C blocks into limbo ---------------------
            {
                std::vector<Long> roff ((size_t) numFronts + 1) ;
                roff [0] = rhead ;
                for (Long p = sStart ; p < sEnd ; p++)
                {
                    const Long f = Post [p] ;
                    const Long fn = Rp [f+1] - Rp [f] ;
                    const Long frank = std::min (Fm [f], Super [f+1] - Super [f]) ;
                    num->Rblock [f] = roff [p - sStart] ;
                    roff [p - sStart + 1] = roff [p - sStart] + r_block_size (fn, frank) ;
                }
                rhead = roff [numFronts] ;
                std::atomic<int> oom (0) ;
                parallel_for (numFronts, [&] (Long k)
                {
                    const Long f = Post [sStart + k] ;
                    const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                    const Long frank = std::min (fm, fp) ;
                    const double *cpuF = fronts [k].cpuR ;
                    pack_r_block (cpuF, fn, frank, num->Rstack + roff [k]) ;
                    if (fronts [k].is_staged)
                    {
                        const Long cn = fn - fp, cm = std::min (fm - frank, cn) ;
                        const double *C = cpuF + frank * fn + fp ;
                        double *L = (double *) calloc ((size_t) std::max ((Long) 1, cm * cn), sizeof (double)) ;
                        if (!L) { oom = 1 ; return ; }
                        for (Long i = 0 ; i < cm ; i++) for (Long j = i ; j < cn ; j++) L [i*cn + j] = C [i*fn + j] ;
                        limbo [f] = L ;
                    }
                }) ;
                if (oom) { code = 1 ; goto done ; }
                for (Long p = sStart ; p < sEnd ; p++)
                {
                    const Long f = Post [p] ;
                    const Long fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                    // assembly traffic: 24 B per S entry, 16 B per C entry + 4 B per map int
                    num->assembleBytes += 24.0 * (double) fronts [p - sStart].Scount ;
                    if (Cm [f] > 0 && fn - fp > 0 && Parent [f] < nf)
                    {
                        const double cmm = (double) Cm [f], cnn = (double) (fn - fp) ;
                        num->assembleBytes += 16.0 * (cmm * (cmm + 1) / 2 + cmm * (cnn - cmm)) + 4.0 * (cmm + cnn) ;
                    }
                }
            }
        }

        TICK ("R pack") ;
        // ---- ranks and dead columns (no rank detection on the device) ----------------------------
        for (Long f = 0 ; f < nf ; f++)
        {
            const Long fm = Fm [f], fp = Super [f+1] - Super [f] ;
            const Long frank = std::min (fm, fp) ;
            for (Long k = fm ; k < fp ; k++) num->Rdead [Super [f] + k] = 1 ;      // pivot columns past the bottom
            num->sumfrank += frank ;
            num->maxfrank = std::max (num->maxfrank, frank) ;
        }
    }
    catch (std::bad_alloc &)
    {
        code = 1 ;
    }

done:
    for (size_t i = 0 ; i < limbo.size ( ) ; i++) free (limbo [i]) ;
    free (fronts) ;
    if (dF) (void) hipFree (dF) ;
    if (dS) (void) hipFree (dS) ;
    if (dRimap) (void) hipFree (dRimap) ;
    if (dRjmap) (void) hipFree (dRjmap) ;
    if (hR [0]) (void) hipHostFree (hR [0]) ;
    if (hS) (void) hipHostFree (hS) ;
    if (code != 0) { spqr_amd_numeric_free (num) ; return code ; }
    *num_out = num ;
    return 0 ;
}

// R blocks -> compressed-column R (the R-only part of spqr_rcount / spqr_rconvert,
// SPQR/Source/spqr_rconvert.cpp:127-260 with keepH = false, n1rows = 0).  Fronts whose
// Rblock offset is negative (factorized by another process) contribute their rows as
// structural positions with no values only if skip_missing is 0; with it they are left out.
extern "C" int spqr_amd_rconvert_blocks
(
    const spqr_amd_symbolic *S, const double *Rstack, const int64_t *Rblock, const char *Rdead,
    int64_t *Rcp, int64_t *Ri, double *Rx
)
{
    if (!S || !Rstack || !Rblock || !Rdead || !Rcp) return 3 ;
    const Long nf = S->nf, n = S->n ;
    const bool fill = (Ri != NULL && Rx != NULL) ;
    std::vector<Long> next ;
    if (fill) next.assign (Rcp, Rcp + n) ;
    else for (Long j = 0 ; j <= n ; j++) Rcp [j] = 0 ;
    Long row1 = 0 ;
    for (Long f = 0 ; f < nf ; f++)
    {
        const double *R = (Rblock [f] >= 0) ? Rstack + Rblock [f] : NULL ;
        const Long col1 = S->Super [f], fp = S->Super [f+1] - col1 ;
        const Long pr = S->Rp [f], fn = S->Rp [f+1] - pr ;
        Long rm = 0 ;
        for (Long k = 0 ; k < fn ; k++)
        {
            Long j ;
            if (k < fp) { j = col1 + k ; if (!Rdead [j]) rm++ ; }
            else j = S->Rj [pr + k] ;
            if (!R) continue ;
            for (Long i = 0 ; i < rm ; i++)
            {
                const double x = *R++ ;
                if (fill) { Ri [next [j]] = row1 + i ; Rx [next [j]] = x ; next [j]++ ; }
                else Rcp [j+1]++ ;
            }
        }
        row1 += rm ;
    }
    if (!fill) for (Long j = 0 ; j < n ; j++) Rcp [j+1] += Rcp [j] ;
    return 0 ;
}

extern "C" int spqr_amd_rconvert
(
    const spqr_amd_symbolic *S, const spqr_amd_numeric *num, int64_t *Rcp, int64_t *Ri, double *Rx
)
{
    if (!num) return 3 ;
    return spqr_amd_rconvert_blocks (S, num->Rstack, num->Rblock, num->Rdead, Rcp, Ri, Rx) ;
}

// -----------------------------------------------------------------------------
// subtree driver: one process's share of the assembly tree
// -----------------------------------------------------------------------------

// Factorize the fronts f with mine[f] != 0 in one stage on the current device.
// Children of those fronts that belong to somebody else arrive as compact
// cm-by-cn contribution blocks in limbo_in[c] (host); fronts whose parent
// belongs to somebody else leave theirs in num->Cstack at Cblock[f].  This is
// the reference's stage / limbo mechanism (spqrgpu_kernel.cpp:556-611,687-724)
// with "a later stage" generalised to "another process".
extern "C" int spqrgpu_amd_factorize_subset
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    const char *mine, const double *const *limbo_in, spqr_amd_numeric **num_out
)
{
    if (!S || !Ap || !Ax || !mine || !num_out) return 3 ;
    *num_out = NULL ;
    int code = 0 ;
    const Long nf = S->nf, n = S->n, m = S->m ;
    const Long *Super = S->Super.data ( ), *Rp = S->Rp.data ( ), *Post = S->Post.data ( ) ;
    const Long *Parent = S->Parent.data ( ), *Childp = S->Childp.data ( ), *Child = S->Child.data ( ) ;
    const Long *Fm = S->Fm.data ( ), *Cm = S->Cm.data ( ), *Sleft = S->Sleft.data ( ), *Sp = S->Sp.data ( ) ;

    spqr_amd_numeric *num = (spqr_amd_numeric *) calloc (1, sizeof (spqr_amd_numeric)) ;
    if (!num) return 1 ;
    double *dF = NULL ; gpuqr_sentry *dS = NULL ; int32_t *dRimap = NULL, *dRjmap = NULL ;
    double *hR = NULL ; gpuqr_sentry *hS = NULL ;
    gpuqr_front *fronts = NULL ;

    try
    {
        // ---- layout of this process's arena -----------------------------------------------
        std::vector<Long> rel (nf, -1), FOff (nf, 0), ROff (nf, 0), SOff (nf, 0), SGlob (nf + 1, 0) ;
        Long nmine = 0, nlimbo = 0 ;
        size_t fsum = 0, rsum = 0, ssum = 0, rpack = 0, cpack = 0 ;
        {
            Long sacc = 0 ;
            for (Long k = 0 ; k < nf ; k++)          // packed S is laid out front after front in postorder
            {
                const Long f = Post [k] ;
                SGlob [f] = sacc ;
                sacc += std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
            }
        }
        for (Long k = 0 ; k < nf ; k++)
        {
            const Long f = Post [k] ;
            if (!mine [f]) continue ;
            const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
            const Long frank = std::min (fm, fp) ;
            rel [f] = nmine++ ;
            FOff [f] = (Long) fsum ; ROff [f] = (Long) rsum ; SOff [f] = (Long) ssum ;
            fsum += (size_t) (fm * fn) ;
            for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
            {
                const Long c = Child [cp] ;
                if (mine [c]) continue ;
                nlimbo++ ;
                fsum += (size_t) (Cm [c] * ((Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]))) ;
            }
            rsum += (size_t) ((frank + Cm [f]) * fn) ;
            ssum += (size_t) std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
            rpack += (size_t) (frank*(frank+1)/2 + frank*(fn-frank)) ;
            if (Parent [f] < nf && !mine [Parent [f]]) cpack += (size_t) (Cm [f] * (fn - fp)) ;
        }

        num->nf = nf ;
        num->rsize = (Long) rpack ;
        num->Rstack = (double *) malloc (std::max ((size_t) 1, rpack) * sizeof (double)) ;
        num->Rblock = (Long *) malloc (std::max ((Long) 1, nf) * sizeof (Long)) ;
        num->Rdead = (char *) calloc (std::max ((Long) 1, n), 1) ;
        num->Cstack = (double *) calloc (std::max ((size_t) 1, cpack), sizeof (double)) ;
        num->Cblock = (Long *) malloc (std::max ((Long) 1, nf) * sizeof (Long)) ;
        if (!num->Rstack || !num->Rblock || !num->Rdead || !num->Cstack || !num->Cblock) { code = 1 ; goto done ; }
        for (Long f = 0 ; f < nf ; f++) { num->Rblock [f] = -1 ; num->Cblock [f] = -1 ; }

        // ---- S values, maps, packed S (whole problem: pattern work only) ---------------------
        std::vector<double> Sx ((size_t) S->anz) ;
        {
            std::vector<Long> w (Sp, Sp + m) ;
            for (Long k = 0 ; k < n ; k++)
            {
                const Long col = S->Qfill [k] ;
                for (Long p = Ap [col] ; p < Ap [col+1] ; p++) Sx [w [S->PLinv [Ai [p]]]++] = Ax [p] ;
            }
        }
        const Long RimapSize = std::max ((Long) 1, S->RimapSize), RjmapSize = std::max ((Long) 1, S->RjmapSize) ;
        std::vector<Long> Stair ((size_t) std::max ((Long) 1, Rp [nf])), Fmap ((size_t) std::max ((Long) 1, n)) ;
        std::vector<int32_t> Rimap ((size_t) RimapSize), Rjmap ((size_t) RjmapSize) ;
        std::vector<gpuqr_sentry> allS ((size_t) std::max ((Long) 1, S->anz)) ;
        spqrgpu_amd_build_assembly_maps (nf, n, Fmap.data ( ), Post, Super, Rp, S->Rj.data ( ), Sleft, Sp,
            S->Sj.data ( ), Sx.data ( ), Fm, Cm, Childp, Child, Stair.data ( ), Rjmap.data ( ),
            S->RjmapOffsets.data ( ), Rimap.data ( ), S->RimapOffsets.data ( ), allS.data ( )) ;

        if (nmine > 0)
        {
            if (hipHostMalloc ((void **) &hS, std::max ((size_t) 1, ssum) * sizeof (gpuqr_sentry), hipHostMallocDefault) != hipSuccess) { hS = NULL ; code = 1 ; goto done ; }
            for (Long f = 0 ; f < nf ; f++)
            {
                if (!mine [f]) continue ;
                const Long cnt = std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
                if (cnt > 0) memcpy (hS + SOff [f], allS.data ( ) + SGlob [f], (size_t) cnt * sizeof (gpuqr_sentry)) ;
            }
            if (hipMalloc ((void **) &dRimap, RimapSize * sizeof (int32_t)) != hipSuccess) { dRimap = NULL ; code = 1 ; goto done ; }
            if (hipMalloc ((void **) &dRjmap, RjmapSize * sizeof (int32_t)) != hipSuccess) { dRjmap = NULL ; code = 1 ; goto done ; }
            HIPCHK (hipMemcpy (dRimap, Rimap.data ( ), RimapSize * sizeof (int32_t), hipMemcpyHostToDevice)) ;
            HIPCHK (hipMemcpy (dRjmap, Rjmap.data ( ), RjmapSize * sizeof (int32_t), hipMemcpyHostToDevice)) ;
            if (hipMalloc ((void **) &dF, std::max ((size_t) 1, fsum) * sizeof (double)) != hipSuccess) { dF = NULL ; code = 1 ; goto done ; }
            if (hipMalloc ((void **) &dS, std::max ((size_t) 1, ssum) * sizeof (gpuqr_sentry)) != hipSuccess) { dS = NULL ; code = 1 ; goto done ; }
            if (hipHostMalloc ((void **) &hR, std::max ((size_t) 1, rsum) * sizeof (double), hipHostMallocDefault) != hipSuccess) { hR = NULL ; code = 1 ; goto done ; }
            fronts = (gpuqr_front *) malloc ((size_t) (nmine + nlimbo) * sizeof (gpuqr_front)) ;
            if (!fronts) { code = 1 ; goto done ; }
            HIPCHK (hipMemset (dF, 0, std::max ((size_t) 1, fsum) * sizeof (double))) ;
            if (ssum > 0) HIPCHK (hipMemcpy (dS, hS, ssum * sizeof (gpuqr_sentry), hipMemcpyHostToDevice)) ;

            Long next_child = nmine ;
            for (Long f = 0 ; f < nf ; f++)
            {
                if (!mine [f]) continue ;
                const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                const Long frank = std::min (fm, fp) ;
                const Long pid = Parent [f] ;
                const bool parent_mine = (pid < nf && mine [pid]) ;
                gpuqr_front &fr = fronts [rel [f]] ;
                memset (&fr, 0, sizeof (fr)) ;
                fr.fids = rel [f] ; fr.pids = parent_mine ? rel [pid] : -1 ; fr.fidg = f ; fr.pidg = pid ;
                fr.fm = fm ; fr.fn = fn ; fr.fp = fp ; fr.rank = frank ; fr.ldn = fn ;
                fr.Stair = Stair.data ( ) + Rp [f] ;
                fr.gpuF = dF + FOff [f] ;
                fr.cpuR = hR + ROff [f] ;
                fr.is_sparse = 1 ;
                fr.nc = (int32_t) (Childp [f+1] - Childp [f]) ;
                fr.Scount = std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
                fr.gpuS = dS + SOff [f] ;
                fr.cn = (int32_t) (fn - fp) ; fr.cm = (int32_t) Cm [f] ;
                if (parent_mine)
                {
                    fr.pn = (int32_t) (Rp [pid+1] - Rp [pid]) ;
                    fr.gpuRjmap = dRjmap + S->RjmapOffsets [f] ;
                    fr.gpuRimap = dRimap + S->RimapOffsets [f] ;
                    fr.gpuC = fr.gpuF + frank * fn + fp ;
                }
                else if (pid < nf)
                {
                    fr.is_staged = 1 ;
                }
                Long child_off = FOff [f] + fm * fn ;
                for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
                {
                    const Long c = Child [cp] ;
                    if (mine [c]) continue ;
                    const Long ccn = (Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]) ;
                    const Long ccm = Cm [c] ;
                    gpuqr_front &ch = fronts [next_child] ;
                    memset (&ch, 0, sizeof (ch)) ;
                    ch.fids = next_child ; ch.pids = rel [f] ; ch.fidg = c ; ch.pidg = f ;
                    ch.fm = ccm ; ch.fn = ccn ; ch.ldn = ccn ;
                    ch.is_sparse = 1 ; ch.push_only = 1 ;
                    ch.pn = (int32_t) fn ; ch.cm = (int32_t) ccm ; ch.cn = (int32_t) ccn ;
                    ch.gpuRimap = dRimap + S->RimapOffsets [c] ;
                    ch.gpuRjmap = dRjmap + S->RjmapOffsets [c] ;
                    ch.gpuF = ch.gpuC = dF + child_off ;
                    if (ccm * ccn > 0)
                    {
                        if (!limbo_in || !limbo_in [c]) { code = 3 ; goto done ; }
                        HIPCHK (hipMemcpy (ch.gpuF, limbo_in [c], (size_t) (ccm * ccn) * sizeof (double), hipMemcpyHostToDevice)) ;
                    }
                    child_off += ccm * ccn ;
                    next_child++ ;
                }
            }

            gpuqr_stats stats ;
            int ecode = gpuqr_factor_sparse (S->gpuMemorySize, fronts, next_child, Parent, Childp, Child, &stats) ;
            if (ecode != GPUQR_SUCCESS) { code = (ecode == GPUQR_OUTOFMEMORY) ? 1 : 2 ; goto done ; }
            num->gpuKernelTime = stats.kernelTime ;
            num->gpuFlops = stats.flopsActual ;
            num->gpuNumKernelLaunches = stats.numLaunches ;
            num->numStagesRun = 1 ;

            {
                std::vector<Long> mylist, roff, coff ;
                Long rhead = 0, chead = 0 ;
                for (Long f = 0 ; f < nf ; f++)
                {
                    if (!mine [f]) continue ;
                    const Long fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                    const Long frank = std::min (Fm [f], fp) ;
                    mylist.push_back (f) ; roff.push_back (rhead) ; coff.push_back (chead) ;
                    num->Rblock [f] = rhead ;
                    rhead += r_block_size (fn, frank) ;
                    if (fronts [rel [f]].is_staged) { num->Cblock [f] = chead ; chead += Cm [f] * (fn - fp) ; }
                }
                parallel_for ((Long) mylist.size ( ), [&] (Long k)
                {
                    const Long f = mylist [k] ;
                    const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                    const Long frank = std::min (fm, fp) ;
                    const double *cpuF = fronts [rel [f]].cpuR ;
                    pack_r_block (cpuF, fn, frank, num->Rstack + roff [k]) ;
                    if (fronts [rel [f]].is_staged)
                    {
                        const Long cn = fn - fp, cm = Cm [f] ;
                        const double *C = cpuF + frank * fn + fp ;
                        double *L = num->Cstack + coff [k] ;
                        for (Long i = 0 ; i < cm ; i++) for (Long j = i ; j < cn ; j++) L [i*cn + j] = C [i*fn + j] ;
                    }
                }) ;
            }
        }
        for (Long f = 0 ; f < nf ; f++)
        {
            const Long fm = Fm [f], fp = Super [f+1] - Super [f] ;
            for (Long k = fm ; k < fp ; k++) num->Rdead [Super [f] + k] = 1 ;
            if (mine [f]) { num->sumfrank += std::min (fm, fp) ; num->maxfrank = std::max (num->maxfrank, std::min (fm, fp)) ; }
        }
    }
    catch (std::bad_alloc &)
    {
        code = 1 ;
    }

done:
    free (fronts) ;
    if (dF) (void) hipFree (dF) ;
    if (dS) (void) hipFree (dS) ;
    if (dRimap) (void) hipFree (dRimap) ;
    if (dRjmap) (void) hipFree (dRjmap) ;
    if (hR) (void) hipHostFree (hR) ;
    if (hS) (void) hipHostFree (hS) ;
    if (code != 0) { spqr_amd_numeric_free (num) ; return code ; }
    *num_out = num ;
    return 0 ;
}
