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
#include <math.h>

#include "../../include/spqrgpu_amd.h"
#include "spqr_symbolic.h"

typedef int64_t Long ;

// Host memory comes from the engine's allocator hooks (gpuqr_set_allocator), as the reference takes
// all of its own from SuiteSparse_config's table, so that a failing allocator installed by a test
// reaches every allocation of the glue too.
static inline void *host_malloc (size_t bytes) { return qr_host_malloc (bytes) ; }
static inline void *host_calloc (size_t n, size_t size) { return qr_host_calloc (n, size) ; }
static inline void host_free (void *p) { if (p) qr_host_free (p) ; }

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

#include <thread>
#include <atomic>

// Run body(k) for k in [0,n) on a few host threads (dynamic, one index at a time).
template <class Body> static void parallel_for (Long n, Body body, Long serial_below = 64)
{
    unsigned hw = std::thread::hardware_concurrency ( ) ;
    const char *env = getenv ("OMP_NUM_THREADS") ;
    long want = env ? atol (env) : 16 ;
    if (want < 1) want = 1 ;
    unsigned nt = (unsigned) std::min<long> (want, hw ? hw : 1) ;
    if (nt > 16) nt = 16 ;
    if (n < serial_below || n < 2 || nt <= 1) { for (Long k = 0 ; k < n ; k++) body (k) ; return ; }
    std::atomic<Long> next (0) ;
    std::vector<std::thread> pool ;
    const Long grain = (n >= 1024) ? 16 : 1 ;
    auto work = [&] ( ) { for (Long k = next.fetch_add (grain) ; k < n ; k = next.fetch_add (grain))
        for (Long q = k ; q < std::min (k + grain, n) ; q++) body (q) ; } ;
    for (unsigned t = 1 ; t < nt ; t++) pool.emplace_back (work) ;
    work ( ) ;
    for (size_t t = 0 ; t < pool.size ( ) ; t++) pool [t].join ( ) ;
}

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
    (void) Fm ;
    // where the S entries of the k-th front (in Post order) start: they are written front after front
    std::vector<Long> soff ((size_t) numFronts + 1, 0) ;
    for (Long k = 0 ; k < numFronts ; k++)
    {
        const Long f = Post [k] ;
        soff [(size_t) k + 1] = soff [(size_t) k] + (Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
    }
    // One front at a time, and no front touches what another one writes (its own staircase, its
    // children's maps, its own stretch of cpuS): many fronts are done by several threads, each with
    // a column map of its own (the caller's Fmap serves the first).
    auto one_front = [&] (Long k, Long *Fmap)
    {
        const Long f = Post [k] ;
        Long snext = soff [(size_t) k] ;
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
    } ;
    unsigned nt = 1 ;
    if (numFronts >= 4096)
    {
        const unsigned hw = std::thread::hardware_concurrency ( ) ;
        const char *env = getenv ("OMP_NUM_THREADS") ;
        const long want = env ? atol (env) : 16 ;
        nt = (unsigned) std::max<long> (1, std::min<long> (std::min<long> (want, hw ? hw : 1), 8)) ;
    }
    std::vector<Long *> maps (nt, (Long *) NULL) ;
    maps [0] = Fmap ;
    for (unsigned t = 1 ; t < nt ; t++)
    {
        maps [t] = (Long *) malloc ((size_t) std::max ((Long) 1, n) * sizeof (Long)) ;
        if (!maps [t]) { nt = t ; break ; }         // fewer threads, then
    }
    std::atomic<Long> next (0) ;
    const Long grain = 64 ;
    auto work = [&] (unsigned t)
    {
        for (Long k0 = next.fetch_add (grain) ; k0 < numFronts ; k0 = next.fetch_add (grain))
            for (Long k = k0 ; k < std::min (k0 + grain, numFronts) ; k++) one_front (k, maps [t]) ;
    } ;
    {
        std::vector<std::thread> pool ;
        try { for (unsigned t = 1 ; t < nt ; t++) pool.emplace_back (work, t) ; } catch (...) { }
        work (0) ;
        for (size_t t = 0 ; t < pool.size ( ) ; t++) pool [t].join ( ) ;
    }
    for (unsigned t = 1 ; t < maps.size ( ) ; t++) free (maps [t]) ;
}

// -----------------------------------------------------------------------------
// numeric driver
// -----------------------------------------------------------------------------

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

struct QrPackRTask ;
extern "C" hipError_t qr_launch_pack_r (const QrPackRTask *d_tasks, const long long *d_tile_prefix, int nfronts,
    long long ntiles, hipStream_t stream) ;
struct PackRTask { const double *F ; double *out ; int32_t ld, fn, frank, c0, c1, pad ; } ;     // = QrPackRTask (qr_tasks.h)
extern "C" hipError_t qr_launch_build_s (void *d_S, const long long *d_findex, const long long *d_aperm,
    const double *d_Ax, long long count, hipStream_t stream) ;

static void release_numeric_cache (spqr_amd_symbolic::NumericCache *c)
{
    if (c->d_Rimap) (void) hipFree (c->d_Rimap) ;
    if (c->d_Rjmap) (void) hipFree (c->d_Rjmap) ;
    if (c->d_findex) (void) hipFree (c->d_findex) ;
    if (c->d_aperm) (void) hipFree (c->d_aperm) ;
    c->d_Rimap = c->d_Rjmap = c->d_findex = c->d_aperm = NULL ;
    c->built = false ;
    c->stage_worst_known = false ; c->fit_gen = 0 ;
    c->fronts_key = 0 ; c->fronts_rstack = NULL ;
    hostvec<gpuqr_front> ( ).swap (c->fronts) ;
    c->lay_ok = false ; c->lay_sruns_ok = false ;
    hostvec<int64_t> ( ).swap (c->lay_sruns) ;
    hostvec<int64_t> ( ).swap (c->lay_list) ; hostvec<int64_t> ( ).swap (c->lay_FOff) ; hostvec<int64_t> ( ).swap (c->lay_SOff) ;
    hostvec<int64_t> ( ).swap (c->lay_ROff) ; hostvec<int64_t> ( ).swap (c->lay_rel) ;
}

// Maps, staircases and the destination / source of every S entry: pattern only, so done once per
// symbolic object and device.  Runs the restated spqrgpu_buildAssemblyMaps with the position of
// each entry in place of its value, which gives the permutation from the caller's Ax to packed S.
// Returns 0, 1 (out of memory) or 2 (device error).
// 64-bit fingerprint of a column pattern.  Eight interleaved multiply-xor chains (one chain is bound
// by the latency of its multiply, 9 ms for the 10 M entries of a 1024 x 1024 grid), on up to four
// threads for large patterns; the position of every entry enters through the chain it falls into and
// the order inside the chain.
static uint64_t hash_span (const int64_t *x, Long len, uint64_t seed)
{
    const uint64_t M = 0xff51afd7ed558ccdull ;
    uint64_t h [8] ;
    for (int l = 0 ; l < 8 ; l++) h [l] = seed + 0x9E3779B97F4A7C15ull * (uint64_t) (l + 1) ;
    Long k = 0 ;
    for ( ; k + 8 <= len ; k += 8)
        for (int l = 0 ; l < 8 ; l++) { h [l] = (h [l] ^ (uint64_t) x [k + l]) * M ; h [l] ^= h [l] >> 29 ; }
    for ( ; k < len ; k++) { h [k & 7] = (h [k & 7] ^ (uint64_t) x [k]) * M ; h [k & 7] ^= h [k & 7] >> 29 ; }
    uint64_t r = seed ^ (uint64_t) len ;
    for (int l = 0 ; l < 8 ; l++) { r = (r ^ h [l]) * M ; r ^= r >> 32 ; }
    return r ;
}

static uint64_t pattern_fingerprint (const int64_t *Ap, const int64_t *Ai, Long n, Long anz)
{
    uint64_t h = hash_span (Ap, n + 1, 0x243F6A8885A308D3ull ^ (uint64_t) n) ;
    const int nth = (anz >= ((Long) 1 << 20)) ? 4 : 1 ;
    uint64_t part [4] = {0, 0, 0, 0} ;
    if (nth == 1) part [0] = hash_span (Ai, anz, 1) ;
    else
    {
        std::thread th [3] ;
        const Long chunk = (anz + nth - 1) / nth ;
        int started = 0 ;
        try
        {
            for (int t = 1 ; t < nth ; t++)
            {
                th [t-1] = std::thread ([=, &part] ( ) { part [t] = hash_span (Ai + t * chunk, std::min (chunk, anz - t * chunk), (uint64_t) t + 1) ; }) ;
                started = t ;
            }
        }
        catch (...) { }
        part [0] = hash_span (Ai, chunk, 1) ;
        for (int t = 1 ; t <= started ; t++) th [t-1].join ( ) ;
        for (int t = started + 1 ; t < nth ; t++) part [t] = hash_span (Ai + t * chunk, std::min (chunk, anz - t * chunk), (uint64_t) t + 1) ;
    }
    for (int t = 0 ; t < nth ; t++) { h = (h ^ part [t]) * 0xff51afd7ed558ccdull ; h ^= h >> 32 ; }
    return h ;
}

// Pageable host memory -> device through two page-locked pieces that take turns: the copy of one
// piece into its page-locked buffer overlaps the DMA of the other.  (A plain hipMemcpy from pageable
// memory page-locks the caller's buffer first.  The first copy a process makes on the null stream costs
// about 150 ms either way: the runtime sets its copy queues up then.)
struct StagedUpload
{
    static const size_t piece = (size_t) 8 << 20 ;
    void *pin [2] = {NULL, NULL} ;
    hipEvent_t done [2] = {NULL, NULL} ;
    bool used [2] = {false, false} ;
    int turn = 0 ;
    StagedUpload ( )
    {
        for (int i = 0 ; i < 2 ; i++)
        {
            if (hipHostMalloc (&pin [i], piece, hipHostMallocDefault) != hipSuccess) { pin [i] = NULL ; (void) hipGetLastError ( ) ; }
            if (hipEventCreateWithFlags (&done [i], hipEventDisableTiming) != hipSuccess) done [i] = NULL ;
        }
    }
    bool ok ( ) const { return pin [0] && pin [1] && done [0] && done [1] ; }
    hipError_t send (void *dst, const void *src, size_t bytes)
    {
        for (size_t off = 0 ; off < bytes ; off += piece)
        {
            const size_t len = std::min (piece, bytes - off) ;
            const int i = turn ; turn ^= 1 ;
            hipError_t e ;
            if (used [i] && (e = hipEventSynchronize (done [i])) != hipSuccess) return e ;
            memcpy (pin [i], (const char *) src + off, len) ;
            if ((e = hipMemcpyAsync ((char *) dst + off, pin [i], len, hipMemcpyHostToDevice, 0)) != hipSuccess) return e ;
            if ((e = hipEventRecord (done [i], 0)) != hipSuccess) return e ;
            used [i] = true ;
        }
        return hipSuccess ;
    }
    hipError_t finish ( ) { return hipStreamSynchronize (0) ; }
    ~StagedUpload ( )
    {
        (void) hipStreamSynchronize (0) ;
        for (int i = 0 ; i < 2 ; i++) { if (pin [i]) (void) hipHostFree (pin [i]) ; if (done [i]) (void) hipEventDestroy (done [i]) ; }
    }
} ;

static int build_numeric_cache (const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai)
{
    spqr_amd_symbolic::NumericCache &C = S->cache ;
    int dev = -1 ;
    if (hipGetDevice (&dev) != hipSuccess) return 2 ;
    // Ap == NULL: the caller hands the values of S = A(P,Q) in row form (SPQR's Sx, made by
    // spqr_stranspose2), which are in the symbolic object's own order.
    const bool from_sx = (Ap == NULL) ;
    if (!from_sx && Ap [S->n] != S->anz) return 3 ;             // not the matrix that was analysed
    if (!from_sx && ((Long) S->Qfill.size ( ) < S->n || (Long) S->PLinv.size ( ) < S->m)) return 3 ;   // made from arrays: Sx only
    // The cache is derived from the pattern of the first matrix factorized with this object: a later
    // call with another pattern (same size or not) must not reuse it silently.
    const uint64_t fp = from_sx ? 0x5358ull : pattern_fingerprint (Ap, Ai, S->n, S->anz) ;
    if (C.built && C.device == dev && C.pattern == fp) return 0 ;
    if (C.built) release_numeric_cache (&C) ;
    C.pattern = fp ;
    C.release = release_numeric_cache ;
    const bool verbose = getenv ("GPUQR_VERBOSE") != NULL ;
    double tlast = now_ms ( ) ;
    const Long nf = S->nf, n = S->n, m = S->m, anz = S->anz ;
    const Long *Sp = S->Sp.data ( ), *Super = S->Super.data ( ), *Sleft = S->Sleft.data ( ), *Post = S->Post.data ( ) ;
    // position in Ax of every entry of S = A(P,Q) in row form (spqr_stranspose2's scatter)
    hostvec<Long> sxperm ((size_t) std::max ((Long) 1, anz)) ;
    hostvec<double> Sx ((size_t) std::max ((Long) 1, anz)) ;
    if (from_sx)
    {
        for (Long q = 0 ; q < anz ; q++) { sxperm [q] = q ; Sx [q] = (double) q ; }
    }
    else
    {
        hostvec<Long> w (Sp, Sp + m) ;
        for (Long k = 0 ; k < n ; k++)
        {
            const Long col = S->Qfill [k] ;
            for (Long p = Ap [col] ; p < Ap [col+1] ; p++) sxperm [w [S->PLinv [Ai [p]]]++] = p ;
        }
        for (Long q = 0 ; q < anz ; q++) Sx [q] = (double) q ;      // exact below 2^53
    }
    TICK ("  cache: A -> S permutation") ;
    const Long RimapSize = std::max ((Long) 1, S->RimapSize), RjmapSize = std::max ((Long) 1, S->RjmapSize) ;
    C.Stair.assign ((size_t) std::max ((Long) 1, S->Rp [nf]), 0) ;
    C.Rimap.assign ((size_t) RimapSize, 0) ; C.Rjmap.assign ((size_t) RjmapSize, 0) ;
    hostvec<Long> Fmap ((size_t) std::max ((Long) 1, n)) ;
    hostvec<gpuqr_sentry> allS ((size_t) std::max ((Long) 1, anz)) ;
    spqrgpu_amd_build_assembly_maps (nf, n, Fmap.data ( ), Post, Super, S->Rp.data ( ), S->Rj.data ( ), Sleft, Sp,
        S->Sj.data ( ), Sx.data ( ), S->Fm.data ( ), S->Cm.data ( ), S->Childp.data ( ), S->Child.data ( ), C.Stair.data ( ),
        C.Rjmap.data ( ), S->RjmapOffsets.data ( ), C.Rimap.data ( ), S->RimapOffsets.data ( ), allS.data ( )) ;
    TICK ("  cache: assembly maps") ;
    hostvec<Long> findex ((size_t) std::max ((Long) 1, anz)), aperm ((size_t) std::max ((Long) 1, anz)) ;
    {
        const Long chunk = (Long) 1 << 18 ;
        parallel_for ((anz + chunk - 1) / chunk, [&] (Long c)
        {
            for (Long k = c * chunk ; k < std::min (anz, (c + 1) * chunk) ; k++)
                { findex [k] = allS [k].findex ; aperm [k] = sxperm [(Long) allS [k].value] ; }
        }, 2) ;
    }
    C.SGlob.assign ((size_t) nf + 1, 0) ;
    {
        Long sacc = 0 ;
        for (Long k = 0 ; k < nf ; k++)
        {
            const Long f = Post [k] ;
            C.SGlob [f] = sacc ;
            sacc += std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ;
        }
        C.SGlob [nf] = sacc ;
    }
    TICK ("  cache: findex, aperm") ;
    const size_t nz = (size_t) std::max ((Long) 1, anz) ;
    if (hipMalloc (&C.d_Rimap, (size_t) RimapSize * sizeof (int32_t)) != hipSuccess) { C.d_Rimap = NULL ; release_numeric_cache (&C) ; return 1 ; }
    if (hipMalloc (&C.d_Rjmap, (size_t) RjmapSize * sizeof (int32_t)) != hipSuccess) { C.d_Rjmap = NULL ; release_numeric_cache (&C) ; return 1 ; }
    if (hipMalloc (&C.d_findex, nz * sizeof (Long)) != hipSuccess) { C.d_findex = NULL ; release_numeric_cache (&C) ; return 1 ; }
    if (hipMalloc (&C.d_aperm, nz * sizeof (Long)) != hipSuccess) { C.d_aperm = NULL ; release_numeric_cache (&C) ; return 1 ; }
    TICK ("  cache: device allocations") ;
    {
        StagedUpload up ;
        TICK ("  cache: staging buffers") ;
        if (!up.ok ( )
         || up.send (C.d_Rimap, C.Rimap.data ( ), (size_t) RimapSize * sizeof (int32_t)) != hipSuccess
         || up.send (C.d_Rjmap, C.Rjmap.data ( ), (size_t) RjmapSize * sizeof (int32_t)) != hipSuccess
         || up.send (C.d_findex, findex.data ( ), nz * sizeof (Long)) != hipSuccess
         || up.send (C.d_aperm, aperm.data ( ), nz * sizeof (Long)) != hipSuccess
         || up.finish ( ) != hipSuccess)
            { release_numeric_cache (&C) ; return 2 ; }
    }
    TICK ("  cache: upload") ;
    C.device = dev ;
    C.built = true ;
    return 0 ;
}

// The page-locked R stack of the last freed numeric object is kept for the next factorization (one
// per process, until spqrgpu_amd_finalize): page-locking 25 GB for the stack of a 96^3 grid costs
// 30-100 ms per factorization, 900 ms the first time.
#include <mutex>
#include <atomic>
// Host memory this driver holds page-locked for R stacks at this moment, whole stacks and locked ends of
// pageable ones, kept ones included: what the limit below is held against.
static std::atomic<long long> g_locked_now (0) ;
struct PinnedStackCache
{
    std::mutex lock ;
    void *kept = NULL ; size_t kept_bytes = 0 ;
    std::vector<std::pair<void *, size_t> > lent ;     // stacks handed out that are larger than their numeric object says
    void *take (size_t bytes)
    {
        std::lock_guard<std::mutex> g (lock) ;
        if (!kept || kept_bytes < bytes || kept_bytes > 2 * bytes + ((size_t) 64 << 20)) return NULL ;
        void *p = kept ;
        try { lent.push_back (std::make_pair (p, kept_bytes)) ; } catch (...) { return NULL ; }
        kept = NULL ; kept_bytes = 0 ;
        return p ;
    }
    void give (void *p, size_t bytes)
    {
        void *drop = NULL ; size_t drop_bytes = 0 ;
        {
            std::lock_guard<std::mutex> g (lock) ;
            for (size_t i = 0 ; i < lent.size ( ) ; i++)
                if (lent [i].first == p) { bytes = lent [i].second ; lent.erase (lent.begin ( ) + i) ; break ; }
            drop = kept ; drop_bytes = kept_bytes ;
            kept = p ; kept_bytes = bytes ;
        }
        if (drop) { (void) hipHostFree (drop) ; g_locked_now -= (long long) drop_bytes ; }
    }
    void clear ( )
    {
        void *drop = NULL ; size_t drop_bytes = 0 ;
        { std::lock_guard<std::mutex> g (lock) ; drop = kept ; drop_bytes = kept_bytes ; kept = NULL ; kept_bytes = 0 ; }
        if (drop) { (void) hipHostFree (drop) ; g_locked_now -= (long long) drop_bytes ; }
    }
} ;
static PinnedStackCache g_pinned_stack ;

// The same for pageable stacks of 1 GiB and more (what a stack beyond the page-locking limit is): the
// 70 GB of a 128^3 grid cost about a second per factorization in page faults on first touch and in
// giving the pages back.  These stacks come from malloc directly, not through the allocator hooks, so
// that one kept across a change of hooks is freed by what allocated it.
struct PageableStackCache
{
    // (GPUQR_KEEP_PAGEABLE_FROM_MB: testing, the size from which a pageable stack is one of these)
    static size_t min_bytes (void)
    {
        static size_t v = 0 ; static bool read = false ;
        if (!read) { const char *e = getenv ("GPUQR_KEEP_PAGEABLE_FROM_MB") ; v = e ? (size_t) atol (e) << 20 : (size_t) 1 << 30 ; read = true ; }
        return v ;
    }
    struct Item { void *p ; size_t bytes ; char *locked ; size_t locked_bytes ; } ;
    std::mutex lock ;
    Item kept = {NULL, 0, NULL, 0} ;
    std::vector<Item> lent ;
    static void let_go (Item &it)
    {
        if (it.locked) { (void) hipHostUnregister (it.locked) ; (void) hipGetLastError ( ) ; g_locked_now -= (long long) it.locked_bytes ; }
        if (it.p) free (it.p) ;
        it = Item {NULL, 0, NULL, 0} ;
    }
    void *get (size_t bytes)
    {
        std::lock_guard<std::mutex> g (lock) ;
        Item it = {NULL, 0, NULL, 0} ;
        if (kept.p && kept.bytes >= bytes && kept.bytes <= 2 * bytes + ((size_t) 64 << 20)) { it = kept ; kept = Item {NULL, 0, NULL, 0} ; }
        else { it.p = malloc (std::max (bytes, (size_t) 1)) ; it.bytes = bytes ; }
        if (!it.p) return NULL ;
        try { lent.push_back (it) ; } catch (...) { let_go (it) ; return NULL ; }
        return it.p ;
    }
    bool give (void *p)                 // false: not one of these stacks
    {
        Item drop = {NULL, 0, NULL, 0} ;
        {
            std::lock_guard<std::mutex> g (lock) ;
            size_t i = 0 ;
            while (i < lent.size ( ) && lent [i].p != p) i++ ;
            if (i == lent.size ( )) return false ;
            drop = kept ;
            kept = lent [i] ;
            lent.erase (lent.begin ( ) + i) ;
        }
        let_go (drop) ;
        return true ;
    }
    // Page-lock [lo, lo + bytes) of a stack that is out on loan, so that packed R can be sent there by DMA
    // from inside the engine.  A range locked before stays locked if it is the same one (the stack of
    // the factorization before, same problem), and is let go otherwise.
    bool lock_range (void *stack, char *lo, size_t bytes)
    {
        Item it = {NULL, 0, NULL, 0} ;
        {
            std::lock_guard<std::mutex> g (lock) ;
            for (size_t i = 0 ; i < lent.size ( ) ; i++) if (lent [i].p == stack) it = lent [i] ;
        }
        if (!it.p) return false ;
        if (it.locked == lo && it.locked_bytes == bytes) return true ;
        if (it.locked)
        {
            (void) hipHostUnregister (it.locked) ; (void) hipGetLastError ( ) ;
            g_locked_now -= (long long) it.locked_bytes ;
            it.locked = NULL ; it.locked_bytes = 0 ;
        }
        if (bytes > 0 && hipHostRegister (lo, bytes, hipHostRegisterDefault) == hipSuccess)
        {
            it.locked = lo ; it.locked_bytes = bytes ;
            g_locked_now += (long long) bytes ;
        }
        else (void) hipGetLastError ( ) ;
        {
            std::lock_guard<std::mutex> g (lock) ;
            for (size_t i = 0 ; i < lent.size ( ) ; i++) if (lent [i].p == stack) lent [i] = it ;
        }
        return it.locked != NULL ;
    }
    size_t locked_of (void *stack)
    {
        std::lock_guard<std::mutex> g (lock) ;
        for (size_t i = 0 ; i < lent.size ( ) ; i++) if (lent [i].p == stack) return lent [i].locked_bytes ;
        return 0 ;
    }
    void unlock_kept ( )                // the kept stack stays, what was page-locked of it is let go
    {
        std::lock_guard<std::mutex> g (lock) ;
        if (kept.locked)
        {
            (void) hipHostUnregister (kept.locked) ; (void) hipGetLastError ( ) ;
            g_locked_now -= (long long) kept.locked_bytes ;
            kept.locked = NULL ; kept.locked_bytes = 0 ;
        }
    }
    void clear ( )
    {
        Item drop = {NULL, 0, NULL, 0} ;
        { std::lock_guard<std::mutex> g (lock) ; drop = kept ; kept = Item {NULL, 0, NULL, 0} ; }
        let_go (drop) ;
    }
} ;
static PageableStackCache g_pageable_stack ;

// The largest R stack this driver asks the system to page-lock (GPUQR_PIN_LIMIT_GIB, default 48); a
// larger one is pageable and filled through bounce buffers.  Systems refuse pinned memory beyond some
// total (64 GB on the machine this was developed on), and an allocation just above that total does not
// fail: 70.4 GB were granted after a second, and the process then died inside the runtime when its
// first copy went to create a hardware queue (the 87 GB of a larger problem are refused at once and
// harmlessly).  Asking only for what leaves the runtime room avoids both.
static size_t pin_limit_bytes (void)
{
    static size_t limit = 0 ;
    if (limit == 0)
    {
        const char *e = getenv ("GPUQR_PIN_LIMIT_GIB") ;
        const double gib = (e && atof (e) > 0) ? atof (e) : 48.0 ;
        limit = (size_t) (gib * 1073741824.0) ;
    }
    return limit ;
}

// Is there room for `bytes` more of page-locked memory under the limit?  Stacks that are only kept for a
// next factorization make way first.
static bool pin_room (size_t bytes)
{
    const long long limit = (long long) pin_limit_bytes ( ) ;
    if ((long long) bytes + g_locked_now.load ( ) <= limit) return true ;
    g_pinned_stack.clear ( ) ;
    g_pageable_stack.unlock_kept ( ) ;
    return (long long) bytes + g_locked_now.load ( ) <= limit ;
}

extern "C" void spqr_amd_numeric_free (spqr_amd_numeric *num)
{
    if (!num) return ;
    if (num->Rstack && !num->RstackBorrowed)
    {
        if (num->RstackPinned) g_pinned_stack.give (num->Rstack, (size_t) std::max ((Long) 1, num->rsize) * sizeof (double)) ;
        else if (!g_pageable_stack.give (num->Rstack)) host_free (num->Rstack) ;
    }
    host_free (num->Rblock) ; host_free (num->Rdead) ; host_free (num->Cstack) ; host_free (num->Cblock) ;
    host_free (num) ;
}

// Grow-only device / pinned buffer
struct Buf
{
    void *p = NULL ; size_t cap = 0 ; bool pinned = false ;
    hipError_t ensure (size_t bytes)
    {
        if (bytes <= cap) return hipSuccess ;
        release ( ) ;
        bytes = std::max (bytes, (size_t) 256) ;
        hipError_t e = pinned ? hipHostMalloc (&p, bytes, hipHostMallocDefault) : hipMalloc (&p, bytes) ;
        if (e != hipSuccess) { p = NULL ; return e ; }
        cap = bytes ;
        return hipSuccess ;
    }
    void release ( ) { if (p) { if (pinned) (void) hipHostFree (p) ; else (void) hipFree (p) ; } p = NULL ; cap = 0 ; }
    ~Buf ( ) { release ( ) ; }
} ;

struct Workspaces { Buf dF, dS, dR, dR2, dAx, dTasks, dPrefix, dLimbo, hBounce [2] ; } ;

// Where contribution blocks wait on the device between groups: pieces of one grow-only buffer handed
// out first-fit.  The same sequence of requests gets the same addresses in every factorization, so the
// front descriptors of a later group (which name the waiting blocks of its children) are the same
// from call to call and the engine finds the schedule it made for them (with a hipMalloc per block
// the groups after the first were planned again in every factorization: 0.3 s each at 128^3).
#include <map>
struct DevArena
{
    char *base = NULL ; size_t cap = 0 ;
    std::map<size_t, size_t> free_at ;          // offset -> length of every free piece
    void reset (void *p, size_t bytes) { base = (char *) p ; cap = p ? bytes : 0 ; free_at.clear ( ) ; if (cap) free_at [0] = cap ; }
    static size_t rounded (size_t bytes) { return (bytes + 255) & ~(size_t) 255 ; }
    bool owns (const void *p) const { return base && (const char *) p >= base && (const char *) p < base + cap ; }
    void *take (size_t bytes)
    {
        bytes = rounded (bytes) ;
        for (auto it = free_at.begin ( ) ; it != free_at.end ( ) ; ++it)
        {
            if (it->second < bytes) continue ;
            const size_t off = it->first, len = it->second ;
            free_at.erase (it) ;
            if (len > bytes) free_at [off + bytes] = len - bytes ;
            return base + off ;
        }
        return NULL ;
    }
    void give (void *p, size_t bytes)
    {
        size_t off = (size_t) ((char *) p - base), len = rounded (bytes) ;
        auto next = free_at.lower_bound (off) ;
        if (next != free_at.begin ( ))
        {
            auto prev = std::prev (next) ;
            if (prev->first + prev->second == off) { off = prev->first ; len += prev->second ; free_at.erase (prev) ; }
        }
        if (next != free_at.end ( ) && off + len == next->first) { len += next->second ; free_at.erase (next) ; }
        free_at [off] = len ;
    }
} ;
// Goes up whenever the driver gives device memory back: what was seen to fit before may not fit now.
static uint64_t g_ws_gen = 1 ;

// Device -> pageable host memory through two pinned bounce buffers: the DMA of one piece overlaps
// the host threads copying the previous piece out (a direct copy into pageable memory ran at
// 2.5 GB/s for the 87 GB of R of the 128^3 problem).
static hipError_t copy_to_pageable (void *dst, const void *src_dev, size_t bytes, Buf bounce [2], hipStream_t stream = 0)
{
    const size_t piece = (size_t) 256 << 20 ;
    hipError_t e ;
    for (int b = 0 ; b < 2 ; b++) { bounce [b].pinned = true ; if ((e = bounce [b].ensure (piece)) != hipSuccess) return e ; }
    size_t done = 0 ;
    int cur = 0 ;
    size_t len_prev = 0 ; char *dst_prev = NULL ; int prev = -1 ;
    while (done < bytes || prev >= 0)
    {
        size_t len = 0 ;
        if (done < bytes)
        {
            len = std::min (piece, bytes - done) ;
            if ((e = hipMemcpyAsync (bounce [cur].p, (const char *) src_dev + done, len, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e ;
        }
        if (prev >= 0)
        {
            // the previous piece has landed (its copy was waited for below): spread it over the host threads
            const size_t nchunk = (len_prev + ((size_t) 4 << 20) - 1) / ((size_t) 4 << 20) ;
            const char *sp = (const char *) bounce [prev].p ;
            parallel_for ((Long) nchunk, [&] (Long k)
            {
                const size_t o = (size_t) k * ((size_t) 4 << 20) ;
                memcpy (dst_prev + o, sp + o, std::min ((size_t) 4 << 20, len_prev - o)) ;
            }, 1) ;
            prev = -1 ;
        }
        if (len > 0)
        {
            if ((e = hipStreamSynchronize (stream)) != hipSuccess) return e ;
            prev = cur ; len_prev = len ; dst_prev = (char *) dst + done ;
            done += len ; cur ^= 1 ;
        }
    }
    return hipSuccess ;
}
#define MAX_WS_DEVICES 64
static Workspaces g_ws [MAX_WS_DEVICES] ;

// Frees the cached device workspaces of every device (they are grow-only between calls).
extern "C" void spqrgpu_amd_finalize (void)
{
    int cur = -1 ;
    (void) hipGetDevice (&cur) ;
    for (int d = 0 ; d < MAX_WS_DEVICES ; d++)
    {
        Workspaces &W = g_ws [d] ;
        if (!W.dF.p && !W.dS.p && !W.dR.p && !W.dR2.p && !W.dAx.p && !W.dTasks.p && !W.dPrefix.p && !W.dLimbo.p) continue ;
        if (hipSetDevice (d) != hipSuccess) continue ;
        W.dF.release ( ) ; W.dS.release ( ) ; W.dR.release ( ) ; W.dR2.release ( ) ; W.dAx.release ( ) ; W.dTasks.release ( ) ; W.dPrefix.release ( ) ;
        W.dLimbo.release ( ) ;
        W.hBounce [0].release ( ) ; W.hBounce [1].release ( ) ;
    }
    g_pinned_stack.clear ( ) ;
    g_pageable_stack.clear ( ) ;
    g_ws_gen++ ;
    if (cur >= 0) (void) hipSetDevice (cur) ;
}

struct gpuqr_replay ;
struct QrRowCopy { const double *src ; double *dst ; } ;        // = qr_tasks.h
extern "C" hipError_t qr_launch_copy_rows (const QrRowCopy *d_tasks, long long ntasks, int ncols,
    long long sstride, long long dstride, hipStream_t stream) ;
struct ColNormTask { const double *x ; int64_t nrows ; int64_t ld ; } ;       // = QrColNormTask of qr_tasks.h
extern "C" hipError_t qr_launch_col_norms (const ColNormTask *d_tasks, long long ntasks, double *d_out, hipStream_t stream) ;

// A factorization that keeps H (spqrgpu_amd_factorize_keep): the front arena with the Householder
// vectors in place and a right-hand-side block of nrhs columns behind every front, the engine's
// replay object (VT tiles, update tasks of the blocks), and the three row lists that take a
// right-hand side through the tree: rows of B into the blocks, finished rows of a child into its
// parent's block (by the round after which they are final), rows of Q'B out.
struct spqr_amd_qt
{
    int device = -1 ;
    Long m = 0, n = 0, nrhs = 0, rank = 0 ;
    Buf dF, dB, dC, dScatter, dPush, dGather ;
    gpuqr_replay *rp = NULL ;
    Long nscatter = 0, ngather = 0 ;
    hostvec<Long> push_ptr ;        // pushes that follow round r: [push_ptr [r], push_ptr [r+1])
    ~spqr_amd_qt ( ) { if (rp) gpuqr_replay_free (rp) ; }
} ;

// What a factorization with a tolerance knows after its first pass: which pivot columns are dead.
// A front with dead columns is factorized again with its columns in the order
// [live pivot columns | non-pivot columns | dead pivot columns]: the live ones are reduced first, on
// consecutive rows (which is SPQR's "squeezed" R, spqr_front.cpp:424-479: a dead column does not use
// up a row), the contribution block sits right behind them, and the dead columns ride along as
// columns that are updated but whose pivots nobody reads.
struct RankPlan
{
    hostvec<char> dead ;        // size n: pivot column found dead at the tolerance (cumulative over passes)
    hostvec<char> Rdead ;       // size n: dead, or a live pivot column past the bottom of its front
    hostvec<char> touched ;     // per front: has dead columns (its columns are permuted)
    hostvec<Long> live ;        // per front: live pivot columns
    hostvec<Long> colpos ;      // per front column (at Rp [f] + j): position in the permuted front
    hostvec<Long> rsize ;       // per front: entries of its squeezed, column-packed R block
    hostvec<Long> Stair ;       // staircases; full for touched fronts
    // size n, filled by the pass that runs with this plan: for a column in `dead`, the 2-norm of its
    // rows from the would-be pivot row down, i.e. the number the CPU rule compares with the tolerance
    mutable hostvec<double> deadnorm ;
    void *d_findex = NULL, *d_Rjmap = NULL ;
    ~RankPlan ( ) { if (d_findex) (void) hipFree (d_findex) ; if (d_Rjmap) (void) hipFree (d_Rjmap) ; }
} ;

// Walk the pivot columns of front f in their original order, as spqr_rcount does
// (SPQR/Source/spqr_rcount.cpp): body (k, rm, is_live) is called with rm = rows of R the column
// holds (for a live column including its own pivot row).
template <class Body> static inline void walk_pivots (Long fm, Long fp, Long col1, const char *dead, Body body)
{
    Long rm = 0 ;
    for (Long k = 0 ; k < fp ; k++)
    {
        const bool live = !(dead && dead [col1 + k]) && rm < fm ;
        if (live) rm++ ;
        body (k, rm, live) ;
    }
}

// The numeric factorization of a set of fronts, group by group.  Mirrors
// spqrgpu_kernel.cpp:63-782: build S and the maps, then per group zero the
// front arena, ship S, describe the fronts (plus push-only fronts for
// contribution blocks that come from outside the group), run the engine, and
// hand R back.
//
// Groups.  mine == NULL: all fronts, one group per stage of the staging.
// mine != NULL (one process's share of the tree): the fronts with mine[f] != 0,
// cut into groups that fit the declared device memory the way the staging cuts
// the whole tree (in postorder, a group is closed when the next front with the
// blocks it needs from outside no longer fits; gpuMemorySize = 0 puts every
// front in a group of its own).
//
// Contribution blocks.  A block whose parent is factorized by a later group of
// this call waits in limbo: on the device when it has the room, on the host
// otherwise.  A block whose parent is not in this call at all leaves through
// c_out_dev (device, compact cm-by-cn, at Cblock[f]) or, when that is NULL,
// num->Cstack (host).  A block that arrives from outside the call comes from
// limbo_in[c]: a device address used where it lies (limbo_in_dev), or host
// memory that is copied in.
//
// Device layout of a group, chosen here (the staging's FOffsets price the memory
// but do not bind the placement): fronts one after another in postorder, each
// with its rows padded to an even length so that every row starts on a 16-byte
// boundary, each followed by the compact contribution blocks of those children
// whose blocks have to be copied in.  The arenas are sized for the largest group
// before the first one runs, so that blocks waiting on the device can never be
// in the way of a later, larger arena.  R never visits the host unpacked: a kernel
// writes the column-packed upper trapezoids of all fronts of the group into one
// device block laid out like the final stack.
static int factorize_impl
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    const char *mine, const double *const *limbo_in, const bool limbo_in_dev, double *c_out_dev,
    double *Rstack_user, int64_t Rstack_room, const RankPlan *rp, spqr_amd_qt *keep,
    spqr_amd_numeric **num_out
)
{
    // Ap == NULL: Ax holds the values of S in row form (see build_numeric_cache)
    *num_out = NULL ;
    int code = 0 ;
    const bool verbose = getenv ("GPUQR_VERBOSE") != NULL ;
    double tlast = now_ms ( ) ;
    const Long nf = S->nf, n = S->n ;
    const Long *Super = S->Super.data ( ), *Rp = S->Rp.data ( ), *Post = S->Post.data ( ) ;
    const Long *Parent = S->Parent.data ( ), *Childp = S->Childp.data ( ), *Child = S->Child.data ( ) ;
    const Long *Fm = S->Fm.data ( ), *Cm = S->Cm.data ( ), *Sleft = S->Sleft.data ( ), *Sp = S->Sp.data ( ) ;

    spqr_amd_numeric *num = (spqr_amd_numeric *) host_calloc (1, sizeof (spqr_amd_numeric)) ;
    if (!num) return 1 ;
    // device workspaces live in a per-device cache between calls: allocating and freeing tens of
    // gigabytes costs seconds (5-6 s per call at 96^3), far more than anything else on the host side
    int wdev = 0 ;
    if (hipGetDevice (&wdev) != hipSuccess || wdev < 0 || wdev >= MAX_WS_DEVICES) { host_free (num) ; return 2 ; }
    Workspaces &WS = g_ws [wdev] ;
    // (a factorization that keeps H owns its front arena: the fronts must outlive the call)
    Buf &dF = keep ? keep->dF : WS.dF, &dS = WS.dS, &dAx = WS.dAx, &dTasks = WS.dTasks, &dPrefix = WS.dPrefix ;
    // A pageable R stack is filled through bounce buffers; between groups that copy runs on a thread of
    // its own beside the next group, which then packs its R into the other of two device blocks.
    std::thread r_thread ;
    hipError_t r_thread_err = hipSuccess ;
    const void *r_thread_src = NULL ;   // the device block the helper thread is copying out of
    auto join_r = [&] ( ) { if (r_thread.joinable ( )) r_thread.join ( ) ; r_thread_src = NULL ; return r_thread_err ; } ;
    hostvec<gpuqr_front> fronts ;
    hostvec<double *> limbo ;           // contribution blocks waiting on the host for a later group
    hostvec<double *> limbo_dev ;       // ... or on the device, when it has the room (no PCIe round trip)
    hostvec<double *> used_dev ;        // device limbo blocks consumed by the running group
    hostvec<size_t> used_bytes ;
    DevArena arena ;                    // the waiting blocks' share of the device (see DevArena)
    // A pageable stack (one beyond the page-locking limit) with its last groups' share page-locked: those
    // groups send their R from inside the engine, the groups before them through the bounce buffers.
    bool stack_kept_kind = false ;      // the stack is one of g_pageable_stack's
    Long tail_first = -1 ;              // first group whose R lies in the page-locked end of the stack (-1: none)
    bool tail_locked = false ;
    std::thread lock_thread ;
    bool registered = false ;           // the caller's stack was page-locked by this call
    // A large page-locked stack is allocated by a helper thread while this one builds the maps and lays
    // the fronts out (page-locking runs at about 28 GB/s: 0.9 s for the 25 GB of a 96^3 grid).
    struct PinJob
    {
        std::thread th ; void *ptr = NULL ; bool ok = false, started = false ;
        void start (size_t bytes)
        {
            int dev = 0 ;
            if (hipGetDevice (&dev) != hipSuccess) return ;
            try
            {
                th = std::thread ([this, dev, bytes] ( )
                {
                    ok = hipSetDevice (dev) == hipSuccess && hipHostMalloc (&ptr, bytes, hipHostMallocDefault) == hipSuccess ;
                    if (!ok) { ptr = NULL ; (void) hipGetLastError ( ) ; }
                }) ;
                started = true ;
            }
            catch (...) { started = false ; }
        }
        void join ( ) { if (th.joinable ( )) th.join ( ) ; }
        ~PinJob ( ) { join ( ) ; if (ptr) (void) hipHostFree (ptr) ; }     // (ptr is taken over on success)
    } pin_job ;

    try
    {
        if (!mine && S->numStages <= 0 && nf > 0) { code = 2 ; goto done ; }
        if (mine && S->gpuMemorySize == 1 && nf > 0) { code = 2 ; goto done ; }    // the "nothing fits" test value

        // ---- what this call factorizes, and the size of its R ------------------------------
        auto S_count = [&] (Long f) { return std::max ((Long) 0, Sp [Sleft [Super [f+1]]] - Sp [Sleft [Super [f]]]) ; } ;
        auto is_mine = [&] (Long f) { return mine == NULL || mine [f] != 0 ; } ;
        // pivot columns a front really has, the rows of its R block, the rows of its contribution block
        auto npiv_of = [&] (Long f) { return rp ? rp->live [f] : (Super [f+1] - Super [f]) ; } ;
        auto rank_of = [&] (Long f) { return std::min (Fm [f], npiv_of (f)) ; } ;
        auto cm_of = [&] (Long f)
        {
            const Long cn = (Rp [f+1] - Rp [f]) - (Super [f+1] - Super [f]) ;
            return std::min (Cm [f], std::min (std::max (Fm [f] - rank_of (f), (Long) 0), cn)) ;
        } ;
        auto rsize_of = [&] (Long f) { return rp ? rp->rsize [f] : r_block_size (Rp [f+1] - Rp [f], rank_of (f)) ; } ;
        auto c_size = [&] (Long c) { return cm_of (c) * ((Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c])) ; } ;
        Long rtotal = 0, ctotal = 0 ;
        for (Long f = 0 ; f < nf ; f++)
        {
            if (!is_mine (f)) continue ;
            const Long fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
            rtotal += rsize_of (f) ;
            if (mine && Parent [f] < nf && !mine [Parent [f]]) ctotal += cm_of (f) * (fn - fp) ;
        }
        num->nf = nf ;
        num->rsize = rtotal ;
        // pinned, so that the packed R arrives by DMA; pageable if the system refuses that much pinned memory
        num->RstackPinned = 1 ;
        if (Rstack_user)
        {
            // the caller's stack (SPQR's Work[0].Stack_head): page-locked in place for the duration of
            // the call when the system allows it, filled through bounce buffers otherwise
            if (Rstack_room < rtotal) { code = 1 ; goto done ; }
            num->Rstack = Rstack_user ;
            num->RstackBorrowed = 1 ;
            if (rtotal > 0 && getenv ("GPUQR_PAGEABLE_R") == NULL
                && pin_room ((size_t) rtotal * sizeof (double))
                && hipHostRegister (Rstack_user, (size_t) rtotal * sizeof (double), hipHostRegisterDefault) == hipSuccess)
            {
                registered = true ;
                g_locked_now += (long long) ((size_t) rtotal * sizeof (double)) ;
            }
            else { (void) hipGetLastError ( ) ; num->RstackPinned = 0 ; }
        }
        else
        if (getenv ("GPUQR_PAGEABLE_R") == NULL
            && (num->Rstack = (double *) g_pinned_stack.take ((size_t) std::max ((Long) 1, rtotal) * sizeof (double))) != NULL)
        {
            // the page-locked stack of an earlier factorization
        }
        else
        if (getenv ("GPUQR_PAGEABLE_R") == NULL && (size_t) rtotal * sizeof (double) >= ((size_t) 256 << 20)
            && pin_room ((size_t) rtotal * sizeof (double))
            && (pin_job.start ((size_t) rtotal * sizeof (double)), pin_job.started))
        {
            // (joined below, before the first use of the stack)
        }
        else
        if (getenv ("GPUQR_PAGEABLE_R") != NULL       // test knob: take the fallback path
            || !pin_room ((size_t) std::max ((Long) 1, rtotal) * sizeof (double))
            || hipHostMalloc ((void **) &num->Rstack, (size_t) std::max ((Long) 1, rtotal) * sizeof (double), hipHostMallocDefault) != hipSuccess)
        {
            (void) hipGetLastError ( ) ;
            num->RstackPinned = 0 ;
            const size_t rbytes = (size_t) std::max ((Long) 1, rtotal) * sizeof (double) ;
            stack_kept_kind = rbytes >= PageableStackCache::min_bytes ( ) ;
            num->Rstack = (double *) (stack_kept_kind ? g_pageable_stack.get (rbytes) : host_malloc (rbytes)) ;
            if (!num->Rstack) { code = 1 ; goto done ; }
        }
        else g_locked_now += (long long) ((size_t) std::max ((Long) 1, rtotal) * sizeof (double)) ;     // (page-locked here and now)
        num->Rblock = (Long *) host_malloc ((size_t) std::max ((Long) 1, nf) * sizeof (Long)) ;
        num->Rdead = (char *) host_calloc ((size_t) std::max ((Long) 1, n), 1) ;
        if (!num->Rblock || !num->Rdead) { code = 1 ; goto done ; }
        for (Long f = 0 ; f < nf ; f++) num->Rblock [f] = -1 ;
        if (mine)
        {
            if (!c_out_dev)
            {
                num->Cstack = (double *) host_calloc ((size_t) std::max ((Long) 1, ctotal), sizeof (double)) ;
                if (!num->Cstack) { code = 1 ; goto done ; }
            }
            num->Cblock = (Long *) host_malloc ((size_t) std::max ((Long) 1, nf) * sizeof (Long)) ;
            if (!num->Cblock) { code = 1 ; goto done ; }
            for (Long f = 0 ; f < nf ; f++) num->Cblock [f] = -1 ;
        }
        num->csize = ctotal ;

        TICK ("allocation") ;
        // ---- maps, staircases, S pattern: once per symbolic object; this call's values of A ------
        {
            const int cc = build_numeric_cache (S, Ap, Ai) ;        // (a failed host allocation inside it lands in the catch below)
            if (cc != 0) { code = cc ; goto done ; }
        }
        const spqr_amd_symbolic::NumericCache &NC = S->cache ;
        const hostvec<Long> &Stair = NC.Stair ;
        const hostvec<Long> &SGlob = NC.SGlob ;
        TICK ("maps + S pattern (cached)") ;
        if (dAx.ensure ((size_t) std::max ((Long) 1, S->anz) * sizeof (double)) != hipSuccess) { code = 1 ; goto done ; }
        if (S->anz > 0) HIPCHK (hipMemcpyAsync (dAx.p, Ax, (size_t) S->anz * sizeof (double), hipMemcpyHostToDevice, 0)) ;
        TICK ("values to the device") ;

        limbo.assign ((size_t) std::max ((Long) 1, nf), (double *) NULL) ;
        limbo_dev.assign ((size_t) std::max ((Long) 1, nf), (double *) NULL) ;
        hostvec<Long> rel ((size_t) std::max ((Long) 1, nf), -1), pitch ((size_t) std::max ((Long) 1, nf), 0) ;
        // Row pitch: even, so that every row starts on a 16-byte boundary (the update kernel moves 16 B
        // per lane); from GPUQR_LINE_ALIGN_FROM columns on a multiple of 16 doubles, so that rows start
        // on 128-byte lines and no row segment of a column block straddles one more line than it fills
        // (measured on the 96^3 grid before this: 2.0 GB of HBM traffic per update launch for 1.64 GB
        // of algorithmic bytes).
        static Long line_from = -1 ;
        if (line_from < 0) { const char *e = getenv ("GPUQR_LINE_ALIGN_FROM") ; line_from = e ? atol (e) : 96 ; }
        hostvec<Long> rhs_col ((size_t) std::max ((Long) 1, nf), 0) ;
        const Long rhs_n = keep ? keep->nrhs : 0 ;
        for (Long f = 0 ; f < nf ; f++)
        {
            const Long fn = Rp [f+1] - Rp [f] ;
            pitch [f] = (line_from > 0 && fn >= line_from) ? ((fn + 15) & ~(Long) 15) : (fn + (fn & 1)) ;
            if (keep)
            {
                // the right-hand-side block sits behind the front's columns, on a 128-byte line
                rhs_col [f] = (fn + 15) & ~(Long) 15 ;
                pitch [f] = rhs_col [f] + ((rhs_n + 15) & ~(Long) 15) ;
            }
        }

        // ---- groups ------------------------------------------------------------------------------
        // Without a subset the groups are the stages of the symbolic analysis (the reference's
        // arithmetic, bit for bit).  That arithmetic prices the reference's scheduler, not this engine,
        // and the declared memory may be all the device has (SPQR/Demo/qrdemo_gpu.cpp:34-35 declares
        // what cudaMemGetInfo reports): when the largest stage does not fit what the device can give
        // here and now, the fronts are cut again, greedily in postorder like a subset's, with every
        // front priced by what this driver and this engine really allocate for it.  Contribution blocks
        // then cross more group boundaries; R does not change (the arithmetic of a front does not depend
        // on its group).
        hostvec<Long> grp ((size_t) std::max ((Long) 1, nf), -1) ;
        Long ngroups = 0 ;
        bool by_stage = !mine ;          // groups = stages: the fronts of a group are Post [Stagingp [g] .. Stagingp [g+1])
        auto front_need = [&] (Long f)
        {
            // the front, its S entries, its packed R (two blocks take turns), the engine's records and tiles
            return (size_t) (Fm [f] * pitch [f]) * 8 + (size_t) S_count (f) * 16 + (size_t) rsize_of (f) * 16
                + (size_t) gpuqr_front_workspace_bytes (Fm [f], Rp [f+1] - Rp [f]) ;
        } ;
        auto greedy_groups = [&] (size_t mem, bool one_each)
        {
            size_t acc = 0 ;
            Long g = 0 ;
            bool open = false ;
            for (Long k = 0 ; k < nf ; k++)
            {
                const Long f = Post [k] ;
                grp [f] = -1 ;
                if (mine && !mine [f]) continue ;
                size_t need = front_need (f) ;
                size_t kids_here = 0 ;      // blocks of children in the open group: they would have to be copied in
                for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
                {
                    const Long c = Child [cp] ;
                    if (!((!mine || mine [c]) && grp [c] == g && open)) need += (size_t) c_size (c) * 8 ;
                    else kids_here += (size_t) c_size (c) * 8 ;
                }
                if (open && (one_each || acc + need >= mem))
                {
                    g++ ; acc = 0 ; need += kids_here ;
                }
                grp [f] = g ; acc += need ; open = true ;
            }
            return open ? g + 1 : (Long) 0 ;
        } ;
        if (!mine)
        {
            for (Long f = 0 ; f < nf ; f++) grp [f] = S->StageMap [f] ;
            ngroups = S->numStages ;
            // what the largest stage takes, against what the device has for it (worked out once per
            // symbolic object; the device is asked again only after workspaces were given back)
            spqr_amd_symbolic::NumericCache &NCm = S->cache ;
            static const char *force = getenv ("GPUQR_REGROUP_BYTES") ;       // test knob: regroup at this many bytes
            if (!(NCm.fit_gen == g_ws_gen && !rp && !keep && !force))
            {
                if (!NCm.stage_worst_known || rp || keep)
                {
                    size_t worst = 0 ;
                    hostvec<size_t> per ((size_t) std::max ((Long) 1, ngroups), (size_t) 0) ;
                    for (Long f = 0 ; f < nf ; f++)
                    {
                        const Long g = grp [f] ;
                        if (g < 0 || g >= ngroups) continue ;
                        per [g] += front_need (f) ;
                        const Long pid = Parent [f] ;
                        if (pid < nf && grp [pid] != g) per [grp [pid]] += (size_t) c_size (f) * 8 ;   // waits, then is copied in
                    }
                    for (Long g = 0 ; g < ngroups ; g++) worst = std::max (worst, per [g]) ;
                    NCm.stage_worst = worst ;
                    NCm.stage_worst_known = !(rp || keep) ;
                }
                const size_t worst = NCm.stage_worst ;
                size_t free_b = 0, total_b = 0 ;
                if (hipMemGetInfo (&free_b, &total_b) != hipSuccess) { (void) hipGetLastError ( ) ; free_b = total_b = 0 ; }
                // the workspaces this driver already holds are reused, what the engine holds is not counted
                // (it may have to grow); a margin for the runtime's own needs and for blocks waiting on the device
                const size_t held = dF.cap + dS.cap + WS.dR.cap + WS.dR2.cap ;
                const size_t margin = (size_t) 1 << 30 ;
                size_t room = free_b + held ;
                room = (room > margin) ? room - margin : 0 ;
                if (force && atof (force) > 0) room = std::min (room, (size_t) atof (force)) ;
                if (total_b > 0 && nf > 0 && worst > room)
                {
                    if (verbose) fprintf (stderr, "  [spqrgpu] largest stage needs %.2f GB, the device has %.2f GB: fronts grouped again\n",
                        (double) worst / 1e9, (double) room / 1e9) ;
                    by_stage = false ;
                    ngroups = greedy_groups ((size_t) (0.92 * (double) room), false) ;
                }
                else if (!rp && !keep) NCm.fit_gen = g_ws_gen ;
            }
        }
        else
        {
            // what the staging reserves outside the front arena (spqrgpu_computeFrontStaging.cpp:90-118)
            size_t mem = S->gpuMemorySize ;
            const size_t reserve [4] = { 2 * gpuqr_max_queue_size (mem) * 104, (size_t) S->RimapSize * sizeof (int),
                (size_t) S->RjmapSize * sizeof (int), (size_t) 1024 * 1024 } ;
            for (int k = 0 ; k < 4 ; k++) if (mem > 1) mem = (mem > reserve [k]) ? (mem - reserve [k]) : 0 ;
            ngroups = greedy_groups (mem, S->gpuMemorySize == 0) ;
        }
        if (keep && (ngroups != 1 || mine || rp)) { code = 3 ; goto done ; }      // H is kept for single-stage factorizations only

        // ---- layout of a group --------------------------------------------------------------------
        hostvec<Long> list, FOff, SOff, ROff ;
        Long fsum = 0, ssum = 0, rsum = 0, nlimbo = 0 ;
        auto in_group = [&] (Long f, Long g) { return f < nf && grp [f] == g ; } ;
        // a child's block is used where it lies when it waits on the device, copied behind its parent otherwise
        auto child_in_place = [&] (Long c) { return (mine && !mine [c]) ? limbo_in_dev : (limbo_dev [c] != NULL) ; } ;
        auto layout_group = [&] (Long g, bool final_pass)
        {
            // One group that is the one stage of the analysis, nothing that changes sizes from call to call
            // (no rank plan, no subset, no right-hand-side block): the layout follows from the symbolic
            // object alone and is kept with it (two passes over the 110,287 fronts of the 2D 1024^2 problem
            // were 10 of a re-factorization's 80 ms).
            spqr_amd_symbolic::NumericCache &NCl = S->cache ;
            const bool lay_cacheable = by_stage && ngroups == 1 && g == 0 && !rp && !mine && !keep && nf > 0 ;
            if (lay_cacheable && NCl.lay_ok && NCl.lay_line_from == line_from)
            {
                list = NCl.lay_list ; FOff = NCl.lay_FOff ; SOff = NCl.lay_SOff ; ROff = NCl.lay_ROff ;
                fsum = NCl.lay_fsum ; ssum = NCl.lay_ssum ; rsum = NCl.lay_rsum ; nlimbo = 0 ;
                if (final_pass) rel = NCl.lay_rel ;
                return ;
            }
            list.clear ( ) ;
            if (!by_stage) { for (Long k = 0 ; k < nf ; k++) if (grp [Post [k]] == g) list.push_back (Post [k]) ; }
            else for (Long p = S->Stagingp [g] ; p < S->Stagingp [g+1] ; p++) list.push_back (Post [p]) ;
            const Long numFronts = (Long) list.size ( ) ;
            FOff.assign ((size_t) numFronts, 0) ; SOff.assign ((size_t) numFronts, 0) ; ROff.assign ((size_t) numFronts, 0) ;
            fsum = ssum = rsum = nlimbo = 0 ;
            for (Long k = 0 ; k < numFronts ; k++)
            {
                const Long f = list [k] ;
                if (final_pass) rel [f] = k ;
                if ((pitch [f] & 15) == 0) fsum = (fsum + 15) & ~(Long) 15 ;      // line-aligned rows need a line-aligned base
                FOff [k] = fsum ; SOff [k] = ssum ; ROff [k] = rsum ;
                fsum += Fm [f] * pitch [f] ;
                for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
                {
                    const Long c = Child [cp] ;
                    if (in_group (c, g)) continue ;
                    nlimbo++ ;
                    // (the sizing pass cannot know yet which blocks will find room on the device: it
                    // prices every one of them as copied in)
                    if (final_pass && child_in_place (c)) continue ;
                    fsum += c_size (c) ;
                    fsum += (fsum & 1) ;
                }
                ssum += S_count (f) ;
                rsum += rsize_of (f) ;
            }
            if (lay_cacheable && final_pass && nlimbo == 0)
            {
                NCl.lay_list = list ; NCl.lay_FOff = FOff ; NCl.lay_SOff = SOff ; NCl.lay_ROff = ROff ; NCl.lay_rel = rel ;
                NCl.lay_fsum = fsum ; NCl.lay_ssum = ssum ; NCl.lay_rsum = rsum ;
                NCl.lay_line_from = line_from ; NCl.lay_ok = true ;
            }
        } ;

        TICK ("groups") ;
        // ---- the page-locked stack, if a helper thread went for it -------------------------------
        if (pin_job.started)
        {
            pin_job.join ( ) ;
            pin_job.started = false ;
            if (pin_job.ok)
            {
                num->Rstack = (double *) pin_job.ptr ; pin_job.ptr = NULL ;
                g_locked_now += (long long) ((size_t) std::max ((Long) 1, rtotal) * sizeof (double)) ;
            }
            else
            {
                num->RstackPinned = 0 ;
                num->Rstack = (double *) host_malloc ((size_t) std::max ((Long) 1, rtotal) * sizeof (double)) ;
                if (!num->Rstack) { code = 1 ; goto done ; }
            }
            TICK ("page-locked stack (rest of it)") ;
        }

        // ---- arenas sized for the largest group, once ---------------------------------------------
        // (a pageable stack and several groups: two device blocks for R take turns, so that a helper thread
        // can copy one group's R out while the next group runs; given up when the device has no room for the second)
        bool two_r = !num->RstackPinned && ngroups > 1 ;
        for (int attempt = 0 ; ; attempt++)
        {
            two_r = !num->RstackPinned && ngroups > 1 ;
            Long fmax = 1, smax = 1, rmax [2] = {1, 1} ;
            for (Long g = 0 ; g < ngroups ; g++)
            {
                layout_group (g, false) ;
                fmax = std::max (fmax, fsum) ; smax = std::max (smax, ssum) ;
                Long &rm = rmax [(two_r && (g & 1)) ? 1 : 0] ;
                rm = std::max (rm, rsum) ;
            }
            auto need = [&] (Buf &b, size_t bytes, const char *what)
            {
                if (b.ensure (bytes) == hipSuccess) return true ;
                (void) hipGetLastError ( ) ;
                if (verbose) fprintf (stderr, "  [spqrgpu] no room on the device for %s (%.2f GB)\n", what, (double) bytes / 1e9) ;
                return false ;
            } ;
            bool ok = need (dF, (size_t) fmax * sizeof (double), "the fronts of the largest group")
                   && need (dS, (size_t) smax * sizeof (gpuqr_sentry), "the S entries of the largest group") ;
            if (ok && two_r && (!need (WS.dR, (size_t) rmax [0] * sizeof (double), "the packed R of a group")
                       || !need (WS.dR2, (size_t) rmax [1] * sizeof (double), "the second block of packed R")))
            {
                WS.dR2.release ( ) ;
                two_r = false ;
                rmax [0] = std::max (rmax [0], rmax [1]) ;
            }
            if (ok && !two_r) ok = need (WS.dR, (size_t) rmax [0] * sizeof (double), "the packed R of the largest group") ;
            if (ok) break ;
            // The stages were taken to fit (an earlier call found them to, or the estimate said so) and the
            // device says otherwise now: everything is handed back and the fronts are grouped by what is
            // really free.  Once; a group that still does not fit is out of memory.
            if (attempt > 0 || nf == 0) { code = 1 ; goto done ; }
            dF.release ( ) ; dS.release ( ) ; WS.dR.release ( ) ; WS.dR2.release ( ) ; WS.dLimbo.release ( ) ;
            g_ws_gen++ ;
            S->cache.fit_gen = 0 ;
            // (a call for a subset of the fronts, or one that keeps H, cannot be cut differently: it tries once
            // more with every buffer of earlier calls given back, the waiting blocks' included)
            if (mine || keep) continue ;
            size_t free_b = 0, total_b = 0 ;
            if (hipMemGetInfo (&free_b, &total_b) != hipSuccess) { (void) hipGetLastError ( ) ; code = 1 ; goto done ; }
            const size_t margin = (size_t) 1 << 30 ;
            const size_t room = (free_b > margin) ? free_b - margin : 0 ;
            if (verbose) fprintf (stderr, "  [spqrgpu] fronts grouped again for %.2f GB\n", (double) room / 1e9) ;
            by_stage = false ;
            ngroups = greedy_groups ((size_t) (0.85 * (double) room), false) ;
        }
        // ---- room for the blocks that wait between groups ----------------------------------------
        if (ngroups > 1 && getenv ("GPUQR_HOST_LIMBO") == NULL && getenv ("GPUQR_NO_LIMBO_ARENA") == NULL)
        {
            // most bytes waiting at any one time: a block is made when its front's group is through and
            // given back when its parent's group is
            hostvec<size_t> made ((size_t) ngroups, 0), gone ((size_t) ngroups, 0) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                if (!is_mine (f)) continue ;
                const Long pf = Parent [f] ;
                if (pf >= nf || !is_mine (pf) || grp [pf] == grp [f]) continue ;
                const Long cn = (Rp [f+1] - Rp [f]) - (Super [f+1] - Super [f]) ;
                const size_t b = DevArena::rounded ((size_t) (cm_of (f) * cn) * sizeof (double)) ;
                if (b == 0 || grp [f] < 0 || grp [pf] < 0) continue ;
                made [grp [f]] += b ; gone [grp [pf]] += b ;
            }
            size_t live = 0, peak = 0 ;
            for (Long g = 0 ; g < ngroups ; g++) { live -= std::min (live, gone [g]) ; live += made [g] ; peak = std::max (peak, live) ; }
            if (peak > 0)
            {
                const size_t want = peak + peak / 4 + ((size_t) 1 << 20) ;      // (first fit leaves holes)
                bool have = WS.dLimbo.cap >= want ;
                if (!have)
                {
                    size_t free_b = 0, total_b = 0 ;
                    if (hipMemGetInfo (&free_b, &total_b) == hipSuccess && free_b + WS.dLimbo.cap >= want + ((size_t) 2 << 30))
                        have = WS.dLimbo.ensure (want) == hipSuccess ;
                    (void) hipGetLastError ( ) ;
                }
                if (have) arena.reset (WS.dLimbo.p, WS.dLimbo.cap) ;
            }
        }
        // ---- the end of a pageable stack, page-locked ---------------------------------------------
        if (stack_kept_kind && !num->RstackPinned && ngroups > 1 && !keep && !rp
            && getenv ("GPUQR_LATE_R") == NULL && getenv ("GPUQR_PAGEABLE_R") == NULL)
        {
            hostvec<Long> rs ((size_t) ngroups, 0) ;
            for (Long g = 0 ; g < ngroups ; g++) { layout_group (g, false) ; rs [g] = rsum ; }
            // (under the limit, together with what else is page-locked now; a page-locked stack that is only
            // kept for a next factorization makes way)
            g_pinned_stack.clear ( ) ;
            const long long others = g_locked_now.load ( ) - (long long) g_pageable_stack.locked_of (num->Rstack) ;
            const long long budget = (long long) pin_limit_bytes ( ) - std::max (others, (long long) 0) ;
            Long tail = 0, g0 = ngroups ;
            while (g0 > 0 && (long long) ((size_t) (tail + rs [g0-1]) * sizeof (double)) <= budget) { tail += rs [g0-1] ; g0-- ; }
            if (g0 < ngroups && tail > 0)
            {
                Long before = 0 ;
                for (Long g = 0 ; g < g0 ; g++) before += rs [g] ;
                char *lo = (char *) (num->Rstack + before) ;
                const size_t bytes = (size_t) tail * sizeof (double) ;
                void *stack = num->Rstack ;
                bool *okp = &tail_locked ;
                tail_first = g0 ;
                // (beside the first groups: page-locking touches every page of the range)
                try { lock_thread = std::thread ([stack, lo, bytes, okp, wdev] ( ) { if (hipSetDevice (wdev) == hipSuccess) *okp = g_pageable_stack.lock_range (stack, lo, bytes) ; }) ; }
                catch (...) { tail_locked = g_pageable_stack.lock_range (stack, lo, bytes) ; }
                if (verbose) fprintf (stderr, "  [spqrgpu] R of groups %ld..%ld goes out from inside the engine: %.3f GB at the end of the pageable stack page-locked\n",
                    (long) g0, (long) (ngroups - 1), (double) bytes / 1e9) ;
            }
        }
        TICK ("workspace allocation") ;

        Long rhead = 0, chead = 0 ;
        // With a page-locked stack the engine itself packs R and sends it (GPUQR_LATE_R: afterwards, in one copy)
        const bool r_in_engine = num->RstackPinned && !keep && getenv ("GPUQR_LATE_R") == NULL ;
        hostvec<PackRTask> ptasks ;
        hostvec<long long> prefix ;
        hostvec<double> rows_host ;             // unpacked R rows of the group's fronts with dead columns
        hostvec<Long> rows_at ;
        for (Long g = 0 ; g < ngroups ; g++)
        {
            // this group's R: from inside the engine (the stack, or its end from this group on, is page-locked)
            // or packed and copied once the group is through
            if (g == tail_first && lock_thread.joinable ( )) lock_thread.join ( ) ;
            const bool rie = r_in_engine || (tail_first >= 0 && g >= tail_first && tail_locked) ;
            // ---- fronts of the group, in postorder -----------------------------------------
            layout_group (g, true) ;
            const Long numFronts = (Long) list.size ( ) ;
            if (numFronts == 0) { num->numStagesRun++ ; continue ; }
            Long rows_off = 0 ;
            rows_at.assign ((size_t) numFronts, -1) ;
            if (rp)
            {
                Long need = 0 ;
                for (Long k = 0 ; k < numFronts ; k++)
                    if (rp->touched [list [k]])     // (the engine also copies the contribution rows of a front whose parent is elsewhere)
                        need += (rank_of (list [k]) + cm_of (list [k])) * (Rp [list [k]+1] - Rp [list [k]]) ;
                rows_host.assign ((size_t) std::max ((Long) 1, need), 0.0) ;
            }
            Buf &dR = (two_r && (g & 1)) ? WS.dR2 : WS.dR ;
            double *F0 = (double *) dF.p ;
            gpuqr_sentry *S0 = (gpuqr_sentry *) dS.p ;
            // A single group with nothing arriving from outside, on the same workspaces as the call
            // before: the descriptors are the ones that call made (only the host address of each packed
            // R block may have moved with the stack).
            spqr_amd_symbolic::NumericCache &NCf = S->cache ;
            uint64_t fronts_key = 0 ;
            if (ngroups == 1 && nlimbo == 0 && !rp && !mine && !keep)
            {
                const uint64_t parts [8] = {S->uid, (uint64_t) (uintptr_t) dF.p, (uint64_t) (uintptr_t) dS.p,
                    (uint64_t) (uintptr_t) NC.d_Rimap, (uint64_t) (uintptr_t) NC.d_Rjmap, (uint64_t) numFronts,
                    (uint64_t) (uintptr_t) dR.p, (uint64_t) (r_in_engine ? 1 : 0)} ;
                fronts_key = 1469598103934665603ull ;
                for (int i = 0 ; i < 8 ; i++) for (int b = 0 ; b < 8 ; b++) { fronts_key ^= (parts [i] >> (8*b)) & 0xff ; fronts_key *= 1099511628211ull ; }
                fronts_key |= 1 ;
            }
            const bool reuse_fronts = fronts_key != 0 && NCf.fronts_key == fronts_key && (Long) NCf.fronts.size ( ) == numFronts ;
            if (!reuse_fronts) fronts.assign ((size_t) (numFronts + nlimbo), gpuqr_front ( )) ;
            else
            {
                fronts.swap (NCf.fronts) ;
                NCf.fronts_key = 0 ;
                const double *stack_now = r_in_engine ? num->Rstack : NULL ;
                if (NCf.fronts_rstack != stack_now)
                    for (Long k = 0 ; k < numFronts ; k++) if (fronts [k].cpuRpack != NULL) fronts [k].cpuRpack = num->Rstack + rhead + ROff [k] ;
            }

            // (the engine is about to pack into this group's block of packed R: a helper thread still copying
            // an earlier group's R out of that block comes first)
            if (rie && !r_in_engine && r_thread_src == dR.p && join_r ( ) != hipSuccess) { code = 2 ; goto done ; }
            TICK ("  group: layout") ;
            HIPCHK (hipMemsetAsync (F0, 0, (size_t) std::max ((Long) 1, fsum) * sizeof (double), 0)) ;
            // S entries of the group, built on the device; runs of fronts that are neighbours in
            // postorder go in one launch
            {
                // (the runs are kept with the layout where that is kept: see layout_group)
                spqr_amd_symbolic::NumericCache &NCl = S->cache ;
                const bool keep_runs = by_stage && ngroups == 1 && !rp && !mine && !keep && NCl.lay_ok ;
                hostvec<Long> runs_now ;
                hostvec<Long> &runs = (keep_runs ? NCl.lay_sruns : runs_now) ;
                if (!(keep_runs && NCl.lay_sruns_ok))
                {
                    runs.clear ( ) ;
                    for (Long k = 0 ; k < numFronts ; )
                    {
                        Long k2 = k, cnt = 0 ;
                        while (k2 < numFronts && SGlob [list [k2]] == SGlob [list [k]] + cnt) { cnt += S_count (list [k2]) ; k2++ ; }
                        if (cnt > 0) { runs.push_back (k) ; runs.push_back (cnt) ; }
                        k = k2 ;
                    }
                    if (keep_runs) NCl.lay_sruns_ok = true ;
                }
                for (size_t i = 0 ; i + 1 < runs.size ( ) ; i += 2)
                {
                    const Long k = runs [i], cnt = runs [i+1] ;
                    HIPCHK (qr_launch_build_s (S0 + SOff [k], (const long long *) (rp ? rp->d_findex : NC.d_findex) + SGlob [list [k]],
                        (const long long *) NC.d_aperm + SGlob [list [k]], (const double *) dAx.p, cnt, 0)) ;
                }
            }

            TICK ("  group: zero fill and S launches") ;
            Long next_child = numFronts ;
            for (Long k = 0 ; k < numFronts && !reuse_fronts ; k++)
            {
                const Long f = list [k] ;
                const Long fm = Fm [f], fn = Rp [f+1] - Rp [f], fp = npiv_of (f) ;
                const Long frank = rank_of (f) ;
                const Long pid = Parent [f] ;
                const bool parent_here = in_group (pid, g) ;
                const int32_t *rjmap_base = (const int32_t *) (rp ? rp->d_Rjmap : NC.d_Rjmap) ;
                const bool host_packed = rp && rp->touched [f] ;    // R of a front with dead columns is packed by the host

                gpuqr_front &fr = fronts [k] ;
                memset (&fr, 0, sizeof (fr)) ;
                fr.fids = k ; fr.pids = parent_here ? rel [pid] : -1 ; fr.fidg = f ; fr.pidg = pid ;
                fr.fm = fm ; fr.fn = fn ; fr.fp = fp ; fr.rank = frank ;
                fr.ldn = pitch [f] ;
                fr.Stair = (int64_t *) (rp ? rp->Stair.data ( ) : Stair.data ( )) + Rp [f] ;
                fr.gpuF = F0 + FOff [k] ;
                fr.cpuR = NULL ;                    // R is packed on the device, see below
                if (host_packed)
                {
                    fr.cpuR = rows_host.data ( ) + rows_off ;       // the first frank rows, unpacked
                    rows_at [k] = rows_off ;
                    rows_off += (frank + cm_of (f)) * fn ;
                }
                else if (rie)
                {
                    // the engine packs the block once the front is final and sends it while the rest runs
                    fr.gpuRpack = (double *) dR.p + ROff [k] ;
                    fr.cpuRpack = num->Rstack + rhead + ROff [k] ;
                }
                fr.is_sparse = 1 ;
                fr.rhs_col = (int32_t) rhs_col [f] ; fr.rhs_n = (int32_t) rhs_n ;
                fr.nc = (int32_t) (Childp [f+1] - Childp [f]) ;
                fr.Scount = S_count (f) ;
                fr.gpuS = S0 + SOff [k] ;
                fr.cn = (int32_t) ((Rp [f+1] - Rp [f]) - (Super [f+1] - Super [f])) ;
                fr.cm = (int32_t) cm_of (f) ;
                if (parent_here)
                {
                    fr.pn = (int32_t) pitch [pid] ;
                    fr.gpuRjmap = rjmap_base + S->RjmapOffsets [f] ;
                    fr.gpuRimap = (int32_t *) NC.d_Rimap + S->RimapOffsets [f] ;
                    // (with dead columns the rows of a parent are permuted on the device only)
                    fr.cpuRimap = rp ? NULL : NC.Rimap.data ( ) + S->RimapOffsets [f] ;
                    fr.gpuC = fr.gpuF + frank * pitch [f] + fp ;
                }
                else if (pid < nf)
                {
                    fr.is_staged = 1 ;      // the parent is factorized in a later group or elsewhere
                }

                // children from outside the group
                Long child_off = FOff [k] + fm * pitch [f] ;
                for (Long cp = Childp [f] ; cp < Childp [f+1] ; cp++)
                {
                    const Long c = Child [cp] ;
                    if (in_group (c, g)) continue ;
                    const Long ccn = (Rp [c+1] - Rp [c]) - (Super [c+1] - Super [c]) ;
                    const Long ccm = cm_of (c) ;
                    const bool external = mine && !mine [c] ;
                    gpuqr_front &ch = fronts [next_child] ;
                    memset (&ch, 0, sizeof (ch)) ;
                    ch.fids = next_child ; ch.pids = k ; ch.fidg = c ; ch.pidg = f ;
                    ch.fm = ccm ; ch.fn = ccn ; ch.fp = 0 ; ch.rank = 0 ; ch.ldn = ccn ;
                    ch.is_sparse = 1 ; ch.push_only = 1 ;
                    ch.pn = (int32_t) pitch [f] ; ch.cm = (int32_t) ccm ; ch.cn = (int32_t) ccn ;
                    ch.gpuRimap = (int32_t *) NC.d_Rimap + S->RimapOffsets [c] ;
                    ch.cpuRimap = rp ? NULL : NC.Rimap.data ( ) + S->RimapOffsets [c] ;
                    ch.gpuRjmap = rjmap_base + S->RjmapOffsets [c] ;
                    const bool in_place = child_in_place (c) ;
                    if (in_place && external)
                    {
                        // the caller's device block (it arrived over the fabric, or an earlier call left it)
                        double *src = (limbo_in && ccm * ccn > 0) ? (double *) limbo_in [c] : NULL ;
                        if (ccm * ccn > 0 && !src) { code = 3 ; goto done ; }
                        ch.gpuF = ch.gpuC = src ? src : F0 ;
                    }
                    else if (in_place)
                    {
                        // the block never left the device: assemble straight from where it waits
                        ch.gpuF = ch.gpuC = limbo_dev [c] ;
                        used_dev.push_back (limbo_dev [c]) ;
                        used_bytes.push_back ((size_t) (ccm * ccn) * sizeof (double)) ;
                        limbo_dev [c] = NULL ;
                    }
                    else
                    {
                        ch.gpuF = ch.gpuC = F0 + child_off ;
                        if (ccm * ccn > 0)
                        {
                            const double *src = external ? (limbo_in ? limbo_in [c] : NULL) : limbo [c] ;
                            if (!src) { code = external ? 3 : 2 ; goto done ; }
                            HIPCHK (hipMemcpy (ch.gpuF, src, (size_t) (ccm * ccn) * sizeof (double), hipMemcpyHostToDevice)) ;
                        }
                        if (!external) { host_free (limbo [c]) ; limbo [c] = NULL ; }
                        child_off += ccm * ccn ;
                        child_off += (child_off & 1) ;
                    }
                    next_child++ ;
                }
            }
            HIPCHK (hipStreamSynchronize (0)) ;
            TICK ("group setup (memset, S, fronts)") ;

            // ---- the device work ---------------------------------------------------------------
            gpuqr_stats stats ;
            // One group and nothing arriving from outside it: every call on this symbolic object and
            // these workspaces describes the same problem to the engine, which may then keep its schedule.
            uint64_t plan_key = 0 ;
            if (ngroups == 1 && nlimbo == 0 && !rp && getenv ("GPUQR_NO_PLAN_CACHE") == NULL)
            {
                const uint64_t parts [7] = {S->uid, (uint64_t) (uintptr_t) dF.p, (uint64_t) (uintptr_t) dS.p,
                    (uint64_t) (uintptr_t) NC.d_Rimap, (uint64_t) (uintptr_t) NC.d_Rjmap, (uint64_t) numFronts,
                    r_in_engine ? (uint64_t) (uintptr_t) dR.p : 0} ;
                plan_key = 1469598103934665603ull ;
                for (int i = 0 ; i < 7 ; i++) for (int b = 0 ; b < 8 ; b++) { plan_key ^= (parts [i] >> (8*b)) & 0xff ; plan_key *= 1099511628211ull ; }
                if (mine) for (Long f = 0 ; f < nf ; f++) { plan_key ^= (uint64_t) (mine [f] != 0) ; plan_key *= 1099511628211ull ; }
                plan_key |= 1 ;
            }
            int ecode = keep ? gpuqr_factor_sparse_keep (fronts.data ( ), next_child, &stats, &keep->rp)
                : gpuqr_factor_sparse_keyed (S->gpuMemorySize, fronts.data ( ), next_child, Parent, Childp, Child, &stats, plan_key) ;
            TICK ("engine") ;
            if (ecode != GPUQR_SUCCESS) { code = (ecode == GPUQR_OUTOFMEMORY) ? 1 : 2 ; goto done ; }
            num->gpuKernelTime += stats.kernelTime ;
            num->gpuFlops += stats.flopsActual ;
            num->gpuNumKernelLaunches += stats.numLaunches ;
            num->numStagesRun++ ;
            for (size_t i = 0 ; i < used_dev.size ( ) ; i++)
            {
                if (arena.owns (used_dev [i])) arena.give (used_dev [i], used_bytes [i]) ;
                else (void) hipFree (used_dev [i]) ;
            }
            used_dev.clear ( ) ; used_bytes.clear ( ) ;

            if (rp)
            {
                // ---- what the CPU rule would have seen in the columns this pass took for dead ------
                // A dead pivot column k with g live pivot columns before it was tested on rows g.. of
                // the front as they stood after those g columns.  Every reflector of this pass that
                // came later worked on rows g.. only and left that norm alone; the last of them, the
                // column's own at its position q in the permuted front, gathered what lay below row q
                // into row q (the rows below hold its Householder vector now).  So the norm of rows
                // g..q of the column in the reduced front is still that number, and the caller
                // compares it with the tolerance.
                hostvec<ColNormTask> ctasks ;
                hostvec<Long> ccols ;
                for (Long k = 0 ; k < numFronts ; k++)
                {
                    const Long f = list [k] ;
                    if (!rp->touched [f]) continue ;
                    const Long *pos = rp->colpos.data ( ) + Rp [f] ;
                    walk_pivots (Fm [f], Super [f+1] - Super [f], Super [f], rp->dead.data ( ), [&] (Long kk, Long rm, bool)
                    {
                        if (!rp->dead [Super [f] + kk]) return ;
                        ColNormTask ct ;
                        ct.x = fronts [k].gpuF + rm * pitch [f] + pos [kk] ;
                        ct.nrows = std::min (pos [kk] + 1, Fm [f]) - rm ; ct.ld = pitch [f] ;
                        if (ct.nrows > 0) { ctasks.push_back (ct) ; ccols.push_back (Super [f] + kk) ; }
                        else rp->deadnorm [Super [f] + kk] = 0 ;
                    }) ;
                }
                const size_t nt = ctasks.size ( ) ;
                if (nt > 0)
                {
                    hostvec<double> norms (nt) ;
                    if (dTasks.ensure (nt * sizeof (ColNormTask)) != hipSuccess) { code = 1 ; goto done ; }
                    if (dPrefix.ensure (nt * sizeof (double)) != hipSuccess) { code = 1 ; goto done ; }
                    HIPCHK (hipMemcpyAsync (dTasks.p, ctasks.data ( ), nt * sizeof (ColNormTask), hipMemcpyHostToDevice, 0)) ;
                    HIPCHK (qr_launch_col_norms ((const ColNormTask *) dTasks.p, (long long) nt, (double *) dPrefix.p, 0)) ;
                    HIPCHK (hipMemcpyAsync (norms.data ( ), dPrefix.p, nt * sizeof (double), hipMemcpyDeviceToHost, 0)) ;
                    HIPCHK (hipStreamSynchronize (0)) ;
                    for (size_t i = 0 ; i < nt ; i++) rp->deadnorm [ccols [i]] = norms [i] ;
                }
            }

            // ---- R: packed on the device, one copy onto the stack ---------------------------------
            if (rie)
            {
                for (Long k = 0 ; k < numFronts ; k++) num->Rblock [list [k]] = rhead + ROff [k] ;
                rhead += rsum ;
            }
            else
            {
                ptasks.assign ((size_t) numFronts, PackRTask ( )) ;
                prefix.assign ((size_t) numFronts + 1, 0) ;
                for (Long k = 0 ; k < numFronts ; k++)
                {
                    const Long f = list [k] ;
                    const Long fn = Rp [f+1] - Rp [f] ;
                    const Long frank = (rp && rp->touched [f]) ? 0 : rank_of (f) ;
                    PackRTask &pt = ptasks [k] ;
                    pt.F = F0 + FOff [k] ; pt.out = (double *) dR.p + ROff [k] ;
                    pt.ld = (int32_t) pitch [f] ; pt.fn = (int32_t) fn ; pt.frank = (int32_t) frank ; pt.c0 = 0 ; pt.c1 = (int32_t) fn ; pt.pad = 0 ;
                    prefix [k+1] = prefix [k] + (frank + 31) / 32 ;
                    num->Rblock [f] = rhead + ROff [k] ;
                }
                if (dTasks.ensure ((size_t) numFronts * sizeof (PackRTask)) != hipSuccess) { code = 1 ; goto done ; }
                if (dPrefix.ensure ((size_t) (numFronts + 1) * sizeof (long long)) != hipSuccess) { code = 1 ; goto done ; }
                HIPCHK (hipMemcpyAsync (dTasks.p, ptasks.data ( ), (size_t) numFronts * sizeof (PackRTask), hipMemcpyHostToDevice, 0)) ;
                HIPCHK (hipMemcpyAsync (dPrefix.p, prefix.data ( ), (size_t) (numFronts + 1) * sizeof (long long), hipMemcpyHostToDevice, 0)) ;
                // The two device blocks take turns, and the copy of a stage is joined where the next stage starts
                // its own -- which a stage without any R (fronts without rows) never does: the stage after it
                // would then pack into the block that the helper thread is still reading.
                if (r_thread_src == dR.p && join_r ( ) != hipSuccess) { code = 2 ; goto done ; }
                HIPCHK (qr_launch_pack_r ((const QrPackRTask *) dTasks.p, (const long long *) dPrefix.p, (int) numFronts, prefix [numFronts], 0)) ;
                if (rsum > 0 && num->RstackPinned)
                    HIPCHK (hipMemcpyAsync (num->Rstack + rhead, dR.p, (size_t) rsum * sizeof (double), hipMemcpyDeviceToHost, 0)) ;
                else if (rsum > 0)
                {
                    HIPCHK (hipStreamSynchronize (0)) ;
                    if (join_r ( ) != hipSuccess) { code = 2 ; goto done ; }       // one copy at a time: they share the bounce buffers
                    double *dst = num->Rstack + rhead ;
                    const void *src = dR.p ;
                    const size_t bytes = (size_t) rsum * sizeof (double) ;
                    if (g + 1 < ngroups && !rp && getenv ("GPUQR_SERIAL_R") == NULL)
                    {
                        bool started = false ;
                        try
                        {
                            r_thread = std::thread ([&WS, &r_thread_err, dst, src, bytes, wdev] ( )
                            {
                                r_thread_err = hipSetDevice (wdev) ;
                                if (r_thread_err == hipSuccess) r_thread_err = copy_to_pageable (dst, src, bytes, WS.hBounce) ;
                            }) ;
                            started = true ;
                        }
                        catch (...) { }         // no thread to be had: copy here and now
                        if (started) r_thread_src = src ;
                        else HIPCHK (copy_to_pageable (dst, src, bytes, WS.hBounce)) ;
                    }
                    else HIPCHK (copy_to_pageable (dst, src, bytes, WS.hBounce)) ;
                }
                rhead += rsum ;
            }
            {
                // contribution blocks that leave the group: compact cm-by-cn
                for (Long k = 0 ; k < numFronts ; k++)
                {
                    if (!fronts [k].is_staged) continue ;
                    const Long f = list [k] ;
                    const Long fn = Rp [f+1] - Rp [f], fp = npiv_of (f) ;
                    const Long frank = rank_of (f) ;
                    const Long cn = fn - (Super [f+1] - Super [f]) ;
                    const bool leaves_call = mine && !mine [Parent [f]] ;
                    const Long cm = cm_of (f) ;
                    const double *Csrc = fronts [k].gpuF + frank * pitch [f] + fp ;
                    double *L ;
                    if (leaves_call && c_out_dev)
                    {
                        // to the caller's device buffer; what lies below the diagonal is never read by an assembly
                        num->Cblock [f] = chead ;
                        if (cm > 0 && cn > 0)
                            HIPCHK (hipMemcpy2DAsync (c_out_dev + chead, (size_t) cn * sizeof (double), Csrc,
                                (size_t) pitch [f] * sizeof (double), (size_t) cn * sizeof (double), (size_t) cm, hipMemcpyDeviceToDevice, 0)) ;
                        chead += cm * cn ;
                        continue ;
                    }
                    else if (leaves_call) { L = num->Cstack + chead ; num->Cblock [f] = chead ; chead += cm * cn ; }
                    else if (cm > 0 && cn > 0 && getenv ("GPUQR_HOST_LIMBO") == NULL
                        && ((limbo_dev [f] = (double *) arena.take ((size_t) (cm * cn) * sizeof (double))) != NULL
                            || hipMalloc ((void **) &limbo_dev [f], (size_t) (cm * cn) * sizeof (double)) == hipSuccess))
                    {
                        // room on the device: the block waits there
                        HIPCHK (hipMemcpy2DAsync (limbo_dev [f], (size_t) cn * sizeof (double), Csrc,
                            (size_t) pitch [f] * sizeof (double), (size_t) cn * sizeof (double), (size_t) cm, hipMemcpyDeviceToDevice, 0)) ;
                        continue ;
                    }
                    else
                    {
                        (void) hipGetLastError ( ) ;
                        limbo_dev [f] = NULL ;
                        L = (double *) host_calloc ((size_t) std::max ((Long) 1, cm * cn), sizeof (double)) ;
                        if (!L) { code = 1 ; goto done ; }
                        limbo [f] = L ;
                    }
                    if (cm > 0 && cn > 0)
                    {
                        HIPCHK (hipMemcpy2D (L, (size_t) cn * sizeof (double), Csrc,
                            (size_t) pitch [f] * sizeof (double), (size_t) cn * sizeof (double), (size_t) cm, hipMemcpyDeviceToHost)) ;
                        for (Long i = 1 ; i < cm ; i++) for (Long j = 0 ; j < std::min (i, cn) ; j++) L [i*cn + j] = 0 ;
                    }
                }
                HIPCHK (hipStreamSynchronize (0)) ;

                if (rp)
                {
                    // R of the fronts with dead columns: squeezed and column-packed here, from the rows
                    // the engine copied out, in the original column order (spqr_rhpack's layout)
                    for (Long k = 0 ; k < numFronts ; k++)
                    {
                        const Long f = list [k] ;
                        if (rows_at [k] < 0) continue ;
                        const Long fn = Rp [f+1] - Rp [f], fpo = Super [f+1] - Super [f] ;
                        const double *rows = rows_host.data ( ) + rows_at [k] ;
                        const Long *pos = rp->colpos.data ( ) + Rp [f] ;
                        double *out = num->Rstack + num->Rblock [f] ;
                        Long rm_all = 0 ;
                        walk_pivots (Fm [f], fpo, Super [f], rp->dead.data ( ), [&] (Long kk, Long rm, bool)
                        {
                            for (Long i = 0 ; i < rm ; i++) *out++ = rows [i*fn + pos [kk]] ;
                            rm_all = rm ;
                        }) ;
                        for (Long j = fpo ; j < fn ; j++) for (Long i = 0 ; i < rm_all ; i++) *out++ = rows [i*fn + pos [j]] ;
                    }
                }

                for (Long k = 0 ; k < numFronts ; k++)
                {
                    const Long f = list [k] ;
                    const Long fn = Rp [f+1] - Rp [f], fp = Super [f+1] - Super [f] ;
                    // assembly traffic: 24 B per S entry, 16 B per C entry + 4 B per map int
                    num->assembleBytes += 24.0 * (double) fronts [k].Scount ;
                    if (Cm [f] > 0 && fn - fp > 0 && Parent [f] < nf)
                    {
                        const double cmm = (double) Cm [f], cnn = (double) (fn - fp) ;
                        num->assembleBytes += 16.0 * (cmm * (cmm + 1) / 2 + cmm * (cnn - cmm)) + 4.0 * (cmm + cnn) ;
                    }
                }
            }
            TICK ("R pack + copy") ;
            if (fronts_key != 0)
            {
                // kept for the next factorization on this symbolic object and these workspaces
                NCf.fronts.swap (fronts) ;
                NCf.fronts_key = fronts_key ;
                NCf.fronts_rstack = r_in_engine ? num->Rstack : NULL ;
            }

            if (keep)
            {
                // ---- the three row lists of Q'B ------------------------------------------------------
                hostvec<QrRowCopy> scat, push, gath ;
                hostvec<Long> push_round ;
                // rows of S (= rows of A, permuted) into the fronts: S row `row` sits in front row
                // findex / fn of its first entry; entries are packed front by front in postorder,
                // pivot column by pivot column, row by row (spqrgpu_amd_build_assembly_maps)
                hostvec<Long> findex ((size_t) std::max ((Long) 1, S->anz)) ;
                if (S->anz > 0) HIPCHK (hipMemcpy (findex.data ( ), NC.d_findex, (size_t) S->anz * sizeof (Long), hipMemcpyDeviceToHost)) ;
                hostvec<Long> P ((size_t) std::max ((Long) 1, S->m), 0) ;
                for (Long i = 0 ; i < S->m ; i++) P [S->PLinv [i]] = i ;
                const double *dB = (const double *) keep->dB.p ;
                Long q = 0 ;
                for (Long kk = 0 ; kk < nf ; kk++)
                {
                    const Long f = Post [kk], k = rel [f] ;
                    const Long fn = Rp [f+1] - Rp [f] ;
                    double *Ff = F0 + FOff [k] ;
                    for (Long row = Sleft [Super [f]] ; row < Sleft [Super [f+1]] ; row++)
                    {
                        if (Sp [row+1] > Sp [row])
                        {
                            const Long i = findex [q] / fn ;
                            QrRowCopy t ; t.src = dB + P [row] ; t.dst = Ff + i * pitch [f] + rhs_col [f] ;
                            scat.push_back (t) ;
                        }
                        q += Sp [row+1] - Sp [row] ;
                    }
                }
                Long row1 = 0 ;
                const int32_t nrounds = gpuqr_replay_rounds (keep->rp) ;
                for (Long f = 0 ; f < nf ; f++)
                {
                    const Long k = rel [f], frank = rank_of (f), pid = Parent [f] ;
                    double *Ff = F0 + FOff [k] ;
                    for (Long i = 0 ; i < frank ; i++)
                    {
                        QrRowCopy t ; t.src = Ff + i * pitch [f] + rhs_col [f] ; t.dst = (double *) keep->dC.p + row1 + i ;
                        gath.push_back (t) ;
                    }
                    row1 += frank ;
                    if (pid < nf)
                    {
                        double *Fp = F0 + FOff [rel [pid]] ;
                        const int32_t *Rimap = NC.Rimap.data ( ) + S->RimapOffsets [f] ;
                        const Long after = gpuqr_replay_front_done (keep->rp, k) ;
                        for (Long i = 0 ; i < cm_of (f) ; i++)
                        {
                            QrRowCopy t ; t.src = Ff + (frank + i) * pitch [f] + rhs_col [f] ;
                            t.dst = Fp + (Long) Rimap [i] * pitch [pid] + rhs_col [pid] ;
                            push.push_back (t) ;
                            push_round.push_back (std::max ((Long) 0, std::min ((Long) nrounds - 1, after))) ;
                        }
                    }
                }
                // pushes by the round they follow (counting sort)
                keep->push_ptr.assign ((size_t) nrounds + 2, 0) ;
                for (size_t i = 0 ; i < push.size ( ) ; i++) keep->push_ptr [push_round [i] + 1]++ ;
                for (int32_t r = 0 ; r <= nrounds ; r++) keep->push_ptr [r+1] += keep->push_ptr [r] ;
                hostvec<QrRowCopy> sorted (std::max (push.size ( ), (size_t) 1)) ;
                {
                    hostvec<Long> pos (keep->push_ptr) ;
                    for (size_t i = 0 ; i < push.size ( ) ; i++) sorted [pos [push_round [i]]++] = push [i] ;
                }
                keep->nscatter = (Long) scat.size ( ) ; keep->ngather = (Long) gath.size ( ) ; keep->rank = row1 ;
                if (keep->dScatter.ensure (std::max (scat.size ( ), (size_t) 1) * sizeof (QrRowCopy)) != hipSuccess
                 || keep->dPush.ensure (sorted.size ( ) * sizeof (QrRowCopy)) != hipSuccess
                 || keep->dGather.ensure (std::max (gath.size ( ), (size_t) 1) * sizeof (QrRowCopy)) != hipSuccess) { code = 1 ; goto done ; }
                HIPCHK (hipMemcpy (keep->dScatter.p, scat.data ( ), scat.size ( ) * sizeof (QrRowCopy), hipMemcpyHostToDevice)) ;
                HIPCHK (hipMemcpy (keep->dPush.p, sorted.data ( ), push.size ( ) * sizeof (QrRowCopy), hipMemcpyHostToDevice)) ;
                HIPCHK (hipMemcpy (keep->dGather.p, gath.data ( ), gath.size ( ) * sizeof (QrRowCopy), hipMemcpyHostToDevice)) ;
                TICK ("row lists of the kept factorization") ;
            }
        }

        // ---- ranks and dead columns (no rank detection on the device) ----------------------------
        for (Long f = 0 ; f < nf ; f++)
        {
            const Long fm = Fm [f], fp = Super [f+1] - Super [f] ;
            const Long frank = rank_of (f) ;
            if (rp) { for (Long k = 0 ; k < fp ; k++) num->Rdead [Super [f] + k] = rp->Rdead [Super [f] + k] ; }
            else for (Long k = fm ; k < fp ; k++) num->Rdead [Super [f] + k] = 1 ;      // pivot columns past the bottom
            if (is_mine (f)) { num->sumfrank += frank ; num->maxfrank = std::max (num->maxfrank, frank) ; }
        }
    }
    catch (std::bad_alloc &)
    {
        code = 1 ;
    }

done:
    if (lock_thread.joinable ( )) lock_thread.join ( ) ;
    if (join_r ( ) != hipSuccess && code == 0) code = 2 ;
    if (registered)
    {
        (void) hipHostUnregister (Rstack_user) ;
        g_locked_now -= (long long) ((size_t) num->rsize * sizeof (double)) ;
        num->RstackPinned = 0 ;
    }
    for (size_t i = 0 ; i < limbo.size ( ) ; i++) host_free (limbo [i]) ;
    for (size_t i = 0 ; i < limbo_dev.size ( ) ; i++) if (limbo_dev [i] && !arena.owns (limbo_dev [i])) (void) hipFree (limbo_dev [i]) ;
    for (size_t i = 0 ; i < used_dev.size ( ) ; i++) if (!arena.owns (used_dev [i])) (void) hipFree (used_dev [i]) ;
    if (code != 0) { spqr_amd_numeric_free (num) ; return code ; }
    *num_out = num ;
    return 0 ;
}

// One call = the whole numeric factorization of A = Q R on the device, stage by stage.
extern "C" int spqrgpu_amd_factorize
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    spqr_amd_numeric **num_out
)
{
    if (!S || !Ap || !Ax || !num_out || (!Ai && S->anz > 0)) return 3 ;
    return factorize_impl (S, Ap, Ai, Ax, NULL, NULL, false, NULL, NULL, 0, NULL, NULL, num_out) ;
}

// -----------------------------------------------------------------------------
// rank detection
// -----------------------------------------------------------------------------

// SPQR's default tolerance, 20 (m+n) eps max_j ||A(:,j)||_2 (SPQR/Source/spqr_tol.cpp:14-27,
// spqr_maxcolnorm.cpp).
extern "C" double spqr_amd_default_tol (int64_t m, int64_t n, const int64_t *Ap, const double *Ax)
{
    if (!Ap || (!Ax && n > 0 && Ap [n] > 0)) return -1 ;
    double maxnorm = 0 ;
    for (Long j = 0 ; j < n ; j++)
    {
        double ss = 0 ;
        for (Long p = Ap [j] ; p < Ap [j+1] ; p++) ss += Ax [p] * Ax [p] ;
        maxnorm = std::max (maxnorm, sqrt (ss)) ;
    }
    return 20.0 * (double) (m + n) * 2.220446049250313e-16 * maxnorm ;
}

// What a pass says about the plan it ran with (rp; NULL for the first pass, which takes every pivot
// column for alive): decisions that must change are changed in `dead`, their number is returned; 0
// means that the pass confirms its whole plan, which is then the CPU rule's.  The R blocks are
// column-packed, squeezed by the plan's dead columns.
//
// The CPU rule (spqr_front.cpp:418-479) decides a column from the columns before it: those of its front
// and, through the contribution blocks, those of the fronts below.  So in a front whose descendants
// have all confirmed their plans the pass has done exactly what the rule does up to the first column
// whose outcome contradicts the plan, and for those columns its numbers are the rule's numbers:
// |R_kk| of a live column, and for a dead one the norm of its rows from the would-be pivot row down
// (rp->deadnorm).  That first contradiction is settled for good.  If it is a live column that dies, at
// row g say, the rule's numbers can be read a little further: it keeps testing rows g.. of the
// following columns as long as they die too, every reflector the pass made since works inside rows
// g.., so the norm of a column's entries from row g down to its own diagonal (to the bottom of the
// front for a column that found no row left; plus what rp->deadnorm holds for a dead one) is the
// number the rule tests.  A run of dead columns -- everything behind the last independent row of a
// front, say -- is settled in one pass that way, and so is the first column that ends the run.
// Behind that, and in fronts above a contradiction, the pass's numbers are not the rule's: there the
// same tests only make guesses for the next plan (good ones for columns that depend on earlier
// columns; a column guessed alive once is not guessed alive again, so guesses cannot chase each other
// for ever), which a later pass settles.  Every pass that changes anything settles at least one more
// column, so the passes end.
static Long review_pass (const spqr_amd_symbolic *S, const spqr_amd_numeric *num, const RankPlan *rp,
    double tol, char *dead, char *revived)
{
    Long changed = 0 ;
    const Long nf = S->nf ;
    hostvec<char> clean ((size_t) std::max ((Long) 1, nf), 0) ;     // this pass confirmed the front and all below it
    hostvec<Long> flips ;
    auto norm_of = [] (const double *x, Long from, Long to, double extra)
    {
        double big = fabs (extra), ssq = 0 ;
        for (Long i = from ; i < to ; i++) big = std::max (big, fabs (x [i])) ;
        if (!(big > 0)) return big ;            // zero, or a NaN that the comparisons below treat as "not small"
        for (Long i = from ; i < to ; i++) { const double v = x [i] / big ; ssq += v * v ; }
        { const double v = extra / big ; ssq += v * v ; }
        return big * sqrt (ssq) ;
    } ;
    for (Long f = 0 ; f < nf ; f++)             // postorder: children before parents
    {
        if (num->Rblock [f] < 0) continue ;
        bool anchored = true ;
        for (Long q = S->Childp [f] ; q < S->Childp [f+1] ; q++) anchored = anchored && clean [S->Child [q]] ;
        const double *R = num->Rstack + num->Rblock [f] ;
        const Long col1 = S->Super [f] ;
        enum { EXACT, RUN, GUESS } phase = anchored ? EXACT : GUESS ;
        Long p = 0, run_row = -1 ;
        flips.clear ( ) ;
        // (the walk must see the packing as it is: `dead` changes only after it)
        walk_pivots (S->Fm [f], S->Super [f+1] - col1, col1, dead, [&] (Long k, Long rm, bool live)
        {
            const double *col = R + p ;         // the rm entries of this column: rows 0..rm-1
            const Long j = col1 + k ;
            p += rm ;
            if (dead [j])
            {
                // dead by the plan: it holds the rm rows above its would-be pivot row
                const double below = rp ? rp->deadnorm [j] : 0 ;
                if (phase == EXACT) { if (!(below <= tol)) { flips.push_back (j) ; phase = GUESS ; } }
                else if (phase == RUN) { if (!(norm_of (col, run_row, rm, below) <= tol)) { flips.push_back (j) ; phase = GUESS ; } }
                else if (!(below <= tol) && !revived [j]) { flips.push_back (j) ; revived [j] = 1 ; }
                return ;
            }
            if (phase == EXACT)
            {
                if (live && fabs (col [rm-1]) <= tol) { flips.push_back (j) ; phase = RUN ; run_row = rm - 1 ; }
            }
            else if (phase == RUN)
            {
                if (norm_of (col, run_row, rm, 0.0) <= tol) flips.push_back (j) ;
                else phase = GUESS ;
            }
            else if (live && fabs (col [rm-1]) <= tol) flips.push_back (j) ;
        }) ;
        for (size_t i = 0 ; i < flips.size ( ) ; i++) dead [flips [i]] = !dead [flips [i]] ;
        changed += (Long) flips.size ( ) ;
        clean [f] = anchored && flips.empty ( ) ;
    }
    return changed ;
}

// The plan of a pass that knows the dead columns.  Returns 0, 1 (out of memory), 2 (device).
static int build_rank_plan (const spqr_amd_symbolic *S, const char *dead, RankPlan &rp)
{
    const Long nf = S->nf, n = S->n ;
    const spqr_amd_symbolic::NumericCache &NC = S->cache ;
    rp.dead.assign (dead, dead + std::max ((Long) 1, n)) ;
    rp.Rdead.assign ((size_t) std::max ((Long) 1, n), 0) ;
    rp.deadnorm.assign ((size_t) std::max ((Long) 1, n), 0.0) ;
    rp.touched.assign ((size_t) std::max ((Long) 1, nf), 0) ;
    rp.live.assign ((size_t) std::max ((Long) 1, nf), 0) ;
    rp.rsize.assign ((size_t) std::max ((Long) 1, nf), 0) ;
    rp.colpos.assign ((size_t) std::max ((Long) 1, S->Rp [nf]), 0) ;
    rp.Stair.assign (NC.Stair.begin ( ), NC.Stair.end ( )) ;
    for (Long f = 0 ; f < nf ; f++)
    {
        const Long col1 = S->Super [f], fp = S->Super [f+1] - col1, fn = S->Rp [f+1] - S->Rp [f], fm = S->Fm [f] ;
        Long ndead = 0 ;
        for (Long k = 0 ; k < fp ; k++) ndead += (dead [col1 + k] != 0) ;
        const Long L = fp - ndead, cn = fn - fp ;
        rp.live [f] = L ;
        rp.touched [f] = (ndead > 0) ;
        Long *pos = rp.colpos.data ( ) + S->Rp [f] ;
        Long nl = 0, nd = 0, size = 0, rm_all = 0 ;
        for (Long k = 0 ; k < fp ; k++) pos [k] = dead [col1 + k] ? (L + cn + nd++) : nl++ ;
        for (Long j = fp ; j < fn ; j++) pos [j] = L + (j - fp) ;
        walk_pivots (fm, fp, col1, dead, [&] (Long k, Long rm, bool live)
        {
            size += rm ; rm_all = rm ;
            rp.Rdead [col1 + k] = live ? 0 : 1 ;        // dead at the tolerance, or past the bottom of the front
        }) ;
        rp.rsize [f] = size + rm_all * cn ;
        if (ndead > 0) for (Long j = 0 ; j < fn ; j++) rp.Stair [S->Rp [f] + j] = fm ;   // no staircase to exploit after the permutation
    }
    // where the S entries and the children's columns land in the permuted fronts
    const Long anz = S->anz ;
    hostvec<Long> findex ((size_t) std::max ((Long) 1, anz)) ;
    hostvec<int32_t> rjmap (NC.Rjmap.begin ( ), NC.Rjmap.end ( )) ;
    if (anz > 0 && hipMemcpy (findex.data ( ), NC.d_findex, (size_t) anz * sizeof (Long), hipMemcpyDeviceToHost) != hipSuccess) return 2 ;
    for (Long f = 0 ; f < nf ; f++)
    {
        const Long fn = S->Rp [f+1] - S->Rp [f] ;
        const Long *pos = rp.colpos.data ( ) + S->Rp [f] ;
        if (rp.touched [f] && fn > 0)
            for (Long q = NC.SGlob [f] ; q < NC.SGlob [f] + std::max ((Long) 0, S->Sp [S->Sleft [S->Super [f+1]]] - S->Sp [S->Sleft [S->Super [f]]]) ; q++)
            {
                const Long i = findex [q] / fn, j = findex [q] - i * fn ;
                findex [q] = i * fn + pos [j] ;
            }
        const Long pid = S->Parent [f] ;
        if (pid < nf && rp.touched [pid])
        {
            const Long *ppos = rp.colpos.data ( ) + S->Rp [pid] ;
            int32_t *m = rjmap.data ( ) + S->RjmapOffsets [f] ;
            for (Long j = 0 ; j < fn - (S->Super [f+1] - S->Super [f]) ; j++) m [j] = (int32_t) ppos [m [j]] ;
        }
    }
    if (hipMalloc (&rp.d_findex, (size_t) std::max ((Long) 1, anz) * sizeof (Long)) != hipSuccess) { rp.d_findex = NULL ; return 1 ; }
    if (hipMalloc (&rp.d_Rjmap, std::max (rjmap.size ( ), (size_t) 1) * sizeof (int32_t)) != hipSuccess) { rp.d_Rjmap = NULL ; return 1 ; }
    if (hipMemcpy (rp.d_findex, findex.data ( ), (size_t) anz * sizeof (Long), hipMemcpyHostToDevice) != hipSuccess
     || hipMemcpy (rp.d_Rjmap, rjmap.data ( ), rjmap.size ( ) * sizeof (int32_t), hipMemcpyHostToDevice) != hipSuccess) return 2 ;
    return 0 ;
}

// Numeric factorization with SPQR's rank detection (SPQR/Source/spqr_front.cpp:424-479), which the
// reference leaves to its CPU path (spqr_analyze.cpp:106-125 turns the GPU off when tol >= 0).  First an
// ordinary factorization; if it shows dead pivot columns, the fronts are factorized again with the dead
// columns moved out of the way (RankPlan), which gives SPQR's squeezed R: rank rows, Rdead, dead columns
// holding the rows above their would-be pivot.
// Every pass is reviewed against the CPU rule in both directions (review_pass): live columns that must
// die, dead ones that must come back -- the first pass spends a row on every column, dead ones
// included, and can so starve the column after a dead one of the only row it had.  The passes end when
// a pass confirms its whole plan, which is then the CPU rule's plan.  Two or three passes for clear-cut
// deficiencies, 64 are allowed.  tol < 0: no rank detection.
extern "C" int spqrgpu_amd_factorize_tol
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax, double tol,
    spqr_amd_numeric **num_out
)
{
    if (!S || !Ap || !Ax || !num_out || (!Ai && S->anz > 0)) return 3 ;
    if (tol < 0) return factorize_impl (S, Ap, Ai, Ax, NULL, NULL, false, NULL, NULL, 0, NULL, NULL, num_out) ;
    if (!S->do_rank_detection) return 3 ;           // the fronts have no room for rows that dead columns free
    spqr_amd_numeric *num = NULL ;
    int code = factorize_impl (S, Ap, Ai, Ax, NULL, NULL, false, NULL, NULL, 0, NULL, NULL, &num) ;
    if (code != 0) return code ;
    try
    {
        hostvec<char> dead ((size_t) std::max ((Long) 1, S->n), 0) ;
        hostvec<char> revived ((size_t) std::max ((Long) 1, S->n), 0) ;
        double wscale = 0, wssq = 1 ;       // norm of the dead columns' norms, scaled as dnrm2 (spqr_front.cpp:431-451)
        const bool verbose = getenv ("GPUQR_VERBOSE") != NULL ;
        Long passes = 1 ;
        Long changed = review_pass (S, num, NULL, tol, dead.data ( ), revived.data ( )) ;
        while (changed > 0)
        {
            if (passes >= 64) { spqr_amd_numeric_free (num) ; return 2 ; }
            spqr_amd_numeric_free (num) ;
            num = NULL ;
            RankPlan rp ;
            code = build_rank_plan (S, dead.data ( ), rp) ;
            if (code == 0) code = factorize_impl (S, Ap, Ai, Ax, NULL, NULL, false, NULL, NULL, 0, &rp, NULL, &num) ;
            if (code != 0) return code ;
            passes++ ;
            changed = review_pass (S, num, &rp, tol, dead.data ( ), revived.data ( )) ;
            if (verbose)
            {
                Long nplan = 0, nnow = 0 ;
                for (Long j = 0 ; j < S->n ; j++) { nplan += (rp.dead [j] != 0) ; nnow += (dead [j] != 0) ; }
                fprintf (stderr, "  [spqrgpu] tolerance pass %ld: plan had %ld dead columns, %ld decisions changed, %ld dead now\n",
                    (long) passes, (long) nplan, (long) changed, (long) nnow) ;
                if (getenv ("GPUQR_VERBOSE") [0] == '2')
                    for (Long j = 0 ; j < S->n ; j++)
                        if ((rp.dead [j] != 0) != (dead [j] != 0))
                            fprintf (stderr, "      column %ld: %s (norm %.3e, tol %.3e)\n", (long) j,
                                dead [j] ? "dies" : "comes back", rp.dead [j] ? rp.deadnorm [j] : -1.0, tol) ;
            }
            wscale = 0 ; wssq = 1 ;
            for (Long j = 0 ; j < S->n ; j++)
            {
                const double wk = rp.dead [j] ? rp.deadnorm [j] : 0 ;
                if (!(wk > 0)) continue ;
                if (wscale < wk) { const double rr = wscale / wk ; wssq = 1 + wssq * rr * rr ; wscale = wk ; }
                else { const double rr = wk / wscale ; wssq += rr * rr ; }
            }
        }
        num->numPasses = passes ;
        num->norm_E_fro = wscale * sqrt (wssq) ;
    }
    catch (std::bad_alloc &)
    {
        if (num) spqr_amd_numeric_free (num) ;
        return 1 ;
    }
    *num_out = num ;
    return 0 ;
}

// -----------------------------------------------------------------------------
// H kept on the device: Q'B for right-hand sides that arrive after the factorization
// -----------------------------------------------------------------------------

extern "C" void spqr_amd_qt_free (spqr_amd_qt *qt)
{
    if (!qt) return ;
    int cur = -1 ;
    (void) hipGetDevice (&cur) ;
    const int dev = qt->device ;            // read before the object is gone
    if (dev >= 0 && dev != cur) (void) hipSetDevice (dev) ;
    qt->~spqr_amd_qt ( ) ;
    host_free (qt) ;
    if (cur >= 0 && dev >= 0 && dev != cur) (void) hipSetDevice (cur) ;
}

extern "C" int spqrgpu_amd_factorize_keep
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax, int64_t nrhs,
    spqr_amd_numeric **num_out, spqr_amd_qt **qt_out
)
{
    if (qt_out) *qt_out = NULL ;
    if (!S || !Ap || !Ax || !num_out || !qt_out || (!Ai && S->anz > 0) || nrhs < 1 || nrhs > 256) return 3 ;
    if (S->do_rank_detection || (Long) S->PLinv.size ( ) < S->m) return 3 ;
    spqr_amd_qt *qt = (spqr_amd_qt *) host_calloc (1, sizeof (spqr_amd_qt)) ;
    if (!qt) return 1 ;
    new (qt) spqr_amd_qt ( ) ;
    int code = 0 ;
    if (hipGetDevice (&qt->device) != hipSuccess) code = 2 ;
    qt->m = S->m ; qt->n = S->n ; qt->nrhs = nrhs ;
    // B and Q'B on the device; sized before the factorization, whose row lists point into them
    Long rank = 0 ;
    for (Long f = 0 ; f < S->nf ; f++) rank += std::min (S->Fm [f], S->Super [f+1] - S->Super [f]) ;
    if (code == 0 && (qt->dB.ensure ((size_t) std::max ((Long) 1, S->m * nrhs) * sizeof (double)) != hipSuccess
        || qt->dC.ensure ((size_t) std::max ((Long) 1, rank * nrhs) * sizeof (double)) != hipSuccess)) code = 1 ;
    if (code == 0)
    {
        try { code = factorize_impl (S, Ap, Ai, Ax, NULL, NULL, false, NULL, NULL, 0, NULL, qt, num_out) ; }
        catch (std::bad_alloc &) { code = 1 ; }
    }
    if (code != 0) { spqr_amd_qt_free (qt) ; return code ; }
    *qt_out = qt ;
    return 0 ;
}

extern "C" int64_t spqr_amd_qt_rank (const spqr_amd_qt *qt) { return qt ? qt->rank : -1 ; }

// C = the first `rank` rows of Q'B, in the row order of R (the rows a solve needs: R x = C), for B
// m-by-nrhs: the rows of B go to the right-hand-side blocks of the fronts that hold the matching
// rows of S, every front's block reflectors are applied to its block in the rounds of the
// factorization's own schedule, a front's rows below its rank go on to its parent's block, and the
// rows above it are the result.  What SuiteSparseQR_qmult (SPQR_QTX) does with H on the host
// (SPQR/Source/SuiteSparseQR_qmult.cpp, spqr_happly.cpp), restricted to the rows of R.
extern "C" int spqrgpu_amd_qtmult
(
    spqr_amd_qt *qt, const double *B, int64_t ldb, double *C, int64_t ldc
)
{
    if (!qt || !qt->rp || !B || !C || ldb < qt->m || ldc < qt->rank) return 3 ;
    int code = 0 ;
    int cur = -1 ;
    if (hipGetDevice (&cur) != hipSuccess || cur != qt->device) return 3 ;
    const Long m = qt->m, k = qt->nrhs ;
    if (ldb == m) HIPCHK (hipMemcpyAsync (qt->dB.p, B, (size_t) (m * k) * sizeof (double), hipMemcpyHostToDevice, 0)) ;
    else HIPCHK (hipMemcpy2DAsync (qt->dB.p, (size_t) m * sizeof (double), B, (size_t) ldb * sizeof (double),
        (size_t) m * sizeof (double), (size_t) k, hipMemcpyHostToDevice, 0)) ;
    HIPCHK (qr_launch_copy_rows ((const QrRowCopy *) qt->dScatter.p, qt->nscatter, (int) k, m, 1, 0)) ;
    {
        const int32_t nrounds = gpuqr_replay_rounds (qt->rp) ;
        for (int32_t r = 0 ; r < nrounds ; r++)
        {
            if (gpuqr_replay_round (qt->rp, r, (void *) 0) != GPUQR_SUCCESS) { code = 2 ; goto done ; }
            const Long p0 = qt->push_ptr [r], p1 = qt->push_ptr [r+1] ;
            if (p1 > p0) HIPCHK (qr_launch_copy_rows ((const QrRowCopy *) qt->dPush.p + p0, p1 - p0, (int) k, 1, 1, 0)) ;
            if ((r & 511) == 511) HIPCHK (hipStreamSynchronize (0)) ;      // bound on the launches left queued
        }
    }
    HIPCHK (qr_launch_copy_rows ((const QrRowCopy *) qt->dGather.p, qt->ngather, (int) k, 1, qt->rank, 0)) ;
    if (qt->rank > 0)
    {
        if (ldc == qt->rank) HIPCHK (hipMemcpyAsync (C, qt->dC.p, (size_t) (qt->rank * k) * sizeof (double), hipMemcpyDeviceToHost, 0)) ;
        else HIPCHK (hipMemcpy2DAsync (C, (size_t) ldc * sizeof (double), qt->dC.p, (size_t) qt->rank * sizeof (double),
            (size_t) qt->rank * sizeof (double), (size_t) k, hipMemcpyDeviceToHost, 0)) ;
    }
done:
    if (hipStreamSynchronize (0) != hipSuccess && code == 0) code = 2 ;
    return code ;
}

// What SPQR's numeric phase hands to spqrgpu_kernel and takes back from it, on plain arrays
// (SPQR/SPQRGPU/spqrgpu_kernel.cpp:63-782): Sx are the values of S = A(P,Q) in row form
// (Blob->Sx); the R blocks are written column-packed, front after front in postorder, stage after
// stage, starting at Stack_head (Work[0].Stack_head), and Rblock_offset[f] says where front f's
// block begins relative to it (the adapter turns that into QRnum->Rblock[f]); Rdead[k] is set to 1
// for pivot columns past the bottom of their front (entries are only ever set, never cleared).
extern "C" int spqrgpu_amd_kernel
(
    const spqr_amd_symbolic *S, const double *Sx,
    double *Stack_head, int64_t stack_room,
    int64_t *Rblock_offset, char *Rdead,
    int64_t *sumfrank, int64_t *maxfrank, int64_t *stack_used,
    double *gpuKernelTime, int64_t *gpuFlops, int64_t *gpuNumKernelLaunches
)
{
    if (!S || (!Sx && S->anz > 0) || (!Stack_head && S->rnz > 0) || !Rblock_offset || !Rdead) return 3 ;
    spqr_amd_numeric *num = NULL ;
    const int code = factorize_impl (S, NULL, NULL, Sx, NULL, NULL, false, NULL, Stack_head, stack_room, NULL, NULL, &num) ;
    if (code != 0) return code ;
    for (Long f = 0 ; f < S->nf ; f++) Rblock_offset [f] = num->Rblock [f] ;
    for (Long k = 0 ; k < S->n ; k++) if (num->Rdead [k]) Rdead [k] = 1 ;
    if (sumfrank) *sumfrank = num->sumfrank ;
    if (maxfrank) *maxfrank = num->maxfrank ;
    if (stack_used) *stack_used = num->rsize ;
    if (gpuKernelTime) *gpuKernelTime = num->gpuKernelTime ;
    if (gpuFlops) *gpuFlops = num->gpuFlops ;
    if (gpuNumKernelLaunches) *gpuNumKernelLaunches = num->gpuNumKernelLaunches ;
    spqr_amd_numeric_free (num) ;
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
    hostvec<Long> next ;
    if (fill) { try { next.assign (Rcp, Rcp + n) ; } catch (std::bad_alloc &) { return 1 ; } }
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

// X = E (R \ C) on the packed R blocks, front by front from the last one back: the multifrontal part of
// spqr_rsolve (SPQR/Source/spqr_rsolve.cpp:136-338 with keepH = false, no singletons).  C is rank-by-nrhs
// (leading dimension ldc) in the row order of R, X is nacols-by-nrhs with leading dimension nacols.
// Columns of the factorization from nacols on are right-hand sides that were appended to A and are
// passed over (:264-267); dead pivot columns get the basic solution's zeros (:209-221); Qfill == NULL
// gives X = R \ C.
extern "C" int spqr_amd_rsolve
(
    const spqr_amd_symbolic *S, const spqr_amd_numeric *num, const int64_t *Qfill, int64_t nacols,
    int64_t nrhs, const double *C, int64_t ldc, double *X
)
{
    if (!S || !num || !num->Rstack || !num->Rblock || !num->Rdead || nrhs < 0 || nacols < 0 || nacols > S->n) return 3 ;
    if (nrhs > 0 && (!C || !X)) return 3 ;
    const Long nf = S->nf, n = nacols, rank = num->sumfrank ;
    if (ldc < std::max ((Long) 1, rank)) return 3 ;
    for (Long f = 0 ; f < nf ; f++) if (num->Rblock [f] < 0) return 3 ;      // a part of R lives elsewhere
    // X has nacols rows: the permutation may only send the first nacols columns there
    if (Qfill) for (Long j = 0 ; j < n ; j++) if (Qfill [j] < 0 || Qfill [j] >= n) return 3 ;
    try
    {
        const Long maxfrank = std::max ((Long) 1, num->maxfrank) ;
        hostvec<const double *> Rcolp ((size_t) maxfrank) ;
        hostvec<Long> Rlive ((size_t) maxfrank) ;
        hostvec<double> W ((size_t) (maxfrank * std::max ((Long) 1, nrhs))) ;
        for (Long kk = 0 ; kk < nrhs ; kk++) for (Long i = 0 ; i < n ; i++) X [i + kk * n] = 0 ;
        const char *Rdead = num->Rdead ;
        Long row2 = rank ;
        for (Long f = nf - 1 ; f >= 0 ; f--)
        {
            const double *R = num->Rstack + num->Rblock [f] ;
            const Long col1 = S->Super [f], fp = S->Super [f+1] - col1 ;
            const Long pr = S->Rp [f], fn = S->Rp [f+1] - pr ;
            // the live pivot columns of the block: one row of R each
            Long rm = 0, k = 0 ;
            for ( ; k < fp ; k++)
            {
                if (!Rdead [col1 + k])
                {
                    if (rm >= maxfrank) return 3 ;
                    Rcolp [rm] = R ; Rlive [rm] = col1 + k ; rm++ ;
                }
                R += rm ;
            }
            const Long row1 = row2 - rm ;
            if (row1 < 0) return 3 ;
            // W = C (row1:row2-1, :)
            for (Long kk = 0 ; kk < nrhs ; kk++)
                for (Long i = 0 ; i < rm ; i++) W [i + kk * rm] = C [row1 + i + kk * ldc] ;
            // W -= R2 x2: the columns right of the pivots, whose x is known from the fronts above
            for ( ; k < fn ; k++)
            {
                const Long j = S->Rj [pr + k] ;
                if (j >= n) break ;             // an appended right-hand side, and so are all further ones
                const Long ii = Qfill ? Qfill [j] : j ;
                if (!Rdead [j])
                    for (Long kk = 0 ; kk < nrhs ; kk++)
                    {
                        const double xi = X [ii + kk * n] ;
                        if (xi != 0) for (Long i = 0 ; i < rm ; i++) W [i + kk * rm] -= R [i] * xi ;
                    }
                R += rm ;
            }
            // the squeezed triangle
            for (k = rm - 1 ; k >= 0 ; k--)
            {
                const double *Rk = Rcolp [k] ;
                const Long j = Rlive [k] ;
                if (j >= n) continue ;
                const Long ii = Qfill ? Qfill [j] : j ;
                for (Long kk = 0 ; kk < nrhs ; kk++)
                {
                    const double xi = W [k + kk * rm] / Rk [k] ;
                    X [ii + kk * n] = xi ;
                    if (xi != 0) for (Long i = 0 ; i < k ; i++) W [i + kk * rm] -= Rk [i] * xi ;
                }
            }
            row2 = row1 ;
        }
        if (row2 != 0) return 3 ;
    }
    catch (std::bad_alloc &) { return 1 ; }
    return 0 ;
}

// X = R' \ B by forward substitution on the packed R blocks of num, on the host: the multifrontal part of
// spqr_private_rtsolve (SPQR/Source/SuiteSparseQR_expert.cpp:367-470, system SPQR_RTX_EQUALS_B; the
// caller applies E' when SPQR_RTX_EQUALS_ETB is wanted).  B is n-by-nrhs in the factorization's column
// order (leading dimension ldb >= n), X rank-by-nrhs (leading dimension ldx >= rank) in the row order
// of R.  Fronts in increasing order: a front first takes what the fronts before it contributed to its
// pivot columns, solves its triangle, then passes its share on to its non-pivotal columns.  Entries of
// B in dead columns are ignored (they belong to no row of R).
extern "C" int spqr_amd_rtsolve
(
    const spqr_amd_symbolic *S, const spqr_amd_numeric *num, int64_t nrhs, const double *B, int64_t ldb,
    double *X, int64_t ldx
)
{
    if (!S || !num || !num->Rstack || !num->Rblock || !num->Rdead || nrhs < 0) return 3 ;
    if (nrhs > 0 && (!B || !X)) return 3 ;
    const Long nf = S->nf, n = S->n, rank = num->sumfrank ;
    if (ldb < std::max ((Long) 1, n) || ldx < std::max ((Long) 1, rank)) return 3 ;
    for (Long f = 0 ; f < nf ; f++) if (num->Rblock [f] < 0) return 3 ;
    try
    {
        hostvec<double> G ((size_t) std::max ((Long) 1, n)) ;
        const char *Rdead = num->Rdead ;
        for (Long kk = 0 ; kk < nrhs ; kk++)
        {
            for (Long j = 0 ; j < n ; j++) G [j] = B [j + kk * ldb] ;
            double *x = X + kk * ldx ;
            Long row1 = 0 ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const double *R = num->Rstack + num->Rblock [f] ;
                const Long col1 = S->Super [f], fp = S->Super [f+1] - col1 ;
                const Long pr = S->Rp [f], fn = S->Rp [f+1] - pr ;
                Long rm = 0 ;
                for (Long k = 0 ; k < fn ; k++)
                {
                    const bool live = (k < fp && !Rdead [col1 + k]) ;
                    if (live) rm++ ;
                    const Long j = S->Rj [pr + k] ;
                    // column k of the block: rm entries, rows row1 .. row1+rm-1; the last one is the
                    // diagonal when the column is a live pivot
                    const Long above = live ? rm - 1 : rm ;
                    double sum = 0 ;
                    for (Long i = 0 ; i < above ; i++) sum += R [i] * x [row1 + i] ;
                    if (live)
                    {
                        if (row1 + rm > rank) return 3 ;
                        x [row1 + rm - 1] = (G [j] - sum) / R [rm - 1] ;
                    }
                    else if (j < n) G [j] -= sum ;
                    R += rm ;
                }
                row1 += rm ;
            }
            if (row1 != rank) return 3 ;
        }
    }
    catch (std::bad_alloc &) { return 1 ; }
    return 0 ;
}

// t = R y (transposed == 0; y of size n in the factorization's column order, t of size rank) or
// z = R' t (transposed != 0; z of size n), on the packed R blocks of num, on a few host threads.
// The two products are what check_r_factor needs of R (SPQR/Tcov/qrtest.cpp:1218-1257 forms
// R'R - (AP)'(AP) explicitly; R' (R y) against (AP)' ((AP) y) is the same test through a vector, at
// sizes where the explicit product cannot be formed).  Fronts own disjoint rows of R, so R y needs no
// synchronisation; for R' t every thread sums into a vector of its own.
extern "C" int spqr_amd_rmult
(
    const spqr_amd_symbolic *S, const spqr_amd_numeric *num, int transposed, const double *in, double *out
)
{
    if (!S || !num || !num->Rstack || !num->Rblock || !num->Rdead || !in || !out) return 3 ;
    const Long nf = S->nf, n = S->n, rank = num->sumfrank ;
    for (Long f = 0 ; f < nf ; f++) if (num->Rblock [f] < 0) return 3 ;
    try
    {
        // first row of every front, block sizes for the split
        hostvec<Long> row1 ((size_t) nf + 1), work ((size_t) nf + 1) ;
        const char *Rdead = num->Rdead ;
        Long r = 0, w = 0 ;
        for (Long f = 0 ; f < nf ; f++)
        {
            row1 [f] = r ; work [f] = w ;
            const Long col1 = S->Super [f], fp = S->Super [f+1] - col1, fn = S->Rp [f+1] - S->Rp [f] ;
            Long rm = 0 ;
            for (Long k = 0 ; k < fp ; k++) if (!Rdead [col1 + k]) rm++ ;
            r += rm ; w += rm * fn + 1 ;
        }
        row1 [nf] = r ; work [nf] = w ;
        if (r != rank) return 3 ;
        unsigned hw = std::thread::hardware_concurrency ( ) ;
        const char *env = getenv ("OMP_NUM_THREADS") ;
        if (env && atoi (env) > 0) hw = (unsigned) atoi (env) ;
        const unsigned nt = (unsigned) std::max<Long> (1, std::min<Long> ((Long) std::min (hw, 32u), w / 2000000 + 1)) ;
        hostvec<double> zs ;
        if (transposed && nt > 1) zs.assign ((size_t) n * (nt - 1), 0.0) ;
        if (transposed) for (Long j = 0 ; j < n ; j++) out [j] = 0 ;
        else for (Long i = 0 ; i < rank ; i++) out [i] = 0 ;
        auto body = [&] (unsigned t)
        {
            const Long wlo = w * t / nt, whi = w * (t + 1) / nt ;
            double *z = (!transposed || t == 0) ? out : zs.data ( ) + (size_t) n * (t - 1) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                if (work [f] < wlo || work [f] >= whi) continue ;
                const double *R = num->Rstack + num->Rblock [f] ;
                const Long col1 = S->Super [f], fp = S->Super [f+1] - col1 ;
                const Long pr = S->Rp [f], fn = S->Rp [f+1] - pr, r1 = row1 [f] ;
                Long rm = 0 ;
                for (Long k = 0 ; k < fn ; k++)
                {
                    if (k < fp && !Rdead [col1 + k]) rm++ ;
                    const Long j = S->Rj [pr + k] ;
                    if (transposed)
                    {
                        double sum = 0 ;
                        for (Long i = 0 ; i < rm ; i++) sum += R [i] * in [r1 + i] ;
                        z [j] += sum ;
                    }
                    else
                    {
                        const double yj = in [j] ;
                        if (yj != 0) for (Long i = 0 ; i < rm ; i++) z [r1 + i] += R [i] * yj ;
                    }
                    R += rm ;
                }
            }
        } ;
        if (nt == 1) body (0) ;
        else
        {
            std::vector<std::thread> th ;
            unsigned started = 0 ;
            try { for (unsigned t = 1 ; t < nt ; t++) { th.emplace_back (body, t) ; started = t ; } }
            catch (...) { }
            body (0) ;
            for (auto &x : th) x.join ( ) ;
            for (unsigned t = started + 1 ; t < nt ; t++) body (t) ;     // threads that could not be had
            if (transposed)
                for (unsigned t = 1 ; t < nt ; t++)
                {
                    const double *z = zs.data ( ) + (size_t) n * (t - 1) ;
                    for (Long j = 0 ; j < n ; j++) out [j] += z [j] ;
                }
        }
    }
    catch (std::bad_alloc &) { return 1 ; }
    return 0 ;
}

// min || A x - B ||_2 from a kept factorization: X = E (R \ (Q'B)), Q'B on the device
// (spqrgpu_amd_qtmult), the back substitution on the host (spqr_amd_rsolve).  What
// SuiteSparseQR_solve (SPQR_RETX_EQUALS_B) composed with SuiteSparseQR_qmult (SPQR_QTX) gives on a
// factorization object in the reference (SPQR/Source/SuiteSparseQR_expert.cpp), where both need the CPU.
// B is m-by-nrhs with leading dimension ldb, nrhs the count given to spqrgpu_amd_factorize_keep;
// X is n-by-nrhs with leading dimension n.
extern "C" int spqrgpu_amd_solve
(
    spqr_amd_qt *qt, const spqr_amd_symbolic *S, const spqr_amd_numeric *num,
    const double *B, int64_t ldb, double *X
)
{
    if (!qt || !S || !num || !B || !X || qt->n != S->n || qt->m != S->m || (Long) S->Qfill.size ( ) < S->n) return 3 ;
    if (num->sumfrank != qt->rank) return 3 ;
    const Long rank = qt->rank, k = qt->nrhs ;
    double *Cq = (double *) host_malloc ((size_t) std::max ((Long) 1, rank * k) * sizeof (double)) ;
    if (!Cq) return 1 ;
    int code = spqrgpu_amd_qtmult (qt, B, ldb, Cq, std::max ((Long) 1, rank)) ;
    if (code == 0) code = spqr_amd_rsolve (S, num, S->Qfill.data ( ), S->n, k, Cq, std::max ((Long) 1, rank), X) ;
    host_free (Cq) ;
    return code ;
}

// -----------------------------------------------------------------------------
// subtree driver: one process's share of the assembly tree
// -----------------------------------------------------------------------------

// Factorize the fronts f with mine[f] != 0 on the current device (in as many groups as the declared
// device memory asks for).  Children of those fronts that belong to somebody else arrive as compact
// cm-by-cn contribution blocks in limbo_in[c] (host); fronts whose parent belongs to somebody else
// leave theirs in num->Cstack at Cblock[f].  This is the reference's stage / limbo mechanism
// (spqrgpu_kernel.cpp:556-611,687-724) with "a later stage" generalised to "another process".
extern "C" int spqrgpu_amd_factorize_subset
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    const char *mine, const double *const *limbo_in, spqr_amd_numeric **num_out
)
{
    // (Ap == NULL: Ax holds the values of S = A(P,Q) in row form, SPQR's Sx, as in spqrgpu_amd_kernel)
    if (!S || !Ax || !mine || !num_out || (Ap && !Ai && S->anz > 0)) return 3 ;
    return factorize_impl (S, Ap, Ai, Ax, mine, limbo_in, false, NULL, NULL, 0, NULL, NULL, num_out) ;
}

// The same with the contribution blocks staying on the device on both sides: limbo_in[c] are device
// addresses on the current device (a block received from another GPU, or left by an earlier call),
// used where they lie; the blocks that leave are written to c_out_dev (device, room for
// spqrgpu_amd_subset_csize doubles) at Cblock[f].  Nothing crosses host memory.
extern "C" int spqrgpu_amd_factorize_subset_dev
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    const char *mine, const double *const *limbo_in_dev, double *c_out_dev, spqr_amd_numeric **num_out
)
{
    if (!S || !Ax || !mine || !num_out || (Ap && !Ai && S->anz > 0)) return 3 ;
    if (!c_out_dev && spqrgpu_amd_subset_csize (S, mine) > 0) return 3 ;
    return factorize_impl (S, Ap, Ai, Ax, mine, limbo_in_dev, true, c_out_dev, NULL, 0, NULL, NULL, num_out) ;
}

// Doubles of contribution blocks that leave a subset call: the fronts of the subset whose parent is
// not in it.
extern "C" int64_t spqrgpu_amd_subset_csize (const spqr_amd_symbolic *S, const char *mine)
{
    if (!S || !mine) return -1 ;
    Long total = 0 ;
    for (Long f = 0 ; f < S->nf ; f++)
    {
        if (!mine [f] || S->Parent [f] >= S->nf || mine [S->Parent [f]]) continue ;
        total += S->Cm [f] * ((S->Rp [f+1] - S->Rp [f]) - (S->Super [f+1] - S->Super [f])) ;
    }
    return total ;
}
