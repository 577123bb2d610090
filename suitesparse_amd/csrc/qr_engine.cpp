// Host side of the front-batch QR engine: builds the complete task schedule
// for a batch of fronts, uploads it once, replays it round by round on one HIP
// stream, and pulls the R factors back while later rounds still run.
//
// Replaces the reference's host loop fillWorkQueue -> transferData ->
// launchKernel -> postProcess (GPUQREngine/Source/GPUQREngine_Internal.cpp:57-73)
// and the Scheduler / BucketList / LLBundle classes behind it
// (GPUQREngine/Include/GPUQREngine_Scheduler.hpp:30-167,
//  GPUQREngine_BucketList.hpp:25-166, GPUQREngine_LLBundle.hpp:27-94).
//
// Numerical method per front (fm x fn, row-major, staircase Stair): right-
// looking blocked Householder QR with 32-column panels.  A panel's rows
// [32p, e_p) are cut into chunks of 256 rows that are factorized
// independently (level 0); the 32-row R factors on top of the chunks are then
// merged 8 at a time (levels 1, 2, ...) until one R remains in the diagonal
// block - a communication-avoiding (TSQR) panel, the generalisation of the
// reference's bundles of three 32-row tiles (its tile schedule:
// GPUQREngine/Source/BucketList/BucketList_CreateBundles.cpp:15-54,
// LLBundle_Advance.cpp:19-80).  Every factorize task leaves a VT tile, and
// apply tasks push its block reflector through the trailing columns, level by
// level.  Round numbers encode the dependencies: a task of round r reads only
// data produced in rounds < r.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>
#include <algorithm>
#include <new>

#include "qr_tasks.h"
#include "../../include/gpuqrengine_amd.h"

extern "C" int        qr_task_kernel_id (int type, int tier) ;
extern "C" hipError_t qr_launch_pack_r (const QrPackRTask *d_tasks, const long long *d_tile_prefix, int nfronts,
    long long ntiles, hipStream_t stream) ;
extern "C" hipError_t qr_launch_tasks (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream) ;
extern "C" hipError_t qr_launch_tasks_ev (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream,
    hipEvent_t stop) ;
extern "C" hipError_t qr_launch_tasks_timed (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream,
    hipEvent_t start, hipEvent_t stop) ;
extern "C" hipError_t qr_kernels_init (void) ;
extern "C" hipError_t qr_launch_persistent (const QrTask *d_tasks, const QrQueueRef *d_refs, int nrefs,
    int *d_counters, int nworkgroups, hipStream_t stream) ;

#define QR_NKERNELS 13     // 0..8 one per task kind and tier; 9..12 the launches that merge tiers (12: and assembly)

// -----------------------------------------------------------------------------
// allocator hooks
// -----------------------------------------------------------------------------

static void *(*g_malloc) (size_t)         = malloc ;
static void *(*g_calloc) (size_t, size_t) = calloc ;
static void  (*g_free)   (void *)         = free ;

extern "C" void gpuqr_set_allocator (void *(*m) (size_t), void *(*c) (size_t, size_t), void (*f) (void *))
{
    g_malloc = m ? m : malloc ;
    g_calloc = c ? c : calloc ;
    g_free   = f ? f : free ;
}

// Host arena of the entry point that allocates nothing: while it is switched on, the planner's
// vectors are carved out of this static block (bump pointer, nothing is given back before the call
// ends) instead of the heap.  Running out of it is reported like any failed allocation.
#define GPUQR_HOST_ARENA_BYTES ((size_t) 128 << 20)
static unsigned char g_host_arena [GPUQR_HOST_ARENA_BYTES] __attribute__ ((aligned (64))) ;
static size_t g_host_arena_pos = 0 ;
// The switch belongs to the calling thread: vectors that other threads allocate meanwhile (a call on
// another device) keep coming from the heap.  The block itself is one per process, so the calls that
// use it take turns (g_host_arena_turn).
static thread_local bool g_host_arena_on = false ;
static std::mutex g_host_arena_turn ;

static inline bool in_host_arena (const void *p)
{
    return (const unsigned char *) p >= g_host_arena && (const unsigned char *) p < g_host_arena + GPUQR_HOST_ARENA_BYTES ;
}

struct HostArenaScope
{
    std::lock_guard<std::mutex> turn ;
    HostArenaScope ( ) : turn (g_host_arena_turn) { g_host_arena_pos = 0 ; g_host_arena_on = true ; }
    ~HostArenaScope ( ) { g_host_arena_on = false ; }
} ;

// the hooks for the other translation units of the library
extern "C" void *qr_host_malloc (size_t bytes) { return g_malloc (bytes) ; }
extern "C" void *qr_host_calloc (size_t n, size_t size) { return g_calloc (n, size) ; }
extern "C" void  qr_host_free (void *p) { g_free (p) ; }

// std::vector storage that honours the hooks and reports failure by bad_alloc
template <class T> struct HookAlloc
{
    typedef T value_type ;
    HookAlloc ( ) { }
    template <class U> HookAlloc (const HookAlloc<U> &) { }
    T *allocate (size_t n)
    {
        if (n > (size_t) -1 / sizeof (T)) throw std::bad_alloc ( ) ;
        if (g_host_arena_on)
        {
            const size_t bytes = (n * sizeof (T) + 63) & ~(size_t) 63 ;
            if (bytes > GPUQR_HOST_ARENA_BYTES - g_host_arena_pos) throw std::bad_alloc ( ) ;
            void *p = g_host_arena + g_host_arena_pos ;
            g_host_arena_pos += bytes ;
            return (T *) p ;
        }
        void *p = g_malloc (n * sizeof (T)) ;
        if (!p) throw std::bad_alloc ( ) ;
        return (T *) p ;
    }
    void deallocate (T *p, size_t) { if (!in_host_arena (p)) g_free (p) ; }
    template <class U> bool operator== (const HookAlloc<U> &) const { return true ; }
    template <class U> bool operator!= (const HookAlloc<U> &) const { return false ; }
} ;
template <class T> using hvec = std::vector<T, HookAlloc<T> > ;

// -----------------------------------------------------------------------------
// small entry points
// -----------------------------------------------------------------------------

extern "C" int gpuqr_version (void) { return 1000 ; }

extern "C" int gpuqr_device_count (void)
{
    int n = 0 ;
    if (hipGetDeviceCount (&n) != hipSuccess) return 0 ;
    return n ;
}

extern "C" size_t gpuqr_max_queue_size (size_t gpuMemorySize)
{
    // ssgpu_maxQueueSize.cpp:6-27: two queue entries per 64 tiles of 32x32
    // doubles of device memory, never fewer than 50000
    const size_t min_queue = 50000 ;
    const size_t tile_bytes = 32 * 32 * sizeof (double) ;
    size_t q = min_queue ;
    if (gpuMemorySize > 1) q = 2 * (gpuMemorySize / tile_bytes) / 64 ;
    return (q < min_queue) ? min_queue : q ;
}

static int64_t *find_staircase_ld (const double *F, int64_t fm, int64_t fn, int64_t ld) ;

extern "C" int64_t *gpuqr_find_staircase (const double *F, int64_t fm, int64_t fn)
{
    return find_staircase_ld (F, fm, fn, fn) ;
}

static int64_t *find_staircase_ld (const double *F, int64_t fm, int64_t fn, int64_t ld)
{
    // GPUQREngine_ExpertDense.cpp:196-217: Stair[j] is the last row of column
    // j holding a nonzero, never above the previous column's stair
    if (!F || fn < 0) return NULL ;
    int64_t *Stair = (int64_t *) g_malloc ((size_t) (fn > 0 ? fn : 1) * sizeof (int64_t)) ;
    if (!Stair) return NULL ;
    int64_t last = 0 ;
    for (int64_t j = 0 ; j < fn ; j++)
    {
        int64_t i = fm - 1 ;
        while (i > last && F [i*ld + j] == 0.0) i-- ;
        Stair [j] = last = i ;
    }
    return Stair ;
}

// -----------------------------------------------------------------------------
// schedule
// -----------------------------------------------------------------------------

// Stream classes.  Every launch goes to the stream of its class; launches of one class run in
// round order, and a launch waits for the other classes only as far back as its tasks depend on
// them, so the three streams drift apart as far as the data dependencies allow.
//   0  critical path of a front: panel factorizations and the updates of the next panel's columns
//   1  bulk updates by 256-row tasks (the bandwidth carrier)
//   2  bulk updates by the smaller row tiers, S-assembly and pack-assembly
#define QR_NCLASS 3
// launch slots of a round: 0..3 factorize tiers, 4..7 bulk apply tiers, 8 assembly, 9..12 apply tiers
// on the critical path
#define QR_NSLOTS 13

struct SchedTask
{
    QrTask  t ;
    int32_t round ;
    int32_t slot ;      // launch slot within the round
    int32_t dep [QR_NCLASS] ;   // latest round of each class that this task needs finished (-1: none)
    int64_t vt ;        // VT tile index or -1
    int32_t nblk ;      // workgroups this task is run by (apply: column blocks; else 1)
    int32_t wait_ctr ;  // work queue: counter (= front) whose earlier rounds this task waits for
    int32_t signal_ctr ;// work queue: counter this task reports to (the parent's, for pack assembly)
    int32_t is_rhs ;    // update task whose columns are the front's right-hand-side block (kept for Q'B replays)
    double  flops ;     // algorithmic flops of this task
    double  bytes ;     // algorithmic (minimum) bytes it moves to and from memory
#ifdef GPUQR_EXPERIMENT_DRY
    int32_t dry ;       // timing experiment only (see below): the update does its arithmetic without touching the block
#endif
} ;

#ifdef GPUQR_EXPERIMENT_DRY
// Timing experiment, compiled into a library of its own (make exp), never into the product: what would
// a step take if the first-level updates of every second panel cost their matrix arithmetic alone,
// i.e. if two consecutive 32-wide panels shared one trip of the trailing block through the registers
// (the 64-reflector update)?  With GPUQR_EXPERIMENT_DRY=1 those update tasks are marked; the kernel then
// works on zeros instead of loading the block and stores nothing.  The results of such a run are
// meaningless; its launch times are those of the same schedule with that traffic removed.
static bool experiment_dry (void)
{
    static int on = -1 ;
    if (on < 0) { const char *e = getenv ("GPUQR_EXPERIMENT_DRY") ; on = (e && *e && *e != '0') ? 1 : 0 ; }
    return on == 1 ;
}
#endif

// -----------------------------------------------------------------------------
// optional per-kernel profile (HIP events around every launch)
// -----------------------------------------------------------------------------

// 0: one launch per (round, kernel) on two streams; 1: one persistent work-queue kernel per call;
// 2: the engine picks per call
static int g_mode = -1 ;
extern "C" void gpuqr_set_mode (int mode) { g_mode = mode ; }
static int engine_mode (void)
{
    if (g_mode < 0) { const char *e = getenv ("GPUQR_MODE") ; g_mode = e ? atoi (e) : 0 ; }
    return g_mode ;
}

struct KernelProfile { double ms, flops, bytes ; int64_t launches, tasks ; } ;
static int g_profile_on = 0 ;
static KernelProfile g_profile [QR_NKERNELS] ;

extern "C" void gpuqr_profile_enable (int on) { g_profile_on = on ; }
extern "C" void gpuqr_profile_reset (void) { memset (g_profile, 0, sizeof (g_profile)) ; }
// Serial debugging mode: off, on by GPUQR_SERIAL=1 in the environment or by gpuqr_set_serial (1).
static int g_serial = -1 ;
static inline bool serial_mode (void)
{
    if (g_serial < 0) { const char *e = getenv ("GPUQR_SERIAL") ; g_serial = (e && *e && *e != '0') ? 1 : 0 ; }
    return g_serial == 1 ;
}
extern "C" void gpuqr_set_serial (int on) { g_serial = on ? 1 : 0 ; }

extern "C" int gpuqr_profile_get (int kid, double *ms, int64_t *launches, int64_t *tasks, double *flops, double *bytes)
{
    if (kid < 0 || kid >= QR_NKERNELS) return -1 ;
    if (ms) *ms = g_profile [kid].ms ;
    if (launches) *launches = g_profile [kid].launches ;
    if (tasks) *tasks = g_profile [kid].tasks ;
    if (flops) *flops = g_profile [kid].flops ;
    if (bytes) *bytes = g_profile [kid].bytes ;
    return 0 ;
}

struct Plan
{
    hvec<SchedTask> tasks ;
    int64_t n_vt = 0 ;
    int64_t flops = 0 ;
    int32_t nrounds = 0 ;
} ;

static inline int tier_of (int K)
{
    return (K <= 32) ? 0 : (K <= 64) ? 1 : (K <= 128) ? 2 : 3 ;
}

// The reference's own flop counts per task (GPUQREngine/Source/TaskDescriptor/TaskDescriptor_flops.cpp),
// with its tile constants replaced by the task's dimensions: flopsFactorizeVT (:18-28) is the formula
// for m = 32*numTiles rows, n = v = 32 columns, flopsFactorize (:34-42) the one without T for any m, n
// with v = min(m, n), flopsApply (:48-53) k*n*(4m - k + 3) for k = 32 reflectors of m rows on n columns.
// Here a panel task has K <= 256 rows (not <= 96) and w <= 32 columns, so the counts are the
// reference's formulas at (K, w): equal to the reference's numbers on its own tile shapes.
static int64_t flops_factorize (int64_t K, int64_t w, bool with_t)
{
    const int64_t m = K, n = w, v = (m < n) ? m : n ;
    if (m <= 0 || n <= 0) return 0 ;
    if (with_t)
        return 2 * (m-1) + v * (6 + 4*m*n + 5*n + 2*m) + ((-4*n - 2*m - 5) * (v*(v+1)/2))
            + 3 * (v*(v+1)*(2*v+1)/6) + (- 2*(m-v+1)*(n-v) - 2*(m-v-1)) ;
    return 2 * (m-1) + v * (6 + 4*m*n + 5*n + 2*m) + ((-4*n - 2*m - 4) * v*(v+1)/2)
        + 2 * v*(v+1)*(2*v+1)/6 + (- 2*(m-v+1)*(n-v) - 2*(m-v-1)) ;
}

static inline int64_t flops_apply (int64_t K, int64_t w, int64_t n)
{
    // TaskDescriptor_flops.cpp:48-53 with m = K rows and k = w reflectors
    return w * n * (4*K - w + 3) ;
}

extern "C" int64_t gpuqr_flops_factorize (int64_t m, int64_t n, int with_t) { return flops_factorize (m, n, with_t != 0) ; }
extern "C" int64_t gpuqr_flops_apply (int64_t m, int64_t k, int64_t n) { return flops_apply (m, k, n) ; }

struct Seg { int32_t row, n ; } ;

static inline int slot_class (int slot)
{
    return (slot < 4 || slot >= 9) ? 0 : ((slot == 7) ? 1 : 2) ;
}
static inline int slot_kernel (int slot) { return (slot < 9) ? slot : slot - 5 ; }

// Look-ahead: trailing widths from this many column tiles up get a separate update task for the
// columns of the next panel, which runs on the critical stream ahead of the bulk of the update.
// Off by default: at 50-60 us per panel step the three extra launches on the critical path cost
// more than the overlap returns (one 4096 x 2048 front: 16.0 ms with, 13.8 ms without; four
// 8192 x 4096 fronts: 55.7 ms with, 56.7 ms without).  GPUQR_LOOKAHEAD_TILES=n turns it on.
static int lookahead_min_tiles (void)
{
    static int v = -1 ;
    if (v < 0) { const char *e = getenv ("GPUQR_LOOKAHEAD_TILES") ; v = e ? atoi (e) : 0 ; if (v < 2) v = 1 << 30 ; }
    return v ;
}

// A block of contribution-block rows that lands in a front: the first (lowest) row of the front
// it writes, and the round of its pack-assembly tasks.  Later rows of the block land further down.
struct AsmEvent { int32_t min_row, round ; } ;

// Emit all factorize/apply tasks of one front.  last [c] is, on entry, the latest round of class c
// whose work the front's first tasks must wait for (assembly into it), and on return the latest
// round of class c used by the front.  Every task gets round = 1 + the latest round it depends on.
// ev [0..nev), sorted by min_row: assembly that the front's panels wait for one by one, a panel
// only for the blocks that reach into the rows it works on.  rows_done (optional) receives, per
// panel and class, the latest round after which the rows of the panel's diagonal block are final;
// fact_done (optional), per panel, the round of its last factorize task: columns [0, 32 (p + 1)) of R are
// final from then on (a column's rows above the panel were finished by earlier panels' updates, which
// the panel's factorization waited for).
static void plan_front
(
    Plan &plan, double *F, int64_t fm, int64_t fn, int64_t ld, const int64_t *Stair,
    int32_t last [QR_NCLASS], bool emit, int64_t rhs_col = 0, int64_t rhs_n = 0,
    const AsmEvent *ev = NULL, size_t nev = 0, hvec<int32_t> *rows_done = NULL, hvec<int32_t> *fact_done = NULL
)
{
    if (fm <= 0 || fn <= 0) return ;
    const int64_t mn = (fm < fn) ? fm : fn ;
    const int64_t npanels = (mn + QR_PANEL_W - 1) / QR_PANEL_W ;
    const int64_t ntiles  = (fn + QR_PANEL_W - 1) / QR_PANEL_W ;
    // Right-hand-side columns [rhs_col, rhs_col + rhs_n) of the front's storage (beyond its fn columns):
    // every panel's reflectors are applied to them, none of them is ever factorized.  They are
    // tracked as one more column tile.
    const bool has_rhs = (rhs_n > 0) ;
    // latest round of each class that wrote column tile t
    hvec<int32_t> lastw [QR_NCLASS] ;
    for (int c = 0 ; c < QR_NCLASS ; c++) lastw [c].assign ((size_t) ntiles + 1, last [c]) ;
    hvec<Seg> tops, next ;
    int64_t e_prev = 0 ;
    const int la_min = lookahead_min_tiles ( ) ;
    size_t ei = 0 ;             // assembly blocks [0, ei) reach into rows that some panel so far works on
    int32_t asm_run = -1 ;      // latest round among them
    if (rows_done) rows_done->assign ((size_t) npanels * QR_NCLASS, -1) ;
    if (fact_done) fact_done->assign ((size_t) npanels, -1) ;       // round of each panel's last factorize task

    for (int64_t p = 0 ; p < npanels ; p++)
    {
        const int64_t c0 = p * QR_PANEL_W ;
        const int64_t w = (fn - c0 < QR_PANEL_W) ? (fn - c0) : QR_PANEL_W ;
        // rows [r0, e) take part: r0 = c0 (everything above is final R), e from the staircase
        const int64_t r0 = c0 ;
        int64_t e = fm ;
        if (Stair != NULL)
        {
            e = Stair [c0 + w - 1] + 1 ;
            if (e > fm) e = fm ;
        }
        int64_t emin = (r0 + w < fm) ? (r0 + w) : fm ;    // at least the diagonal block
        if (e < emin) e = emin ;
        if (e < e_prev) e = e_prev ;                      // fill from earlier panels
        e_prev = e ;
        const int64_t m = e - r0 ;
        const bool trailing = (c0 + w < fn) ;
        const bool wants_vt = trailing || has_rhs ;
        while (ei < nev && ev [ei].min_row < e) { asm_run = std::max (asm_run, ev [ei].round) ; ei++ ; }
        last [2] = std::max (last [2], asm_run) ;

        // ---- factorize tasks, level by level ---------------------------------
        struct FRef { int64_t vt ; int32_t K ; int32_t nseg ; Seg segs [QR_MAX_SEG] ; int32_t level ; } ;
        hvec<FRef> frefs ;
        tops.clear ( ) ;
        const int64_t nchunks = (m + QR_MAX_K - 1) / QR_MAX_K ;
        int32_t level = 0 ;
        int32_t depF [QR_NCLASS] ;
        int32_t rF0 = 0 ;
        for (int c = 0 ; c < QR_NCLASS ; c++)
        {
            depF [c] = lastw [c][p] ;
            if (c == 2) depF [c] = std::max (depF [c], asm_run) ;
            rF0 = std::max (rF0, depF [c] + 1) ;
        }
        for (int64_t c = 0 ; c < nchunks ; c++)
        {
            FRef fr ;
            int64_t a = r0 + c * QR_MAX_K ;
            int64_t b = (a + QR_MAX_K < e) ? (a + QR_MAX_K) : e ;
            fr.K = (int32_t) (b - a) ; fr.nseg = 1 ; fr.level = 0 ;
            fr.segs [0].row = (int32_t) a ; fr.segs [0].n = fr.K ;
            fr.vt = wants_vt ? plan.n_vt++ : -1 ;
            frefs.push_back (fr) ;
            Seg s ; s.row = (int32_t) a ; s.n = (int32_t) ((fr.K < w) ? fr.K : w) ;
            tops.push_back (s) ;
        }
        while (tops.size ( ) > 1)
        {
            level++ ;
            next.clear ( ) ;
            for (size_t g = 0 ; g < tops.size ( ) ; g += QR_MAX_SEG)
            {
                size_t gend = std::min (g + (size_t) QR_MAX_SEG, tops.size ( )) ;
                if (gend - g == 1) { next.push_back (tops [g]) ; continue ; }
                FRef fr ; fr.nseg = (int32_t) (gend - g) ; fr.K = 0 ; fr.level = level ;
                for (size_t s = g ; s < gend ; s++) { fr.segs [s-g] = tops [s] ; fr.K += tops [s].n ; }
                fr.vt = wants_vt ? plan.n_vt++ : -1 ;
                frefs.push_back (fr) ;
                Seg s = tops [g] ; s.n = (int32_t) ((fr.K < w) ? fr.K : w) ;
                next.push_back (s) ;
            }
            tops.swap (next) ;
        }
        const int32_t nlevels = level + 1 ;

        for (size_t i = 0 ; i < frefs.size ( ) ; i++)
        {
            const FRef &fr = frefs [i] ;
            plan.flops += flops_factorize (fr.K, w, wants_vt) ;
            if (!emit) continue ;
            SchedTask st ;
            memset (&st, 0, sizeof (st)) ;
            st.t.type = QR_TASK_FACTORIZE ;
            st.t.tier = tier_of (fr.K) ;
            QrPanelTask &pt = st.t.u.panel ;
            pt.F = F ; pt.VT = NULL ; pt.ld = (int32_t) ld ; pt.col0 = (int32_t) c0 ; pt.w = (int32_t) w ;
            pt.K = fr.K ; pt.nseg = fr.nseg ; pt.c_begin = pt.c_end = 0 ;
            for (int s = 0 ; s < fr.nseg ; s++) { pt.seg_row [s] = fr.segs [s].row ; pt.seg_n [s] = fr.segs [s].n ; }
            st.round = rF0 + fr.level ;
            st.slot = st.t.tier ;
            for (int c = 0 ; c < QR_NCLASS ; c++) st.dep [c] = depF [c] ;
            if (fr.level > 0) st.dep [0] = rF0 + fr.level - 1 ;
            st.vt = fr.vt ;
            st.nblk = 1 ;
            st.flops = (double) flops_factorize (fr.K, w, wants_vt) ;
            st.bytes = 16.0 * fr.K * w + (wants_vt ? 8.0 * QR_VT_DOUBLES : 0.0) ;
            plan.tasks.push_back (st) ;
        }
        lastw [0][p] = rF0 + nlevels - 1 ;
        last [0] = std::max (last [0], rF0 + nlevels - 1) ;
        if (fact_done) (*fact_done) [(size_t) p] = rF0 + nlevels - 1 ;

        // ---- apply tasks ---------------------------------------------------------
        if (trailing)
        {
            // One apply task per factorize task and column range; the kernel cuts a range into
            // column blocks, one workgroup each.  A wide trailing matrix is cut in two ranges: the
            // next panel's tile, updated on the critical stream so that the next factorization can
            // start at once, and the rest.
            const int64_t t0 = p + 1 ;
            const bool split = (p + 1 < npanels) && (ntiles - t0 >= la_min) ;
            for (int part = 0 ; part < (split ? 2 : 1) ; part++)
            {
                const bool critical = split && part == 0 ;
                const int64_t ta = (split && part == 1) ? t0 + 1 : t0 ;             // first tile
                const int64_t tb = critical ? t0 + 1 : ntiles ;                     // one past the last
                const int64_t bt = critical ? 1 : 4 ;   // column tiles per workgroup: 128 columns (64 and 256 measured slower)
                const int64_t cb = ta * QR_PANEL_W ;
                const int64_t ce = std::min (tb * QR_PANEL_W, fn) ;
                const int64_t nblk = (tb - ta + bt - 1) / bt ;
                for (int32_t lv = 0 ; lv < nlevels ; lv++)
                {
                    // what this level's tasks wait for: the last writers of their columns, and the
                    // factorize tasks of the level
                    int32_t dep [QR_NCLASS] = {-1, -1, -1} ;
                    for (int c = 0 ; c < QR_NCLASS ; c++)
                        for (int64_t t = ta ; t < tb ; t++) dep [c] = std::max (dep [c], lastw [c][t]) ;
                    dep [0] = std::max (dep [0], rF0 + lv) ;
                    dep [2] = std::max (dep [2], asm_run) ;
                    const int32_t round = 1 + std::max (dep [0], std::max (dep [1], dep [2])) ;
                    int32_t used [QR_NCLASS] = {-1, -1, -1} ;
                    for (size_t i = 0 ; i < frefs.size ( ) ; i++)
                    {
                        const FRef &fr = frefs [i] ;
                        if (fr.level != lv) continue ;
                        plan.flops += flops_apply (fr.K, w, ce - cb) ;
                        const int tier = tier_of (fr.K) ;
                        const int slot = critical ? 9 + tier : 4 + tier ;
                        used [slot_class (slot)] = round ;
                        if (!emit) continue ;
                        SchedTask st ;
                        memset (&st, 0, sizeof (st)) ;
                        st.t.type = QR_TASK_APPLY ;
                        st.t.tier = tier ;
                        QrPanelTask &pt = st.t.u.panel ;
                        pt.F = F ; pt.VT = NULL ; pt.ld = (int32_t) ld ; pt.col0 = (int32_t) c0 ; pt.w = (int32_t) w ;
                        pt.K = fr.K ; pt.nseg = fr.nseg ; pt.c_begin = (int32_t) cb ; pt.c_end = (int32_t) ce ;
                        pt.cb_cols = (int32_t) (bt * QR_PANEL_W) ;
                        for (int s = 0 ; s < fr.nseg ; s++) { pt.seg_row [s] = fr.segs [s].row ; pt.seg_n [s] = fr.segs [s].n ; }
                        st.round = round ;
                        st.slot = slot ;
                        for (int c = 0 ; c < QR_NCLASS ; c++) st.dep [c] = dep [c] ;
                        st.vt = fr.vt ;
                        st.nblk = (int32_t) nblk ;
#ifdef GPUQR_EXPERIMENT_DRY
                        st.dry = (experiment_dry ( ) && lv == 0 && (p & 1) && !critical) ? 1 : 0 ;
#endif
                        st.flops = (double) flops_apply (fr.K, w, ce - cb) ;
                        // A read + written once; V (K x 32) and the VT tile read once per workgroup
                        st.bytes = 16.0 * fr.K * (double) (ce - cb) + (double) nblk * (8.0 * fr.K * w + 8.0 * QR_VT_DOUBLES) ;
                        plan.tasks.push_back (st) ;
                    }
                    for (int c = 0 ; c < QR_NCLASS ; c++)
                    {
                        if (used [c] < 0) continue ;
                        for (int64_t t = ta ; t < tb ; t++) lastw [c][t] = used [c] ;
                        last [c] = std::max (last [c], used [c]) ;
                    }
                }
            }
        }
        // ---- the same reflectors on the right-hand-side block ----------------------------
        if (has_rhs)
        {
            const int64_t tr = ntiles ;         // the tile that stands for the block
            for (int32_t lv = 0 ; lv < nlevels ; lv++)
            {
                int32_t dep [QR_NCLASS] ;
                for (int c = 0 ; c < QR_NCLASS ; c++) dep [c] = lastw [c][tr] ;
                dep [0] = std::max (dep [0], rF0 + lv) ;
                dep [2] = std::max (dep [2], asm_run) ;
                const int32_t round = 1 + std::max (dep [0], std::max (dep [1], dep [2])) ;
                int32_t used [QR_NCLASS] = {-1, -1, -1} ;
                for (size_t i = 0 ; i < frefs.size ( ) ; i++)
                {
                    const FRef &fr = frefs [i] ;
                    if (fr.level != lv) continue ;
                    plan.flops += flops_apply (fr.K, w, rhs_n) ;
                    const int tier = tier_of (fr.K) ;
                    const int slot = 4 + tier ;
                    used [slot_class (slot)] = round ;
                    if (!emit) continue ;
                    SchedTask st ;
                    memset (&st, 0, sizeof (st)) ;
                    st.t.type = QR_TASK_APPLY ;
                    st.t.tier = tier ;
                    QrPanelTask &pt = st.t.u.panel ;
                    pt.F = F ; pt.VT = NULL ; pt.ld = (int32_t) ld ; pt.col0 = (int32_t) c0 ; pt.w = (int32_t) w ;
                    pt.K = fr.K ; pt.nseg = fr.nseg ; pt.c_begin = (int32_t) rhs_col ; pt.c_end = (int32_t) (rhs_col + rhs_n) ;
                    pt.cb_cols = (int32_t) ((rhs_n + 15) / 16 * 16) ;
                    for (int s = 0 ; s < fr.nseg ; s++) { pt.seg_row [s] = fr.segs [s].row ; pt.seg_n [s] = fr.segs [s].n ; }
                    st.round = round ;
                    st.slot = slot ;
                    for (int c = 0 ; c < QR_NCLASS ; c++) st.dep [c] = dep [c] ;
                    st.vt = fr.vt ;
                    st.nblk = 1 ;
                    st.is_rhs = 1 ;
                    st.flops = (double) flops_apply (fr.K, w, rhs_n) ;
                    st.bytes = 16.0 * fr.K * (double) rhs_n + 8.0 * fr.K * w + 8.0 * QR_VT_DOUBLES ;
                    plan.tasks.push_back (st) ;
                }
                for (int c = 0 ; c < QR_NCLASS ; c++)
                {
                    if (used [c] < 0) continue ;
                    lastw [c][tr] = used [c] ;
                    last [c] = std::max (last [c], used [c]) ;
                }
            }
        }
        // everything that writes rows [c0, c0 + w) has been emitted by now
        if (rows_done) for (int c = 0 ; c < QR_NCLASS ; c++) (*rows_done) [(size_t) p * QR_NCLASS + c] = last [c] ;
    }
}

// Device memory the engine itself takes for one fm-by-fn front (full staircase: an upper bound): a
// 128-byte record per task, 8 bytes per workgroup of an update task, one 8 KB VT tile per panel task
// that somebody applies.  Counted the way plan_front cuts the work (256-row chunks, merges of 8,
// 128-column update blocks), in closed form per panel.  A caller that must fit a batch of fronts into a
// given amount of memory adds this to the fronts themselves.
extern "C" int64_t gpuqr_front_workspace_bytes (int64_t fm, int64_t fn)
{
    if (fm <= 0 || fn <= 0) return 0 ;
    const int64_t mn = (fm < fn) ? fm : fn ;
    const int64_t npanels = (mn + QR_PANEL_W - 1) / QR_PANEL_W ;
    int64_t tasks = 0, refs = 0, vts = 0 ;
    for (int64_t p = 0 ; p < npanels ; p++)
    {
        const int64_t c0 = p * QR_PANEL_W ;
        const int64_t w = (fn - c0 < QR_PANEL_W) ? (fn - c0) : QR_PANEL_W ;
        const int64_t m = fm - c0 ;
        int64_t nodes = 0 ;
        for (int64_t t = (m + QR_MAX_K - 1) / QR_MAX_K ; ; t = (t + QR_MAX_SEG - 1) / QR_MAX_SEG)
        {
            nodes += t ;
            if (t <= 1) break ;
        }
        const int64_t trailing = fn - c0 - w ;
        tasks += nodes ;
        if (trailing > 0)
        {
            tasks += nodes ; vts += nodes ;
            refs += nodes * ((trailing + 127) / 128 + 1) ;
        }
        refs += nodes ;
    }
    return tasks * (int64_t) sizeof (QrTask) + refs * (int64_t) sizeof (QrRef) + vts * (int64_t) (QR_VT_DOUBLES * sizeof (double)) ;
}

extern "C" int64_t gpuqr_front_flops (int64_t fm, int64_t fn, const int64_t *Stair)
{
    try
    {
        Plan plan ;
        { int32_t last [QR_NCLASS] = {-1, -1, -1} ; plan_front (plan, NULL, fm, fn, fn, Stair, last, false) ; }
        return plan.flops ;
    }
    catch (std::bad_alloc &) { return -1 ; }
}

// -----------------------------------------------------------------------------
// the engine proper: fronts already on the device
// -----------------------------------------------------------------------------

#define HIP_TRY(call) do { hipError_t e_ = (call) ; if (e_ != hipSuccess) { \
    if (getenv ("GPUQR_VERBOSE")) fprintf (stderr, "gpuqr: %s failed: %s (%s:%d)\n", #call, \
        hipGetErrorString (e_), __FILE__, __LINE__) ; \
    code = (e_ == hipErrorOutOfMemory) ? GPUQR_OUTOFMEMORY : GPUQR_GPUERROR ; goto cleanup ; } } while (0)

// -----------------------------------------------------------------------------
// per-device resources kept between calls (streams, events, grow-only buffers):
// allocating and freeing device memory synchronises the whole device, so the
// engine does it only when a call needs more than any earlier one
// -----------------------------------------------------------------------------

// What the launch loop needs of a schedule; kept per device so that a caller who promises an
// unchanged problem (same key) skips planning, sorting and the upload.
struct LaunchPlan
{
    hvec<int64_t> rptr ;            // first workgroup of every (round, slot) bucket
    hvec<int32_t> bdep ;            // per bucket and class: latest round depended on
    hvec<int32_t> done_last ;       // per front and class: latest round used
    hvec<double> bflops, bbytes ;   // per bucket: flops and algorithmic bytes of its tasks
    hvec<int64_t> seg_ptr ;         // per front f: [2f] first entry of seg_round, [2f+1] number of entries (0: R goes out whole)
    hvec<int32_t> seg_round ;       // per piece of QR_RSEG_COLS pivot columns: round from which those columns of R are final
    int32_t nrounds = 0 ;
    int64_t nbuckets = 0, ntasks = 0, nrefs = 0 ;
    int64_t flops = 0 ;
    void swap_with (LaunchPlan &o)
    {
        rptr.swap (o.rptr) ; bdep.swap (o.bdep) ; done_last.swap (o.done_last) ; bflops.swap (o.bflops) ; bbytes.swap (o.bbytes) ;
        seg_ptr.swap (o.seg_ptr) ; seg_round.swap (o.seg_round) ;
        std::swap (nrounds, o.nrounds) ; std::swap (nbuckets, o.nbuckets) ; std::swap (ntasks, o.ntasks) ;
        std::swap (nrefs, o.nrefs) ; std::swap (flops, o.flops) ;
    }
} ;

// The packed R of a front with at least QR_RSEG_MIN (4096) pivot rows leaves in pieces of QR_RSEG_COLS (512)
// columns, each as soon as its columns are final, instead of all at once when the front is done: the
// largest fronts finish last, and their R was what the call still had to send after its last kernel
// (96^3 grid: 124 -> 42 ms between the last kernel and the end of the step; nothing for a front of
// 2048 columns, where the extra pack launches cost what the earlier copies save).
// GPUQR_R_PIECES=0: never; =n > 1: from n pivot rows on, in pieces of n/4 columns (for tests).
static int g_rseg_cols = 0, g_rseg_min = 0 ;
static void rseg_setup (void)
{
    if (g_rseg_cols != 0) return ;
    const char *e = getenv ("GPUQR_R_PIECES") ;
    const int v = e ? atoi (e) : 1 ;
    if (v == 0) { g_rseg_cols = 512 ; g_rseg_min = INT32_MAX ; }
    else if (v > 1) { g_rseg_min = std::max (v, 128) ; g_rseg_cols = std::max (32, g_rseg_min / 4 / 32 * 32) ; }
    else { g_rseg_cols = 512 ; g_rseg_min = 4096 ; }
}
#define QR_RSEG_COLS g_rseg_cols
#define QR_RSEG_MIN  g_rseg_min

#define QR_PLAN_SLOTS 4      // displaced schedules kept per device

// bound on the launches left queued at a time: a checkpoint every QR_CK_EVERY launches, QR_CK_SLOTS of them in flight
#define QR_CK_EVERY 512
#define QR_CK_SLOTS 5

struct DeviceCache
{
    bool ready = false ;
    LaunchPlan lp ;
    uint64_t plan_key = 0 ;         // key of the schedule whose tasks sit in d_tasks / d_refs (0: none)
    int64_t plan_nf = 0 ;
    hipStream_t sCompute = NULL, sApply = NULL, sCopy = NULL ;
    hipEvent_t evStart = NULL, evStop = NULL ;
    std::vector<hipEvent_t> evPool [3] ;    // untimed events per task class, reused by every call
    hipEvent_t evCheck [QR_CK_SLOTS][2] = {} ;  // checkpoints of the two compute streams that bound the queued launches
    void *d_tasks = NULL, *d_refs = NULL, *d_vt = NULL, *h_stage = NULL, *d_counters = NULL, *d_pack = NULL ;
    size_t cap_tasks = 0, cap_refs = 0, cap_vt = 0, cap_stage = 0, cap_counters = 0, cap_pack = 0 ;
    // Schedules of earlier calls that another call has displaced, each with its own task and workgroup
    // buffers on the device: a caller that alternates between a few problems (the stages of a staged
    // factorization, the levels of a tree-mapped one: several calls per factorization, the same ones in
    // every factorization) finds each of them again.  Their tasks point into the VT pool, which is
    // shared: a schedule parked while the pool lay elsewhere is not taken back.
    struct Parked
    {
        LaunchPlan lp ; uint64_t key = 0 ; int64_t nf = 0 ;
        void *d_tasks = NULL, *d_refs = NULL, *vt_base = NULL ; size_t cap_tasks = 0, cap_refs = 0 ; uint64_t stamp = 0 ;
    } ;
    Parked parked [QR_PLAN_SLOTS] ;
    void *plan_vt_base = NULL ;     // where the VT pool lay when the active schedule was staged
    uint64_t stamp = 0 ;
    int num_cus = 0 ;
    char *scratch = NULL ;          // this device's copy of the fixed workspace (hipGetSymbolAddress), never freed
    size_t scratch_bytes = 0 ;

    hipError_t ensure (void **p, size_t *cap, size_t need, bool pinned_host)
    {
        if (need <= *cap) return hipSuccess ;
        hipError_t e ;
        if (*p) { e = pinned_host ? hipHostFree (*p) : hipFree (*p) ; *p = NULL ; *cap = 0 ; if (e != hipSuccess) return e ; }
        size_t want = need + need / 4 ;
        e = pinned_host ? hipHostMalloc (p, want, hipHostMallocDefault) : hipMalloc (p, want) ;
        if (e != hipSuccess) { *p = NULL ; return e ; }
        *cap = want ;
        return hipSuccess ;
    }
    // k-th untimed event of class c's pool (created on first use).  Each class has a ring of events: the
    // k-th event of a call is slot k mod ring, so an old slot is recorded again by a later launch of the
    // same class, on the same stream.  A wait on a reused slot then waits for a later launch of a stream
    // that runs in order: never too little.  Why a ring: a process that held more than about 4,500
    // recorded events kept every new process on the machine from initialising the device (they died in
    // their first copy); 3 x 1024 stays clear of that.
    hipEvent_t pooled_event (int c, size_t k)
    {
        static size_t ring = 0 ;
        if (ring == 0) { const char *e = getenv ("GPUQR_EVENT_RING") ; ring = e ? (size_t) std::max (16, atoi (e)) : 1024 ; }
        std::vector<hipEvent_t> &pool = evPool [c] ;
        k %= ring ;
        while (pool.size ( ) <= k)
        {
            hipEvent_t ev = NULL ;
            if (hipEventCreateWithFlags (&ev, hipEventDisableTiming) != hipSuccess) return NULL ;
            pool.push_back (ev) ;
        }
        return pool [k] ;
    }
    // Make the schedule with this key the active one if it is here.  True: dc->lp, d_tasks, d_refs are its.
    bool activate (uint64_t key, int64_t nf)
    {
        if (key == 0) return false ;
        if (plan_key == key && plan_nf == nf && plan_vt_base == d_vt) return true ;
        for (int i = 0 ; i < QR_PLAN_SLOTS ; i++)
        {
            Parked &q = parked [i] ;
            if (q.key != key || q.nf != nf || q.vt_base != d_vt) continue ;
            // swap the active schedule with the parked one
            lp.swap_with (q.lp) ;
            std::swap (plan_key, q.key) ; std::swap (plan_nf, q.nf) ;
            std::swap (d_tasks, q.d_tasks) ; std::swap (d_refs, q.d_refs) ;
            std::swap (cap_tasks, q.cap_tasks) ; std::swap (cap_refs, q.cap_refs) ;
            std::swap (plan_vt_base, q.vt_base) ;
            q.stamp = ++stamp ;
            return true ;
        }
        return false ;
    }
    // The active schedule is about to be replaced: keep it aside if it is worth keeping.
    void park_active ( )
    {
        if (plan_key == 0 || plan_vt_base != d_vt || d_tasks == NULL) { plan_key = 0 ; return ; }
        int slot = 0 ;
        for (int i = 0 ; i < QR_PLAN_SLOTS ; i++)
        {
            if (parked [i].key == 0 || parked [i].vt_base != d_vt) { slot = i ; break ; }
            if (parked [i].stamp < parked [slot].stamp) slot = i ;
        }
        Parked &q = parked [slot] ;
        // the slot's own buffers become the active ones (to be overwritten, grown if too small)
        lp.swap_with (q.lp) ;
        q.key = plan_key ; q.nf = plan_nf ; q.vt_base = plan_vt_base ; q.stamp = ++stamp ;
        std::swap (d_tasks, q.d_tasks) ; std::swap (d_refs, q.d_refs) ;
        std::swap (cap_tasks, q.cap_tasks) ; std::swap (cap_refs, q.cap_refs) ;
        plan_key = 0 ; plan_nf = 0 ; plan_vt_base = NULL ;
    }
    void release ( )
    {
        for (int i = 0 ; i < QR_PLAN_SLOTS ; i++)
        {
            if (parked [i].d_tasks) (void) hipFree (parked [i].d_tasks) ;
            if (parked [i].d_refs) (void) hipFree (parked [i].d_refs) ;
        }
        if (sCompute) (void) hipStreamDestroy (sCompute) ;
        if (sApply) (void) hipStreamDestroy (sApply) ;
        if (sCopy) (void) hipStreamDestroy (sCopy) ;
        for (int c = 0 ; c < 3 ; c++) { for (size_t i = 0 ; i < evPool [c].size ( ) ; i++) (void) hipEventDestroy (evPool [c][i]) ; evPool [c].clear ( ) ; }
        for (int i = 0 ; i < QR_CK_SLOTS ; i++) for (int k = 0 ; k < 2 ; k++) { if (evCheck [i][k]) (void) hipEventDestroy (evCheck [i][k]) ; evCheck [i][k] = NULL ; }
        if (evStart) (void) hipEventDestroy (evStart) ;
        if (evStop) (void) hipEventDestroy (evStop) ;
        if (d_tasks) (void) hipFree (d_tasks) ;
        if (d_refs) (void) hipFree (d_refs) ;
        if (d_vt) (void) hipFree (d_vt) ;
        if (d_counters) (void) hipFree (d_counters) ;
        if (d_pack) (void) hipFree (d_pack) ;
        if (h_stage) (void) hipHostFree (h_stage) ;
        *this = DeviceCache ( ) ;
    }
} ;

#define GPUQR_MAX_DEVICES 64
static DeviceCache g_cache [GPUQR_MAX_DEVICES] ;

static DeviceCache *device_cache (void)
{
    int dev = -1 ;
    if (hipGetDevice (&dev) != hipSuccess || dev < 0 || dev >= GPUQR_MAX_DEVICES) return NULL ;
    DeviceCache *dc = &g_cache [dev] ;
    if (!dc->ready)
    {
        // No stream priorities.  A high-priority panel stream bought under 1 % (its few long workgroups
        // take the compute-unit slots that free up first; measured again with contribution blocks
        // travelling row block by row block: 88.4 against 87.9 ms on the 1024 x 1024 grid, 594 against
        // 591 ms on the 64^3 grid), and with several processes on one device it
        // stalled all of them for good: every launch queued, no stream ever finishing, reproducibly
        // with three processes holding contexts and gone with equal priorities.  A high-priority queue
        // parked on a barrier packet that waits for a kernel of the same process's low-priority queue
        // appears to keep the other processes' low-priority queues from being scheduled.
        if (hipStreamCreateWithFlags (&dc->sCompute, hipStreamNonBlocking) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (hipStreamCreateWithFlags (&dc->sApply, hipStreamNonBlocking) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (hipStreamCreateWithFlags (&dc->sCopy, hipStreamNonBlocking) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (hipEventCreate (&dc->evStart) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (hipEventCreate (&dc->evStop) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (qr_kernels_init ( ) != hipSuccess) { dc->release ( ) ; return NULL ; }
        if (hipDeviceGetAttribute (&dc->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || dc->num_cus <= 0) dc->num_cus = 256 ;
        dc->ready = true ;
    }
    return dc ;
}

extern "C" void gpuqr_finalize (void)
{
    int cur = -1 ;
    (void) hipGetDevice (&cur) ;
    for (int d = 0 ; d < GPUQR_MAX_DEVICES ; d++)
    {
        if (!g_cache [d].ready) continue ;
        if (hipSetDevice (d) == hipSuccess) g_cache [d].release ( ) ;
    }
    if (cur >= 0) (void) hipSetDevice (cur) ;
}

// Row pitch of a device-resident front: fn, or ldn when the caller padded the rows.
static inline int64_t front_pitch (const gpuqr_front &fr)
{
    return (!fr.is_col_major && fr.ldn > fr.fn) ? fr.ldn : fr.fn ;
}

// The first nrows rows of a device front to packed host rows of fn entries.
static hipError_t pull_rows (const gpuqr_front &fr, int64_t nrows, hipStream_t stream)
{
    const int64_t ld = front_pitch (fr) ;
    if (nrows <= 0 || fr.fn <= 0) return hipSuccess ;
    if (ld == fr.fn)
        return hipMemcpyAsync (fr.cpuR, fr.gpuF, (size_t) (nrows * fr.fn) * sizeof (double), hipMemcpyDeviceToHost, stream) ;
    return hipMemcpy2DAsync (fr.cpuR, (size_t) fr.fn * sizeof (double), fr.gpuF, (size_t) ld * sizeof (double),
        (size_t) fr.fn * sizeof (double), (size_t) nrows, hipMemcpyDeviceToHost, stream) ;
}

#include <chrono>
static double eng_now_ms (void)
{
    return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now ( ).time_since_epoch ( )).count ( ) ;
}
// GPUQR_VERBOSE=2 prints where the host time of an engine call goes
#define ENG_TICK(label) do { if (eng_verbose) { double t_ = eng_now_ms ( ) ; fprintf (stderr, "    [engine] %-24s %9.2f ms\n", label, t_ - eng_tlast) ; eng_tlast = t_ ; } } while (0)

// R in the caller's packed layout (gpuqr_front.gpuRpack / cpuRpack).  Fronts are grouped by the
// rounds after which they are final; a group is packed by one launch on the copy stream and sent by
// one copy per run of blocks that are neighbours in both address spaces, while the compute streams
// go on with the rest of the tree.
struct RPackOut
{
    struct Block { int32_t group ; int64_t f ; const char *dev ; char *host ; size_t bytes ; int64_t task ; } ;
    hvec<Block> blocks ;            // sorted by (group, device address)
    hvec<long long> prefix ;        // per group: running count of 32-row tiles, restarting at 0
    hvec<int64_t> gprefix ;         // per block: where its group's counts begin in prefix
    const QrPackRTask *d_tasks = NULL ;
    const long long *d_prefix = NULL ;
    size_t pos = 0 ;

    // group_of (f): group of front f; groups are issued in increasing order.  seg_ptr / seg_round (may be
    // empty): fronts whose R leaves in pieces of QR_RSEG_COLS columns, with the round from which each piece
    // is final; group_of_round maps such a round to its group.
    template <typename G, typename GR> hipError_t build (DeviceCache *dc, const gpuqr_front *fronts, int64_t nf, G group_of,
        GR group_of_round, const hvec<int64_t> &seg_ptr, const hvec<int32_t> &seg_round, hipStream_t stream)
    {
        rseg_setup ( ) ;
        hvec<QrPackRTask> tasks_unsorted ;
        auto packed_at = [] (int64_t c, int64_t rr) { return (c <= rr) ? c * (c + 1) / 2 : rr * (rr + 1) / 2 + (c - rr) * rr ; } ;
        for (int64_t f = 0 ; f < nf ; f++)
        {
            const gpuqr_front &fr = fronts [f] ;
            if (fr.gpuRpack == NULL || (fr.is_sparse && fr.push_only) || fr.rank <= 0 || fr.fn <= 0 || fr.fm <= 0) continue ;
            const int64_t rr = std::min (fr.rank, fr.fm) ;
            // (a front that goes out in pieces has one round per piece, the last piece included)
            const int64_t nent = ((int64_t) seg_ptr.size ( ) >= 2 * nf) ? seg_ptr [(size_t) 2 * f + 1] : 0 ;
            const int64_t nseg = (nent > 0) ? nent - 1 : 0 ;
            for (int64_t k = 0 ; k <= nseg ; k++)
            {
                // piece k: columns [k * QR_RSEG_COLS, (k + 1) * QR_RSEG_COLS), the last piece up to fn
                const int64_t c0 = k * QR_RSEG_COLS, c1 = (k < nseg) ? (k + 1) * QR_RSEG_COLS : fr.fn ;
                Block c ;
                c.group = (k < nent) ? group_of_round (seg_round [(size_t) (seg_ptr [(size_t) 2 * f] + k)]) : group_of (f) ;
                c.group = std::min (c.group, group_of (f)) ;
                c.f = f ;
                const size_t off = (size_t) packed_at (c0, rr) * sizeof (double) ;
                c.dev = (const char *) fr.gpuRpack + off ;
                c.host = fr.cpuRpack ? (char *) fr.cpuRpack + off : NULL ;
                c.bytes = (size_t) (packed_at (c1, rr) - packed_at (c0, rr)) * sizeof (double) ;
                c.task = (int64_t) tasks_unsorted.size ( ) ;
                blocks.push_back (c) ;
                QrPackRTask t ;
                t.F = fr.gpuF ; t.out = fr.gpuRpack ; t.ld = (int32_t) front_pitch (fr) ; t.fn = (int32_t) fr.fn ;
                t.frank = (int32_t) rr ; t.c0 = (int32_t) c0 ; t.c1 = (int32_t) c1 ; t.pad = 0 ;
                tasks_unsorted.push_back (t) ;
            }
        }
        if (blocks.empty ( )) return hipSuccess ;
        std::sort (blocks.begin ( ), blocks.end ( ), [] (const Block &a, const Block &b)
            { return (a.group != b.group) ? (a.group < b.group) : (a.dev < b.dev) ; }) ;
        hvec<QrPackRTask> tasks (blocks.size ( )) ;
        gprefix.assign (blocks.size ( ), 0) ;
        for (size_t i = 0 ; i < blocks.size ( ) ; i++)
        {
            tasks [i] = tasks_unsorted [(size_t) blocks [i].task] ;
            // row tiles that hold entries of columns < c1
            const int64_t rows = std::min<int64_t> (tasks [i].frank, tasks [i].c1) ;
            if (i == 0 || blocks [i].group != blocks [i-1].group) prefix.push_back (0) ;
            gprefix [i] = (i > 0 && blocks [i].group == blocks [i-1].group) ? gprefix [i-1] : (int64_t) prefix.size ( ) - 1 ;
            prefix.push_back (prefix.back ( ) + (rows + 31) / 32) ;
        }
        const size_t tbytes = tasks.size ( ) * sizeof (QrPackRTask), pbytes = prefix.size ( ) * sizeof (long long) ;
        hipError_t e = dc->ensure (&dc->d_pack, &dc->cap_pack, tbytes + pbytes, false) ;
        if (e != hipSuccess) return e ;
        d_tasks = (const QrPackRTask *) dc->d_pack ;
        d_prefix = (const long long *) ((char *) dc->d_pack + tbytes) ;
        e = hipMemcpyAsync (dc->d_pack, tasks.data ( ), tbytes, hipMemcpyHostToDevice, stream) ;
        if (e != hipSuccess) return e ;
        e = hipMemcpyAsync ((char *) dc->d_pack + tbytes, prefix.data ( ), pbytes, hipMemcpyHostToDevice, stream) ;
        if (e != hipSuccess) return e ;
        return hipStreamSynchronize (stream) ;      // the host arrays go away
    }

    bool due (int32_t group) const { return pos < blocks.size ( ) && blocks [pos].group <= group ; }

    // pack and send the groups up to `group`; the stream has already been told what to wait for
    hipError_t issue (int32_t group, hipStream_t stream)
    {
        while (due (group))
        {
            size_t end = pos ;
            while (end < blocks.size ( ) && blocks [end].group == blocks [pos].group) end++ ;
            const int64_t g0 = gprefix [pos] ;       // the group's counts: prefix [g0 + (pos - first of group)]
            hipError_t e = qr_launch_pack_r (d_tasks + pos, d_prefix + g0, (int) (end - pos), prefix [g0 + (int64_t) (end - pos)], stream) ;
            if (e != hipSuccess) return e ;
            for (size_t i = pos ; i < end ; )
            {
                size_t j = i + 1 ;
                size_t bytes = blocks [i].bytes ;
                if (blocks [i].host == NULL) { i = j ; continue ; }
                while (j < end && blocks [j].host != NULL && blocks [j].dev == blocks [i].dev + bytes && blocks [j].host == blocks [i].host + bytes)
                {
                    bytes += blocks [j].bytes ; j++ ;
                }
                e = hipMemcpyAsync (blocks [i].host, blocks [i].dev, bytes, hipMemcpyDeviceToHost, stream) ;
                if (e != hipSuccess) return e ;
                i = j ;
            }
            pos = end ;
        }
        return hipSuccess ;
    }
} ;

// Queue the launches of the schedule in dc->lp (tasks and workgroup lists already on their way to
// the device on the critical stream) and wait for them.
static int launch_rounds (DeviceCache *dc, gpuqr_front *fronts, int64_t nf, gpuqr_stats *stats,
    const bool eng_verbose, double &eng_tlast)
{
    int code = GPUQR_SUCCESS ;
    const LaunchPlan &lp = dc->lp ;
    const hvec<double> &bflops = lp.bflops, &bbytes = lp.bbytes ;
    const hvec<int64_t> &rptr = lp.rptr ;
    const hvec<int32_t> &bdep = lp.bdep, &done_last = lp.done_last ;
    const int32_t nrounds = lp.nrounds ;
    const int64_t nbuckets = lp.nbuckets ;
    QrTask *d_tasks = (QrTask *) dc->d_tasks ;
    QrRef *d_refs = (QrRef *) dc->d_refs ;
    hipStream_t sCompute = dc->sCompute, sApply = dc->sApply, sCopy = dc->sCopy ;
    hipEvent_t evStart = dc->evStart, evStop = dc->evStop ;
    hvec<hipEvent_t> evProf ;
    hvec<int64_t> evProfBucket ;      // per profiled launch: first bucket, workgroups, kernel id
    hvec<int64_t> evProfN ;
    hvec<int> evProfKid ;
    int64_t launches = 0, synced_at = 0, ck_at = 0, ck_n = 0 ;
    float ms = 0 ;
    try
    {
        {
            // ---- fronts grouped by the round after which their data is final -------------
            auto final_round = [&] (int64_t f)
            {
                const int32_t *d = &done_last [f*QR_NCLASS] ;
                return std::max (0, std::max (d [0], std::max (d [1], d [2]))) ;
            } ;
            hvec<int64_t> pull_order ;
            for (int64_t f = 0 ; f < nf ; f++)
                if (fronts [f].cpuR != NULL && !(fronts [f].is_sparse && fronts [f].push_only)) pull_order.push_back (f) ;
            std::sort (pull_order.begin ( ), pull_order.end ( ), [&] (int64_t a, int64_t b)
                {
                    int32_t ra = std::min (final_round (a), nrounds - 1), rb = std::min (final_round (b), nrounds - 1) ;
                    return (ra != rb) ? (ra < rb) : (a < b) ;
                }) ;

            // Two compute streams for three classes.  The launches of a class are queued in round
            // order; before a launch its stream waits for the event of the latest round of each
            // class on the other stream that the tasks of the launch depend on, and for nothing else.
            // With the copy stream and the null stream that the SPQR driver uses, a process holds four
            // streams: one per hardware queue HIP gives a process by default, so no two of them are
            // multiplexed onto one queue.
            // Class 2 joins the lighter of the two streams.  Rough launch times (us per 512 workgroups,
            // measured): factorize 40/40/50/62 by tier, updates 12/17/28/80 by tier.
            int small_on ;
            {
                static const double us [QR_NSLOTS] = {40, 40, 50, 62, 12, 17, 28, 80, 55, 12, 17, 28, 80} ;
                double est [QR_NCLASS] = {0, 0, 0} ;
                for (int64_t b = 0 ; b < nbuckets ; b++)
                {
                    const int64_t n = rptr [b+1] - rptr [b] ;
                    if (n > 0) est [slot_class ((int) (b % QR_NSLOTS))] += us [b % QR_NSLOTS] * (double) ((n + 511) / 512) ;
                }
                small_on = est [1] < est [0] ;
                const char *e = getenv ("GPUQR_SMALL_ON") ;
                if (e) small_on = atoi (e) ;
            }
            hipStream_t cstream [QR_NCLASS] = {sCompute, sApply, small_on ? sApply : sCompute} ;
            HIP_TRY (hipEventRecord (evStart, sCompute)) ;
            for (int c = 1 ; c < QR_NCLASS ; c++) HIP_TRY (hipStreamWaitEvent (cstream [c], evStart, 0)) ;    // the task upload
            size_t pi = 0 ;
            size_t nev [QR_NCLASS] = {0, 0, 0} ;
            // R blocks packed for the caller: fronts grouped by the rounds in which they become final
            static int rgroup = 0 ;
            if (rgroup == 0) { const char *e = getenv ("GPUQR_RCOPY_GROUP") ; rgroup = e ? std::max (1, atoi (e)) : 256 ; }
            RPackOut rout ;
            double rout_ms = 0 ; int rout_calls = 0 ;
            HIP_TRY (rout.build (dc, fronts, nf, [&] (int64_t f) { return std::min (final_round (f), nrounds - 1) / rgroup ; },
                [&] (int32_t round) { return std::min (std::max (round, 0), nrounds - 1) / rgroup ; },
                lp.seg_ptr, lp.seg_round, sCopy)) ;
            // evAt [c][r]: event recorded after the last launch of class c in the latest round <= r that had one
            hvec<hipEvent_t> evAt [QR_NCLASS] ;
            for (int c = 0 ; c < QR_NCLASS ; c++) evAt [c].assign ((size_t) nrounds, (hipEvent_t) NULL) ;
            // evRound [c][r]: the round in which evAt [c][r] was recorded (-1: none).  Events come from a ring
            // and are recorded again later, so "already waited for" is decided by rounds, not by handles.
            hvec<int32_t> evRound [QR_NCLASS] ;
            for (int c = 0 ; c < QR_NCLASS ; c++) evRound [c].assign ((size_t) nrounds, -1) ;
            int32_t wround [QR_NCLASS][QR_NCLASS] ;    // class x's stream has waited for class y's launches up to this round
            for (int x = 0 ; x < QR_NCLASS ; x++) for (int y = 0 ; y < QR_NCLASS ; y++) wround [x][y] = -1 ;
            static const int slot_order [QR_NSLOTS] = {0, 1, 2, 3, 9, 10, 11, 12, 7, 4, 5, 6, 8} ;
            // Row tiers that are neighbours in the slot order and share a class go out as one launch
            // (slots 0..3: all panel tiers; 4..6: the small update tiers), unless the per-kernel profile
            // wants them apart.
            const bool merge_tiers = getenv ("GPUQR_SPLIT_TIERS") == NULL ;
            static int ride_below = -1 ;
            if (ride_below < 0) { const char *e = getenv ("GPUQR_RIDE_BELOW") ; ride_below = e ? atoi (e) : 512 ; }
            static int asm_join = -1 ;      // assembly launches below this many workgroups join the round's updates
            if (asm_join < 0) { const char *e = getenv ("GPUQR_ASM_JOIN_BELOW") ; asm_join = e ? atoi (e) : 1024 ; }
            for (int32_t r = 0 ; r < nrounds ; r++)
            {
                bool launched [QR_NCLASS] = {false, false, false} ;
                bool rode = false ;
                struct RoundLaunch { int kid, x ; int64_t b, n ; int32_t dep [QR_NCLASS] ; } rls [QR_NSLOTS] ;
                int nrl = 0 ;
                for (int si = 0 ; si < QR_NSLOTS ; si++)
                {
                    const int slot = slot_order [si] ;
                    int64_t b = (int64_t) r * QR_NSLOTS + slot ;
                    int64_t n = rptr [b+1] - rptr [b] ;
                    int kid = slot_kernel (slot) ;
                    int32_t dmerged [QR_NCLASS] = {bdep [b*QR_NCLASS], bdep [b*QR_NCLASS + 1], bdep [b*QR_NCLASS + 2]} ;
                    if (merge_tiers && (slot == 0 || slot == 4 || slot == 9))
                    {
                        const int last = (slot == 0) ? 3 : ((slot == 4) ? 6 : 12) ;
                        const int64_t bl = (int64_t) r * QR_NSLOTS + last ;
                        n = rptr [bl+1] - rptr [b] ;
                        kid = (slot == 0) ? 9 : ((slot == 4) ? 10 : 11) ;      // 11: every update tier (the look-ahead updates)
                        // a single tier keeps its own kernel (fewer registers, more workgroups per CU)
                        int present = 0, only = slot ;
                        for (int64_t bb = b ; bb <= bl ; bb++) if (rptr [bb+1] > rptr [bb]) { present++ ; only = (int) (bb % QR_NSLOTS) ; }
                        if (present == 1) kid = slot_kernel (only) ;
                        for (int64_t bb = b + 1 ; bb <= bl ; bb++)
                            for (int c = 0 ; c < QR_NCLASS ; c++) dmerged [c] = std::max (dmerged [c], bdep [bb*QR_NCLASS + c]) ;
                    }
                    else if (merge_tiers && ((slot >= 1 && slot <= 3) || slot == 5 || slot == 6 || slot >= 10)) continue ;
                    // a 256-row update launch too small to fill the chip takes the small tiers of its round
                    // along (slots 4..7 are neighbours in the workgroup list)
                    const int64_t b4 = (int64_t) r * QR_NSLOTS + 4, b7 = (int64_t) r * QR_NSLOTS + 7 ;
                    const bool ride = merge_tiers && ride_below > 0 && rptr [b7+1] > rptr [b7] && rptr [b7] > rptr [b4]
                        && rptr [b7+1] - rptr [b7] < ride_below ;
                    // A round's few assembly workgroups join its update launch when they follow it in the
                    // workgroup list (slot 8 comes right after slots 4..7) and on its stream anyway.
                    const int64_t b8 = (int64_t) r * QR_NSLOTS + 8 ;
                    const int64_t n8 = rptr [b8+1] - rptr [b8] ;
                    const bool asm_small = merge_tiers && asm_join > 0 && n8 > 0 && n8 < asm_join ;
                    const bool asm_rides = asm_small && ride && cstream [2] == cstream [1] ;
                    const bool asm_with_small = asm_small && !ride && rptr [b7+1] == rptr [b7] && rptr [b7] > rptr [b4] ;
                    if (ride && slot == 4) continue ;           // launched with slot 7
                    if ((asm_rides || asm_with_small) && slot == 8) continue ;      // launched with the updates
                    if (ride && slot == 7)
                    {
                        rode = true ;
                        b = b4 ;
                        const int64_t bend = asm_rides ? b8 : b7 ;
                        n = rptr [bend+1] - rptr [b4] ;
                        kid = asm_rides ? 12 : 11 ;
                        for (int64_t bb = b4 ; bb <= bend ; bb++)
                            for (int c = 0 ; c < QR_NCLASS ; c++) dmerged [c] = std::max (dmerged [c], bdep [bb*QR_NCLASS + c]) ;
                    }
                    if (asm_with_small && slot == 4)
                    {
                        n = rptr [b8+1] - rptr [b4] ;
                        kid = 12 ;
                        for (int c = 0 ; c < QR_NCLASS ; c++) dmerged [c] = std::max (dmerged [c], bdep [b8*QR_NCLASS + c]) ;
                    }
                    if (n <= 0) continue ;
                    RoundLaunch &rl = rls [nrl++] ;
                    rl.kid = kid ; rl.b = b ; rl.n = n ; rl.x = slot_class (slot) ;
                    for (int c = 0 ; c < QR_NCLASS ; c++) rl.dep [c] = dmerged [c] ;
                }
                // The event of a class in a round is the completion signal of its last launch (attached to
                // the launch itself): a separate event record is one more packet for the queue to work
                // through, measured at 2.4 us per launch on one stream and 4-6 us per hop between two.
                int last_of [QR_NCLASS] = {-1, -1, -1} ;
                for (int i = 0 ; i < nrl ; i++) last_of [rls [i].x] = i ;
                hipEvent_t ev_of [QR_NCLASS] = {NULL, NULL, NULL} ;
                for (int i = 0 ; i < nrl ; i++)
                {
                    const int kid = rls [i].kid ;
                    const int64_t b = rls [i].b, n = rls [i].n ;
                    const int x = rls [i].x ;
                    const int32_t *dmerged = rls [i].dep ;
                    hipStream_t st = cstream [x] ;
                    for (int y = 0 ; y < QR_NCLASS ; y++)
                    {
                        const int32_t d = dmerged [y] ;
                        if (y == x || d < 0 || cstream [y] == st) continue ;
                        hipEvent_t ev = evAt [y][d] ;
                        if (ev == NULL || evRound [y][d] <= wround [x][y]) continue ;
                        HIP_TRY (hipStreamWaitEvent (st, ev, 0)) ;
                        wround [x][y] = evRound [y][d] ;
                    }
                    if (g_profile_on)
                    {
                        hipEvent_t e0, e1 ;
                        HIP_TRY (hipEventCreate (&e0)) ; evProf.push_back (e0) ;
                        HIP_TRY (hipEventCreate (&e1)) ; evProf.push_back (e1) ;
                        evProfBucket.push_back (b) ; evProfN.push_back (n) ; evProfKid.push_back (kid) ;
                        HIP_TRY (qr_launch_tasks_timed (kid, d_tasks, d_refs + rptr [b], (int) n, st, e0, e1)) ;
                    }
                    else if (i == last_of [x])
                    {
                        hipEvent_t ev = dc->pooled_event (x, nev [x]++) ;
                        if (!ev) { code = GPUQR_GPUERROR ; goto cleanup ; }
                        HIP_TRY (qr_launch_tasks_ev (kid, d_tasks, d_refs + rptr [b], (int) n, st, ev)) ;
                        ev_of [x] = ev ;
                    }
                    else
                    {
                        HIP_TRY (qr_launch_tasks (kid, d_tasks, d_refs + rptr [b], (int) n, st)) ;
                    }
                    launched [x] = true ;
                    launches++ ;
                    // GPUQR_SERIAL=1 (gpuqr_set_serial): the debugging mode of the reference's
                    // -DFORCE_SYNCHRONIZE (GPUQREngine/Source/Scheduler/Scheduler_LaunchKernel.cpp:55-58):
                    // every launch is waited for and checked before the next one goes out, and a failure
                    // names the launch it belongs to.
                    if (serial_mode ( ))
                    {
                        hipError_t e1_ = hipStreamSynchronize (st) ;
                        hipError_t e2_ = hipGetLastError ( ) ;
                        if (e1_ != hipSuccess || e2_ != hipSuccess)
                        {
                            fprintf (stderr, "gpuqr: launch %lld failed (round %d, kernel slot %d, %lld workgroups): %s\n",
                                (long long) launches, (int) r, kid, (long long) n,
                                hipGetErrorString (e1_ != hipSuccess ? e1_ : e2_)) ;
                            code = GPUQR_GPUERROR ; goto cleanup ;
                        }
                    }
                }
                for (int c = 0 ; c < QR_NCLASS ; c++)
                {
                    if (launched [c])
                    {
                        hipEvent_t ev = ev_of [c] ;
                        if (ev == NULL)
                        {
                            // (profiled call: its launches carry timing events, so this one is recorded apart)
                            ev = dc->pooled_event (c, nev [c]++) ;
                            if (!ev) { code = GPUQR_GPUERROR ; goto cleanup ; }
                            HIP_TRY (hipEventRecord (ev, cstream [c])) ;
                        }
                        evAt [c][r] = ev ; evRound [c][r] = r ;
                        // a class that shares the stream is covered by the same event
                        for (int c2 = 0 ; c2 < c ; c2++) if (cstream [c2] == cstream [c]) { evAt [c2][r] = ev ; evRound [c2][r] = r ; }
                    }
                    else
                    {
                        evAt [c][r] = (r > 0) ? evAt [c][r-1] : (hipEvent_t) NULL ;
                        evRound [c][r] = (r > 0) ? evRound [c][r-1] : -1 ;
                    }
                }
                if (rode && cstream [2] != cstream [1] && launched [2])
                {
                    // Class-2 update tasks of this round ran on the bulk stream, and the round's assembly
                    // launch ran on the class-2 stream.  The class-2 stream is told to wait for the bulk
                    // launch and a fresh event is recorded behind that wait, so whoever waits for class 2
                    // of this round waits for both.
                    HIP_TRY (hipStreamWaitEvent (cstream [2], evAt [1][r], 0)) ;
                    hipEvent_t ev = dc->pooled_event (2, nev [2]++) ;
                    if (!ev) { code = GPUQR_GPUERROR ; goto cleanup ; }
                    HIP_TRY (hipEventRecord (ev, cstream [2])) ;
                    evAt [2][r] = ev ; evRound [2][r] = r ;
                    for (int c2 = 0 ; c2 < 2 ; c2++) if (cstream [c2] == cstream [2]) { evAt [c2][r] = ev ; evRound [c2][r] = r ; }
                }
                else if (rode)
                {
                    // only the bulk stream ran class-2 work this round; the class-2 stream stays in order behind it
                    if (cstream [2] != cstream [1]) HIP_TRY (hipStreamWaitEvent (cstream [2], evAt [1][r], 0)) ;
                    evAt [2][r] = evAt [1][r] ; evRound [2][r] = evRound [1][r] ;
                }
                if (pi < pull_order.size ( ) && std::min (final_round (pull_order [pi]), nrounds - 1) <= r)
                {
                    for (int c = 0 ; c < QR_NCLASS ; c++) if (evAt [c][r]) HIP_TRY (hipStreamWaitEvent (sCopy, evAt [c][r], 0)) ;
                    while (pi < pull_order.size ( ) && std::min (final_round (pull_order [pi]), nrounds - 1) <= r)
                    {
                        gpuqr_front &fr = fronts [pull_order [pi++]] ;
                        int64_t nrows = fr.rank ;
                        if (fr.is_sparse && fr.is_staged) nrows += fr.cm ;
                        if (nrows > fr.fm) nrows = fr.fm ;
                        HIP_TRY (pull_rows (fr, nrows, sCopy)) ;
                    }
                }
                // No more than about 2,500 launches are left queued at any time.  With some 4,500 or more
                // launches of one process waiting in its queues, every process that initialised the device
                // meanwhile died in its first copy (seen with 5,963 queued launches, not with 4,444; a wait
                // every 3,000 cured it), and a profiler attached to the process crashed.  Every 512 launches
                // both compute streams record a checkpoint; before a checkpoint slot is used again, five
                // checkpoints later, the host waits for it.  The device still has 2,000 launches queued then.
                if (launches - ck_at >= QR_CK_EVERY)
                {
                    hipEvent_t *slot = dc->evCheck [ck_n % QR_CK_SLOTS] ;
                    if (!slot [0])
                    {
                        HIP_TRY (hipEventCreateWithFlags (&slot [0], hipEventDisableTiming)) ;
                        HIP_TRY (hipEventCreateWithFlags (&slot [1], hipEventDisableTiming)) ;
                    }
                    else if (ck_n >= QR_CK_SLOTS)
                    {
                        HIP_TRY (hipEventSynchronize (slot [0])) ;
                        HIP_TRY (hipEventSynchronize (slot [1])) ;
                    }
                    HIP_TRY (hipEventRecord (slot [0], sCompute)) ;
                    HIP_TRY (hipEventRecord (slot [1], sApply)) ;
                    ck_n++ ;
                    ck_at = launches ;
                }
                // GPUQR_SYNC_EVERY=n: wait for the device every n launches (off; for experiments).
                static int sync_every = -1 ;
                if (sync_every < 0) { const char *e = getenv ("GPUQR_SYNC_EVERY") ; sync_every = e ? atoi (e) : 0 ; }
                if (sync_every > 0 && launches - synced_at >= sync_every)
                {
                    HIP_TRY (hipStreamSynchronize (sCompute)) ;
                    HIP_TRY (hipStreamSynchronize (sApply)) ;
                    synced_at = launches ;
                }
                // R blocks of the fronts that became final in this group of rounds
                if (((r + 1) % rgroup == 0 || r == nrounds - 1) && rout.due (r / rgroup))
                {
                    const double t0 = eng_verbose ? eng_now_ms ( ) : 0 ;
                    for (int c = 0 ; c < QR_NCLASS ; c++) if (evAt [c][r]) HIP_TRY (hipStreamWaitEvent (sCopy, evAt [c][r], 0)) ;
                    HIP_TRY (rout.issue (r / rgroup, sCopy)) ;
                    if (eng_verbose) { rout_ms += eng_now_ms ( ) - t0 ; rout_calls++ ; }
                }
            }
            for (int c = 1 ; c < QR_NCLASS ; c++)
                if (nrounds > 0 && evAt [c][nrounds-1]) HIP_TRY (hipStreamWaitEvent (sCompute, evAt [c][nrounds-1], 0)) ;
            HIP_TRY (hipEventRecord (evStop, sCompute)) ;
            ENG_TICK ("launch loop (host)") ;
            if (eng_verbose && rout_calls > 0) fprintf (stderr, "    [engine] of which R hand-off: %d groups, %.2f ms, %d rounds\n", rout_calls, rout_ms, (int) nrounds) ;
            HIP_TRY (hipStreamSynchronize (sCompute)) ;
            HIP_TRY (hipStreamSynchronize (sApply)) ;
            HIP_TRY (hipStreamSynchronize (sCopy)) ;
            HIP_TRY (hipEventElapsedTime (&ms, evStart, evStop)) ;
            ENG_TICK ("wait for the device") ;
            // GPUQR_PROFILE_DUMP=path: one line per launch (round, kernel id, workgroups, microseconds, flops)
            FILE *dump = (g_profile_on && getenv ("GPUQR_PROFILE_DUMP")) ? fopen (getenv ("GPUQR_PROFILE_DUMP"), "w") : NULL ;
            for (size_t i = 0 ; i < evProfBucket.size ( ) ; i++)
            {
                float t = 0 ;
                HIP_TRY (hipEventElapsedTime (&t, evProf [2*i], evProf [2*i+1])) ;
                const int64_t b = evProfBucket [i] ;
                KernelProfile &kp = g_profile [evProfKid [i]] ;
                // a merged launch covers the neighbouring buckets up to its workgroup count
                double fl = 0, by = 0 ;
                for (int64_t bb = b ; bb < nbuckets && rptr [bb] < rptr [b] + evProfN [i] ; bb++) { fl += bflops [bb] ; by += bbytes [bb] ; }
                kp.ms += t ; kp.launches++ ; kp.tasks += evProfN [i] ;      // workgroups
                if (dump) fprintf (dump, "%ld %d %ld %.3f %.0f\n", (long) (b / QR_NSLOTS), evProfKid [i],
                    (long) evProfN [i], t * 1e3, fl) ;
                kp.flops += fl ; kp.bytes += by ;
            }
            if (dump) fclose (dump) ;
        }
        if (stats)
        {
            stats->kernelTime = ms ;
            stats->numLaunches = launches ;
            stats->flopsActual = lp.flops ;
        }
    }
    catch (std::bad_alloc &)
    {
        code = GPUQR_OUTOFMEMORY ;
    }

cleanup:
    (void) hipStreamSynchronize (sCompute) ;
    (void) hipStreamSynchronize (sApply) ;
    (void) hipStreamSynchronize (sCopy) ;
    for (size_t i = 0 ; i < evProf.size ( ) ; i++) (void) hipEventDestroy (evProf [i]) ;
    if (code != GPUQR_SUCCESS) dc->plan_key = 0 ;
    return code ;
}

// Fingerprint of everything the schedule and the task descriptors are built from: the front
// descriptors as they stand (shapes, flags, every pointer), the contents of the staircases and of
// the host row maps (the planner orders the row-block hand-over of a contribution block by the
// values of cpuRimap, so two batches that differ only there need different schedules).
// Two calls with the same fingerprint differ at most in the values stored behind the pointers.
static uint64_t problem_fingerprint (const gpuqr_front *fronts, int64_t nf)
{
    uint64_t h = 0x243F6A8885A308D3ull ^ (uint64_t) nf ;
    auto mix = [&h] (const void *p, size_t bytes)
    {
        const unsigned char *b = (const unsigned char *) p ;
        size_t i = 0 ;
        for ( ; i + 8 <= bytes ; i += 8)
        {
            uint64_t w ; memcpy (&w, b + i, 8) ;
            h = (h ^ w) * 0x9E3779B97F4A7C15ull ; h ^= h >> 29 ;
        }
        uint64_t w = 0 ;
        if (i < bytes) { memcpy (&w, b + i, bytes - i) ; h = (h ^ w) * 0x9E3779B97F4A7C15ull ; h ^= h >> 29 ; }
    } ;
    for (int64_t f = 0 ; f < nf ; f++)
    {
        mix (&fronts [f], sizeof (gpuqr_front)) ;
        if (fronts [f].Stair != NULL && fronts [f].fn > 0) mix (fronts [f].Stair, (size_t) fronts [f].fn * sizeof (int64_t)) ;
        if (fronts [f].cpuRimap != NULL && fronts [f].cm > 0) mix (fronts [f].cpuRimap, (size_t) fronts [f].cm * sizeof (int32_t)) ;
    }
    return h | 1 ;
}

// Everything the launch loops need of one engine call: the tasks, and the workgroups sorted into
// (round, launch slot) buckets.
struct Schedule
{
    Plan plan ;
    hvec<int32_t> done_last ;       // per front and class: latest round used
    hvec<int64_t> seg_ptr ;         // see LaunchPlan
    hvec<int32_t> seg_round ;
    hvec<int64_t> rptr ;            // first workgroup of every (round, slot) bucket
    hvec<int32_t> bdep ;            // per bucket and class: latest round depended on
    hvec<double> bflops, bbytes ;   // per bucket: flops and algorithmic bytes of its tasks
    int32_t nrounds = 0 ;
    int64_t nbuckets = 0, ntasks = 0, nrefs = 0 ;
} ;

// body (c) for c in [0, n), on a few host threads when there are many chunks (c is handed out one at a time)
template <class Body> static void run_chunks (int64_t n, Body body)
{
    unsigned nt = 1 ;
    if (n >= 8 && !g_host_arena_on)         // (the entry point that allocates nothing starts no threads either)
    {
        const unsigned hw = std::thread::hardware_concurrency ( ) ;
        const char *env = getenv ("OMP_NUM_THREADS") ;
        const long want = env ? atol (env) : 16 ;
        nt = (unsigned) std::max<long> (1, std::min<long> (std::min<long> (want, hw ? hw : 1), 8)) ;
    }
    std::atomic<int64_t> next (0) ;
    auto work = [&] ( ) { for (int64_t c = next.fetch_add (1) ; c < n ; c = next.fetch_add (1)) body (c) ; } ;
    std::vector<std::thread> pool ;
    try { for (unsigned t = 1 ; t < nt ; t++) pool.emplace_back (work) ; } catch (...) { }
    work ( ) ;
    for (size_t t = 0 ; t < pool.size ( ) ; t++) pool [t].join ( ) ;
}

// Validate the fronts, order them children first, emit their tasks and sort the workgroups by
// (round, launch slot).  Host work only.
// How many schedules this process has planned so far (a call that finds the schedule of an earlier call
// on the device plans nothing): lets a caller see whether its re-factorizations replay.
static std::atomic<long long> g_schedules_built (0) ;
extern "C" int64_t gpuqr_schedules_built (void) { return (int64_t) g_schedules_built.load ( ) ; }

static int build_schedule (gpuqr_front *fronts, int64_t nf, Schedule &sc, const bool eng_verbose, double &eng_tlast)
{
    g_schedules_built++ ;
    Plan &plan = sc.plan ;
    hvec<int32_t> &done_last = sc.done_last, &bdep = sc.bdep ;
    hvec<int64_t> &rptr = sc.rptr ;
    hvec<double> &bflops = sc.bflops, &bbytes = sc.bbytes ;
    int32_t &nrounds = sc.nrounds ;
    int64_t &nbuckets = sc.nbuckets, &ntasks = sc.ntasks, &nrefs = sc.nrefs ;
    {
        // ---- validate ---------------------------------------------------------------
        for (int64_t f = 0 ; f < nf ; f++)
        {
            gpuqr_front &fr = fronts [f] ;
            if (fr.fids != f) return GPUQR_GPUERROR ;
            if (fr.pids < -1 || fr.pids >= nf || fr.pids == f) return GPUQR_GPUERROR ;
            if (fr.fm < 0 || fr.fn < 0 || fr.fm > INT32_MAX || fr.fn > INT32_MAX) return GPUQR_GPUERROR ;
            if (fr.fm > 0 && fr.fn > 0 && fr.gpuF == NULL) return GPUQR_GPUERROR ;
            if (front_pitch (fr) > INT32_MAX) return GPUQR_GPUERROR ;
            if (fr.rhs_n < 0 || (fr.rhs_n > 0 && (fr.rhs_n > 256 || fr.rhs_col < fr.fn || fr.rhs_col + fr.rhs_n > front_pitch (fr))))
                return GPUQR_GPUERROR ;
            if (fr.is_sparse && fr.pids >= 0 && fr.cm > 0 && fr.cn > 0)
            {
                if (!fr.gpuRimap || !fr.gpuRjmap || !fr.gpuC) return GPUQR_GPUERROR ;
                if (fronts [fr.pids].gpuF == NULL) return GPUQR_GPUERROR ;
            }
            if (fr.is_sparse && fr.Scount > 0 && !fr.gpuS) return GPUQR_GPUERROR ;
        }

        // ---- children-first order from pids -----------------------------------------
        hvec<int64_t> nchild ((size_t) nf, 0), order ;
        for (int64_t f = 0 ; f < nf ; f++) if (fronts [f].pids >= 0) nchild [fronts [f].pids]++ ;
        order.reserve ((size_t) nf) ;
        {
            hvec<int64_t> pending (nchild) ;
            hvec<int64_t> stack ;
            for (int64_t f = 0 ; f < nf ; f++) if (pending [f] == 0) stack.push_back (f) ;
            while (!stack.empty ( ))
            {
                int64_t f = stack.back ( ) ; stack.pop_back ( ) ;
                order.push_back (f) ;
                int64_t p = fronts [f].pids ;
                if (p >= 0 && --pending [p] == 0) stack.push_back (p) ;
            }
            if ((int64_t) order.size ( ) != nf) return GPUQR_GPUERROR ;     // cycle in pids
        }

        ENG_TICK ("validate + order") ;
        // ---- build the schedule -----------------------------------------------------
        // per front: latest round of each class that assembled into it; latest round of each class it used
        hvec<int32_t> in_last ((size_t) nf * QR_NCLASS, -1) ;
        done_last.assign ((size_t) nf * QR_NCLASS, -1) ;
        // Contribution blocks travel to the parent in blocks of rows, each as soon as its rows are
        // final, when the caller supplied the host copy of the row map (cpuRimap): the engine then
        // knows the first parent row a block lands in, and a panel of the parent waits only for the
        // blocks that reach into its rows.  Without the map, or with right-hand-side blocks (whose
        // replays move rows between fronts front by front), the parent waits for the whole child.
        // The work queue (modes 1 and 2) counts arrivals per front and takes "n arrivals" for "the first
        // n tasks": a block of rows that arrives early would be counted in place of a task the front
        // still waits for, so the queue keeps whole-front hand-overs.
        bool pipelined = getenv ("GPUQR_NO_PIPELINE") == NULL && engine_mode ( ) == 0 ;
        for (int64_t f = 0 ; f < nf && pipelined ; f++) if (fronts [f].rhs_n > 0) pipelined = false ;
        static int pipe_rows = 0 ;
        if (pipe_rows == 0)
        {
            const char *e = getenv ("GPUQR_PIPE_ROWS") ;
            // rows per block: one panel's worth (measured on the 1024 x 1024 grid: 99.6 / 98.8 / 96.9 ms with
            // 128 / 64 / 32 rows; the 64^3 grid: 660 / 598 / 596 ms)
            pipe_rows = e ? atoi (e) : QR_PANEL_W ;
            if (pipe_rows < QR_PANEL_W || pipe_rows > QR_PACK_TILE || pipe_rows % QR_PANEL_W) pipe_rows = QR_PANEL_W ;
        }
        {
            // an upper bound on the tasks, so that the vector never regrows (regrowth copied 1.6 GB for a
            // 96^3 grid): per panel one factorize and one update task per 256-row chunk and merge
            double est = 0 ;
            for (int64_t f = 0 ; f < nf ; f++)
            {
                const gpuqr_front &fr = fronts [f] ;
                if (fr.fm <= 0 || fr.fn <= 0) continue ;
                const double npan = (double) ((std::min (fr.fm, fr.fn) + QR_PANEL_W - 1) / QR_PANEL_W) ;
                const double nch = (double) ((fr.fm + QR_MAX_K - 1) / QR_MAX_K) ;
                est += npan * (2.3 * nch + 6) * (fr.rhs_n > 0 ? 1.5 : 1.0)
                    + (double) (fr.Scount / QR_SASM_CHUNK + 1)
                    + (double) (fr.cm / QR_PANEL_W + 1) * (double) (fr.cn / QR_PACK_TILE + 1) ;
            }
            if (est < 4e9 / sizeof (SchedTask)) plan.tasks.reserve ((size_t) est) ;
        }
        struct EvNode { AsmEvent ev ; int64_t next ; } ;
        hvec<EvNode> evpool ;
        hvec<int64_t> ev_head ((size_t) nf, -1) ;
        hvec<AsmEvent> evs ;
        hvec<int32_t> rows_done, fact_done ;
        rseg_setup ( ) ;
        sc.seg_ptr.assign ((size_t) 2 * nf, 0) ;      // per front: first entry of seg_round, number of entries
        sc.seg_round.clear ( ) ;
        for (size_t k = 0 ; k < order.size ( ) ; k++)
        {
            const int64_t f = order [k] ;
            gpuqr_front &fr = fronts [f] ;
            const size_t first_task = plan.tasks.size ( ) ;
            int32_t last [QR_NCLASS] ;
            for (int c = 0 ; c < QR_NCLASS ; c++) last [c] = in_last [f*QR_NCLASS + c] ;
            evs.clear ( ) ;
            for (int64_t q = ev_head [f] ; q >= 0 ; q = evpool [q].next) evs.push_back (evpool [q].ev) ;
            std::sort (evs.begin ( ), evs.end ( ), [] (const AsmEvent &a, const AsmEvent &b) { return a.min_row < b.min_row ; }) ;
            rows_done.clear ( ) ;
            if (!(fr.is_sparse && fr.push_only))
            {
                // S assembly, round 0
                if (fr.is_sparse)
                {
                    for (int64_t ps = 0 ; ps < fr.Scount ; ps += QR_SASM_CHUNK)
                    {
                        SchedTask st ; memset (&st, 0, sizeof (st)) ;
                        st.t.type = QR_TASK_SASSEMBLE ;
                        st.t.u.sasm.F = fr.gpuF ; st.t.u.sasm.S = fr.gpuS ;
                        st.t.u.sasm.pstart = ps ;
                        st.t.u.sasm.pend = std::min (ps + (int64_t) QR_SASM_CHUNK, fr.Scount) ;
                        st.t.u.sasm.fn = (int32_t) fr.fn ; st.t.u.sasm.ld = (int32_t) front_pitch (fr) ;
                        st.round = 0 ; st.slot = 8 ; st.vt = -1 ; st.nblk = 1 ;
                        st.dep [0] = st.dep [1] = st.dep [2] = -1 ;
                        st.flops = 0 ;
                        st.bytes = 24.0 * (double) (st.t.u.sasm.pend - st.t.u.sasm.pstart) ;   // 16 read + 8 written
                        plan.tasks.push_back (st) ;
                    }
                    if (fr.Scount > 0) last [2] = std::max (last [2], 0) ;
                }
                plan_front (plan, fr.gpuF, fr.fm, fr.fn, front_pitch (fr), fr.Stair, last, true, fr.rhs_col, fr.rhs_n,
                    evs.data ( ), evs.size ( ), &rows_done, &fact_done) ;
                // pieces of R: the first rr / QR_RSEG_COLS pieces hold pivot columns only
                const int64_t rr = std::min (fr.rank, fr.fm) ;
                sc.seg_ptr [(size_t) 2 * f] = (int64_t) sc.seg_round.size ( ) ;
                if (fr.gpuRpack != NULL && rr > 0 && !fact_done.empty ( ) && !rows_done.empty ( ))
                {
                    if (rr >= QR_RSEG_MIN)
                        for (int64_t c1 = QR_RSEG_COLS ; c1 <= rr && c1 < fr.fn ; c1 += QR_RSEG_COLS)
                            sc.seg_round.push_back (fact_done [(size_t) std::min<int64_t> (c1 / QR_PANEL_W - 1, (int64_t) fact_done.size ( ) - 1)]) ;
                    // the last piece (the rest of the pivot columns and every other column; the whole R of a
                    // smaller front): its rows are final once the panel of the last row of R has been applied
                    // to all columns, long before the front is through with its contribution block
                    const size_t pl = (size_t) std::min<int64_t> ((rr - 1) / QR_PANEL_W, (int64_t) fact_done.size ( ) - 1) ;
                    int32_t rlast = -1 ;
                    for (int c = 0 ; c < QR_NCLASS ; c++) rlast = std::max (rlast, rows_done [pl * QR_NCLASS + c]) ;
                    sc.seg_round.push_back (rlast) ;
                }
                sc.seg_ptr [(size_t) 2 * f + 1] = (int64_t) sc.seg_round.size ( ) - sc.seg_ptr [(size_t) 2 * f] ;
            }
            // blocks that no panel reached into (rows below every panel's) still belong to the front
            for (size_t q = 0 ; q < evs.size ( ) ; q++) last [2] = std::max (last [2], evs [q].round) ;
            for (int c = 0 ; c < QR_NCLASS ; c++) done_last [f*QR_NCLASS + c] = last [c] ;
            for (size_t i = first_task ; i < plan.tasks.size ( ) ; i++)
            {
                plan.tasks [i].wait_ctr = (int32_t) f ; plan.tasks [i].signal_ctr = (int32_t) f ;
            }
            // pack assembly into the parent
            if (fr.is_sparse && fr.pids >= 0 && fr.cm > 0 && fr.cn > 0)
            {
                const int32_t cld = fr.push_only ? fr.cn : (int32_t) front_pitch (fr) ;
                // first row of C inside the front (when C lies inside it, which the engine cannot take for granted)
                int64_t crow0 = -1 ;
                const bool by_rows = pipelined && fr.cpuRimap != NULL ;
                if (by_rows && !fr.push_only && !rows_done.empty ( ) && (uintptr_t) fr.gpuC >= (uintptr_t) fr.gpuF
                    && ((uintptr_t) fr.gpuC - (uintptr_t) fr.gpuF) % sizeof (double) == 0)
                {
                    const int64_t off = (int64_t) (((uintptr_t) fr.gpuC - (uintptr_t) fr.gpuF) / sizeof (double)) ;
                    const int64_t r = off / front_pitch (fr) ;
                    const int64_t mn = std::min (fr.fm, fr.fn) ;
                    if (r + fr.cm <= mn) crow0 = r ;
                }
                const int32_t rstep = by_rows ? pipe_rows : QR_PACK_TILE ;
                int32_t round_max = -1 ;
                for (int32_t ci = 0 ; ci < fr.cm ; ci += rstep)
                {
                    const int32_t ciend = std::min (ci + rstep, fr.cm) ;
                    int32_t dep [QR_NCLASS] ;
                    for (int c = 0 ; c < QR_NCLASS ; c++) dep [c] = last [c] ;
                    if (crow0 >= 0)
                    {
                        // the rows of this block are final once the panel of its last row is through
                        const int64_t pl = (crow0 + ciend - 1) / QR_PANEL_W ;
                        for (int c = 0 ; c < QR_NCLASS ; c++) dep [c] = rows_done [(size_t) pl * QR_NCLASS + c] ;
                    }
                    const int32_t round = 1 + std::max (dep [0], std::max (dep [1], dep [2])) ;
                    round_max = std::max (round_max, round) ;
                    for (int32_t cj = (ci / QR_PACK_TILE) * QR_PACK_TILE ; cj < fr.cn ; cj += QR_PACK_TILE)
                    {
                        SchedTask st ; memset (&st, 0, sizeof (st)) ;
                        st.t.type = QR_TASK_PACKASSEMBLE ;
                        QrPackAssembleTask &pk = st.t.u.pack ;
                        pk.C = fr.gpuC ; pk.P = fronts [fr.pids].gpuF ;
                        pk.Rjmap = fr.gpuRjmap ; pk.Rimap = fr.gpuRimap ;
                        pk.cld = cld ;
                        pk.pn = (int32_t) std::max<int64_t> (fr.pn, front_pitch (fronts [fr.pids])) ;
                        pk.cistart = ci ; pk.ciend = ciend ;
                        pk.cjstart = cj ; pk.cjend = std::min (cj + QR_PACK_TILE, fr.cn) ;
                        st.round = round ; st.slot = 8 ; st.vt = -1 ; st.nblk = 1 ;
                        for (int c = 0 ; c < QR_NCLASS ; c++) st.dep [c] = dep [c] ;
                        st.wait_ctr = (int32_t) f ; st.signal_ctr = (int32_t) fr.pids ;
                        {
                            // entries of the tile on or above the diagonal: 8 read + 8 written, plus the map ints
                            double cnt = 0 ;
                            for (int32_t i = pk.cistart ; i < pk.ciend ; i++)
                                cnt += (double) std::max (0, pk.cjend - std::max (pk.cjstart, i)) ;
                            st.flops = 0 ;
                            st.bytes = 16.0 * cnt + 4.0 * ((pk.ciend - pk.cistart) + (pk.cjend - pk.cjstart)) ;
                        }
                        plan.tasks.push_back (st) ;
                    }
                    if (by_rows)
                    {
                        EvNode nd ;
                        nd.ev.min_row = fr.cpuRimap [ci] ; nd.ev.round = round ;
                        nd.next = ev_head [fr.pids] ;
                        ev_head [fr.pids] = (int64_t) evpool.size ( ) ;
                        evpool.push_back (nd) ;
                    }
                }
                if (!by_rows)
                {
                    int32_t &pl = in_last [fr.pids*QR_NCLASS + 2] ;
                    pl = std::max (pl, round_max) ;
                }
                done_last [f*QR_NCLASS + 2] = std::max (done_last [f*QR_NCLASS + 2], round_max) ;
            }
        }

        ENG_TICK ("plan") ;
        ntasks = (int64_t) plan.tasks.size ( ) ;
        nrounds = 0 ;
        for (int64_t i = 0 ; i < ntasks ; i++) nrounds = std::max (nrounds, plan.tasks [i].round + 1) ;

        nbuckets = (int64_t) nrounds * QR_NSLOTS ;
        // An update launch that cannot fill the chip runs as long as one of its workgroups does:
        // there the column blocks are made narrower (more, shorter workgroups).  The threshold is in
        // workgroups per launch; one workgroup slot = half a compute unit.
        {
            static int thr = -1 ;
            if (thr < 0) { const char *e = getenv ("GPUQR_NARROW_BELOW") ; thr = e ? atoi (e) : 128 ; }
            hvec<int64_t> total ((size_t) nbuckets, 0) ;
            for (int64_t i = 0 ; i < ntasks ; i++)
                if (plan.tasks [i].t.type == QR_TASK_APPLY) total [(int64_t) plan.tasks [i].round * QR_NSLOTS + plan.tasks [i].slot] += plan.tasks [i].nblk ;
            for (int64_t i = 0 ; i < ntasks ; i++)
            {
                SchedTask &st = plan.tasks [i] ;
                if (st.t.type != QR_TASK_APPLY || st.is_rhs) continue ;
                QrPanelTask &pt = st.t.u.panel ;
                int64_t tot = total [(int64_t) st.round * QR_NSLOTS + st.slot] ;
                const int32_t width = pt.c_end - pt.c_begin ;
                while (tot < thr && pt.cb_cols > 32 && width > pt.cb_cols / 2)
                {
                    pt.cb_cols /= 2 ;
                    tot *= 2 ;
                }
                // and a launch that fills the chip many times over takes wider blocks: V and T are
                // staged once per workgroup, so their share of the traffic halves
                static int wide = -1 ;
                if (wide < 0) { const char *e = getenv ("GPUQR_WIDE_ABOVE") ; wide = e ? atoi (e) : 1536 ; }
                if (wide > 0 && tot >= wide && pt.cb_cols == 128 && width >= 256) pt.cb_cols = 256 ;
                const int32_t nb = (width + pt.cb_cols - 1) / pt.cb_cols ;
                // the same number of blocks, evened out (a last block of 32 columns would leave half of
                // its workgroup's waves idle); 16-column granularity keeps the 16-byte accesses aligned
                pt.cb_cols = (((width + nb - 1) / nb) + 15) / 16 * 16 ;
                st.bytes += (double) (nb - st.nblk) * (8.0 * pt.K * pt.w + 8.0 * QR_VT_DOUBLES) ;
                st.nblk = nb ;
            }
        }
        // counting sort of the workgroups by (round, launch slot): one launch per non-empty bucket
        rptr.assign ((size_t) nbuckets + 1, 0) ;
        // latest round of each class that some task of the bucket depends on
        bdep.assign ((size_t) nbuckets * QR_NCLASS, -1) ;
        for (int64_t i = 0 ; i < ntasks ; i++)
        {
            const SchedTask &st = plan.tasks [i] ;
            const int64_t b = (int64_t) st.round * QR_NSLOTS + st.slot ;
            rptr [b + 1] += st.nblk ;
            for (int c = 0 ; c < QR_NCLASS ; c++) bdep [b*QR_NCLASS + c] = std::max (bdep [b*QR_NCLASS + c], st.dep [c]) ;
        }
        for (int64_t b = 0 ; b < nbuckets ; b++) rptr [b+1] += rptr [b] ;
        nrefs = rptr [nbuckets] ;
        {
            bflops.assign ((size_t) nbuckets, 0.0) ; bbytes.assign ((size_t) nbuckets, 0.0) ;
            for (int64_t i = 0 ; i < ntasks ; i++)
            {
                const SchedTask &st = plan.tasks [i] ;
                const int64_t b = (int64_t) st.round * QR_NSLOTS + st.slot ;
                bflops [b] += st.flops ; bbytes [b] += st.bytes ;
            }
        }

        if (getenv ("GPUQR_PLAN_STATS"))
        {
            // diagnostic: how full the 256-row apply workgroups are
            double wg = 0, wgK [4] = {0,0,0,0}, wgodd = 0, wgmis = 0, wgseg = 0, colsum = 0 ;
            for (int64_t i = 0 ; i < ntasks ; i++)
            {
                const SchedTask &st = plan.tasks [i] ;
                if (st.t.type != QR_TASK_APPLY || st.t.tier != 3) continue ;
                const QrPanelTask &pt = st.t.u.panel ;
                wg += st.nblk ; wgK [(pt.K - 129) / 32] += st.nblk ;
                if (pt.ld & 1) wgodd += st.nblk ; else if (((uintptr_t) pt.F >> 3) & 1) wgmis += st.nblk ;
                if (pt.nseg > 1) wgseg += st.nblk ;
                colsum += pt.c_end - pt.c_begin ;
            }
            fprintf (stderr, "plan: apply<16> workgroups %.0f; K 129-160 %.1f%% 161-192 %.1f%% 193-224 %.1f%% 225-256 %.1f%%; odd ld %.1f%%, "
                "even ld odd base %.1f%%; gathered %.1f%%; mean columns per workgroup %.1f; rounds %d\n", wg, 100*wgK[0]/wg, 100*wgK[1]/wg,
                100*wgK[2]/wg, 100*wgK[3]/wg, 100*wgodd/wg, 100*wgmis/wg, 100*wgseg/wg, colsum/wg, nrounds) ;
        }

        ENG_TICK ("bucket sort") ;
    }
    return GPUQR_SUCCESS ;
}

// Point the panel tasks at their VT tiles, check the bounds the kernels index LDS and registers
// by, and write the task records and the (task, block) list of the workgroups in launch order.
static int stage_tasks (Schedule &sc, double *d_vt, QrTask *h_tasks, QrRef *h_refs)
{
    // where each task's workgroups go in the launch-ordered list: tasks of a bucket in task order
    hvec<int64_t> pos (sc.rptr), at ((size_t) sc.ntasks) ;
    for (int64_t i = 0 ; i < sc.ntasks ; i++)
    {
        const SchedTask &st = sc.plan.tasks [i] ;
        int64_t &q = pos [(int64_t) st.round * QR_NSLOTS + st.slot] ;
        at [(size_t) i] = q ;
        q += st.nblk ;
    }
    std::atomic<int> bad (0) ;
    const int64_t chunk = 8192 ;
    run_chunks ((sc.ntasks + chunk - 1) / chunk, [&] (int64_t c)
    {
        for (int64_t i = c * chunk ; i < std::min (sc.ntasks, (c + 1) * chunk) ; i++)
        {
            SchedTask &st = sc.plan.tasks [i] ;
            if (st.t.type == QR_TASK_FACTORIZE || st.t.type == QR_TASK_APPLY)
            {
                QrPanelTask &pt = st.t.u.panel ;
                pt.VT = (st.vt >= 0) ? d_vt + (size_t) st.vt * QR_VT_DOUBLES : NULL ;
                bool ok = pt.K >= 1 && pt.K <= QR_MAX_K && pt.w >= 1 && pt.w <= QR_PANEL_W
                    && pt.nseg >= 1 && pt.nseg <= QR_MAX_SEG && pt.col0 >= 0 && pt.col0 + pt.w <= pt.ld
                    && (st.t.type == QR_TASK_FACTORIZE || (pt.VT != NULL && pt.c_begin >= 0
                        && pt.c_begin < pt.c_end && pt.c_end <= pt.ld && pt.cb_cols >= 16
                        && (int64_t) (st.nblk - 1) * pt.cb_cols < pt.c_end - pt.c_begin)) ;
                int acc = 0 ;
                for (int s = 0 ; s < pt.nseg && ok ; s++) { ok = pt.seg_row [s] >= 0 && pt.seg_n [s] >= 1 ; acc += pt.seg_n [s] ; }
                if (!ok || acc != pt.K) bad.store (1) ;
            }
            h_tasks [i] = st.t ;
#ifdef GPUQR_EXPERIMENT_DRY
            if (st.t.type == QR_TASK_APPLY && st.dry) h_tasks [i].u.panel.w |= 64 ;
#endif
            int64_t q = at [(size_t) i] ;
            for (int32_t b = 0 ; b < st.nblk ; b++) { h_refs [q].task = (int32_t) i ; h_refs [q].blk = b ; q++ ; }
        }
    }) ;
    return bad.load ( ) ? GPUQR_GPUERROR : GPUQR_SUCCESS ;
}

static int engine_run (gpuqr_front *fronts, int64_t nf, gpuqr_stats *stats, uint64_t plan_key)
{
    const char *eng_env = getenv ("GPUQR_VERBOSE") ;
    const bool eng_verbose = eng_env && atoi (eng_env) >= 2 ;
    double eng_tlast = eng_now_ms ( ) ;
    int code = GPUQR_SUCCESS ;
    QrTask *d_tasks = NULL ;
    QrRef *d_refs = NULL ;
    double *d_vt = NULL ;
    hipStream_t sCompute = NULL, sApply = NULL, sCopy = NULL ;
    hipEvent_t evStart = NULL, evStop = NULL ;
    DeviceCache *dc = NULL ;
    int64_t launches = 0 ;
    float ms = 0 ;

    if (stats) { stats->kernelTime = 0 ; stats->numLaunches = 0 ; stats->flopsActual = 0 ; }
    if (nf <= 0) return GPUQR_SUCCESS ;
    dc = device_cache ( ) ;
    if (!dc) return GPUQR_GPUERROR ;
    LaunchPlan &lp = dc->lp ;

    // Without a key from the caller the engine makes its own from the problem description, so that
    // a repeated call (same shapes, staircases and buffers, new values) skips planning, sorting and
    // the upload of the tasks just the same.
    static int auto_key = -1 ;
    if (auto_key < 0) auto_key = (getenv ("GPUQR_NO_PLAN_CACHE") == NULL) ? 1 : 0 ;
    if (plan_key == 0 && auto_key && engine_mode ( ) != 1)
    {
        plan_key = problem_fingerprint (fronts, nf) ;
        ENG_TICK ("fingerprint") ;
    }
    // A schedule belongs to the mode it was made in: the engine's own choice between launches and the
    // work queue (mode 2) must not be pre-empted by the launches that mode 0 left behind for the same
    // problem, so the mode is part of every key.
    if (plan_key != 0) plan_key = (plan_key ^ (0x9E3779B97F4A7C15ull * (uint64_t) (engine_mode ( ) + 1))) | 1 ;

    // The caller vouches, by passing the same nonzero key, that nothing about the problem but the
    // values in the fronts has changed since the call that left its schedule on the device.
    if (plan_key != 0 && engine_mode ( ) != 1 && dc->activate (plan_key, nf) && lp.ntasks > 0)
        return launch_rounds (dc, fronts, nf, stats, eng_verbose, eng_tlast) ;
    dc->park_active ( ) ;       // the active buffers are about to be overwritten: the schedule in them moves aside

    try
    {
        Schedule sc ;
        code = build_schedule (fronts, nf, sc, eng_verbose, eng_tlast) ;
        if (code != GPUQR_SUCCESS) return code ;
        Plan &plan = sc.plan ;
        hvec<int64_t> &rptr = sc.rptr ;
        const int64_t ntasks = sc.ntasks, nrefs = sc.nrefs, nbuckets = sc.nbuckets ;
        sCompute = dc->sCompute ; sApply = dc->sApply ; sCopy = dc->sCopy ; evStart = dc->evStart ; evStop = dc->evStop ;

        // How launch-bound the schedule is: time of the launches if each ran alone (a launch lasts at
        // least as long as one of its workgroups) over the time of the same work with the chip full.
        double launch_bound = 1 ;
        {
            static const double us [QR_NSLOTS] = {40, 40, 50, 62, 12, 17, 28, 80, 55, 12, 17, 28, 80} ;
            double t_launch = 0, t_work = 0 ;
            for (int64_t b = 0 ; b < nbuckets ; b++)
            {
                const int64_t n = rptr [b+1] - rptr [b] ;
                if (n <= 0) continue ;
                t_launch += us [b % QR_NSLOTS] * (double) ((n + 511) / 512) ;
                t_work += us [b % QR_NSLOTS] * (double) n / 512.0 ;
            }
            if (t_work > 0) launch_bound = t_launch / t_work ;
            if (getenv ("GPUQR_PLAN_STATS")) fprintf (stderr, "plan: launch-bound ratio %.2f (launch model %.1f ms, work %.1f ms)\n",
                launch_bound, t_launch / 1e3, t_work / 1e3) ;
        }
        // mode 2: the work queue when the launches would be very badly filled.  Measured (queue vs
        // launches, kernel time): 2D 256^2 grid, ratio 22: 16.2 vs 23.6 ms; one 1024 x 1024 front: 3.2 vs
        // 3.6 ms; 2D 1024^2 grid, ratio 4.8: 155 vs 152 ms; one 4096 x 2048 front, ratio 11: 12.1 vs
        // 11.5 ms; dense batch, ratio 1.6: 79 vs 51 ms.
        const bool use_queue = engine_mode ( ) == 1 || (engine_mode ( ) == 2 && launch_bound >= 12.0) ;
        if (ntasks > 0 && use_queue && !g_profile_on)
        {
            // ---- one persistent kernel: the same tasks in the same order, with arrival counters ----
            // Counter c counts the finished workgroups that report to front c: its own tasks and the
            // pack-assembly tasks of its children.  A task of round r waits until everything reporting
            // to its front from rounds < r has arrived, i.e. rounds stay barriers inside a front, while
            // different fronts run free of each other.
            const size_t task_bytes = (size_t) ntasks * sizeof (QrTask) ;
            const size_t ref_bytes = (size_t) nrefs * sizeof (QrQueueRef) ;
            const size_t ctr_bytes = ((size_t) (QR_QUEUE_FIRST + nf) * sizeof (int) + 15) & ~(size_t) 15 ;
            HIP_TRY (dc->ensure (&dc->d_vt, &dc->cap_vt, (size_t) plan.n_vt * QR_VT_DOUBLES * sizeof (double), false)) ;
            HIP_TRY (dc->ensure (&dc->d_tasks, &dc->cap_tasks, task_bytes, false)) ;
            HIP_TRY (dc->ensure (&dc->d_refs, &dc->cap_refs, ref_bytes, false)) ;
            HIP_TRY (dc->ensure (&dc->d_counters, &dc->cap_counters, ctr_bytes, false)) ;
            HIP_TRY (dc->ensure (&dc->h_stage, &dc->cap_stage, task_bytes + ref_bytes, true)) ;
            d_vt = (double *) dc->d_vt ; d_tasks = (QrTask *) dc->d_tasks ;
            QrQueueRef *d_qrefs = (QrQueueRef *) dc->d_refs ;
            QrTask *h_tasks = (QrTask *) dc->h_stage ;
            QrQueueRef *h_qrefs = (QrQueueRef *) ((char *) dc->h_stage + task_bytes) ;

            // arrivals per (counter, round), as prefix sums over the rounds of each counter
            hvec<int64_t> cptr ((size_t) nf + 1, 0) ;
            for (int64_t i = 0 ; i < ntasks ; i++) cptr [plan.tasks [i].signal_ctr + 1]++ ;
            for (int64_t c = 0 ; c < nf ; c++) cptr [c+1] += cptr [c] ;
            hvec<int64_t> cidx ((size_t) ntasks) ;
            {
                hvec<int64_t> w (cptr) ;
                for (int64_t i = 0 ; i < ntasks ; i++) cidx [w [plan.tasks [i].signal_ctr]++] = i ;
            }
            hvec<int32_t> cround ((size_t) ntasks), ccum ((size_t) ntasks) ;
            for (int64_t c = 0 ; c < nf ; c++)
            {
                std::sort (cidx.begin ( ) + cptr [c], cidx.begin ( ) + cptr [c+1], [&] (int64_t a, int64_t b)
                    { return plan.tasks [a].round < plan.tasks [b].round ; }) ;
                int64_t cum = 0 ;
                for (int64_t k = cptr [c] ; k < cptr [c+1] ; k++)
                {
                    cround [k] = plan.tasks [cidx [k]].round ;
                    ccum [k] = (int32_t) cum ;               // arrivals from entries before this one
                    cum += plan.tasks [cidx [k]].nblk ;
                }
            }
            {
                hvec<int64_t> pos (rptr) ;
                for (int64_t i = 0 ; i < ntasks ; i++)
                {
                    SchedTask &st = plan.tasks [i] ;
                    if (st.t.type == QR_TASK_FACTORIZE || st.t.type == QR_TASK_APPLY)
                    {
                        QrPanelTask &pt = st.t.u.panel ;
                        pt.VT = (st.vt >= 0) ? d_vt + (size_t) st.vt * QR_VT_DOUBLES : NULL ;
                        bool ok = pt.K >= 1 && pt.K <= QR_MAX_K && pt.w >= 1 && pt.w <= QR_PANEL_W
                            && pt.nseg >= 1 && pt.nseg <= QR_MAX_SEG && pt.col0 >= 0 && pt.col0 + pt.w <= pt.ld
                            && (st.t.type == QR_TASK_FACTORIZE || (pt.VT != NULL && pt.c_begin >= 0
                                && pt.c_begin < pt.c_end && pt.c_end <= pt.ld && pt.cb_cols >= 16
                                && (int64_t) (st.nblk - 1) * pt.cb_cols < pt.c_end - pt.c_begin)) ;
                        int acc = 0 ;
                        for (int s = 0 ; s < pt.nseg && ok ; s++) { ok = pt.seg_row [s] >= 0 && pt.seg_n [s] >= 1 ; acc += pt.seg_n [s] ; }
                        if (!ok || acc != pt.K) { code = GPUQR_GPUERROR ; goto cleanup ; }
                    }
                    h_tasks [i] = st.t ;
                    // arrivals at the waited-for counter from rounds before this task's
                    const int64_t c = st.wait_ctr ;
                    const int32_t *lo = cround.data ( ) + cptr [c], *hi = cround.data ( ) + cptr [c+1] ;
                    const int32_t *it = std::lower_bound (lo, hi, st.round) ;
                    int32_t need = 0 ;
                    if (it != hi) need = ccum [cptr [c] + (it - lo)] ;
                    else if (hi != lo)
                    {
                        const int64_t last = cptr [c+1] - 1 ;
                        need = ccum [last] + plan.tasks [cidx [last]].nblk ;
                    }
                    int64_t &q = pos [(int64_t) st.round * QR_NSLOTS + st.slot] ;
                    for (int32_t b = 0 ; b < st.nblk ; b++)
                    {
                        QrQueueRef &qr = h_qrefs [q++] ;
                        qr.task = (int32_t) i ; qr.blk = b ; qr.wait_ctr = st.wait_ctr ; qr.need = need ;
                        qr.signal_ctr = st.signal_ctr ;
                    }
                }
            }
            HIP_TRY (hipMemcpyAsync (d_tasks, h_tasks, task_bytes, hipMemcpyHostToDevice, sCompute)) ;
            HIP_TRY (hipMemcpyAsync (d_qrefs, h_qrefs, ref_bytes, hipMemcpyHostToDevice, sCompute)) ;
            HIP_TRY (hipMemsetAsync (dc->d_counters, 0, ctr_bytes, sCompute)) ;
            HIP_TRY (hipEventRecord (evStart, sCompute)) ;
            {
                int64_t nwg = std::min<int64_t> (nrefs, (int64_t) 2 * dc->num_cus) ;
                HIP_TRY (qr_launch_persistent (d_tasks, d_qrefs, (int) nrefs, (int *) dc->d_counters, (int) nwg, sCompute)) ;
                launches = 1 ;
            }
            HIP_TRY (hipEventRecord (evStop, sCompute)) ;
            int qstate [2] = {0, 0} ;
            HIP_TRY (hipMemcpyAsync (qstate, dc->d_counters, sizeof (qstate), hipMemcpyDeviceToHost, sCompute)) ;
            {
                // R blocks packed for the caller: one group, behind the kernel
                RPackOut rout ;
                static const hvec<int64_t> no_ptr ;
                static const hvec<int32_t> no_round ;
                HIP_TRY (rout.build (dc, fronts, nf, [] (int64_t) { return 0 ; }, [] (int32_t) { return 0 ; }, no_ptr, no_round, sCopy)) ;
                HIP_TRY (rout.issue (0, sCompute)) ;
            }
            for (int64_t f = 0 ; f < nf ; f++)
            {
                gpuqr_front &fr = fronts [f] ;
                if (fr.cpuR == NULL || (fr.is_sparse && fr.push_only)) continue ;
                int64_t nrows = fr.rank ;
                if (fr.is_sparse && fr.is_staged) nrows += fr.cm ;
                if (nrows > fr.fm) nrows = fr.fm ;
                HIP_TRY (pull_rows (fr, nrows, sCompute)) ;
            }
            HIP_TRY (hipStreamSynchronize (sCompute)) ;
            HIP_TRY (hipEventElapsedTime (&ms, evStart, evStop)) ;
            if (qstate [QR_QUEUE_ERROR] != 0 || qstate [QR_QUEUE_HEAD] < nrefs)
            {
                if (getenv ("GPUQR_VERBOSE")) fprintf (stderr, "gpuqr: work queue stopped early (error %d, head %d of %ld)\n",
                    qstate [QR_QUEUE_ERROR], qstate [QR_QUEUE_HEAD], (long) nrefs) ;
                code = GPUQR_GPUERROR ; goto cleanup ;
            }
        }
        else
        if (ntasks > 0)
        {
            const size_t task_bytes = (size_t) ntasks * sizeof (QrTask) ;
            const size_t ref_bytes = (size_t) nrefs * sizeof (QrRef) ;
            HIP_TRY (dc->ensure (&dc->d_vt, &dc->cap_vt, (size_t) plan.n_vt * QR_VT_DOUBLES * sizeof (double), false)) ;
            HIP_TRY (dc->ensure (&dc->d_tasks, &dc->cap_tasks, task_bytes, false)) ;
            HIP_TRY (dc->ensure (&dc->d_refs, &dc->cap_refs, ref_bytes, false)) ;
            HIP_TRY (dc->ensure (&dc->h_stage, &dc->cap_stage, task_bytes + ref_bytes, true)) ;
            d_vt = (double *) dc->d_vt ; d_tasks = (QrTask *) dc->d_tasks ; d_refs = (QrRef *) dc->d_refs ;
            QrTask *h_tasks = (QrTask *) dc->h_stage ;
            QrRef *h_refs = (QrRef *) ((char *) dc->h_stage + task_bytes) ;
            code = stage_tasks (sc, d_vt, h_tasks, h_refs) ;
            if (code != GPUQR_SUCCESS) goto cleanup ;
            ENG_TICK ("task + ref staging") ;
            HIP_TRY (hipMemcpyAsync (d_tasks, h_tasks, task_bytes, hipMemcpyHostToDevice, sCompute)) ;
            HIP_TRY (hipMemcpyAsync (d_refs, h_refs, ref_bytes, hipMemcpyHostToDevice, sCompute)) ;

            // the schedule is on the device: remember whose it is, then run it
            lp.rptr.swap (sc.rptr) ; lp.bdep.swap (sc.bdep) ; lp.done_last.swap (sc.done_last) ;
            lp.seg_ptr.swap (sc.seg_ptr) ; lp.seg_round.swap (sc.seg_round) ;
            lp.nrounds = sc.nrounds ; lp.nbuckets = nbuckets ; lp.ntasks = ntasks ; lp.nrefs = nrefs ; lp.flops = plan.flops ;
            dc->plan_key = plan_key ; dc->plan_nf = nf ; dc->plan_vt_base = dc->d_vt ;
            lp.bflops.swap (sc.bflops) ; lp.bbytes.swap (sc.bbytes) ;
            return launch_rounds (dc, fronts, nf, stats, eng_verbose, eng_tlast) ;
        }
        if (stats)
        {
            stats->kernelTime = ms ;
            stats->numLaunches = launches ;
            stats->flopsActual = plan.flops ;
        }
    }
    catch (std::bad_alloc &)
    {
        code = GPUQR_OUTOFMEMORY ;
    }

cleanup:
    if (sCompute) (void) hipStreamSynchronize (sCompute) ;
    if (sApply) (void) hipStreamSynchronize (sApply) ;
    if (sCopy) (void) hipStreamSynchronize (sCopy) ;
    // streams, events and device buffers stay in the per-device cache for the next call
    return code ;
}

extern "C" int gpuqr_factor_sparse_keyed
(
    size_t gpuMemorySize, gpuqr_front *fronts, int64_t numFronts,
    const int64_t *Parent, const int64_t *Childp, const int64_t *Child, gpuqr_stats *stats, uint64_t plan_key
)
{
    (void) gpuMemorySize ; (void) Parent ; (void) Childp ; (void) Child ;
    if (numFronts < 0 || (numFronts > 0 && fronts == NULL)) return GPUQR_GPUERROR ;
    return engine_run (fronts, numFronts, stats, plan_key) ;
}

extern "C" int gpuqr_factor_sparse
(
    size_t gpuMemorySize, gpuqr_front *fronts, int64_t numFronts,
    const int64_t *Parent, const int64_t *Childp, const int64_t *Child, gpuqr_stats *stats
)
{
    (void) gpuMemorySize ; (void) Parent ; (void) Childp ; (void) Child ;
    if (numFronts < 0 || (numFronts > 0 && fronts == NULL)) return GPUQR_GPUERROR ;
    return engine_run (fronts, numFronts, stats, 0) ;
}

// -----------------------------------------------------------------------------
// the entry point that allocates nothing
// -----------------------------------------------------------------------------

extern "C" hipError_t qr_scratch_address (void **ptr, size_t *bytes) ;

// GPUQR_BACKTRACE=1: a fatal signal in the host code prints the native call stack (return addresses
// relative to the libraries' load addresses, resolvable with llvm-symbolizer) before the default action.
// A debugging aid for machines without a debugger; installed when the library is loaded.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void fatal_signal_backtrace (int sig)
{
    void *frames [64] ;
    const int n = backtrace (frames, 64) ;
    const char msg [] = "[gpuqr] fatal signal, native backtrace:\n" ;
    (void) !write (2, msg, sizeof (msg) - 1) ;
    backtrace_symbols_fd (frames, n, 2) ;
    signal (sig, SIG_DFL) ;
    raise (sig) ;
}
__attribute__ ((constructor)) static void install_backtrace_handler (void)
{
    const char *e = getenv ("GPUQR_BACKTRACE") ;
    if (e && *e && *e != '0')
    {
        void *warm [2] ; (void) backtrace (warm, 2) ;       // loads the unwinder now, not inside the handler
        signal (SIGSEGV, fatal_signal_backtrace) ;
        signal (SIGBUS, fatal_signal_backtrace) ;
        signal (SIGABRT, fatal_signal_backtrace) ;
    }
}

// Per-device set-up of everything the engine keeps between calls: streams, events, the kernels'
// attributes, and the address of the device's copy of the fixed workspace.  Whatever must be created
// is created here, so that gpuqr_factor_sparse_noalloc itself creates nothing.
extern "C" int gpuqr_init (void)
{
    DeviceCache *dc = device_cache ( ) ;
    if (!dc) return GPUQR_GPUERROR ;
    if (dc->scratch == NULL)
    {
        void *p = NULL ; size_t bytes = 0 ;
        if (qr_scratch_address (&p, &bytes) != hipSuccess || p == NULL) return GPUQR_GPUERROR ;
        dc->scratch = (char *) p ;
        dc->scratch_bytes = bytes ;
    }
    return GPUQR_SUCCESS ;
}

extern "C" size_t gpuqr_noalloc_workspace_bytes (void) { return (size_t) QR_SCRATCH_BYTES ; }

static size_t g_noalloc_need = 0 ;
extern "C" size_t gpuqr_noalloc_last_need (void) { return g_noalloc_need ; }

extern "C" int gpuqr_factor_sparse_noalloc (gpuqr_front *fronts, int64_t nf, gpuqr_stats *stats)
{
    if (stats) { stats->kernelTime = 0 ; stats->numLaunches = 0 ; stats->flopsActual = 0 ; }
    if (nf < 0 || (nf > 0 && fronts == NULL)) return GPUQR_GPUERROR ;
    if (nf == 0) return GPUQR_SUCCESS ;
    // first call on this device: set-up (a caller that must not allocate even then calls gpuqr_init before)
    if (gpuqr_init ( ) != GPUQR_SUCCESS) return GPUQR_GPUERROR ;
    DeviceCache *dc = device_cache ( ) ;
    // the packed R hand-off needs a task list of its own on the device: not offered here
    for (int64_t f = 0 ; f < nf ; f++) if (fronts [f].gpuRpack != NULL || fronts [f].cpuRpack != NULL) return GPUQR_GPUERROR ;

    int code = GPUQR_SUCCESS ;
    hipStream_t stream = dc->sCompute ;
    bool eng_verbose = false ; double eng_tlast = 0 ;
    g_noalloc_need = 0 ;
    HostArenaScope arena ;      // from here on the planner's vectors come out of the static host block
    try
    {
        Schedule sc ;
        code = build_schedule (fronts, nf, sc, eng_verbose, eng_tlast) ;
        if (code != GPUQR_SUCCESS) return code ;
        if (sc.ntasks > 0)
        {
            const size_t task_bytes = ((size_t) sc.ntasks * sizeof (QrTask) + 255) & ~(size_t) 255 ;
            const size_t ref_bytes = ((size_t) sc.nrefs * sizeof (QrRef) + 255) & ~(size_t) 255 ;
            const size_t vt_bytes = (size_t) sc.plan.n_vt * QR_VT_DOUBLES * sizeof (double) ;
            g_noalloc_need = task_bytes + ref_bytes + vt_bytes ;
            if (g_noalloc_need > dc->scratch_bytes)
            {
                if (getenv ("GPUQR_VERBOSE")) fprintf (stderr, "gpuqr_factor_sparse_noalloc: the batch needs %zu bytes of "
                    "workspace, the fixed workspace holds %zu\n", g_noalloc_need, dc->scratch_bytes) ;
                return GPUQR_TOOLARGE ;
            }
            // dc->scratch is the address hipGetSymbolAddress gave for this device
            QrTask *d_tasks = (QrTask *) dc->scratch ;
            QrRef *d_refs = (QrRef *) (dc->scratch + task_bytes) ;
            double *d_vt = (double *) (dc->scratch + task_bytes + ref_bytes) ;
            hvec<QrTask> h_tasks ((size_t) sc.ntasks) ;
            hvec<QrRef> h_refs ((size_t) std::max<int64_t> (sc.nrefs, 1)) ;
            code = stage_tasks (sc, d_vt, h_tasks.data ( ), h_refs.data ( )) ;
            if (code != GPUQR_SUCCESS) return code ;
            // Nothing in the workspace has to start at zero: every word a kernel reads from it was
            // written earlier in this call (the records by these copies, a VT tile by the panel task
            // in front of the updates that read it).
            HIP_TRY (hipMemcpyAsync (d_tasks, h_tasks.data ( ), (size_t) sc.ntasks * sizeof (QrTask), hipMemcpyHostToDevice, stream)) ;
            HIP_TRY (hipMemcpyAsync (d_refs, h_refs.data ( ), (size_t) sc.nrefs * sizeof (QrRef), hipMemcpyHostToDevice, stream)) ;
            HIP_TRY (hipEventRecord (dc->evStart, stream)) ;
            // one stream, rounds in order: a task of round r reads only what rounds < r wrote
            int64_t launches = 0 ;
            for (int32_t r = 0 ; r < sc.nrounds ; r++)
            {
                for (int slot = 0 ; slot < QR_NSLOTS ; slot++)
                {
                    const int64_t b = (int64_t) r * QR_NSLOTS + slot ;
                    const int64_t n = sc.rptr [b+1] - sc.rptr [b] ;
                    if (n <= 0) continue ;
                    HIP_TRY (qr_launch_tasks (slot_kernel (slot), d_tasks, d_refs + sc.rptr [b], (int) n, stream)) ;
                    launches++ ;
                    if ((launches & 2047) == 0) HIP_TRY (hipStreamSynchronize (stream)) ;      // bound on the queued launches
                }
            }
            HIP_TRY (hipEventRecord (dc->evStop, stream)) ;
            for (int64_t f = 0 ; f < nf ; f++)
            {
                gpuqr_front &fr = fronts [f] ;
                if (fr.cpuR == NULL || (fr.is_sparse && fr.push_only)) continue ;
                int64_t nrows = fr.rank ;
                if (fr.is_sparse && fr.is_staged) nrows += fr.cm ;
                if (nrows > fr.fm) nrows = fr.fm ;
                HIP_TRY (pull_rows (fr, nrows, stream)) ;
            }
            HIP_TRY (hipStreamSynchronize (stream)) ;
            float ms = 0 ;
            HIP_TRY (hipEventElapsedTime (&ms, dc->evStart, dc->evStop)) ;
            if (stats) { stats->kernelTime = ms ; stats->numLaunches = launches ; }
        }
        if (stats) stats->flopsActual = sc.plan.flops ;
    }
    catch (std::bad_alloc &)
    {
        // the static host block ran out: the batch is too large for the fixed workspaces
        code = GPUQR_TOOLARGE ;
    }

cleanup:
    (void) hipStreamSynchronize (stream) ;
    return code ;
}

// -----------------------------------------------------------------------------
// dense overload: host fronts in, R out
// -----------------------------------------------------------------------------

// -----------------------------------------------------------------------------
// factorization that keeps H, and replays of it on new right-hand sides
// -----------------------------------------------------------------------------

// What a kept factorization owns: the VT tiles (the fronts themselves, with V in place, belong to the
// caller), the update tasks of the right-hand-side blocks in launch order, and when each front is done.
struct gpuqr_replay
{
    void *d_vt = NULL, *d_tasks = NULL, *d_refs = NULL ;
    hvec<int64_t> rptr ;            // first workgroup of every (round, slot) bucket of the replay
    hvec<int32_t> done_round ;      // per front: last round in which the factorization touched it
    int32_t nrounds = 0 ;
    int64_t nrefs = 0 ;
    int device = -1 ;
} ;

extern "C" void gpuqr_replay_free (gpuqr_replay *rp)
{
    if (!rp) return ;
    if (rp->d_vt) (void) hipFree (rp->d_vt) ;
    if (rp->d_tasks) (void) hipFree (rp->d_tasks) ;
    if (rp->d_refs) (void) hipFree (rp->d_refs) ;
    rp->~gpuqr_replay ( ) ;
    g_free (rp) ;
}

extern "C" int32_t gpuqr_replay_rounds (const gpuqr_replay *rp) { return rp ? rp->nrounds : 0 ; }
extern "C" int32_t gpuqr_replay_front_done (const gpuqr_replay *rp, int64_t f)
{
    return (rp && f >= 0 && f < (int64_t) rp->done_round.size ( )) ? rp->done_round [f] : -1 ;
}

// The updates of round r on the right-hand-side blocks, queued on `stream` (no waiting).
extern "C" int gpuqr_replay_round (const gpuqr_replay *rp, int32_t r, void *hip_stream)
{
    hipStream_t stream = (hipStream_t) hip_stream ;
    if (!rp || r < 0 || r >= rp->nrounds) return GPUQR_GPUERROR ;
    for (int slot = 0 ; slot < QR_NSLOTS ; slot++)
    {
        const int64_t b = (int64_t) r * QR_NSLOTS + slot ;
        const int64_t n = rp->rptr [b+1] - rp->rptr [b] ;
        if (n <= 0) continue ;
        if (qr_launch_tasks (slot_kernel (slot), (const QrTask *) rp->d_tasks, (const QrRef *) rp->d_refs + rp->rptr [b],
            (int) n, stream) != hipSuccess) return GPUQR_GPUERROR ;
    }
    return GPUQR_SUCCESS ;
}

// gpuqr_factor_sparse that keeps the Householder vectors usable afterwards.  The fronts carry a
// right-hand-side block (gpuqr_front.rhs_col / rhs_n: columns of the front's storage beyond its fn
// columns, zero or holding a right-hand side already): every panel's block reflector is applied to
// that block too, its VT tile goes to memory the returned object owns, and the update tasks of the
// blocks are kept on the device so that gpuqr_replay_round can push new right-hand sides through the
// same reflectors, round by round, once the caller has put their rows in place.  The fronts must
// stay where they are, untouched, while the object lives.  One stream, rounds in order.
extern "C" int gpuqr_factor_sparse_keep (gpuqr_front *fronts, int64_t nf, gpuqr_stats *stats, gpuqr_replay **out)
{
    if (out) *out = NULL ;
    if (stats) { stats->kernelTime = 0 ; stats->numLaunches = 0 ; stats->flopsActual = 0 ; }
    if (!out || nf < 0 || (nf > 0 && fronts == NULL)) return GPUQR_GPUERROR ;
    DeviceCache *dc = device_cache ( ) ;
    if (!dc) return GPUQR_GPUERROR ;
    int code = GPUQR_SUCCESS ;
    hipStream_t stream = dc->sCompute ;
    bool eng_verbose = false ; double eng_tlast = 0 ;
    gpuqr_replay *rp = (gpuqr_replay *) g_calloc (1, sizeof (gpuqr_replay)) ;
    if (!rp) return GPUQR_OUTOFMEMORY ;
    new (rp) gpuqr_replay ( ) ;
    (void) hipGetDevice (&rp->device) ;
    try
    {
        Schedule sc ;
        code = build_schedule (fronts, nf, sc, eng_verbose, eng_tlast) ;
        if (code != GPUQR_SUCCESS) { gpuqr_replay_free (rp) ; return code ; }
        const size_t task_bytes = (size_t) std::max<int64_t> (sc.ntasks, 1) * sizeof (QrTask) ;
        const size_t ref_bytes = (size_t) std::max<int64_t> (sc.nrefs, 1) * sizeof (QrRef) ;
        const size_t vt_bytes = (size_t) std::max<int64_t> (sc.plan.n_vt, 1) * QR_VT_DOUBLES * sizeof (double) ;
        void *d_refs_all = NULL ;
        if (hipMalloc (&rp->d_vt, vt_bytes) != hipSuccess) { rp->d_vt = NULL ; gpuqr_replay_free (rp) ; return GPUQR_OUTOFMEMORY ; }
        if (hipMalloc (&rp->d_tasks, task_bytes) != hipSuccess) { rp->d_tasks = NULL ; gpuqr_replay_free (rp) ; return GPUQR_OUTOFMEMORY ; }
        if (hipMalloc (&d_refs_all, ref_bytes) != hipSuccess) { gpuqr_replay_free (rp) ; return GPUQR_OUTOFMEMORY ; }
        hvec<QrTask> h_tasks ((size_t) std::max<int64_t> (sc.ntasks, 1)) ;
        hvec<QrRef> h_refs ((size_t) std::max<int64_t> (sc.nrefs, 1)) ;
        code = stage_tasks (sc, (double *) rp->d_vt, h_tasks.data ( ), h_refs.data ( )) ;
        // the replay: the right-hand-side updates alone, in the same (round, slot) buckets
        rp->nrounds = sc.nrounds ;
        rp->rptr.assign ((size_t) sc.nbuckets + 1, 0) ;
        hvec<QrRef> r_refs ;
        if (code == GPUQR_SUCCESS)
        {
            for (int64_t i = 0 ; i < sc.ntasks ; i++)
                if (sc.plan.tasks [i].is_rhs) rp->rptr [(int64_t) sc.plan.tasks [i].round * QR_NSLOTS + sc.plan.tasks [i].slot + 1]++ ;
            for (int64_t b = 0 ; b < sc.nbuckets ; b++) rp->rptr [b+1] += rp->rptr [b] ;
            rp->nrefs = rp->rptr [sc.nbuckets] ;
            r_refs.assign ((size_t) std::max<int64_t> (rp->nrefs, 1), QrRef ( )) ;
            hvec<int64_t> pos (rp->rptr) ;
            for (int64_t i = 0 ; i < sc.ntasks ; i++)
            {
                const SchedTask &st = sc.plan.tasks [i] ;
                if (!st.is_rhs) continue ;
                QrRef &q = r_refs [pos [(int64_t) st.round * QR_NSLOTS + st.slot]++] ;
                q.task = (int32_t) i ; q.blk = 0 ;
            }
            rp->done_round.assign ((size_t) nf, -1) ;
            for (int64_t f = 0 ; f < nf ; f++)
                rp->done_round [f] = std::max (sc.done_last [f*QR_NCLASS], std::max (sc.done_last [f*QR_NCLASS+1], sc.done_last [f*QR_NCLASS+2])) ;
            if (hipMalloc (&rp->d_refs, r_refs.size ( ) * sizeof (QrRef)) != hipSuccess) { rp->d_refs = NULL ; code = GPUQR_OUTOFMEMORY ; }
        }
        if (code == GPUQR_SUCCESS)
        {
            int64_t launches = 0 ;
            float ms = 0 ;
            bool ok = hipMemcpyAsync (rp->d_tasks, h_tasks.data ( ), (size_t) sc.ntasks * sizeof (QrTask), hipMemcpyHostToDevice, stream) == hipSuccess
                && hipMemcpyAsync (d_refs_all, h_refs.data ( ), (size_t) sc.nrefs * sizeof (QrRef), hipMemcpyHostToDevice, stream) == hipSuccess
                && hipMemcpyAsync (rp->d_refs, r_refs.data ( ), r_refs.size ( ) * sizeof (QrRef), hipMemcpyHostToDevice, stream) == hipSuccess
                && hipEventRecord (dc->evStart, stream) == hipSuccess ;
            for (int32_t r = 0 ; ok && r < sc.nrounds ; r++)
                for (int slot = 0 ; ok && slot < QR_NSLOTS ; slot++)
                {
                    const int64_t b = (int64_t) r * QR_NSLOTS + slot ;
                    const int64_t n = sc.rptr [b+1] - sc.rptr [b] ;
                    if (n <= 0) continue ;
                    ok = qr_launch_tasks (slot_kernel (slot), (const QrTask *) rp->d_tasks, (const QrRef *) d_refs_all + sc.rptr [b], (int) n, stream) == hipSuccess ;
                    launches++ ;
                    if (ok && (launches & 2047) == 0) ok = hipStreamSynchronize (stream) == hipSuccess ;   // bound on the queued launches
                }
            ok = ok && hipEventRecord (dc->evStop, stream) == hipSuccess ;
            ok = (hipStreamSynchronize (stream) == hipSuccess) && ok ;
            if (ok) ok = hipEventElapsedTime (&ms, dc->evStart, dc->evStop) == hipSuccess ;
            if (!ok) code = GPUQR_GPUERROR ;
            if (stats) { stats->kernelTime = ms ; stats->numLaunches = launches ; stats->flopsActual = sc.plan.flops ; }
        }
        (void) hipFree (d_refs_all) ;
    }
    catch (std::bad_alloc &)
    {
        code = GPUQR_OUTOFMEMORY ;
    }
    if (code != GPUQR_SUCCESS) { (void) hipStreamSynchronize (stream) ; gpuqr_replay_free (rp) ; return code ; }
    *out = rp ;
    return GPUQR_SUCCESS ;
}

// Row pitch of the device copy of a dense front: even, so that with a 16-byte aligned base every
// row starts on a 16-byte boundary (the update kernel moves two columns per lane); from 96 columns
// on a multiple of 16 doubles, so that rows start on 128-byte lines.
static inline int64_t dense_pitch (int64_t n)
{
    return (n >= 96) ? ((n + 15) & ~(int64_t) 15) : (n + (n & 1)) ;
}

extern "C" int gpuqr_factor_dense
(
    size_t gpuMemorySize, gpuqr_front *userFronts, int64_t nf, gpuqr_stats *stats
)
{
    (void) gpuMemorySize ;
    int code = GPUQR_SUCCESS ;
    if (stats) { stats->kernelTime = 0 ; stats->numLaunches = 0 ; stats->flopsActual = 0 ; }
    if (nf < 0 || (nf > 0 && userFronts == NULL)) return GPUQR_GPUERROR ;
    if (nf == 0) return GPUQR_SUCCESS ;

    gpuqr_front *fronts = (gpuqr_front *) g_calloc ((size_t) nf, sizeof (gpuqr_front)) ;
    int64_t **ownStair = (int64_t **) g_calloc ((size_t) nf, sizeof (int64_t *)) ;
    double *hF = NULL, *hR = NULL, *dF = NULL ;
    size_t FSize = 0, RSize = 0 ;
    if (!fronts || !ownStair) { code = GPUQR_OUTOFMEMORY ; goto cleanup ; }

    for (int64_t f = 0 ; f < nf ; f++)
    {
        gpuqr_front &u = userFronts [f] ;
        if (u.fm < 0 || u.fn < 0 || (u.fm > 0 && u.fn > 0 && (!u.F || !u.cpuR))) { code = GPUQR_GPUERROR ; goto cleanup ; }
        FSize = (FSize + 15) & ~(size_t) 15 ;
        FSize += (size_t) u.fm * (size_t) dense_pitch (u.fn) ;
        RSize += (size_t) std::min (u.fm, u.fn) * u.fn ;
    }
    // pinned staging copies: the H2D/D2H transfers then run asynchronously
    if (FSize > 0)
    {
        if (hipHostMalloc ((void **) &hF, FSize * sizeof (double), hipHostMallocDefault) != hipSuccess) { hF = NULL ; code = GPUQR_OUTOFMEMORY ; goto cleanup ; }
        if (hipHostMalloc ((void **) &hR, std::max (RSize, (size_t) 1) * sizeof (double), hipHostMallocDefault) != hipSuccess) { hR = NULL ; code = GPUQR_OUTOFMEMORY ; goto cleanup ; }
        if (hipMalloc ((void **) &dF, FSize * sizeof (double)) != hipSuccess) { dF = NULL ; code = GPUQR_OUTOFMEMORY ; goto cleanup ; }
    }
    {
        size_t FOff = 0, ROff = 0 ;
        for (int64_t f = 0 ; f < nf ; f++)
        {
            gpuqr_front &u = userFronts [f] ;
            gpuqr_front &fr = fronts [f] ;
            const int64_t m = u.fm, n = u.fn ;
            const int64_t ldn = (u.ldn > 0) ? u.ldn : (u.is_col_major ? m : n) ;
            fr.fids = f ; fr.pids = -1 ; fr.fidg = u.fidg ; fr.pidg = u.pidg ;
            fr.fm = m ; fr.fn = n ; fr.fp = u.fp ; fr.rank = std::min (m, n) ;
            fr.is_sparse = 0 ;
            const int64_t np = dense_pitch (n) ;
            FOff = (FOff + 15) & ~(size_t) 15 ;
            fr.is_col_major = 0 ; fr.ldn = np ;
            double *F = hF + FOff ;
            fr.F = F ; fr.gpuF = dF + FOff ; fr.cpuR = hR + ROff ;
            for (int64_t i = 0 ; i < m ; i++)
            {
                for (int64_t j = 0 ; j < n ; j++)
                    F [i*np + j] = u.is_col_major ? u.F [j*ldn + i] : u.F [i*ldn + j] ;
                for (int64_t j = n ; j < np ; j++) F [i*np + j] = 0 ;
            }
            fr.Stair = u.Stair ;
            if (fr.Stair == NULL && m > 0 && n > 0)
            {
                ownStair [f] = find_staircase_ld (F, m, n, np) ;
                if (!ownStair [f]) { code = GPUQR_OUTOFMEMORY ; goto cleanup ; }
                fr.Stair = ownStair [f] ;
            }
            FOff += (size_t) m * np ;
            ROff += (size_t) fr.rank * n ;
        }
    }
    if (FSize > 0 && hipMemcpy (dF, hF, FSize * sizeof (double), hipMemcpyHostToDevice) != hipSuccess) { code = GPUQR_GPUERROR ; goto cleanup ; }

    code = engine_run (fronts, nf, stats, 0) ;
    if (code != GPUQR_SUCCESS) goto cleanup ;

    for (int64_t f = 0 ; f < nf ; f++)
    {
        gpuqr_front &u = userFronts [f] ;
        const int64_t m = u.fm, n = u.fn, rank = std::min (m, n) ;
        const int64_t ldn = (u.ldn > 0) ? u.ldn : (u.is_col_major ? m : n) ;
        const double *R = fronts [f].cpuR ;
        u.rank = rank ;
        // R(i,j), i < rank, goes to the caller's layout (the reference indexes the
        // column-major case transposed, GPUQREngine_ExpertDense.cpp:146-152, which is
        // only right for square R; here both layouts get the same matrix)
        for (int64_t i = 0 ; i < rank ; i++)
            for (int64_t j = 0 ; j < n ; j++)
            {
                if (u.is_col_major) u.cpuR [j*ldn + i] = R [i*n + j] ;
                else                u.cpuR [i*ldn + j] = R [i*n + j] ;
            }
    }

cleanup:
    if (ownStair) { for (int64_t f = 0 ; f < nf ; f++) if (ownStair [f]) g_free (ownStair [f]) ; g_free (ownStair) ; }
    if (fronts) g_free (fronts) ;
    if (hF) (void) hipHostFree (hF) ;
    if (hR) (void) hipHostFree (hR) ;
    if (dF) (void) hipFree (dF) ;
    return code ;
}
