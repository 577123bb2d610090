/* C ABI of the MI355X multifrontal QR front-batch engine (libgpuqrengine_amd.so).
 *
 * These are the symbols a foreign-function binding (or the C++ shim in
 * GPUQREngine.hpp next to this file) binds in place of the reference's C++
 * overloads.  Plain pointers and sizes only.  Every function is callable from
 * C; none throws.  All sizes/indices are 64-bit like the reference's
 * Int = SuiteSparse_long (GPUQREngine/Include/GPUQREngine_Common.hpp:64).
 */
#ifndef GPUQRENGINE_AMD_H
#define GPUQRENGINE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* result codes, same values and meaning as QREngineResultCode
 * (GPUQREngine/Include/GPUQREngine.hpp:17-22) */
enum
{
    GPUQR_SUCCESS     = 0,
    GPUQR_OUTOFMEMORY = 1,
    GPUQR_GPUERROR    = 2
};

/* One entry of the packed S list: F[findex] = value.  Same 16-byte layout as
 * SEntry (GPUQREngine/Include/GPUQREngine_SEntry.hpp:13-23). */
typedef struct gpuqr_sentry
{
    int64_t findex;
    double  value;
} gpuqr_sentry;

/* Plain-data mirror of Front + SparseMeta
 * (GPUQREngine/Include/GPUQREngine_Front.hpp:20-86,
 *  GPUQREngine/Include/GPUQREngine_SparseMeta.hpp:19-80).
 * Field meaning is the reference's; fields the reference's scheduler kept for
 * its own bookkeeping (state, lastSIndex, lastCiStart, ...) are absent. */
typedef struct gpuqr_front
{
    int64_t fids;          /* front id within the batch; must equal the array index */
    int64_t pids;          /* parent id within the batch, or -1 */
    int64_t fidg, pidg;    /* global ids, carried through untouched */
    int64_t fm, fn, fp;    /* rows, columns, pivotal columns */
    int64_t rank;          /* rows of R pulled to cpuR: min(fm,fn) dense, min(fm,fp) sparse */
    int32_t is_col_major;  /* dense overload only: layout of F and cpuR */
    int32_t pad0;
    int64_t ldn;           /* dense overload only: leading dimension of F and cpuR */
    double *F;             /* dense overload: host input, fm-by-fn */
    double *gpuF;          /* sparse overload: device front, row-major fm-by-fn, caller-zeroed */
    double *cpuR;          /* host destination of R (and C if staged); NULL = leave on device */
    const int64_t *Stair;  /* size fn, Stair[j] = last row of column j that may be nonzero; NULL = find it */

    /* --- SparseMeta --- */
    int32_t is_sparse;     /* 0: dense front, no assembly */
    int32_t is_staged;     /* parent lives in a later stage: pull R and the cm rows of C */
    int32_t push_only;     /* front is just a contribution block (cm-by-cn at gpuC, ld cn) to push */
    int32_t nc;            /* number of children in this batch (informational) */
    const gpuqr_sentry *gpuS; /* device S entries of this front */
    int64_t Scount;
    int32_t cm, cn;        /* contribution block is cm-by-cn */
    int32_t pn;            /* columns of the parent front */
    int32_t pad1;
    const int32_t *gpuRjmap;  /* device, size cn: C column -> parent column */
    const int32_t *gpuRimap;  /* device, size cm: C row -> parent row */
    double *gpuC;          /* device location of C(0,0); its leading dimension is fn (cn if push_only) */
    /* R hand-off in SPQR's stack layout, no counterpart in the reference's Front (its caller packs
     * on the host, SPQR/SPQRGPU/spqrgpu_kernel.cpp:417-439).  When gpuRpack is set, the engine writes
     * the first `rank` rows of the reduced front there as a column-packed upper trapezoid (column j
     * holds min(j+1, rank) entries) as soon as the front is final; when cpuRpack is set too it copies
     * that block to the host while other fronts are still being factorized, neighbouring blocks in
     * one transfer.  Both NULL: nothing happens. */
    double *gpuRpack;
    double *cpuRpack;
    /* Right-hand-side block, no counterpart in the reference's Front (its GPU path gets Q'B only by
     * appending B to A before the analysis, SPQR/Source/SuiteSparseQR.cpp:256-260): columns
     * [rhs_col, rhs_col + rhs_n) of the front's storage, rhs_col >= fn and rhs_col + rhs_n <= ldn.
     * Every panel's block reflector is applied to them; they are never factorized.  rhs_n = 0: none. */
    int32_t rhs_col;
    int32_t rhs_n;
    /* Host copy of gpuRimap (size cm), optional, no counterpart in the reference's Front.  With it the
     * engine knows which rows of the parent each block of C rows lands in, sends the rows of C to the
     * parent as soon as they are final and lets the parent's first panels start while this front is
     * still working on its later ones.  It must hold exactly what gpuRimap holds (the engine orders
     * the work by it and never compares the two).  NULL: the parent waits for the whole front. */
    const int32_t *cpuRimap;
} gpuqr_front;

/* same fields as QREngineStats (GPUQREngine/Include/GPUQREngine_Stats.hpp:13-29) */
typedef struct gpuqr_stats
{
    float   kernelTime;    /* ms spent in the round launches, measured with HIP events */
    int64_t numLaunches;
    int64_t flopsActual;
} gpuqr_stats;

/* Dense overload: replaces
 *   QREngineResultCode GPUQREngine (size_t, Front *, Int, QREngineStats *)
 * (GPUQREngine/Include/GPUQREngine.hpp:26-32,
 *  GPUQREngine/Source/GPUQREngine_ExpertDense.cpp:29-158).
 * Each front's F (host, row- or column-major with leading dimension ldn) is
 * copied to the device, factorized, and the first `rank` rows of the reduced
 * front are written to cpuR in the caller's layout. */
int gpuqr_factor_dense (size_t gpuMemorySize, gpuqr_front *fronts, int64_t numFronts,
                        gpuqr_stats *stats);

/* Sparse overload: replaces
 *   QREngineResultCode GPUQREngine (size_t, Front *, Int, Int *Parent, Int *Childp, Int *Child,
 *                                   QREngineStats *)
 * (GPUQREngine/Include/GPUQREngine.hpp:34-43,
 *  GPUQREngine/Source/GPUQREngine_ExpertSparse.cpp:13-29, GPUQREngine_Internal.cpp:16-88).
 * The caller owns every device buffer the fronts point to; the tree comes
 * from pids.  Parent/Childp/Child are accepted and ignored, as in the
 * reference.  Blocks until every front is reduced and its R (and C rows when
 * is_staged) is in cpuR. */
int gpuqr_factor_sparse (size_t gpuMemorySize, gpuqr_front *fronts, int64_t numFronts,
                         const int64_t *Parent, const int64_t *Childp, const int64_t *Child,
                         gpuqr_stats *stats);

/* Replaces Int *GPUQREngine_FindStaircase (Front *)
 * (GPUQREngine/Source/GPUQREngine_ExpertDense.cpp:196-217).  F is row-major
 * fm-by-fn on the host.  Returns a malloc'd (through the hooks below) array of
 * size fn, or NULL. */
int64_t *gpuqr_find_staircase (const double *F, int64_t fm, int64_t fn);

/* Replaces size_t ssgpu_maxQueueSize (size_t)
 * (GPUQREngine/Source/Scheduler/ssgpu_maxQueueSize.cpp:6-27).  The value feeds
 * the staging arithmetic only; the engine's own task list is not bounded by it. */
size_t gpuqr_max_queue_size (size_t gpuMemorySize);

/* gpuqr_factor_sparse for callers that factorize the same problem again and again (same fronts, same
 * shapes, staircases, tree and device pointers; only the numbers in the fronts change), the usual
 * case of a sparse factorization inside an iteration.  A nonzero plan_key is the caller's promise that
 * every call made with it describes that one problem: the engine then builds and uploads its schedule
 * once and replays it while no call with another key (or none) has intervened on the device.  No
 * counterpart in the reference, which rebuilds its Scheduler on every call
 * (GPUQREngine/Source/GPUQREngine_Internal.cpp:34-51).  plan_key = 0 behaves as gpuqr_factor_sparse:
 * the engine then derives a key itself from the front descriptors and the staircase contents (a
 * 64-bit fingerprint, a few ms for a million fronts' worth), so an unchanged description is
 * replayed all the same; GPUQR_NO_PLAN_CACHE in the environment turns that off. */
int gpuqr_factor_sparse_keyed (size_t gpuMemorySize, gpuqr_front *fronts, int64_t numFronts,
                               const int64_t *Parent, const int64_t *Childp, const int64_t *Child,
                               gpuqr_stats *stats, uint64_t plan_key);

/* gpuqr_factor_sparse that keeps H: the block reflectors stay usable after the call (V in place in
 * the fronts, the VT tiles in memory the returned object owns), and the update tasks of the fronts'
 * right-hand-side blocks (gpuqr_front.rhs_col / rhs_n) can be run again on new contents of those
 * blocks: gpuqr_replay_round (rp, r, stream) queues round r of them, r = 0 .. gpuqr_replay_rounds - 1;
 * gpuqr_replay_front_done (rp, f) is the last round that touches front f (its finished rows may be
 * handed to the parent's block after it).  What the reference would need keepH for and therefore
 * leaves to the CPU (SPQR/Source/SuiteSparseQR_expert.cpp:324-343).  The fronts must stay in place,
 * unmodified, while the object lives.  One stream, rounds in order. */
typedef struct gpuqr_replay gpuqr_replay;
int gpuqr_factor_sparse_keep (gpuqr_front *fronts, int64_t numFronts, gpuqr_stats *stats, gpuqr_replay **out);
int32_t gpuqr_replay_rounds (const gpuqr_replay *rp);
int32_t gpuqr_replay_front_done (const gpuqr_replay *rp, int64_t front);
int gpuqr_replay_round (const gpuqr_replay *rp, int32_t round, void *hip_stream);
void gpuqr_replay_free (gpuqr_replay *rp);

/* Exact flop count of the schedule the engine would run for one fm-by-fn
 * front with staircase Stair (NULL = full), the analogue of getFlops
 * (GPUQREngine/Source/TaskDescriptor/TaskDescriptor_flops.cpp:70-104). */
int64_t gpuqr_front_flops (int64_t fm, int64_t fn, const int64_t *Stair);

/* Debugging mode: every kernel launch is waited for and checked before the next one is issued, and a
 * failure is reported with its round and kernel (stderr) and ends the call with GPUQR_GPUERROR: what
 * the reference builds with -DFORCE_SYNCHRONIZE (GPUQREngine/Source/Scheduler/Scheduler_LaunchKernel.cpp:
 * 55-58).  Also switched on by GPUQR_SERIAL=1 in the environment.  Results are bit-identical to the
 * normal mode's (same tasks, same order within a stream). */
void gpuqr_set_serial (int on);

/* Number of schedules planned in this process so far.  A call whose problem (shapes, staircases, tree,
 * device pointers) the engine has seen before replays the schedule it kept on the device and plans
 * nothing; a caller that factorizes again can see from this counter whether that happened.  No
 * counterpart in the reference, which plans every launch anew. */
int64_t gpuqr_schedules_built (void);

/* Device memory the engine itself allocates for one fm-by-fn front beyond the front (task records,
 * workgroup list, VT tiles), an upper bound.  The reference prices its queue and VT pools inside
 * spqrgpu_computeFrontStaging (SPQR/SPQRGPU/spqrgpu_computeFrontStaging.cpp:90-118,161-164) with the
 * sizes of ITS scheduler; this engine's differ, and a driver that must fit a stage into real memory asks
 * here. */
int64_t gpuqr_front_workspace_bytes (int64_t fm, int64_t fn);

/* The per-task counts that sum is made of: the reference's formulas flopsFactorizeVT / flopsFactorize /
 * flopsApply (TaskDescriptor_flops.cpp:18-28, :34-42, :48-53) evaluated at the task's own dimensions.
 * On the reference's tile shapes (m = 32*numTiles rows, n = 32 columns, k = 32 reflectors) they are the
 * reference's numbers.  They differ from a reference run's total because the tasks differ: a panel
 * task here covers up to 256 rows (the reference's at most 96), so there are fewer merge levels. */
int64_t gpuqr_flops_factorize (int64_t m, int64_t n, int with_t);
int64_t gpuqr_flops_apply (int64_t m, int64_t k, int64_t n);

/* Host allocator hooks, the analogue of SuiteSparse_config's function table
 * (SuiteSparse_config/SuiteSparse_config.h:87-91): every host allocation the
 * engine makes goes through these.  NULL restores malloc/calloc/free. */
void gpuqr_set_allocator (void *(*malloc_fn) (size_t), void *(*calloc_fn) (size_t, size_t),
                          void (*free_fn) (void *));

/* Per-kernel profile of the engine's own launches, measured with HIP events on
 * the stream they run on (no counterpart in the reference beyond its -DTIMING
 * total, GPUQREngine/Source/Scheduler/Scheduler_LaunchKernel.cpp:46-64).
 * Kernel ids: 0..3 factorize (32/64/128/256-row tiers), 4..7 apply (same
 * tiers), 8 assembly, 9 factorize launches that mix tiers, 10 apply launches that mix the three
 * small tiers, 11 apply launches that mix all tiers.  While enabled every engine call adds to the totals:
 * launch time in ms, launches, tasks, algorithmic flops and bytes. */
void gpuqr_profile_enable (int on);
void gpuqr_profile_reset (void);
int  gpuqr_profile_get (int kernel_id, double *ms, int64_t *launches, int64_t *tasks,
                        double *flops, double *bytes);

/* Execution mode: 0 = one launch per (round, kernel) on two streams (default), 1 = the whole call
 * as one persistent kernel that pulls the same task list from a device-resident work queue with
 * per-front arrival counters (the replacement of the reference's Scheduler/BucketList/LLBundle
 * loop, GPUQREngine/Source/GPUQREngine_Internal.cpp:57-73), 2 = the engine picks per call: the queue
 * when the per-round launches would be badly filled (long dependency chains of small tasks), launches
 * otherwise.  Environment GPUQR_MODE sets the initial value. */
void gpuqr_set_mode (int mode);

/* The engine keeps two streams, two events and grow-only task/VT buffers per
 * device between calls (the reference rebuilds its Scheduler every call,
 * GPUQREngine/Source/GPUQREngine_Internal.cpp:34-51).  This frees them. */
void gpuqr_finalize (void);

/* Per-device set-up of what the engine keeps between calls (two compute streams, a copy stream, two
 * events, kernel attributes, the address of the device's fixed workspace), for the current device.
 * Every entry point does this by itself on its first call on a device; a caller that must not
 * allocate inside gpuqr_factor_sparse_noalloc calls it once beforehand.  No counterpart in the
 * reference, whose Scheduler constructor does the equivalent on every call
 * (GPUQREngine/Source/Scheduler/Scheduler.cpp:172-253).  Returns GPUQR_SUCCESS or GPUQR_GPUERROR. */
int gpuqr_init (void);

/* gpuqr_factor_sparse for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE.  Same fronts, same
 * kernels and same results as gpuqr_factor_sparse (device-resident fronts in gpuF, reduced in
 * place; R rows copied to cpuR where it is set), but the task records, the workgroup list and the
 * VT tiles live in a fixed device workspace of gpuqr_noalloc_workspace_bytes() = 32 MiB that is part
 * of the library's code object (each GPU the library runs on holds it until the process ends), and
 * the planner's host memory comes from a static 128 MiB block: after gpuqr_init() on the device the
 * call makes no hipMalloc, hipHostMalloc, hipFree, stream or event creation, and no heap allocation.
 * It runs on one stream, round after round.  Limits: a batch whose records and VT tiles exceed the
 * workspace (or whose plan exceeds the host block) is refused with GPUQR_TOOLARGE before anything
 * is launched (gpuqr_noalloc_last_need() then tells the bytes it would take); fronts that ask for
 * the packed R hand-off (gpuRpack / cpuRpack) are refused with GPUQR_GPUERROR.  All calls on one GPU
 * share that GPU's workspace: calls that overlap on one GPU race, also with each other's results.
 * The static host block is one per process: calls from several threads (one per GPU) take turns on
 * it, from planning to the last launch; calls of the other entry points on other threads are not
 * affected. */
#define GPUQR_TOOLARGE 3
int gpuqr_factor_sparse_noalloc (gpuqr_front *fronts, int64_t numFronts, gpuqr_stats *stats);
size_t gpuqr_noalloc_workspace_bytes (void);
size_t gpuqr_noalloc_last_need (void);

/* Number of HIP devices visible, or 0 (never fails). */
int gpuqr_device_count (void);

/* Library version, major*1000 + minor */
int gpuqr_version (void);

#ifdef __cplusplus
}
#endif
#endif
