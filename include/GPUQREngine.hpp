// Source-compatible C++ face of the engine: the class and function names of
// GPUQREngine/Include/GPUQREngine.hpp:17-48, GPUQREngine_Front.hpp:20-167,
// GPUQREngine_SparseMeta.hpp:19-80, GPUQREngine_SEntry.hpp:13-23 and
// GPUQREngine_Stats.hpp:13-29, implemented as a header-only shim over the C ABI
// in gpuqrengine_amd.h.  Code written against the reference's headers (for
// example SPQR/SPQRGPU/spqrgpu_kernel.cpp:516-555,638) compiles against this
// one unchanged; the overloads convert the Front array to the plain-data
// gpuqr_front records and call libgpuqrengine_amd.so.
#ifndef GPUQRENGINE_HPP
#define GPUQRENGINE_HPP

#include <stddef.h>
#include <stdlib.h>
#include <new>
#include "gpuqrengine_amd.h"

// Int is SuiteSparse_long.  The reference makes it a macro (GPUQREngine_Common.hpp:64), and so does this
// fork's CHOLMOD/Include/cholmod_subtree.h:28-32 (as int, unless DLONG is set), which SuiteSparseQR.hpp
// pulls in ahead of this header: such a macro is dropped here, as the reference's own redefinition does.
#ifndef GPUQRENGINE_INT_DEFINED
#define GPUQRENGINE_INT_DEFINED
#ifdef Int
#undef Int
#endif
typedef long Int ;                      // SuiteSparse_long
#endif

// SuiteSparse_GPURuntime's buffer class (SuiteSparse_GPURuntime/Include/SuiteSparseGPU_Workspace.hpp:10-108).
// The engine has no use for buffers of that kind (it calls hipMalloc directly), but this fork's
// spqr_symbolic keeps pointers to workspaces that its spqrgpu_computeFrontStaging allocates at symbolic
// time (SPQR/Include/SuiteSparseQR.hpp:208-219, SPQR/SPQRGPU/spqrgpu_computeFrontStaging.cpp:251-286)
// and that spqr_1factor lets go of with Workspace::destroy once the numeric factorization is done
// (SPQR/Source/spqr_1factor.cpp:643-655).  That lifetime is what the AMD driver's own symbolic object
// wants too (assembly maps, S permutation, the engine's schedule live in it), so the class here is a
// handle with exactly that protocol: something to keep, and how to let go of it.
class Workspace
{
public:
    void *payload ;
    void (*release) (void *) ;
    Workspace (void *payload_arg, void (*release_arg) (void *)) : payload (payload_arg), release (release_arg) { }
    void *cpu ( ) { return payload ; }
    void *gpu ( ) { return NULL ; }
    static Workspace *destroy (Workspace *address)
    {
        if (address)
        {
            if (address->release && address->payload) address->release (address->payload) ;
            delete address ;
        }
        return NULL ;
    }
} ;
#ifndef EMPTY
#define EMPTY (-1)
#endif

// geometry constants callers size things with (GPUQREngine_Common.hpp:38-45)
#define TILESIZE 32
#define PANELSIZE 3

enum QREngineResultCode
{
    QRENGINE_SUCCESS     = GPUQR_SUCCESS,
    QRENGINE_OUTOFMEMORY = GPUQR_OUTOFMEMORY,
    QRENGINE_GPUERROR    = GPUQR_GPUERROR
} ;

enum FrontState                         // kept for source compatibility; the engine has no per-front FSM
{
    ALLOCATE_WAIT = 0, ASSEMBLE_S = 1, CHILD_WAIT = 2, FACTORIZE = 3, FACTORIZE_COMPLETE = 4,
    PARENT_WAIT = 5, PUSH_ASSEMBLE = 6, CLEANUP = 7, DONE = 8
} ;

struct SEntry
{
    long findex ;
    double value ;
} ;

struct QREngineStats
{
    float kernelTime ;
    Int numLaunches ;
    Int flopsActual ;
    QREngineStats ( ) : kernelTime (0), numLaunches (0), flopsActual (0) { }
} ;

class SparseMeta
{
public:
    int fp, nc ;
    bool isStaged, pushOnly, isSparse ;
    int lastSIndex ;
    SEntry *cpuS, *gpuS ;
    int Scount ;
    int cm, cn, csize, pn, pc, lastCiStart, lastCjStart ;
    int *gpuRjmap, *gpuRimap ;
    double *gpuC, *gpuP ;
    const int *cpuRimap ;       // extension (not in the reference's class): host copy of gpuRimap, may stay NULL
    SparseMeta ( ) : fp (0), nc (0), isStaged (false), pushOnly (false), isSparse (false),
        lastSIndex (0), cpuS (NULL), gpuS (NULL), Scount (0), cm (0), cn (0), csize (0), pn (0),
        pc (0), lastCiStart (0), lastCjStart (0), gpuRjmap (NULL), gpuRimap (NULL), gpuC (NULL),
        gpuP (NULL), cpuRimap (NULL) { }
} ;

class Front
{
public:
    Int fids, pids, fidg, pidg ;
    Int fm, fn, fp, rank ;
    bool isColMajor ;
    Int ldn ;
    double *F, *gpuF, *cpuR ;
    FrontState state ;
    Int *Stair ;
    SparseMeta sparseMeta ;
    bool printMe ;

    void *operator new (size_t, Front *ptr) { return ptr ; }
    void operator delete (void *, Front *) { }

    Front (Int fids_arg, Int pids_arg, Int fm_arg, Int fn_arg, Int fp_arg,
           bool isColMajor_arg = false, Int ldn_arg = EMPTY) :
        fids (fids_arg), pids (pids_arg), fidg (EMPTY), pidg (EMPTY), fm (fm_arg), fn (fn_arg),
        fp (fp_arg), rank (fm_arg < fn_arg ? fm_arg : fn_arg), isColMajor (isColMajor_arg),
        ldn (ldn_arg == EMPTY ? fn_arg : ldn_arg), F (NULL), gpuF (NULL), cpuR (NULL),
        state (ALLOCATE_WAIT), Stair (NULL), sparseMeta ( ), printMe (false) { }

    bool isAllocated ( ) { return gpuF != NULL ; }
    bool isDense ( )     { return !sparseMeta.isSparse ; }
    bool isSparse ( )    { return sparseMeta.isSparse ; }
    bool isStaged ( )    { return isSparse ( ) && sparseMeta.isStaged ; }
    bool isPushOnly ( )  { return isSparse ( ) && sparseMeta.pushOnly ; }
    size_t getNumFrontValues ( ) { return (size_t) (fm * fn) ; }
    size_t getNumRValues ( )     { return (size_t) (rank * fn) ; }
    bool isTooBigForSmallQR ( )  { return fm > PANELSIZE * TILESIZE || fn > TILESIZE ; }
} ;

namespace gpuqr_shim
{
    inline void pack (const Front &f, gpuqr_front &c)
    {
        c = gpuqr_front ( ) ;
        c.fids = f.fids ; c.pids = f.pids ; c.fidg = f.fidg ; c.pidg = f.pidg ;
        c.fm = f.fm ; c.fn = f.fn ; c.fp = f.fp ; c.rank = f.rank ;
        c.is_col_major = f.isColMajor ; c.ldn = f.ldn ;
        c.F = f.F ; c.gpuF = f.gpuF ; c.cpuR = f.cpuR ;
        c.Stair = reinterpret_cast<const int64_t *> (f.Stair) ;
        const SparseMeta &m = f.sparseMeta ;
        c.is_sparse = m.isSparse ; c.is_staged = m.isStaged ; c.push_only = m.pushOnly ; c.nc = m.nc ;
        c.gpuS = reinterpret_cast<const gpuqr_sentry *> (m.gpuS) ; c.Scount = m.Scount ;
        c.cm = m.cm ; c.cn = m.cn ; c.pn = m.pn ;
        c.gpuRjmap = m.gpuRjmap ; c.gpuRimap = m.gpuRimap ; c.gpuC = m.gpuC ;
        c.cpuRimap = m.cpuRimap ;
    }
    inline QREngineResultCode run (size_t mem, Front *fronts, Int n, bool dense,
        const Int *Parent, const Int *Childp, const Int *Child, QREngineStats *stats)
    {
        static_assert (sizeof (Int) == sizeof (int64_t), "Int must be 64 bits") ;
        static_assert (sizeof (SEntry) == sizeof (gpuqr_sentry), "SEntry must be 16 bytes") ;
        gpuqr_front *c = (gpuqr_front *) malloc ((size_t) (n > 0 ? n : 1) * sizeof (gpuqr_front)) ;
        if (!c) return QRENGINE_OUTOFMEMORY ;
        for (Int k = 0 ; k < n ; k++) pack (fronts [k], c [k]) ;
        gpuqr_stats st ;
        int code = dense ? gpuqr_factor_dense (mem, c, n, &st)
                         : gpuqr_factor_sparse (mem, c, n, (const int64_t *) Parent,
                               (const int64_t *) Childp, (const int64_t *) Child, &st) ;
        if (dense) for (Int k = 0 ; k < n ; k++) fronts [k].rank = c [k].rank ;
        if (stats) { stats->kernelTime = st.kernelTime ; stats->numLaunches = st.numLaunches ; stats->flopsActual = st.flopsActual ; }
        free (c) ;
        return (QREngineResultCode) code ;
    }
}

// dense fronts: host F in, cpuR out (GPUQREngine.hpp:26-32)
inline QREngineResultCode GPUQREngine (size_t gpuMemorySize, Front *userFronts, Int numFronts,
    QREngineStats *stats = NULL)
{
    return gpuqr_shim::run (gpuMemorySize, userFronts, numFronts, true, NULL, NULL, NULL, stats) ;
}

// fronts of a sparse factorization, device buffers owned by the caller (GPUQREngine.hpp:34-43)
inline QREngineResultCode GPUQREngine (size_t gpuMemorySize, Front *userFronts, Int numFronts,
    Int *Parent, Int *Childp, Int *Child, QREngineStats *stats = NULL)
{
    return gpuqr_shim::run (gpuMemorySize, userFronts, numFronts, false, Parent, Childp, Child, stats) ;
}

// staircase of a dense row-major front (GPUQREngine.hpp:45-48); free with free()
inline Int *GPUQREngine_FindStaircase (Front *front)
{
    return (Int *) gpuqr_find_staircase (front->F, front->fm, front->fn) ;
}

inline size_t ssgpu_maxQueueSize (size_t gpuMemorySize) { return gpuqr_max_queue_size (gpuMemorySize) ; }

#define GPUQRENGINE_DATE "Nov 2026"
#define GPUQRENGINE_VER_CODE(main,sub) ((main) * 1000 + (sub))
#define GPUQRENGINE_MAIN_VERSION 1
#define GPUQRENGINE_SUB_VERSION 0
#define GPUQRENGINE_SUBSUB_VERSION 5
#define GPUQRENGINE_VERSION GPUQRENGINE_VER_CODE(GPUQRENGINE_MAIN_VERSION,GPUQRENGINE_SUB_VERSION)

#endif
