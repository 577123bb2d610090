// Definition of the opaque symbolic object of include/spqrgpu_amd.h: the fields
// of spqr_symbolic / spqr_gpu the GPU path reads
// (SPQR/Include/SuiteSparseQR.hpp:118-260), as owning vectors.
#ifndef SUITESPARSE_AMD_SPQR_SYMBOLIC_H
#define SUITESPARSE_AMD_SPQR_SYMBOLIC_H

#include <stdint.h>
#include <stddef.h>
#include <vector>
#include <new>
#include "../../include/gpuqrengine_amd.h"

// host memory through the engine's allocator hooks (qr_engine.cpp)
extern "C" void *qr_host_malloc (size_t bytes) ;
extern "C" void *qr_host_calloc (size_t n, size_t size) ;
extern "C" void  qr_host_free (void *p) ;

template <class T> struct HostAlloc
{
    typedef T value_type ;
    HostAlloc ( ) { }
    template <class U> HostAlloc (const HostAlloc<U> &) { }
    T *allocate (size_t n)
    {
        if (n > (size_t) -1 / sizeof (T)) throw std::bad_alloc ( ) ;
        void *p = qr_host_malloc (n * sizeof (T)) ;
        if (!p) throw std::bad_alloc ( ) ;
        return (T *) p ;
    }
    void deallocate (T *p, size_t) { qr_host_free (p) ; }
    template <class U> bool operator== (const HostAlloc<U> &) const { return true ; }
    template <class U> bool operator!= (const HostAlloc<U> &) const { return false ; }
} ;
template <class T> using hostvec = std::vector<T, HostAlloc<T> > ;

struct spqr_amd_symbolic
{
    int64_t m = 0, n = 0, anz = 0, nf = 0, maxfn = 0, rjsize = 0 ;
    int64_t RimapSize = 0, RjmapSize = 0, numStages = 0, rnz = 0 ;
    double  flopcount = 0 ;
    size_t  gpuMemorySize = 0 ;
    int     do_rank_detection = 0 ; // Cm and Fm sized for pivot columns that die (spqr_analyze.cpp:826-842)
    uint64_t uid = 0 ;          // unique per analysis in this process (an address can come back, this cannot)
    std::vector<int64_t> Qfill, PLinv, Sleft, Sp, Sj, Super, Rp, Rj, Parent, Childp, Child, Post,
        Fm, Cm, RjmapOffsets, RimapOffsets, Stagingp, StageMap, FSize, RSize, SSize,
        FOffsets, ROffsets, SOffsets, ColCount ;

    // Pattern-only by-products of the assembly maps (spqrgpu_buildAssemblyMaps.cpp computes them in
    // every numeric call; they depend on the pattern alone).  Built by the first numeric call and
    // kept, host and device side, for the later ones.
    struct NumericCache
    {
        bool built = false ;
        int  device = -1 ;                  // the device that holds the d_* copies
        uint64_t pattern = 0 ;              // fingerprint of the Ap / Ai the cache was built from
        hostvec<int64_t> Stair ;            // per front, at Rp [f]
        hostvec<int32_t> Rimap, Rjmap ;
        hostvec<int64_t> SGlob ;            // first packed S entry of front f (entries in postorder of fronts)
        void *d_Rimap = NULL, *d_Rjmap = NULL ;
        void *d_findex = NULL ;             // int64 per S entry: where it goes in its front
        void *d_aperm = NULL ;              // int64 per S entry: position of its value in the caller's Ax
        void (*release) (NumericCache *) = NULL ;
        // What a re-factorization need not work out again (all of it follows from the symbolic object,
        // the row pitches and the addresses of the driver's workspaces):
        size_t stage_worst = 0 ;            // device bytes the largest stage takes
        bool   stage_worst_known = false ;
        uint64_t fit_gen = 0 ;              // workspace generation at which the stages were seen to fit the device
        hostvec<gpuqr_front> fronts ;       // the front descriptors of a single-group factorization ...
        uint64_t fronts_key = 0 ;           // ... made for the workspaces this key stands for
        const double *fronts_rstack = NULL ;    // ... and for R blocks delivered to this stack
        // ... and where that group's fronts, S entries and R blocks lie in their arenas
        bool lay_ok = false ;
        int64_t lay_line_from = -1 ;            // (the row-pitch rule the layout was made under)
        hostvec<int64_t> lay_list, lay_FOff, lay_SOff, lay_ROff, lay_rel ;
        int64_t lay_fsum = 0, lay_ssum = 0, lay_rsum = 0 ;
        hostvec<int64_t> lay_sruns ;            // runs of fronts whose S entries are built by one launch: (first front, entries) pairs
        bool lay_sruns_ok = false ;
    } ;
    mutable NumericCache cache ;
    // The per-device copies of this object that spqrgpu_amd_factorize_multi works on (each with a numeric
    // cache on its device), kept from one factorization to the next; owned and released by that file.
    mutable void *multi_copies = NULL ;
    mutable void (*multi_release) (void *) = NULL ;
    ~spqr_amd_symbolic ( )
    {
        if (multi_copies && multi_release) multi_release (multi_copies) ;
        if (cache.release) cache.release (&cache) ;
    }
} ;

#endif
