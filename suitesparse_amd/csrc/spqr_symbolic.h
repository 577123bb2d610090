// Definition of the opaque symbolic object of include/spqrgpu_amd.h: the fields
// of spqr_symbolic / spqr_gpu the GPU path reads
// (SPQR/Include/SuiteSparseQR.hpp:118-260), as owning vectors.
#ifndef SUITESPARSE_AMD_SPQR_SYMBOLIC_H
#define SUITESPARSE_AMD_SPQR_SYMBOLIC_H

#include <stdint.h>
#include <stddef.h>
#include <vector>

struct spqr_amd_symbolic
{
    int64_t m = 0, n = 0, anz = 0, nf = 0, maxfn = 0, rjsize = 0 ;
    int64_t RimapSize = 0, RjmapSize = 0, numStages = 0, rnz = 0 ;
    double  flopcount = 0 ;
    size_t  gpuMemorySize = 0 ;
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
        std::vector<int64_t> Stair ;        // per front, at Rp [f]
        std::vector<int32_t> Rimap, Rjmap ;
        std::vector<int64_t> SGlob ;        // first packed S entry of front f (entries in postorder of fronts)
        void *d_Rimap = NULL, *d_Rjmap = NULL ;
        void *d_findex = NULL ;             // int64 per S entry: where it goes in its front
        void *d_aperm = NULL ;              // int64 per S entry: position of its value in the caller's Ax
        void (*release) (NumericCache *) = NULL ;
    } ;
    mutable NumericCache cache ;
    ~spqr_amd_symbolic ( ) { if (cache.release) cache.release (&cache) ; }
} ;

#endif
