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
    std::vector<int64_t> Qfill, PLinv, Sleft, Sp, Sj, Super, Rp, Rj, Parent, Childp, Child, Post,
        Fm, Cm, RjmapOffsets, RimapOffsets, Stagingp, StageMap, FSize, RSize, SSize,
        FOffsets, ROffsets, SOffsets, ColCount ;
} ;

#endif
