// Drop-in translation unit for an SPQR build: the reference's entry point
//
//     void spqrgpu_buildAssemblyMaps (Long numFronts, Long n, Long *Fmap, ..., SEntry *cpuS) ;
//                                                                 (SPQR/Include/spqrgpu.hpp:59-82)
//
// implemented on libgpuqrengine_amd.so; it replaces SPQR/SPQRGPU/spqrgpu_buildAssemblyMaps.cpp.
// adapter/spqrgpu_kernel.cpp does not call it (the AMD driver builds the maps once per symbolic object
// and keeps them on the device); it is here for code that calls the reference's function by name.
// Compiled inside an SPQR tree with -DSUITESPARSE_HIP and -I<this repo>/include.

#ifdef SUITESPARSE_HIP

#include "spqr.hpp"
#include "spqrgpu_amd.h"

void spqrgpu_buildAssemblyMaps
(
    Long numFronts, Long n, Long *Fmap, Long *Post, Long *Super, Long *Rp, Long *Rj,
    Long *Sleft, Long *Sp, Long *Sj, double *Sx, Long *Fm, Long *Cm, Long *Childp, Long *Child,
    Long *CompleteStair, int *CompleteRjmap, Long *RjmapOffsets, int *CompleteRimap, Long *RimapOffsets,
    SEntry *cpuS
)
{
    // SEntry (include/GPUQREngine.hpp) and gpuqr_sentry are the same 16 bytes: findex, value
    spqrgpu_amd_build_assembly_maps (numFronts, n, Fmap, Post, Super, Rp, Rj, Sleft, Sp, Sj, Sx, Fm, Cm,
        Childp, Child, CompleteStair, CompleteRjmap, RjmapOffsets, CompleteRimap, RimapOffsets,
        (gpuqr_sentry *) cpuS) ;
}

#endif
