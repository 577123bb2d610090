// Drop-in translation unit for an SPQR build: the reference's entry point
//
//     void spqrgpu_computeFrontStaging (Long numFronts, Long *Parent, ..., cholmod_common *cc) ;
//                                                                 (SPQR/Include/spqrgpu.hpp:16-57)
//
// implemented on libgpuqrengine_amd.so.  It replaces SPQR/SPQRGPU/spqrgpu_computeFrontStaging.cpp in
// the list of SPQR's sources; spqr_analyze (SPQR/Source/spqr_analyze.cpp:1104-1111) calls it unchanged.
// Compiled inside an SPQR tree with -DSUITESPARSE_HIP and -I<this repo>/include.
//
// This fork's signature carries the symbolic object as well (EXCLUDE_ALLOCATION_TIME, set in
// SPQR/Include/spqr.hpp:11): its version allocates the engine's workspaces into QRsym here
// (spqrgpu_computeFrontStaging.cpp:251-281) so that their allocation is not timed with the
// factorization.  The AMD driver keeps its device workspaces per device between calls by itself; what
// it wants to live as long as QRsym is its own symbolic object (a copy of QRsym's arrays, plus the
// assembly maps, the S permutation and the engine's schedule once the first factorization has made
// them).  It is created here, kept in QRsym->wsMondoF, found there by spqrgpu_kernel, and released by
// the Workspace::destroy that spqr_1factor.cpp:643-655 already calls on that field.

#ifdef SUITESPARSE_HIP

#include <new>
#include "spqr.hpp"
#include "spqrgpu_amd.h"

void spqrgpu_computeFrontStaging
(
    Long numFronts, Long *Parent, Long *Childp, Long *Child,
    Long *Fm, Long *Cm, Long *Rp, Long *Sp, Long *Sleft, Long *Super, Long *Post,
    Long RimapSize, Long RjmapSize,
    bool *feasible, Long *numStages, Long *Stagingp, Long *StageMap,
    size_t *FSize, size_t *RSize, size_t *SSize,
    Long *FOffsets, Long *ROffsets, Long *SOffsets,
#ifdef EXCLUDE_ALLOCATION_TIME
    spqr_symbolic *QRsym,
#endif
    cholmod_common *cc
)
{
    *feasible = spqrgpu_amd_compute_front_staging (numFronts, Parent, Childp, Child, Fm, Cm, Rp, Sp,
        Sleft, Super, Post, RimapSize, RjmapSize, cc->gpuMemorySize, numStages, Stagingp, StageMap,
        FSize, RSize, SSize, FOffsets, ROffsets, SOffsets) != 0 ;
    if (!*feasible)
    {
        // (the caller raises CHOLMOD_GPU_PROBLEM, spqr_analyze.cpp:1118-1126)
        *numStages = 0 ;
        return ;
    }
#ifdef EXCLUDE_ALLOCATION_TIME
    Long fsize = 0, rsize = 0, ssize = 0, maxfronts = 0 ;
    for (Long stage = 0 ; stage < *numStages ; stage++)
    {
        const Long leftover = spqrgpu_amd_numfronts_in_stage (stage, Stagingp, StageMap, Post, Child, Childp) ;
        fsize = MAX (fsize, (Long) FSize [stage]) ;
        rsize = MAX (rsize, (Long) RSize [stage]) ;
        ssize = MAX (ssize, (Long) SSize [stage]) ;
        maxfronts = MAX (maxfronts, Stagingp [stage+1] - Stagingp [stage] + leftover) ;
    }
    QRsym->wsMondoF_size = fsize ;
    QRsym->wsMondoR_size = rsize ;
    QRsym->wsS_size = ssize ;
    QRsym->maxfronts_in_stage = maxfronts ;
    QRsym->fronts [0] = QRsym->fronts [1] = NULL ;
    QRsym->wsMondoF = QRsym->wsMondoR [0] = QRsym->wsMondoR [1] = NULL ;
    QRsym->wsMondoS = QRsym->wsRimap = QRsym->wsRjmap = QRsym->wsS = NULL ;
    spqr_gpu *QRgpu = QRsym->QRgpu ;
    if (QRgpu != NULL)
    {
        // (QRsym's own arrays are complete by now, spqr_analyze.cpp:1104; the staging is what was just computed)
        int status = 0 ;
        spqr_amd_symbolic *sym = spqr_amd_symbolic_from_arrays (QRsym->m, QRsym->n, QRsym->anz, numFronts,
            QRsym->Sp, QRsym->Sj, QRsym->Sleft, QRsym->Super, QRsym->Rp, QRsym->Rj,
            QRsym->Parent, QRsym->Childp, QRsym->Child, QRsym->Post, QRsym->Fm, QRsym->Cm,
            QRgpu->RimapOffsets, RimapSize, QRgpu->RjmapOffsets, RjmapSize,
            *numStages, Stagingp, StageMap, cc->gpuMemorySize, &status) ;
        if (sym != NULL)
        {
            QRsym->wsMondoF = new (std::nothrow) Workspace (sym, [] (void *p) { spqr_amd_symbolic_free ((spqr_amd_symbolic *) p) ; }) ;
            if (QRsym->wsMondoF == NULL) spqr_amd_symbolic_free (sym) ;
        }
        // (without it spqrgpu_kernel makes a short-lived one per call)
    }
#endif
}

#endif
