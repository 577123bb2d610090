// Drop-in translation unit for an SPQR build: the reference's entry point
//
//     void spqrgpu_kernel (spqr_blob <double> *Blob) ;                (SPQR/Include/spqrgpu.hpp:9-14)
//
// implemented on libgpuqrengine_amd.so.  It replaces SPQR/SPQRGPU/spqrgpu_kernel.cpp in the list of
// SPQR's sources; spqr_kernel (SPQR/Source/spqr_kernel.cpp:35-42) calls it unchanged.  It is written
// against SPQR's own headers and is compiled inside an SPQR tree, not in this repository (which
// holds none of SPQR's headers); the work is done by spqrgpu_amd_kernel (include/spqrgpu_amd.h),
// which this repository tests on the same arrays.
//
// Build: add -DSUITESPARSE_HIP (this file's own gate; SPQR's -DSUITESPARSE_CUDA pulls CUDA headers
// into cholmod_core.h) and -I<this repo>/include, link -lgpuqrengine_amd.
//
// What it reads and writes is what the reference's function reads and writes
// (SPQR/SPQRGPU/spqrgpu_kernel.cpp:63-782): QRsym's arrays and QRsym->QRgpu, Blob->Sx,
// cc->gpuMemorySize in; QRnum->Rblock[], QRnum->Rdead[], Work[0].Stack_head / sumfrank / maxfrank,
// cc->gpuKernelTime / gpuFlops / gpuNumKernelLaunches and cc->status out.

#ifdef SUITESPARSE_HIP

#include "spqr.hpp"
#include "spqrgpu_amd.h"
#include <string.h>

void spqrgpu_kernel (spqr_blob <double> *Blob)
{
    spqr_symbolic *QRsym = Blob->QRsym ;
    spqr_numeric <double> *QRnum = Blob->QRnum ;
    spqr_work <double> *Work = Blob->Work ;
    cholmod_common *cc = Blob->cc ;
    spqr_gpu *QRgpu = QRsym->QRgpu ;
    const Long nf = QRsym->nf ;
    const Long stack = 0 ;                      // no mixing of GPU and TBB parallelism

    // The symbolic object of the AMD driver: a copy of the arrays above that also keeps the assembly
    // maps, the S permutation and the engine's schedule between calls.  This fork's analysis made it at
    // symbolic time and keeps it in QRsym->wsMondoF (adapter/spqrgpu_computeFrontStaging.cpp; released
    // by spqr_1factor.cpp:643-655); a stock SPQR 4.5.5, whose spqr_symbolic has no such field, gets a
    // short-lived one here.
    int status = 0 ;
    spqr_amd_symbolic *sym = NULL ;
    bool own_sym = true ;
#ifdef EXCLUDE_ALLOCATION_TIME
    if (QRsym->wsMondoF != NULL && QRsym->wsMondoF->payload != NULL)
    {
        sym = (spqr_amd_symbolic *) QRsym->wsMondoF->payload ;
        own_sym = false ;
    }
#endif
    if (sym == NULL) sym = spqr_amd_symbolic_from_arrays (QRsym->m, QRsym->n, QRsym->anz, nf,
        QRsym->Sp, QRsym->Sj, QRsym->Sleft, QRsym->Super, QRsym->Rp, QRsym->Rj,
        QRsym->Parent, QRsym->Childp, QRsym->Child, QRsym->Post, QRsym->Fm, QRsym->Cm,
        QRgpu->RimapOffsets, QRgpu->RimapSize, QRgpu->RjmapOffsets, QRgpu->RjmapSize,
        QRgpu->numStages, QRgpu->Stagingp, QRgpu->StageMap, cc->gpuMemorySize, &status) ;
    if (!sym)
    {
        ERROR (status == 1 ? CHOLMOD_OUT_OF_MEMORY : CHOLMOD_INVALID, "cannot set up the GPU factorization") ;
        return ;
    }

    double *Stack_head = Work [stack].Stack_head ;
    double *Stack_top = Work [stack].Stack_top ;
    Long *Rblock_offset = (Long *) cholmod_l_malloc (MAX (nf, 1), sizeof (Long), cc) ;
    if (cc->status < CHOLMOD_OK)
    {
        if (own_sym) spqr_amd_symbolic_free (sym) ;
        return ;
    }

    Long sumfrank = 0, maxfrank = 0, used = 0, flops = 0, launches = 0 ;
    double kernel_ms = 0 ;
    int code ;
    if (cc->numGPU > 1)
    {
        // Several devices (cholmod_common's numGPU, CHOLMOD/Include/cholmod_core.h:988; the reference's
        // SPQR never reads it): the assembly tree is mapped onto devices 0 .. numGPU-1, one host thread
        // per device, contribution blocks that change devices go by peer copies; all of R comes back in
        // one numeric object, laid out like the stack, and is copied onto it.
        spqr_amd_numeric *num = NULL ;
        code = spqrgpu_amd_factorize_multi (sym, NULL, NULL, Blob->Sx, (int) cc->numGPU, NULL, &num) ;
        if (code == 0)
        {
            if (num->rsize > (Long) (Stack_top - Stack_head)) code = 1 ;
            else
            {
                if (num->rsize > 0) memcpy (Stack_head, num->Rstack, (size_t) num->rsize * sizeof (double)) ;
                for (Long f = 0 ; f < nf ; f++) Rblock_offset [f] = num->Rblock [f] ;
                for (Long k = 0 ; k < QRsym->n ; k++) if (num->Rdead [k]) QRnum->Rdead [k] = 1 ;
                sumfrank = num->sumfrank ; maxfrank = num->maxfrank ; used = num->rsize ;
                kernel_ms = num->gpuKernelTime ; flops = num->gpuFlops ; launches = num->gpuNumKernelLaunches ;
            }
            spqr_amd_numeric_free (num) ;
        }
    }
    else code = spqrgpu_amd_kernel (sym, Blob->Sx, Stack_head, (Long) (Stack_top - Stack_head),
        Rblock_offset, QRnum->Rdead, &sumfrank, &maxfrank, &used, &kernel_ms, &flops, &launches) ;
    if (code == 0)
    {
        for (Long f = 0 ; f < nf ; f++) QRnum->Rblock [f] = Stack_head + Rblock_offset [f] ;
        Work [stack].Stack_head = Stack_head + used ;
        Work [stack].Stack_top = Stack_top ;
        Work [stack].sumfrank += sumfrank ;
        Work [stack].maxfrank = MAX (Work [stack].maxfrank, maxfrank) ;
        cc->gpuKernelTime = kernel_ms ;
        cc->gpuFlops = flops ;
        cc->gpuNumKernelLaunches = launches ;
    }
    cholmod_l_free (MAX (nf, 1), sizeof (Long), Rblock_offset, cc) ;
    if (own_sym) spqr_amd_symbolic_free (sym) ;
    // (the reference ignores the engine's return code, spqrgpu_kernel.cpp:638; here a failure reaches cc)
    if (code == 1) ERROR (CHOLMOD_OUT_OF_MEMORY, "out of memory") ;
    else if (code != 0) ERROR (CHOLMOD_GPU_PROBLEM, "GPU factorization failed") ;
}

void spqrgpu_kernel (spqr_blob <Complex> *Blob)
{
    // complex case not supported on the GPU, as in the reference (spqrgpu_kernel.cpp:786-795)
    cholmod_common *cc = Blob->cc ;
    ERROR (CHOLMOD_INVALID, "complex case not yet supported on the GPU") ;
}

// void spqrgpu_computeFrontStaging (...) (SPQR/Include/spqrgpu.hpp:16-57) and
// void spqrgpu_buildAssemblyMaps (...) (spqrgpu.hpp:59-82) bind to
// spqrgpu_amd_compute_front_staging / spqrgpu_amd_build_assembly_maps the same way: same arguments in
// the same order, cc->gpuMemorySize passed by value.

#endif
