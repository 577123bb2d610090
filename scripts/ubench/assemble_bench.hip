// Isolated timing of the assembly kernel (k_assemble) against the device's copy rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I suitesparse_amd/csrc scripts/ubench/assemble_bench.hip -o scripts/ubench/assemble_bench
// Three cases, each one launch, inputs resident:
//   pack  N ROWS [runs|scatter]  pack-assembly of one N x N upper-triangular contribution block into a (2N) x (2N) parent
//                  through monotone row and column maps (every second parent row / column), cut into tasks
//                  of ROWS rows x 128 columns as the planner cuts them (ROWS = 128: a whole front handed over
//                  at once; ROWS = 32: the row-block hand-over)
//   sasm  NNZ      S-assembly of NNZ entries (16-byte index/value pairs) into zeroed fronts 4096 columns
//                  wide, 7 entries per row in ascending columns, tasks of 2048 entries
//   copy  BYTES    hipMemcpyAsync device to device of the same number of bytes the pack case moves
// Bytes are the algorithmic ones: 16 per entry of the block + 4 per map int; 24 per S entry.
#include "qr_kernels.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static float time_launch (int kid, QrTask *d_tasks, QrRef *d_refs, int nrefs, int reps)
{
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
    float best = 1e30f ;
    for (int r = 0 ; r < reps + 1 ; r++)
    {
        hipEventRecord (e0) ;
        hipError_t e = qr_launch_tasks (kid, d_tasks, d_refs, nrefs, 0) ;
        hipEventRecord (e1) ; hipEventSynchronize (e1) ;
        if (e != hipSuccess || hipGetLastError ( ) != hipSuccess) { printf ("launch failed\n") ; exit (1) ; }
        float ms ; hipEventElapsedTime (&ms, e0, e1) ;
        if (r > 0 && ms < best) best = ms ;
    }
    return best ;
}

static void upload (std::vector<QrTask> &tasks, QrTask **d_tasks, QrRef **d_refs)
{
    std::vector<QrRef> refs (tasks.size ( )) ;
    for (size_t t = 0 ; t < tasks.size ( ) ; t++) { refs [t].task = (int) t ; refs [t].blk = 0 ; }
    hipMalloc (d_tasks, tasks.size ( ) * sizeof (QrTask)) ;
    hipMalloc (d_refs, refs.size ( ) * sizeof (QrRef)) ;
    hipMemcpy (*d_tasks, tasks.data ( ), tasks.size ( ) * sizeof (QrTask), hipMemcpyHostToDevice) ;
    hipMemcpy (*d_refs, refs.data ( ), refs.size ( ) * sizeof (QrRef), hipMemcpyHostToDevice) ;
}

int main (int argc, char **argv)
{
    const char *what = argc > 1 ? argv [1] : "pack" ;
    if (qr_kernels_init ( ) != hipSuccess) { printf ("init failed\n") ; return 1 ; }
    if (!strcmp (what, "pack"))
    {
        const int N = argc > 2 ? atoi (argv [2]) : 16384 ;
        const int rows = argc > 3 ? atoi (argv [3]) : 128 ;
        const int cld = N + 32 ;                // the block sits inside its child's front
        const int pn = 2 * N ;
        double *C, *P ; int *Rimap, *Rjmap ;
        hipMalloc (&C, (size_t) N * cld * 8) ; hipMemset (C, 0, (size_t) N * cld * 8) ;
        hipMalloc (&P, (size_t) 2 * N * pn * 8) ; hipMemset (P, 0, (size_t) 2 * N * pn * 8) ;
        std::vector<int> map (N) ;
        // "runs" (default): the maps are runs of 48 consecutive indices with gaps of 16, what a child whose
        // columns sit between a parent's own looks like; "scatter": every map entry skips one, no two
        // neighbours in the child are neighbours in the parent (the worst case for the stores)
        const bool scatter = argc > 4 && !strcmp (argv [4], "scatter") ;
        for (int i = 0 ; i < N ; i++) map [i] = scatter ? 2 * i + (i % 3 == 0) : i + (i / 48) * 16 ;
        hipMalloc (&Rimap, N * 4) ; hipMalloc (&Rjmap, N * 4) ;
        hipMemcpy (Rimap, map.data ( ), N * 4, hipMemcpyHostToDevice) ;
        hipMemcpy (Rjmap, map.data ( ), N * 4, hipMemcpyHostToDevice) ;
        std::vector<QrTask> tasks ;
        double bytes = 0 ;
        for (int ci = 0 ; ci < N ; ci += rows)
            for (int cj = (ci / QR_PACK_TILE) * QR_PACK_TILE ; cj < N ; cj += QR_PACK_TILE)
            {
                QrTask q = QrTask ( ) ;
                q.type = QR_TASK_PACKASSEMBLE ;
                QrPackAssembleTask &pk = q.u.pack ;
                pk.C = C ; pk.P = P ; pk.Rjmap = Rjmap ; pk.Rimap = Rimap ; pk.cld = cld ; pk.pn = pn ;
                pk.cistart = ci ; pk.ciend = ci + rows < N ? ci + rows : N ;
                pk.cjstart = cj ; pk.cjend = cj + QR_PACK_TILE < N ? cj + QR_PACK_TILE : N ;
                for (int i = pk.cistart ; i < pk.ciend ; i++)
                    bytes += 16.0 * (double) (pk.cjend - (pk.cjstart > i ? pk.cjstart : i) > 0 ? pk.cjend - (pk.cjstart > i ? pk.cjstart : i) : 0) ;
                bytes += 4.0 * ((pk.ciend - pk.cistart) + (pk.cjend - pk.cjstart)) ;
                tasks.push_back (q) ;
            }
        QrTask *d_tasks ; QrRef *d_refs ;
        upload (tasks, &d_tasks, &d_refs) ;
        const float ms = time_launch (8, d_tasks, d_refs, (int) tasks.size ( ), 5) ;
        printf ("pack-assembly (%s maps) %d x %d upper triangle, tasks of %d rows x 128 cols: %d workgroups  %.3f ms  %.1f MB  %.2f TB/s = %.3f of 8 TB/s\n",
            scatter ? "scattered" : "runs in the", N, N, rows, (int) tasks.size ( ), ms, bytes / 1e6, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0) ;
        // the same bytes through the runtime's device-to-device copy (half read, half written)
        hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
        const size_t half = (size_t) (bytes / 2) & ~(size_t) 255 ;
        float best = 1e30f ;
        for (int r = 0 ; r < 4 ; r++)
        {
            hipEventRecord (e0) ;
            hipMemcpyAsync (P, P + (size_t) N * pn, half, hipMemcpyDeviceToDevice, 0) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float t ; hipEventElapsedTime (&t, e0, e1) ;
            if (r > 0 && t < best) best = t ;
        }
        printf ("device-to-device copy of the same bytes: %.3f ms  %.2f TB/s\n", best, 2.0 * half / best / 1e9) ;
    }
    else if (!strcmp (what, "sasm"))
    {
        const long long nnz = argc > 2 ? atoll (argv [2]) : 16678912LL ;
        const int fn = 4096, per_row = 7 ;
        const long long rowsF = (nnz + per_row - 1) / per_row ;
        double *F ; hipMalloc (&F, (size_t) rowsF * fn * 8) ; hipMemset (F, 0, (size_t) rowsF * fn * 8) ;
        struct SE { long long findex ; double value ; } ;
        std::vector<SE> S ((size_t) nnz) ;
        for (long long k = 0 ; k < nnz ; k++)
        {
            const long long i = k / per_row, j = ((i * 37) % (fn - 600)) + (k % per_row) * 85 ;
            S [k].findex = i * fn + j ; S [k].value = 1.0 + (double) (k % 5) ;
        }
        SE *dS ; hipMalloc (&dS, (size_t) nnz * sizeof (SE)) ;
        hipMemcpy (dS, S.data ( ), (size_t) nnz * sizeof (SE), hipMemcpyHostToDevice) ;
        std::vector<QrTask> tasks ;
        for (long long p = 0 ; p < nnz ; p += 2048)
        {
            QrTask q = QrTask ( ) ;
            q.type = QR_TASK_SASSEMBLE ;
            q.u.sasm.F = F ; q.u.sasm.S = dS ; q.u.sasm.pstart = p ; q.u.sasm.pend = p + 2048 < nnz ? p + 2048 : nnz ;
            q.u.sasm.fn = fn ; q.u.sasm.ld = fn ;
            tasks.push_back (q) ;
        }
        QrTask *d_tasks ; QrRef *d_refs ;
        upload (tasks, &d_tasks, &d_refs) ;
        const float ms = time_launch (8, d_tasks, d_refs, (int) tasks.size ( ), 5) ;
        const double bytes = 24.0 * (double) nnz ;
        printf ("S-assembly of %lld entries into zeroed fronts (%d columns, %d entries a row): %d workgroups  %.3f ms  %.1f MB  %.2f TB/s = %.3f of 8 TB/s"
            "  (every 8-byte store dirties a 128-byte line: %.1f MB of lines written)\n",
            nnz, fn, per_row, (int) tasks.size ( ), ms, bytes / 1e6, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0, 128.0 * nnz / 1e6) ;
    }
    return 0 ;
}
