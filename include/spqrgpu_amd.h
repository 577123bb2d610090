/* C ABI of the SPQR <-> engine glue on MI355X: staging, assembly maps, the
 * per-stage numeric driver, and the small symbolic front end that lets the
 * library run stand-alone on a box without SuiteSparse.
 *
 * Int = int64_t everywhere (SuiteSparse_long, SPQR's Long).
 */
#ifndef SPQRGPU_AMD_H
#define SPQRGPU_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "gpuqrengine_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces void spqrgpu_computeFrontStaging (...)
 * (SPQR/Include/spqrgpu.hpp:16-57, SPQR/SPQRGPU/spqrgpu_computeFrontStaging.cpp:31-249).
 * Same arguments in the same order, with cc->gpuMemorySize passed by value
 * and the return value replacing *feasible (1 feasible, 0 not; on 0
 * *numStages = 0 and the caller raises CHOLMOD_GPU_PROBLEM).  The arithmetic
 * is bit-for-bit the reference's, including sizeof(TaskDescriptor) = 104. */
int spqrgpu_amd_compute_front_staging
(
    int64_t numFronts, const int64_t *Parent, const int64_t *Childp, const int64_t *Child,
    const int64_t *Fm, const int64_t *Cm, const int64_t *Rp, const int64_t *Sp, const int64_t *Sleft,
    const int64_t *Super, const int64_t *Post, int64_t RimapSize, int64_t RjmapSize,
    size_t gpuMemorySize,
    int64_t *numStages, int64_t *Stagingp, int64_t *StageMap,
    size_t *FSize, size_t *RSize, size_t *SSize,
    int64_t *FOffsets, int64_t *ROffsets, int64_t *SOffsets
);

/* Replaces void spqrgpu_buildAssemblyMaps (...)
 * (SPQR/Include/spqrgpu.hpp:59-82, SPQR/SPQRGPU/spqrgpu_buildAssemblyMaps.cpp:8-190). */
void spqrgpu_amd_build_assembly_maps
(
    int64_t numFronts, int64_t n, int64_t *Fmap, const int64_t *Post, const int64_t *Super,
    const int64_t *Rp, const int64_t *Rj, const int64_t *Sleft, const int64_t *Sp, const int64_t *Sj,
    const double *Sx, const int64_t *Fm, const int64_t *Cm, const int64_t *Childp, const int64_t *Child,
    int64_t *CompleteStair, int32_t *CompleteRjmap, const int64_t *RjmapOffsets,
    int32_t *CompleteRimap, const int64_t *RimapOffsets, gpuqr_sentry *cpuS
);

/* Replaces numfronts_in_stage (SPQR/SPQRGPU/spqrgpu_kernel.cpp:26-56): number of
 * children of this stage's fronts that were factorized in an earlier stage. */
int64_t spqrgpu_amd_numfronts_in_stage
(
    int64_t stage, const int64_t *Stagingp, const int64_t *StageMap, const int64_t *Post,
    const int64_t *Child, const int64_t *Childp
);

/* --- symbolic object ------------------------------------------------------- */

/* Opaque mirror of the parts of spqr_symbolic + spqr_gpu the GPU path reads
 * (SPQR/Include/SuiteSparseQR.hpp:118-260). */
typedef struct spqr_amd_symbolic spqr_amd_symbolic;

/* Stand-alone symbolic analysis of A(:,Q) for the GPU path: column elimination
 * tree, postorder, fundamental supernodes (optionally relaxed), front patterns
 * Rp/Rj, S = A(P,Q) in row form with front-local column indices, Fm, Cm, map
 * offsets, canonical flop count, and the staging for gpuMemorySize.  It stands
 * in for spqr_analyze (SPQR/Source/spqr_analyze.cpp:75-1130) on a machine
 * without CHOLMOD; an SPQR build passes its own QRsym arrays to the functions
 * above instead.  A is m-by-n in compressed-column form, Quser a column
 * permutation or NULL.  relax < 0: the arrays SPQR itself would build for the
 * column order Quser (spqr_analyze with SPQR_ORDERING_GIVEN, the call
 * spqr_1factor.cpp:624-641 makes once spqr_1colamd has ordered the columns): no
 * further postordering, CHOLMOD's relaxed supernodes with its default parameters
 * (CHOLMOD/Supernodal/cholmod_super_symbolic.c:676-790), the frontal tree
 * postordered by front width (spqr_analyze.cpp:561-611): Super, Rp, Rj, Parent,
 * Post, Childp, Child, Sp, Sj, Sleft, PLinv, Fm, Cm and the flop count are then
 * those of an SPQR run, entry for entry.  relax >= 0: the columns are postordered
 * here, fundamental supernodes, merged into their parent when that adds at most
 * relax percent of explicit zeros.  *status: 0 ok, 1 out of memory, 2 staging
 * infeasible, 3 invalid input (Quser not a permutation included). */
spqr_amd_symbolic *spqr_amd_analyze
(
    int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *Quser,
    int relax, size_t gpuMemorySize, int *status
);

/* The same analysis with SPQR's do_rank_detection switch (SPQR/Source/spqr_analyze.cpp:75-125,826-842):
 * when it is on, Cm[f] = min (fm, cn) and Fm is sized for it, so that a front whose pivot columns
 * die at a tolerance has room to pass the freed rows up.  spqrgpu_amd_factorize_tol needs an object
 * analysed this way. */
spqr_amd_symbolic *spqr_amd_analyze_rd
(
    int64_t m, int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *Quser,
    int relax, size_t gpuMemorySize, int do_rank_detection, int *status
);

void spqr_amd_symbolic_free (spqr_amd_symbolic *sym);

/* The symbolic object of an SPQR build: the arrays spqrgpu_kernel reads out of QRsym and
 * QRsym->QRgpu (SPQR/SPQRGPU/spqrgpu_kernel.cpp:85-143, SPQR/Include/SuiteSparseQR.hpp:21-38,60-196),
 * copied.  m, n, anz: S = A(P,Q) is m-by-n with anz entries; Sp (m+1), Sj (anz, front-local column
 * indices as spqr_analyze leaves them for the GPU path, SPQR/Source/spqr_analyze.cpp:858-882),
 * Sleft (n+2), Super (nf+1), Rp (nf+1), Rj (Rp[nf]), Parent (nf+1), Childp (nf+2), Child (nf+1),
 * Post (nf), Fm (nf), Cm (nf), RimapOffsets / RjmapOffsets (nf) with their sizes, the staging
 * (numStages, Stagingp (numStages+1), StageMap (nf)) and cc->gpuMemorySize.  An object made this way
 * is factorized with spqrgpu_amd_kernel (values as Sx); it has no Qfill / PLinv, so the entry
 * points that take A's compressed columns refuse it.  *status: 0 ok, 1 out of memory, 3 invalid
 * input. */
spqr_amd_symbolic *spqr_amd_symbolic_from_arrays
(
    int64_t m, int64_t n, int64_t anz, int64_t nf,
    const int64_t *Sp, const int64_t *Sj, const int64_t *Sleft,
    const int64_t *Super, const int64_t *Rp, const int64_t *Rj,
    const int64_t *Parent, const int64_t *Childp, const int64_t *Child, const int64_t *Post,
    const int64_t *Fm, const int64_t *Cm,
    const int64_t *RimapOffsets, int64_t RimapSize, const int64_t *RjmapOffsets, int64_t RjmapSize,
    int64_t numStages, const int64_t *Stagingp, const int64_t *StageMap,
    size_t gpuMemorySize, int *status
);

/* Named access: arrays "Qfill" "PLinv" "Sleft" "Sp" "Sj" "Super" "Rp" "Rj" "Parent" "Childp" "Child"
 * "Post" "Fm" "Cm" "RjmapOffsets" "RimapOffsets" "Stagingp" "StageMap" "FSize" "RSize" "SSize"
 * "FOffsets" "ROffsets" "SOffsets" "ColCount"; returns NULL for an unknown name. */
const int64_t *spqr_amd_symbolic_array (const spqr_amd_symbolic *sym, const char *name, int64_t *len);
/* scalars "m" "n" "anz" "nf" "maxfn" "rjsize" "RimapSize" "RjmapSize" "numStages" "flopcount" "rnz";
 * returns -1 for an unknown name. */
double spqr_amd_symbolic_scalar (const spqr_amd_symbolic *sym, const char *name);

/* --- numeric driver -------------------------------------------------------- */

typedef struct spqr_amd_numeric
{
    int64_t  nf;
    int64_t  rsize;         /* doubles in Rstack */
    double  *Rstack;        /* all R blocks, column-packed upper trapezoidal, as on SPQR's stack; pinned
                               host memory owned by the object: release with spqr_amd_numeric_free only */
    int64_t *Rblock;        /* size nf: offset of front f's block in Rstack */
    char    *Rdead;         /* size n: 1 for pivot columns past the bottom of their front */
    int64_t  sumfrank, maxfrank;
    double   gpuKernelTime; /* ms */
    int64_t  gpuFlops, gpuNumKernelLaunches;
    int64_t  numStagesRun;
    double   assembleBytes; /* algorithmic bytes moved by S- and pack-assembly */
    double  *Cstack;        /* subset driver: contribution blocks leaving this process (compact cm-by-cn) */
    int64_t *Cblock;        /* subset driver: size nf, offset of front f's block in Cstack or -1 */
    int64_t  RstackPinned;  /* 1: Rstack is pinned host memory; 0: the pinned allocation was refused and it is pageable */
    int64_t  csize;         /* subset driver: doubles of contribution blocks that left (in Cstack or the caller's device buffer) */
    int64_t  RstackBorrowed;/* 1: Rstack is the caller's memory (spqrgpu_amd_kernel) and is not freed with the object */
    int64_t  numPasses;     /* spqrgpu_amd_factorize_tol: factorization passes it took (1: no dead column found) */
    double   norm_E_fro;    /* spqrgpu_amd_factorize_tol: 2-norm of the norms of the dead columns, what the CPU path
                               leaves in QRnum->norm_E_fro and cc->SPQR_norm_E_fro (SPQR/Source/spqr_factorize.cpp:546-577) */
} spqr_amd_numeric;

/* Replaces void spqrgpu_kernel (spqr_blob <double> *Blob)
 * (SPQR/Include/spqrgpu.hpp:9-14, SPQR/SPQRGPU/spqrgpu_kernel.cpp:63-782): stage
 * loop, Front construction, limbo contribution blocks, engine call, R packed
 * onto a stack.  Ax are the values of A in the same compressed-column order
 * given to spqr_amd_analyze.  Returns 0, or 1 out of memory, 2 GPU problem. */
int spqrgpu_amd_factorize (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                           const double *Ax, spqr_amd_numeric **num_out);

void spqr_amd_numeric_free (spqr_amd_numeric *num);

/* Numeric factorization with SPQR's rank detection at tolerance tol on the device: what the
 * reference's CPU path does in spqr_front (SPQR/Source/spqr_front.cpp:424-479: a pivot column whose
 * remaining norm is <= tol is dead, Rdead[k] = 1, and does not use up a row, so R comes out squeezed
 * with `rank` rows) and what its GPU path declines to do (SPQR/Source/spqr_analyze.cpp:106-125 turns
 * the GPU off when tol >= 0).  sym must come from spqr_amd_analyze_rd with do_rank_detection = 1.
 * On return num->Rdead flags the dead pivot columns, num->sumfrank is the numerical rank, and the R
 * blocks are squeezed and column-packed as SPQR's (spqr_amd_rconvert reads them).  tol < 0 means no
 * rank detection (SPQR_NO_TOL); spqr_amd_default_tol gives SPQR_DEFAULT_TOL
 * (SPQR/Source/spqr_tol.cpp:14-27).  Returns 0, 1 out of memory, 2 GPU problem, 3 invalid argument. */
int spqrgpu_amd_factorize_tol (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                               const double *Ax, double tol, spqr_amd_numeric **num_out);
double spqr_amd_default_tol (int64_t m, int64_t n, const int64_t *Ap, const double *Ax);

/* --- H kept on the device: Q'B and solves after the factorization ---------------------------------
 *
 * The reference gets Q'B on its GPU path only by appending B to A before the analysis
 * (SPQR/Source/SuiteSparseQR.cpp:256-260); anything that needs H later (SuiteSparseQR_qmult,
 * SuiteSparseQR_solve on a factorization object, SPQR/Source/SuiteSparseQR_expert.cpp:324-343) forces
 * keepH and with it the CPU.  spqrgpu_amd_factorize_keep is spqrgpu_amd_factorize that leaves H where
 * the factorization made it: the fronts (Householder vectors in place) stay in a device arena the
 * returned object owns, with a block of nrhs (1..256) right-hand-side columns behind every front, and
 * the engine keeps the VT tiles and the update tasks of those blocks.  spqrgpu_amd_qtmult then
 * computes, for any B (m-by-nrhs, column-major, leading dimension ldb), the first rank rows of Q'B in
 * the row order of R (C, rank-by-nrhs, leading dimension ldc; rank = spqr_amd_qt_rank): exactly what
 * R x = Q'B needs.  One factorization, any number of right-hand sides.  Single-stage factorizations
 * only (the fronts must all be resident); the symbolic object must not be a rank-detection one.
 * Return codes as spqrgpu_amd_factorize. */
typedef struct spqr_amd_qt spqr_amd_qt;
int spqrgpu_amd_factorize_keep (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                                const double *Ax, int64_t nrhs,
                                spqr_amd_numeric **num_out, spqr_amd_qt **qt_out);
int spqrgpu_amd_qtmult (spqr_amd_qt *qt, const double *B, int64_t ldb, double *C, int64_t ldc);

/* X = E (R \ C) by back substitution on the packed R blocks of num, on the host: the multifrontal part
 * of spqr_rsolve (SPQR/Source/spqr_rsolve.cpp:136-338, keepH = 0, no singletons).  C is rank-by-nrhs
 * (leading dimension ldc >= rank) in the row order of R, e.g. from spqrgpu_amd_qtmult, or the rows
 * 0..rank-1 of the appended columns of R when B was appended to A (SPQR/Source/SuiteSparseQR.cpp:256-260);
 * X is nacols-by-nrhs, leading dimension nacols; nacols <= n is the number of columns of A proper, later
 * columns of the factorization are appended right-hand sides and are passed over.  Qfill (size >= nacols,
 * "Qfill" of spqr_amd_symbolic_array) scatters the solution back to A's column order; NULL gives
 * X = R \ C.  Dead pivot columns get zeros (the basic solution).  Returns 0, 1 out of memory,
 * 3 invalid argument (including a numeric object that holds only a part of R). */
int spqr_amd_rsolve (const spqr_amd_symbolic *sym, const spqr_amd_numeric *num, const int64_t *Qfill,
                     int64_t nacols, int64_t nrhs, const double *C, int64_t ldc, double *X);

/* X = R' \ B by forward substitution on the packed R blocks of num, on the host: the multifrontal part
 * of spqr_private_rtsolve (SPQR/Source/SuiteSparseQR_expert.cpp:367-470; SuiteSparseQR_solve with
 * SPQR_RTX_EQUALS_B, SuiteSparseQR_definitions.h:40).  B is n-by-nrhs in the factorization's column
 * order (ldb >= n), X rank-by-nrhs in the row order of R (ldx >= rank).  With spqr_amd_rsolve it solves
 * the seminormal equations R'R x = A'b.  Returns 0, 1 out of memory, 3 invalid argument. */
int spqr_amd_rtsolve (const spqr_amd_symbolic *sym, const spqr_amd_numeric *num, int64_t nrhs,
                      const double *B, int64_t ldb, double *X, int64_t ldx);

/* out = R in (transposed == 0: in has n entries in the factorization's column order, out rank entries) or
 * out = R' in (transposed != 0: in rank entries, out n entries), on the packed R blocks of num, on the
 * host.  No counterpart in the reference's library; its test suite forms R'R - (AP)'(AP) explicitly
 * (check_r_factor, SPQR/Tcov/qrtest.cpp:1218-1257), these two products make the same comparison
 * through a vector at sizes where that product cannot be formed.  Returns 0, 1 out of memory,
 * 3 invalid argument (including a numeric object that holds only a part of R). */
int spqr_amd_rmult (const spqr_amd_symbolic *sym, const spqr_amd_numeric *num, int transposed,
                    const double *in, double *out);

/* min || A x - B ||_2 from a kept factorization: spqrgpu_amd_qtmult on the device, then spqr_amd_rsolve
 * with the symbolic object's Qfill: what SuiteSparseQR_qmult (SPQR_QTX) followed by SuiteSparseQR_solve
 * (SPQR_RETX_EQUALS_B) do on the reference's factorization object, on the CPU
 * (SPQR/Source/SuiteSparseQR_expert.cpp:324-343).  B is m-by-nrhs (leading dimension ldb) with the nrhs
 * given to spqrgpu_amd_factorize_keep, X n-by-nrhs (leading dimension n); sym must be the object the
 * factorization was made with, from spqr_amd_analyze (it holds the column permutation). */
int spqrgpu_amd_solve (spqr_amd_qt *qt, const spqr_amd_symbolic *sym, const spqr_amd_numeric *num,
                       const double *B, int64_t ldb, double *X);
int64_t spqr_amd_qt_rank (const spqr_amd_qt *qt);
void spqr_amd_qt_free (spqr_amd_qt *qt);

/* The body of void spqrgpu_kernel (spqr_blob <double> *Blob) on plain arrays
 * (SPQR/Include/spqrgpu.hpp:9-14, SPQR/SPQRGPU/spqrgpu_kernel.cpp:63-782; adapter/spqrgpu_kernel.cpp
 * is the translation unit with the reference's signature that calls it).  Sx = Blob->Sx, the values
 * of S in row form.  The R blocks are written column-packed onto the caller's stack from Stack_head
 * on (Work[0].Stack_head; stack_room doubles are available), front after front in postorder, stage
 * after stage, as the reference lays them (spqrgpu_kernel.cpp:399-440); Rblock_offset[f] (size nf) is
 * where front f's block starts relative to Stack_head, *stack_used how far the head moved.
 * Rdead (size n) gets 1 for pivot columns past the bottom of their front (spqrgpu_kernel.cpp:753-771);
 * sumfrank / maxfrank as Work[0]'s.  The three statistics feed cc->gpuKernelTime / gpuFlops /
 * gpuNumKernelLaunches (spqrgpu_kernel.cpp:637-641).  Returns 0, 1 (out of memory: CHOLMOD_OUT_OF_MEMORY),
 * 2 (CHOLMOD_GPU_PROBLEM) or 3 (invalid argument). */
int spqrgpu_amd_kernel (const spqr_amd_symbolic *sym, const double *Sx,
                        double *Stack_head, int64_t stack_room,
                        int64_t *Rblock_offset, char *Rdead,
                        int64_t *sumfrank, int64_t *maxfrank, int64_t *stack_used,
                        double *gpuKernelTime, int64_t *gpuFlops, int64_t *gpuNumKernelLaunches);

/* The numeric driver keeps its device workspaces (front arena, S, packed R) per device between
 * calls, grow-only, as the reference keeps its Workspace objects alive across stages
 * (SPQR/SPQRGPU/spqrgpu_kernel.cpp:221-330 allocates them once per factorization).  This frees them. */
void spqrgpu_amd_finalize (void);

/* R as an n-column compressed-column matrix (rows = sumfrank), the job of
 * spqr_rcount + spqr_rconvert (SPQR/Source/spqr_rconvert.cpp).  First call
 * with Rp only (size n+1) to get the counts, then with Ri/Rx of size Rp[n]. */
int spqr_amd_rconvert (const spqr_amd_symbolic *sym, const spqr_amd_numeric *num,
                       int64_t *Rp, int64_t *Ri, double *Rx);

/* The same conversion on raw arrays, for R blocks gathered from several processes:
 * Rblock[f] < 0 marks a front whose block is absent (it still consumes its rows). */
int spqr_amd_rconvert_blocks (const spqr_amd_symbolic *sym, const double *Rstack, const int64_t *Rblock,
                              const char *Rdead, int64_t *Rp, int64_t *Ri, double *Rx);

/* One process's share of the assembly tree (no counterpart in the single-GPU reference; it
 * generalises its stage / limbo hand-off, spqrgpu_kernel.cpp:556-611,687-724, from "a later stage"
 * to "another GPU").  Fronts with mine[f] != 0 are factorized in one stage on the current device.
 * limbo_in[c] (host, compact cm-by-cn, may be NULL where unused) supplies the contribution block
 * of every child c of those fronts that another process factorized; fronts whose parent belongs
 * to another process leave their block in num->Cstack at num->Cblock[f].  Rblock[f] = -1 for
 * fronts not factorized here. */
int spqrgpu_amd_factorize_subset (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                                  const double *Ax, const char *mine, const double *const *limbo_in,
                                  spqr_amd_numeric **num_out);

/* The same with the contribution blocks on the device on both sides, for the exchange between GPUs
 * (RCCL send/recv or peer copies straight on these buffers; no counterpart in the reference, whose
 * limbo blocks cross host memory, spqrgpu_kernel.cpp:598-603,711-723).  limbo_in_dev[c] is a device
 * address on the current device and the block is assembled from where it lies; the blocks that leave
 * are written to c_out_dev, which has room for spqrgpu_amd_subset_csize (sym, mine) doubles, at
 * num->Cblock[f] (compact cm-by-cn, row-major; entries below the diagonal are unspecified and never
 * read).  num->Cstack stays NULL. */
int spqrgpu_amd_factorize_subset_dev (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                                      const double *Ax, const char *mine,
                                      const double *const *limbo_in_dev, double *c_out_dev,
                                      spqr_amd_numeric **num_out);
int64_t spqrgpu_amd_subset_csize (const spqr_amd_symbolic *sym, const char *mine);

/* --- one process, several GPUs ---------------------------------------------------------------------
 *
 * No counterpart in the reference (its QR is single-GPU; the hook an SPQR build would use is cc->numGPU,
 * CHOLMOD/Include/cholmod_core.h:988, and its multi-GPU Cholesky drives one host thread per device,
 * CHOLMOD/GPU/t_initialize_subtree.c:671-678).
 *
 * spqr_amd_map_tree maps the assembly tree onto nparts devices proportionally to the flops: owner[f] =
 * the part that factorizes front f (a front whose subtree spans several parts goes to the first of
 * them), level[f] = the number of owner changes below f; the fronts (part, level) form one call of the
 * subset driver, which needs blocks from calls of lower levels only.  Optional outputs: the share of the
 * flops in fronts whose subtree spans several parts, and the speed-up bound total / critical path that
 * the mapping allows.  Returns 0, 1 out of memory, 3 invalid argument.
 *
 * spqrgpu_amd_factorize_multi runs that schedule from one process: one host thread per device, one
 * spqrgpu_amd_factorize_subset_dev call per (part, level), contribution blocks whose parent lives on
 * another device copied device to device (hipMemcpyPeer), everything else staying where it is.
 * devices[k] = HIP device of part k (NULL: 0 .. ndev-1); parts may share a device (they then take turns
 * on it).  The result is one numeric object holding all of R on a pageable host stack, as
 * spqrgpu_amd_factorize leaves it; R is the single-device R bit for bit.  sym must have been analysed
 * without rank detection.  Ap == NULL (here and in the two subset entry points): Ax holds the values of
 * S = A(P,Q) in row form instead (SPQR's Blob->Sx, as spqrgpu_amd_kernel takes them), which is what an
 * SPQR adapter passes for cc->numGPU > 1 (adapter/spqrgpu_kernel.cpp).  Return codes as
 * spqrgpu_amd_factorize. */
int spqr_amd_map_tree (const spqr_amd_symbolic *sym, int nparts, int64_t *owner, int64_t *level,
                       double *top_flops_share, double *speedup_bound);
int spqrgpu_amd_factorize_multi (const spqr_amd_symbolic *sym, const int64_t *Ap, const int64_t *Ai,
                                 const double *Ax, int ndev, const int *devices, spqr_amd_numeric **num_out);

#ifdef __cplusplus
}
#endif
#endif
