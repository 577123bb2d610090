// Device task records shared by the host schedule builder and the HIP kernels.
//
// One record describes one workgroup's worth of work.  The whole schedule of a
// GPUQREngine call is built once on the host (it depends only on front shapes,
// staircases and the assembly tree), uploaded once, and then replayed round by
// round: every task of round r depends only on tasks of rounds < r.
//
// Replaces the reference's 104-byte TaskDescriptor
// (GPUQREngine/Include/GPUQREngine_TaskDescriptor.hpp:54-103), which is rebuilt
// and re-uploaded by the host between every pair of launches.
#ifndef SUITESPARSE_AMD_QR_TASKS_H
#define SUITESPARSE_AMD_QR_TASKS_H

#include <stdint.h>

#define QR_PANEL_W        32    // panel (block reflector) width, columns
#define QR_MAX_SEG        8     // row segments gathered by one panel task
#define QR_MAX_K          256   // most rows one panel task factorizes / applies
#define QR_THREADS        256   // workgroup size of every task kernel
#define QR_VT_DOUBLES     1024  // one VT tile: 32x32 doubles (T upper, V1 strictly lower)
#define QR_SASM_CHUNK     2048  // S entries scattered by one S-assembly task
#define QR_PACK_TILE      128   // contribution-block tile edge of one pack-assembly task
#define QR_SCRATCH_BYTES  (32u << 20)   // fixed device workspace of the entry point that allocates nothing

enum QrTaskType
{
    QR_TASK_FACTORIZE    = 0,   // Householder QR of a gathered K x w panel, emits V (in place) and VT
    QR_TASK_APPLY        = 1,   // A <- A - V T' V' A on the same gathered rows, columns [c_begin,c_end)
    QR_TASK_SASSEMBLE    = 2,   // F[findex] = value for a slice of the S entries of a front
    QR_TASK_PACKASSEMBLE = 3    // child contribution block tile -> parent front through Rimap/Rjmap
};

// Panel tasks (factorize / apply): the K rows of the panel are the
// concatenation of nseg row segments of the front (local row r of the panel is
// the r-th row of that concatenation).  Local rows [0,w) form the "diagonal
// block": after factorize the front holds R there (zeros below the diagonal),
// the strictly-lower part of V lives in the VT tile, rows >= w hold V in place.
struct QrPanelTask
{
    double  *F;                 // front, row-major, leading dimension ld
    double  *VT;                // VT tile (QR_VT_DOUBLES doubles) or NULL if nobody applies it
    int32_t  ld;                // = fn
    int32_t  col0;              // first column of the panel
    int32_t  w;                 // panel width (1..32)
    int32_t  K;                 // rows gathered (1..256)
    int32_t  nseg;
    int32_t  c_begin, c_end;    // apply only: column range updated by the whole group of workgroups
    int32_t  cb_cols;           // apply only: columns per workgroup (block blk covers c_begin + blk*cb_cols ...)
    int32_t  seg_row [QR_MAX_SEG];
    int32_t  seg_n   [QR_MAX_SEG];
};

struct QrSAssembleTask
{
    double       *F;            // destination front
    const void   *S;            // packed (int64 findex, double value) pairs
    int64_t       pstart, pend; // slice of S
    int32_t       fn, ld;       // findex = i*fn + j addresses F[i*ld + j]; no conversion when ld == fn
};

struct QrPackAssembleTask
{
    const double *C;            // child contribution block, C(ci,cj) at C[ci*cld + cj]
    double       *P;            // parent front, leading dimension pn
    const int    *Rjmap;        // child C column -> parent column
    const int    *Rimap;        // child C row    -> parent row
    int32_t       cld;          // leading dimension of C (child fn, or cn for push-only fronts)
    int32_t       pn;
    int32_t       cistart, ciend, cjstart, cjend;
};

// One workgroup = one (task, block) pair; only apply tasks have more than one block.
struct QrRef
{
    int32_t task;               // index into the task array of the call
    int32_t blk;                // column block within an apply task, else 0
};

// Entry of the persistent work queue: the workgroup that draws ticket q runs refs[q] once the
// counter wait_ctr has reached `need`, then adds one to signal_ctr.
struct QrQueueRef
{
    int32_t task;
    int32_t blk;
    int32_t wait_ctr;
    int32_t need;
    int32_t signal_ctr;
};

// R hand-off: the first frank rows of a reduced front (row-major, row pitch ld) go to a
// column-packed upper trapezoid: column j holds min(j+1, frank) entries, columns one after another.
struct QrPackRTask
{
    const double *F;            // reduced front
    double       *out;          // its block of the packed stack
    int32_t       ld, fn, frank;
    int32_t       c0, c1;       // columns [c0, c1) of R are packed by this task (a front's R may go out in pieces)
    int32_t       pad;
};

// One row of a right-hand-side block on its way in, up the tree, or out.
struct QrRowCopy
{
    const double *src;
    double       *dst;
};

// Rank detection: the piece of one front column whose 2-norm decides whether the column is dead
// (rows [first, first + nrows) of a row-major front, one entry every `ld` doubles).
struct QrColNormTask
{
    const double *x;            // first entry
    int64_t       nrows;
    int64_t       ld;
};

#define QR_QUEUE_HEAD   0       // counters[0]: next ticket
#define QR_QUEUE_ERROR  1       // counters[1]: nonzero after a spin ran out (no task hangs the device)
#define QR_QUEUE_FIRST  2       // counters[2 + c]: arrivals at counter c

struct QrTask
{
    int32_t type;               // QrTaskType
    int32_t tier;               // panel tasks: log2 (row capacity / 32), 0..3
    union
    {
        QrPanelTask        panel;
        QrSAssembleTask    sasm;
        QrPackAssembleTask pack;
    } u;
};

#endif
