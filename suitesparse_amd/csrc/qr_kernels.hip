// HIP kernels (gfx950 / CDNA4) of the multifrontal QR engine.
//
// Every kernel runs one workgroup per task; all tasks of one launch belong to
// the same round of the schedule.  Task kinds and what they replace in the
// reference:
//
//   factorize_panel  panel Householder QR of up to 256 gathered rows x 32 cols
//                    (GPUQREngine/Include/Kernel/Factorize/factorize_vt.cu:14-529,
//                     which handles at most 96 rows as three 32-row tiles)
//   apply_block      A <- A - V T' V' A on FP64 MFMA (v_mfma_f64_16x16x4_f64)
//                    (GPUQREngine/Include/Kernel/Apply/block_apply.cu:270-550,
//                     block_apply_chunk.cu:36-338)
//   sassemble        F[findex] = value (Kernel/Assemble/sAssemble.cu:5-19)
//   packassemble     child C -> parent front (Kernel/Assemble/packAssemble.cu:5-57)
//
// Fronts are row-major FP64, F(i,j) = F[i*fn + j], as in the reference
// (factorize_vt.cu:101).  Wavefronts are 64 lanes; every workgroup has 256
// threads = 4 waves, one per SIMD of a CU.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "qr_tasks.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

// Pointers read out of a task descriptor are generic to the compiler, and every access through
// one becomes a flat_ instruction that also counts against the LDS counter, so LDS waits would
// stall on outstanding HBM traffic.  All front, VT, map and S pointers are device memory: naming
// the global address space gives global_load / global_store, tracked by vmcnt alone.
typedef __attribute__((address_space(1))) double gdouble ;
typedef __attribute__((address_space(1))) const double gcdouble ;
typedef __attribute__((address_space(1))) const int gcint ;

#define HOUSE_EPS 1e-90     // same "column is already zero" threshold as factorize_vt.cu:12

// -----------------------------------------------------------------------------
// local row -> front row map of a panel task (in LDS)
// -----------------------------------------------------------------------------

// Entries past the task's K rows get `fill`.
__device__ __forceinline__ void build_rowmap (const QrPanelTask &t, int *rowmap, int cap, int fill = -1)
{
    for (int r = threadIdx.x ; r < cap ; r += QR_THREADS)
    {
        int g = fill, acc = 0 ;
        for (int s = 0 ; s < t.nseg ; s++)
        {
            int n = t.seg_n [s] ;
            if (r >= acc && r < acc + n) g = t.seg_row [s] + (r - acc) ;
            acc += n ;
        }
        rowmap [r] = g ;
    }
}

// -----------------------------------------------------------------------------
// 16-byte accesses to a row-major front, one column per lane afterwards
// -----------------------------------------------------------------------------

typedef double d2 __attribute__((ext_vector_type(2))) ;
typedef __attribute__((address_space(1))) d2 __attribute__((aligned(8))) gd2 ;  // rows of odd fn are only 8-byte aligned

// old in the lanes whose bank (group of 4 lanes within a row of 16) is not in BANKS, elsewhere
// src of the lane 8 further round the row
template <int BANKS> __device__ __forceinline__ double row_ror8_merge (double old, double src)
{
    int lo = __builtin_amdgcn_update_dpp (__double2loint (old), __double2loint (src), 0x128, 0xf, BANKS, false) ;
    int hi = __builtin_amdgcn_update_dpp (__double2hiint (old), __double2hiint (src), 0x128, 0xf, BANKS, false) ;
    return __hiloint2double (hi, lo) ;
}

// (row r, columns c and c+1) per lane  <->  (rows r_even and r_even+1, one column) per lane; the
// exchange is its own inverse
__device__ __forceinline__ d2 pair_exchange (d2 v)
{
    d2 o ;
    o.x = row_ror8_merge<0xC> (v.x, v.y) ;
    o.y = row_ror8_merge<0x3> (v.y, v.x) ;
    return o ;
}

// The same exchange when the two lanes are 16 apart (neighbouring rows of 16 lanes): the odd rows
// of the first register trade places with the even rows of the second.  Its own inverse too.
__device__ __forceinline__ d2 pair_exchange16 (d2 v)
{
    auto lo = __builtin_amdgcn_permlane16_swap ((unsigned) __double2loint (v.x), (unsigned) __double2loint (v.y), false, false) ;
    auto hi = __builtin_amdgcn_permlane16_swap ((unsigned) __double2hiint (v.x), (unsigned) __double2hiint (v.y), false, false) ;
    d2 o ;
    o.x = __hiloint2double ((int) hi [0], (int) lo [0]) ;
    o.y = __hiloint2double ((int) hi [1], (int) lo [1]) ;
    return o ;
}

template <int RG> __device__ __forceinline__ d2 pair_exchange_rg (d2 v)
{
    if constexpr (RG == 8) return pair_exchange (v) ; else return pair_exchange16 (v) ;
}

// -----------------------------------------------------------------------------
// factorize_panel
// -----------------------------------------------------------------------------
//
// The rows of a column are dealt to RG lanes (RG = 8 or 16 row groups), so a wave
// owns 64/RG of the 32 panel columns and the workgroup has 32*RG threads (4 or 8
// waves).  Lane l of a wave stands for column cl = l/RG of the wave and row group
// rg = l%RG, and keeps rows rg, rg+RG, ... of its column in registers (RPT of them).
// A sum over the rows of one column is then a sum over RG neighbouring lanes: three
// or four DPP steps, no LDS and no barrier.  The chain of dependent operations per
// column grows with RPT (a lone wave issues a double precision operation only every
// 8 to 10 cycles), which is why the stand-alone kernels use RG = 16: half the
// registers per lane, half the chain.  Per column k:
//   every lane reads the normalised Householder vector v_k (v_kk = 1, zeros
//   above) from LDS, forms its share of v_k'a_c, reduces it over its 8 lanes,
//   and updates its column: a_c -= tau (v_k'a_c) v_k;
//   the 8 lanes of column k+1 then hold that column's final values, so they
//   compute its reflector at once (sigma over 8 lanes, the scalar part, the
//   normalised vector) and publish it to LDS for the next step.
// One barrier per column.  Columns j < k hold v_j, so the same dot products
// give Z(j,k) = -tau_k v_j'v_k, from which a wave that is off the critical
// path builds column k of T during the next step:
// T(0:k-1,k) = T(0:k-1,0:k-1) Z(0:k-1,k), T(k,k) = tau_k (the recurrence of
// factorize_vt.cu:397-410 and spqr_larftb.cpp:13-22).
//
// Householder convention (restated from factorize_vt.cu:272-293, but with v
// scaled to a unit head so that V1 fits the strictly-lower part of the VT
// tile):  sigma = sum_{i>k} a_ik^2;  if sigma <= 1e-90: tau = 0, R_kk = a_kk;
// else s = sqrt(a_kk^2 + sigma), v1 = a_kk - s (a_kk <= 0) or -sigma/(a_kk+s),
// R_kk = s >= 0, v = [1; a_{k+1:,k}/v1], tau = -v1/s.

struct FactorizeLds
{
    int    rowmap [QR_MAX_K] ;
    double vbuf   [2][QR_MAX_K] ;   // double buffered: v_k is read while v_{k+1} is written
    double tau    [32] ;
    double Z      [32][33] ;
    double T      [32][33] ;
} ;

template <int CTRL> __device__ __forceinline__ double dpp_move (double x)
{
    int lo = __builtin_amdgcn_update_dpp (0, __double2loint (x), CTRL, 0xf, 0xf, true) ;
    int hi = __builtin_amdgcn_update_dpp (0, __double2hiint (x), CTRL, 0xf, 0xf, true) ;
    return __hiloint2double (hi, lo) ;
}

// sum over each group of RG neighbouring lanes, result in all of them
template <int RG> __device__ __forceinline__ double sum_rows (double x)
{
    x += dpp_move<0xB1> (x) ;       // quad_perm [1,0,3,2]
    x += dpp_move<0x4E> (x) ;       // quad_perm [2,3,0,1]
    x += dpp_move<0x141> (x) ;      // row_half_mirror: lane i <-> 7-i of each 8
    if constexpr (RG == 16) x += dpp_move<0x140> (x) ;     // row_mirror: lane i <-> 15-i of each 16
    return x ;
}

// Column kc of T, by one wave: lane l works on row l%32 and takes every other group of four
// terms (l/32 says which), the two halves of the wave are added at the end.  Z(i,kc) = 0 for
// i >= kc and T(j,i) = 0 for i < j (both arrays start as zeros), so the sums need no masks and
// may run past kc to the end of a group.  This runs beside the Householder step of its wave and
// must not take longer than the reflector of the owning wave, or it becomes the critical path.
__device__ __forceinline__ void t_column (FactorizeLds &L, const int kc, const int lane)
{
    const int j = lane & 31, h = lane >> 5 ;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0 ;
    #pragma unroll 2
    for (int i = 4*h ; i < kc ; i += 8)
    {
        acc0 += L.T [j][i]   * L.Z [i][kc] ;
        acc1 += L.T [j][i+1] * L.Z [i+1][kc] ;
        acc2 += L.T [j][i+2] * L.Z [i+2][kc] ;
        acc3 += L.T [j][i+3] * L.Z [i+3][kc] ;
    }
    const double x = (acc0 + acc1) + (acc2 + acc3) ;
    // lanes 32..63 of the first register trade places with lanes 0..31 of the second
    auto lo = __builtin_amdgcn_permlane32_swap ((unsigned) __double2loint (x), (unsigned) __double2loint (x), false, false) ;
    auto hi = __builtin_amdgcn_permlane32_swap ((unsigned) __double2hiint (x), (unsigned) __double2hiint (x), false, false) ;
    const double sum = __hiloint2double ((int) hi [0], (int) lo [0]) + __hiloint2double ((int) hi [1], (int) lo [1]) ;
    if (lane < 32) L.T [j][kc] = (j < kc) ? sum : ((j == kc) ? L.tau [kc] : 0.0) ;
}

// The 8 lanes of column k1 (all of them active, nobody else) make its reflector and publish it:
// vb[r] = 0 above row k1, 1 at k1, a_r/v1 below, for the rows of slots >= I0; returns R(k1,k1).
template <int RPT, int RG>
__device__ __forceinline__ double make_reflector
(
    const double (&a) [RPT], FactorizeLds &L, const int k1, const int i0, const int rg
)
{
    double part = 0, part1 = 0, x1 = 0 ;
    #pragma unroll
    for (int i = 0 ; i < RPT ; i++)
    {
        const int r = rg + RG*i ;
        const double x = a [i] ;
        if (r > k1) { if (i & 1) part1 += x*x ; else part += x*x ; }
        else if (r == k1) x1 = x ;
    }
    const double sigma = sum_rows<RG> (part + part1) ;
    x1 = sum_rows<RG> (x1) ;        // one lane holds it, the others add zeros
    // The scalar part sits on the critical path of every column (a lone wave issues a dependent
    // double precision operation only every 16 cycles or so), so it is built from the hardware
    // reciprocal and reciprocal square root seeds with two Newton steps each, and arranged so that
    // only one reciprocal depends on the square root: with q = |x1| + s (no cancellation),
    //   x1 <= 0:  v1 = x1 - s = -q,          1/v1 = -1/q,       tau = -v1/s = q/s
    //   x1 >  0:  v1 = -sigma/(x1 + s) = -sigma/q,  1/v1 = -q/sigma,  tau = sigma/(q s)
    // and 1/sigma is refined next to 1/sqrt(n2), off the chain.
    double s, tau, inv ;
    if (sigma <= HOUSE_EPS)
    {
        s = x1 ; tau = 0 ; inv = 0 ;
    }
    else
    {
        const double n2 = fma (x1, x1, sigma) ;
        double y = __builtin_amdgcn_rsq (n2) ;                  // ~ 1/s
        double rs = __builtin_amdgcn_rcp (sigma) ;              // ~ 1/sigma
        y = fma (0.5 * y, fma (-n2 * y, y, 1.0), y) ;
        rs = fma (rs, fma (-sigma, rs, 1.0), rs) ;
        y = fma (0.5 * y, fma (-n2 * y, y, 1.0), y) ;
        rs = fma (rs, fma (-sigma, rs, 1.0), rs) ;
        s = n2 * y ;
        s = fma (fma (-s, s, n2), 0.5 * y, s) ;                 // s = sqrt (n2)
        const double q = fabs (x1) + s ;
        double rq = __builtin_amdgcn_rcp (q) ;
        const double qs = -q * rs ;                             // 1/v1 when x1 > 0
        rq = fma (rq, fma (-q, rq, 1.0), rq) ;
        rq = fma (rq, fma (-q, rq, 1.0), rq) ;
        inv = (x1 <= 0) ? -rq : qs ;
        tau = (x1 <= 0) ? (q * y) : (sigma * rq * y) ;
    }
    double *vb = L.vbuf [k1 & 1] ;
    #pragma unroll
    for (int i = 0 ; i < RPT ; i++)
    {
        const int r = rg + RG*i ;
        if (i >= i0) vb [r] = (r > k1) ? a [i] * inv : ((r == k1) ? 1.0 : 0.0) ;
    }
    if (rg == 0) L.tau [k1] = tau ;
    return s ;
}

// The RG Householder steps k = RG*IK .. RG*IK+RG-1.  With rows dealt to lanes as
// r = rg + RG*i, the register slot of the pivot row (i = IK) is a compile-time
// constant inside this function: slots i < IK lie entirely above the pivot row
// and are skipped (v is zero there).
template <int RPT, int RG, int IK>
__device__ __forceinline__ void house_steps
(
    double (&a) [RPT], double &s_cur, FactorizeLds &L, const int kmax, const int rg, const int c,
    const int wave, const int lane
)
{
    constexpr int CPW = 64 / RG ;                       // columns per wave
    constexpr int NW = 32 / CPW ;                       // waves
    if constexpr (IK < RPT && IK * RG < 32)
    {
        for (int rgk = 0 ; rgk < RG ; rgk++)
        {
            const int k = RG*IK + rgk ;
            if (k >= kmax) break ;
            const int k1 = k + 1 ;
            const int own1 = (k1 / CPW) & (NW - 1) ;    // the wave that owns column k+1: the critical one
            if (k > 0 && wave == ((own1 + NW/2) & (NW - 1))) t_column (L, k - 1, lane) ;

            const double *vb = L.vbuf [k & 1] ;
            double v [RPT] ;
            #pragma unroll
            for (int i = IK ; i < RPT ; i++) v [i] = vb [rg + RG*i] ;
            const double tau = L.tau [k] ;

            double d0 = 0, d1 = 0 ;
            #pragma unroll
            for (int i = IK ; i < RPT ; i++)
            {
                if ((i - IK) & 1) d1 += v [i] * a [i] ; else d0 += v [i] * a [i] ;
            }
            const double z = -tau * sum_rows<RG> (d0 + d1) ;

            if (c > k)
            {
                #pragma unroll
                for (int i = IK ; i < RPT ; i++) a [i] += v [i] * z ;
            }
            else if (c == k)
            {
                // the column becomes R above and at the diagonal, v below it
                #pragma unroll
                for (int i = IK ; i < RPT ; i++)
                {
                    if (i > IK || rg > rgk) a [i] = v [i] ;
                    else if (rg == rgk)     a [i] = s_cur ;
                }
            }
            else if (rg == 0)
            {
                L.Z [c][k] = z ;
            }

            if (wave == own1 && k1 < kmax)
            {
                if (c == k1) s_cur = make_reflector<RPT, RG> (a, L, k1, (rgk == RG - 1) ? IK + 1 : IK, rg) ;
            }
            __syncthreads ( ) ;
        }
    }
}

template <int RPT, int RG>
__device__ __forceinline__ void factorize_panel (const QrPanelTask &t, char *smem)
{
    constexpr int NT = 32 * RG ;                        // threads of the workgroup
    FactorizeLds &L = *reinterpret_cast<FactorizeLds *>(smem) ;
    const int tid = threadIdx.x ;
    const int lane = tid & 63 ;
    const int wave = __builtin_amdgcn_readfirstlane (tid >> 6) ;    // a scalar: branches on it are uniform
    const int rg = lane & (RG - 1) ;
    const int c = (64 / RG) * wave + lane / RG ;
    const int K = t.K, w = t.w, ld = t.ld ;
    const int kmax = (K < w) ? K : w ;
    gdouble *F = (gdouble *) t.F ;
    gdouble *VT = (gdouble *) t.VT ;

    build_rowmap (t, L.rowmap, RG*RPT) ;
    for (int i = tid ; i < 2*32*33 ; i += NT) (&L.Z[0][0]) [i] = 0 ;             // Z and T
    __syncthreads ( ) ;

    // Load with 16 bytes per lane: the lanes of columns c and c^1 (RG lanes apart) each fetch both
    // columns of one of two rows, then trade one element (see pair_exchange).
    double a [RPT] ;
    const int cb = c & ~1 ;                                     // first column of the lane's pair
    const bool pair0 = cb < w, pair1 = (cb + 1 < w) && (t.col0 + cb + 1 < ld) ;
    // (all loads first, the exchanges after them: an exchange waits for its load)
    #pragma unroll
    for (int i = 0 ; i < RPT ; i += 2)
    {
        const int r = rg + RG * (i + (c & 1)) ;                 // even column: first row of the two
        d2 v = {0.0, 0.0} ;
        if (r < K && pair0)
        {
            gdouble *p = F + (size_t) L.rowmap [r] * ld + t.col0 + cb ;
            if (pair1) v = *(gd2 *) p ; else v.x = p [0] ;
        }
        a [i] = v.x ; a [i+1] = v.y ;
    }
    #pragma unroll
    for (int i = 0 ; i < RPT ; i += 2)
    {
        d2 v = {a [i], a [i+1]} ;
        v = pair_exchange_rg<RG> (v) ;
        a [i] = (c < w) ? v.x : 0.0 ;
        a [i+1] = (c < w) ? v.y : 0.0 ;
    }

    double s_cur = 0 ;
    if (wave == 0 && c == 0) s_cur = make_reflector<RPT, RG> (a, L, 0, 0, rg) ;
    __syncthreads ( ) ;

    house_steps<RPT, RG, 0> (a, s_cur, L, kmax, rg, c, wave, lane) ;
    house_steps<RPT, RG, 1> (a, s_cur, L, kmax, rg, c, wave, lane) ;
    house_steps<RPT, RG, 2> (a, s_cur, L, kmax, rg, c, wave, lane) ;
    house_steps<RPT, RG, 3> (a, s_cur, L, kmax, rg, c, wave, lane) ;

    // ---- write back R (zeros below the diagonal), V2 in place, V1 to the VT tile ----
    #pragma unroll
    for (int i = 0 ; i < RPT ; i += 2)
    {
        d2 o ;
        #pragma unroll
        for (int e = 0 ; e < 2 ; e++)
        {
            const int r = rg + RG * (i + e) ;
            double val = a [i + e] ;
            if (r < w)
            {
                if (VT != NULL && c < r && r < K && c < w) VT [r*32 + c] = val ;
                if (c < r) val = 0.0 ;
            }
            o [e] = val ;
        }
        o = pair_exchange_rg<RG> (o) ;                          // back to (one row, two columns)
        const int r = rg + RG * (i + (c & 1)) ;
        if (r < K && pair0)
        {
            gdouble *p = F + (size_t) L.rowmap [r] * ld + t.col0 + cb ;
            if (pair1) *(gd2 *) p = o ; else p [0] = o.x ;
        }
    }

    // ---- T (only when somebody will apply this panel): its last column, then the tile ----
    if (VT != NULL)
    {
        if (wave == 0) t_column (L, kmax - 1, lane) ;
        __syncthreads ( ) ;
        for (int idx = tid ; idx < 32*32 ; idx += NT)
        {
            const int r = idx >> 5, cc = idx & 31 ;
            if (cc >= r && r < w && cc < w) VT [r*32 + cc] = L.T [r][cc] ;
        }
    }
}

// -----------------------------------------------------------------------------
// apply_block
// -----------------------------------------------------------------------------
//
// V (K x 32, unit lower trapezoidal) and T (32 x 32 upper) are staged once in
// LDS.  Each of the 4 waves then walks 16-column slabs of [c_begin,c_end).  A
// slab (K x 16) lives in registers as MT = K/16 MFMA C/D tiles.  For
// v_mfma_f64_16x16x4_f64 the C/D tile register i of lane l is element
// (m = 4i + l/16, n = l%16), which is exactly the B-operand layout of the
// 4-deep slice i, so the same registers serve as
//   B operand of  W  = V' A      (32 x 16, contraction over the K rows)
//   accumulator of A -= V (T' W) (contraction over the 32 reflector columns)
// and the slab makes one trip from HBM and back with no LDS staging of A.
//
// Memory side: 8 bytes per lane (one element) moves a slab at about half the
// rate of 16 bytes per lane, so a lane loads two neighbouring columns of one
// row: lane l covers row l/8 of an 8-row group and columns 2(l%8), 2(l%8)+1 of
// the slab (8 rows x 128 B per instruction).  The MFMA needs one column per
// lane, so lanes l and l^8 trade one element each (one DPP row rotation by 8):
// afterwards lane l holds rows 2(l/16) and 2(l/16)+1 of the group in column
// 2(l%8) + (l/8)%2.  Neither the order of the rows inside a tile nor the order
// of the columns inside a slab matters to the arithmetic as long as V is read
// in the same row order, so tile register i of lane l stands for
//   row  16 mt + 8 (i/2) + 2 (l/16) + i%2,   column slot l%16.
// LDS row stride 34 doubles keeps the A-operand reads of step 3 conflict free.

#define VS_LD 34

// (a micro-benchmark that includes this file may supply its own forms of these three)
#ifndef MFMA64
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64 ((a), (b), (c), 0, 0, 0)
#endif
#ifdef GPUQR_EXPERIMENT_DRY
// timing experiment (its own library, make exp): a marked update task works on zeros and stores nothing
#define SLAB_LOAD(p) (dry_ ? d2 {0.0, 0.0} : *(p))
#define SLAB_STORE(v, p) do { if (!dry_) *(p) = (v) ; } while (0)
#endif
#ifndef SLAB_LOAD
#define SLAB_LOAD(p) (*(p))
#define SLAB_STORE(v, p) (*(p) = (v))
#endif

template <int MT>
__device__ __forceinline__ void apply_block (const QrPanelTask &t, const int blk, char *smem)
{
    double (*Vs)[VS_LD] = reinterpret_cast<double (*)[VS_LD]>(smem) ;                       // [MT*16][34]
    double (*Ts)[VS_LD] = reinterpret_cast<double (*)[VS_LD]>(smem + (size_t) MT*16*VS_LD*8) ; // [32][34]
    int *rowmap = reinterpret_cast<int *>(smem + (size_t) (MT*16 + 32)*VS_LD*8) ;            // [MT*16]

    const int tid = threadIdx.x ;
#ifdef GPUQR_EXPERIMENT_DRY
    const int K = t.K, w = t.w & 63, ld = t.ld ;
    const bool dry_ = (t.w & 64) != 0 ;
#else
    const int K = t.K, w = t.w, ld = t.ld ;
#endif
    const int kmax = (K < w) ? K : w ;
    gdouble *F = (gdouble *) t.F ;
    gcdouble *VT = (gcdouble *) t.VT ;

    // rows past K point at the task's first row: loads through them stay in bounds, what they
    // bring meets zero rows of V, and nothing is stored through them
    build_rowmap (t, rowmap, MT*16, t.seg_row [0]) ;
    __syncthreads ( ) ;

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane (tid >> 6) ;    // wave: a scalar
    const int lr = lane >> 4, lc = lane & 15 ;
    const int l8 = lane >> 3 ;                  // memory side: row within an 8-row group
    const int j2 = (lane & 7) * 2 ;             // memory side: first of the lane's two columns
    const int pi_lc = 8*(lc >> 3) + 2*(lc & 3) + ((lc >> 2) & 1) ;     // tile row held by MFMA row lc
    const int c_begin = t.c_begin + blk * t.cb_cols ;
    const int c_end = (c_begin + t.cb_cols < t.c_end) ? (c_begin + t.cb_cols) : t.c_end ;
    const int nslab = (c_end - c_begin + 15) >> 4 ;

    // Software pipeline over the wave's slabs: while step 3 finishes tile mt of
    // the current slab, that tile is stored and the same registers are refilled
    // with tile mt of the wave's next slab, so the loads and stores of one slab
    // travel while the matrix pipe works on another.  What a load brings stays
    // in memory order until step 1 of its slab needs it.
    //
    // Loads are unconditional, so that none of them sits in a branch with a wait
    // behind it: a lane whose columns lie (partly) past c_end reads the last two
    // columns before c_end instead (c_end - 2 >= c_begin - 1 >= 31 is a column of
    // the same row), rows past K read the task's first row.  The stray values are
    // finite entries of the front, each column slot is computed independently
    // and rows past K meet zero rows of V, so they change nothing that is stored.
    d2 A [MT][2] ;
    {
        const int slab0 = (wave < nslab) ? wave : 0 ;
        const int col = c_begin + slab0*16 + j2 ;
        const int lcol = (col < c_end - 2) ? col : (c_end - 2) ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                const int r = mt*16 + 8*h + l8 ;
                A [mt][h] = SLAB_LOAD ((gd2 *) (F + (size_t) rowmap [r] * ld + lcol)) ;
            }
        }
    }

    // V and T go to LDS while the first slab is on its way
    // two columns per lane (16 bytes, as for the slabs); col0 is a multiple of 32, so the pair never
    // straddles the panel edge except at an odd panel width, where the second column is dropped
    for (int idx = tid ; idx < MT*16*16 ; idx += QR_THREADS)
    {
        const int r = idx >> 4, c = (idx & 15) * 2 ;
        d2 val = {0.0, 0.0} ;
        if (r < K && c < kmax)
        {
            if (r < w)
            {
                val.x = (c < r) ? VT [r*32 + c] : ((c == r) ? 1.0 : 0.0) ;
                val.y = (c + 1 < r) ? VT [r*32 + c + 1] : ((c + 1 == r) ? 1.0 : 0.0) ;
            }
            else if (c + 1 < t.ld - t.col0)
                val = *(gd2 *) (F + (size_t) rowmap [r] * ld + t.col0 + c) ;
            else
                val.x = F [(size_t) rowmap [r] * ld + t.col0 + c] ;
            if (c + 1 >= kmax) val.y = 0.0 ;
        }
        Vs [r][c] = val.x ; Vs [r][c + 1] = val.y ;
    }
    for (int idx = tid ; idx < 32*16 ; idx += QR_THREADS)
    {
        const int r = idx >> 4, c = (idx & 15) * 2 ;
        typedef __attribute__((address_space(1))) const d2 gcd2 ;
        const d2 tv = *(gcd2 *) (VT + r*32 + c) ;       // the tile is 32 x 32 and 16-byte aligned
        Ts [r][c]     = (c >= r && c < kmax) ? tv.x : 0.0 ;
        Ts [r][c + 1] = (c + 1 >= r && c + 1 < kmax) ? tv.y : 0.0 ;
    }
    __syncthreads ( ) ;

    for (int slab = wave ; slab < nslab ; slab += 4)
    {
        const int col = c_begin + slab*16 + j2 ;
        const bool v0 = col < c_end, v1 = col + 1 < c_end ;
        // after its last slab a wave fetches that slab again (from L2, it has just been written)
        // rather than branch round 2 MT loads
        const int ncol = (slab + 4 < nslab) ? (col + 64) : col ;
        const int nlcol = (ncol < c_end - 2) ? ncol : (c_end - 2) ;

        // step 1: W = V' A, two 16-row tiles of W (two independent accumulation chains)
        v4f64 W0 = {0,0,0,0}, W1 = {0,0,0,0} ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                d2 v = A [mt][h] ;
                v.x = v1 ? v.x : v.y ;      // last column of an odd width: the load was moved one column back
                A [mt][h] = pair_exchange (v) ;
                #pragma unroll
                for (int e = 0 ; e < 2 ; e++)
                {
                    const int r = mt*16 + 8*h + 2*lr + e ;
                    const double b = A [mt][h][e] ;
                    W0 = MFMA64 (Vs [r][lc],      b, W0) ;
                    W1 = MFMA64 (Vs [r][16 + lc], b, W1) ;
                }
            }
        }

        // step 2: X = T' W (T upper triangular: rows 16..31 of T do not reach X0)
        v4f64 X0 = {0,0,0,0}, X1 = {0,0,0,0} ;
        #pragma unroll
        for (int q = 0 ; q < 4 ; q++)
        {
            X0 = MFMA64 (Ts [4*q + lr][lc],      W0 [q], X0) ;
            X1 = MFMA64 (Ts [4*q + lr][16 + lc], W0 [q], X1) ;
        }
        #pragma unroll
        for (int q = 0 ; q < 4 ; q++)
        {
            X1 = MFMA64 (Ts [16 + 4*q + lr][16 + lc], W1 [q], X1) ;
        }
        X0 = -X0 ; X1 = -X1 ;

        // step 3: A += V (-X), tile by tile; store the tile, fetch its successor.
        // The opaque copy of the lane's memory row makes the compiler recompute
        // the addresses from the row map instead of keeping them alive across
        // the MFMA section.
        int l8_st = l8 ;
        asm volatile ("" : "+v" (l8_st)) ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            v4f64 acc = {A [mt][0][0], A [mt][0][1], A [mt][1][0], A [mt][1][1]} ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++)
                acc = MFMA64 (Vs [mt*16 + pi_lc][4*q + lr], X0 [q], acc) ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++)
                acc = MFMA64 (Vs [mt*16 + pi_lc][16 + 4*q + lr], X1 [q], acc) ;
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                const int r = mt*16 + 8*h + l8_st ;
                d2 o = {acc [2*h], acc [2*h + 1]} ;
                o = pair_exchange (o) ;
                gdouble *rowp = F + (size_t) rowmap [r] * ld ;
                if (r < K)
                {
                    if (v1)      SLAB_STORE (o, (gd2 *) (rowp + col)) ;
#ifdef GPUQR_EXPERIMENT_DRY
                    else if (v0 && !dry_) rowp [col] = o.x ;
#else
                    else if (v0) rowp [col] = o.x ;
#endif
                }
                A [mt][h] = SLAB_LOAD ((gd2 *) (rowp + nlcol)) ;
            }
        }
    }
}

// -----------------------------------------------------------------------------
// assembly
// -----------------------------------------------------------------------------

struct SEntryDev { long long findex ; double value ; } ;    // 16 bytes, GPUQREngine_SEntry.hpp:13-23

__device__ void sassemble (const QrSAssembleTask &t)
{
    // an entry is two 8-byte words: index, then the bits of the value
    typedef __attribute__((address_space(1))) const long long gcll ;
    gcll *S = (gcll *) t.S ;
    gdouble *F = (gdouble *) t.F ;
    if (t.ld == t.fn)
    {
        for (int64_t p = t.pstart + threadIdx.x ; p < t.pend ; p += QR_THREADS)
            F [S [2*p]] = __longlong_as_double (S [2*p + 1]) ;
    }
    else
    {
        // rows of the front are padded to ld entries: split the packed index
        for (int64_t p = t.pstart + threadIdx.x ; p < t.pend ; p += QR_THREADS)
        {
            const long long k = S [2*p], i = k / t.fn ;
            F [i * t.ld + (k - i * t.fn)] = __longlong_as_double (S [2*p + 1]) ;
        }
    }
}

__device__ void packassemble (const QrPackAssembleTask &t, char *smem)
{
    int *shRimap = reinterpret_cast<int *>(smem) ;          // [QR_PACK_TILE]
    int *shRjmap = shRimap + QR_PACK_TILE ;                 // [QR_PACK_TILE]
    const int ctm = t.ciend - t.cistart ;
    const int ctn = t.cjend - t.cjstart ;
    gcint *Rimap = (gcint *) t.Rimap, *Rjmap = (gcint *) t.Rjmap ;
    for (int p = threadIdx.x ; p < ctm ; p += QR_THREADS) shRimap [p] = Rimap [t.cistart + p] ;
    for (int p = threadIdx.x ; p < ctn ; p += QR_THREADS) shRjmap [p] = Rjmap [t.cjstart + p] ;
    __syncthreads ( ) ;
    // One wave per row of the tile, lanes along the (contiguous) child columns, two columns (16 bytes)
    // per lane, so that one instruction fetches a whole 128-column row of the tile; only the upper
    // trapezoid cj >= ci carries data (packAssemble.cu:47-55).  Eight rows are fetched before anything
    // is stored (eight 16-byte loads in flight per lane).  Where the parent columns of a lane's pair are
    // neighbours too (the column map is a run of consecutive indices except where the parent has
    // columns of its own in between) the pair is stored with one 16-byte store.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6 ;
    static_assert (QR_PACK_TILE == 128, "one pass of 64 lanes x 2 columns covers the columns of a tile") ;
    const int c0 = 2 * lane, c1 = c0 + 1 ;
    // (columns past the tile's edge are read from its last pair; ctn >= 1, and the pair before the edge
    // is memory of the child's block unless the tile is a single column wide, which reads one double)
    const bool one_col = (ctn < 2) ;
    const int lc0 = one_col ? 0 : ((c1 < ctn) ? c0 : (ctn - 2)) ;
    const int pj0 = (c0 < ctn) ? shRjmap [c0] : 0, pj1 = (c1 < ctn) ? shRjmap [c1] : -1 ;
    const bool pair = (c1 < ctn) && (pj1 == pj0 + 1) ;
    for (int cil0 = wave ; cil0 < ctm ; cil0 += 32)
    {
        d2 v [8] ;
        #pragma unroll
        for (int u = 0 ; u < 8 ; u++)
        {
            // unconditional loads through clamped indices (every entry of the tile's rows is memory
            // of the child's block, also below its diagonal): no branch, no wait behind a load
            const int cil = (cil0 + 4*u < ctm) ? (cil0 + 4*u) : (ctm - 1) ;
            gcdouble *Crow = (gcdouble *) t.C + (size_t) (t.cistart + cil) * t.cld + t.cjstart ;
            if (one_col) { v [u].x = Crow [0] ; v [u].y = 0 ; }
            else v [u] = *(const gd2 *) (Crow + lc0) ;
        }
        #pragma unroll
        for (int u = 0 ; u < 8 ; u++)
        {
            const int cil = cil0 + 4*u ;
            if (cil >= ctm) break ;
            const int ci = t.cistart + cil ;
            gdouble *Prow = (gdouble *) t.P + (size_t) shRimap [cil] * t.pn ;
            d2 val = v [u] ;
            if (!one_col && c1 >= ctn && c0 < ctn) val.x = val.y ;     // odd width: the last pair was read one column back
            const bool ok0 = (c0 < ctn) && (t.cjstart + c0 >= ci), ok1 = (c1 < ctn) && (t.cjstart + c1 >= ci) ;
            if (ok0 && ok1 && pair) *(gd2 *) (Prow + pj0) = val ;
            else
            {
                if (ok0) Prow [pj0] = val.x ;
                if (ok1) Prow [pj1] = val.y ;
            }
        }
    }
}

// -----------------------------------------------------------------------------
// S entries from the pattern and this factorization's values
// -----------------------------------------------------------------------------
//
// S [k] = (findex [k], Ax [aperm [k]]): where an entry of A lands in its front is known from the
// symbolic analysis; only the values change between factorizations, so only Ax crosses PCIe.

__global__ void __launch_bounds__ (QR_THREADS) k_build_s (long long *S, const long long *findex, const long long *aperm,
    const double *Ax, long long count)
{
    const long long k = (long long) blockIdx.x * QR_THREADS + threadIdx.x ;
    if (k < count)
    {
        S [2*k] = findex [k] ;
        S [2*k + 1] = __double_as_longlong (Ax [aperm [k]]) ;
    }
}

extern "C" hipError_t qr_launch_build_s (void *d_S, const long long *d_findex, const long long *d_aperm,
    const double *d_Ax, long long count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess ;
    const long long nb = (count + QR_THREADS - 1) / QR_THREADS ;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue ;
    hipLaunchKernelGGL (k_build_s, dim3 ((unsigned) nb), dim3 (QR_THREADS), 0, stream, (long long *) d_S, d_findex, d_aperm, d_Ax, count) ;
    return hipGetLastError ( ) ;
}

// -----------------------------------------------------------------------------
// R hand-off
// -----------------------------------------------------------------------------
//
// Row-major R rows -> SPQR's column-packed stack layout (the job of the host loop at
// spqrgpu_kernel.cpp:417-439), on the device so that only the packed entries cross PCIe.
// One workgroup per (front, 32-row tile): it walks the 32-column tiles at and right of the
// diagonal, reads each with lanes along the row (contiguous in the front) and writes it with
// lanes along the column (contiguous in the packed block), through a padded LDS tile.
// tile_prefix [f] = number of row tiles of fronts before f.

__global__ void __launch_bounds__ (QR_THREADS) k_pack_r (const QrPackRTask *tasks, const long long *tile_prefix, int nfronts)
{
    __shared__ double tile [32][33] ;
    // front of this workgroup: last f with tile_prefix [f] <= blockIdx.x
    int lo = 0, hi = nfronts - 1 ;
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) >> 1 ;
        if (tile_prefix [mid] <= (long long) blockIdx.x) lo = mid ; else hi = mid - 1 ;
    }
    const QrPackRTask t = tasks [lo] ;
    const int i0 = 32 * (int) ((long long) blockIdx.x - tile_prefix [lo]) ;
    gcdouble *F = (gcdouble *) t.F ;
    gdouble *out = (gdouble *) t.out ;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5 ;
    const long long frank = t.frank, tri = frank * (frank + 1) / 2 ;
    // columns [c0, c1) only, from the diagonal on (c0 is a multiple of 32)
    for (int j0 = (i0 > t.c0) ? i0 : t.c0 ; j0 < t.c1 ; j0 += 32)
    {
        #pragma unroll
        for (int k = 0 ; k < 4 ; k++)
        {
            const int i = i0 + ty + 8*k, j = j0 + tx ;
            tile [ty + 8*k][tx] = (i < t.frank && j < t.c1) ? F [(size_t) i * t.ld + j] : 0.0 ;
        }
        __syncthreads ( ) ;
        #pragma unroll
        for (int k = 0 ; k < 4 ; k++)
        {
            const int i = i0 + tx ;
            const long long j = j0 + ty + 8*k ;
            if (i < t.frank && j < t.c1 && j >= i)
            {
                const long long col = (j < frank) ? (j * (j + 1) / 2) : (tri + (j - frank) * frank) ;
                out [col + i] = tile [tx][ty + 8*k] ;
            }
        }
        __syncthreads ( ) ;
    }
}

extern "C" hipError_t qr_launch_pack_r (const QrPackRTask *d_tasks, const long long *d_tile_prefix, int nfronts,
    long long ntiles, hipStream_t stream)
{
    if (nfronts <= 0 || ntiles <= 0) return hipSuccess ;
    if (ntiles > 0x7fffffffLL) return hipErrorInvalidValue ;
    hipLaunchKernelGGL (k_pack_r, dim3 ((unsigned) ntiles), dim3 (QR_THREADS), 0, stream, d_tasks, d_tile_prefix, nfronts) ;
    return hipGetLastError ( ) ;
}

// -----------------------------------------------------------------------------
// rows of right-hand sides in and out of the fronts
// -----------------------------------------------------------------------------
//
// dst [j * dstride] = src [j * sstride] for j < ncols, one (row, j) per thread: the rows of B scattered
// to the right-hand-side blocks of the fronts (source strided by B's leading dimension), a child's
// finished rows pushed into its parent's block, and the rows of Q'B gathered at the end
// (destination strided).

__global__ void __launch_bounds__ (QR_THREADS) k_copy_rows (const QrRowCopy *tasks, long long ntasks, int ncols,
    long long sstride, long long dstride)
{
    const long long g = (long long) blockIdx.x * QR_THREADS + threadIdx.x ;
    const long long row = g / ncols ;
    const int j = (int) (g - row * ncols) ;
    if (row < ntasks)
    {
        const QrRowCopy t = tasks [row] ;
        ((gdouble *) t.dst) [j * dstride] = ((gcdouble *) t.src) [j * sstride] ;
    }
}

extern "C" hipError_t qr_launch_copy_rows (const QrRowCopy *d_tasks, long long ntasks, int ncols,
    long long sstride, long long dstride, hipStream_t stream)
{
    if (ntasks <= 0 || ncols <= 0) return hipSuccess ;
    const long long nb = (ntasks * ncols + QR_THREADS - 1) / QR_THREADS ;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue ;
    hipLaunchKernelGGL (k_copy_rows, dim3 ((unsigned) nb), dim3 (QR_THREADS), 0, stream, d_tasks, ntasks, ncols, sstride, dstride) ;
    return hipGetLastError ( ) ;
}

// -----------------------------------------------------------------------------
// rank detection: 2-norms of column pieces
// -----------------------------------------------------------------------------
//
// out [t] = || x_t ||_2 for the strided column piece of task t, one wave per task: the largest
// magnitude first, then the sum of squares of the scaled entries (no overflow or underflow of the
// squares, as dnrm2).  The pieces are the rows of dead pivot columns from their would-be pivot row
// down: the norm SPQR's CPU path tests against its tolerance (SPQR/Source/spqr_front.cpp:418-424).

__global__ void __launch_bounds__ (QR_THREADS) k_col_norms (const QrColNormTask *tasks, long long ntasks, double *out)
{
    const int lane = threadIdx.x & 63 ;
    const long long t = (long long) blockIdx.x * (QR_THREADS / 64) + (threadIdx.x >> 6) ;
    if (t >= ntasks) return ;                       // (whole waves leave; nobody waits for them)
    const QrColNormTask task = tasks [t] ;
    gcdouble *x = (gcdouble *) task.x ;
    double big = 0 ;
    int nan = 0 ;
    for (long long i = lane ; i < task.nrows ; i += 64)
    {
        const double a = fabs (x [i * task.ld]) ;
        nan |= (a != a) ;
        big = fmax (big, a) ;
    }
    for (int s = 32 ; s > 0 ; s >>= 1) big = fmax (big, __shfl_xor (big, s, 64)) ;
    nan = __any (nan) ;
    double ssq = 0 ;
    if (big > 0 && !nan)
    {
        const double inv = 1.0 / big ;
        for (long long i = lane ; i < task.nrows ; i += 64) { const double v = x [i * task.ld] * inv ; ssq += v * v ; }
        for (int s = 32 ; s > 0 ; s >>= 1) ssq += __shfl_xor (ssq, s, 64) ;
    }
    // a NaN in the piece gives a NaN (not "at or below" any tolerance), an infinity gives a NaN too (inf * 0)
    if (lane == 0) ((gdouble *) out) [t] = nan ? __longlong_as_double (0x7ff8000000000000LL) : big * sqrt (ssq) ;
}

extern "C" hipError_t qr_launch_col_norms (const QrColNormTask *d_tasks, long long ntasks, double *d_out, hipStream_t stream)
{
    if (ntasks <= 0) return hipSuccess ;
    const long long nb = (ntasks + QR_THREADS / 64 - 1) / (QR_THREADS / 64) ;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue ;
    hipLaunchKernelGGL (k_col_norms, dim3 ((unsigned) nb), dim3 (QR_THREADS), 0, stream, d_tasks, ntasks, d_out) ;
    return hipGetLastError ( ) ;
}

// -----------------------------------------------------------------------------
// kernels: one per task kind and row tier, so each gets its own register budget
// -----------------------------------------------------------------------------

// The stand-alone panel kernels run 16 row groups (512 threads, RPT = rows / 16), except for the
// 32-row tier: there the chain per column is short already and the fourth reduction step costs
// more than the halved row count saves, so it keeps 8 row groups (256 threads).
#define QR_FACT_THREADS 512
template <int RPT, int RG>
__global__ void __launch_bounds__ (32 * RG, 2) k_factorize (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    factorize_panel<RPT, RG> (tasks [refs [blockIdx.x].task].u.panel, smem) ;
}

template <int MT>
__global__ void __launch_bounds__ (QR_THREADS, 2) k_apply (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrRef ref = refs [blockIdx.x] ;
    apply_block<MT> (tasks [ref.task].u.panel, ref.blk, smem) ;
}

// All row tiers in one launch: the panel tasks of a round are few and latency-bound, so four
// launches in a row (one per tier) cost the sum of four latencies where one launch costs the
// longest.  The same holds for the three small tiers of the update.
__global__ void __launch_bounds__ (QR_FACT_THREADS, 2) k_factorize_all (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrTask &t = tasks [refs [blockIdx.x].task] ;
    switch (__builtin_amdgcn_readfirstlane (t.tier))
    {
        // (the upper four waves leave at once: a barrier only counts the waves still running)
        case 0:  if (threadIdx.x < 256) factorize_panel<4, 8> (t.u.panel, smem) ; break ;
        case 1:  factorize_panel<4, 16>  (t.u.panel, smem) ; break ;
        case 2:  factorize_panel<8, 16>  (t.u.panel, smem) ; break ;
        default: factorize_panel<16, 16> (t.u.panel, smem) ; break ;
    }
}

__global__ void __launch_bounds__ (QR_THREADS, 2) k_apply_small (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrRef ref = refs [blockIdx.x] ;
    const QrTask &t = tasks [ref.task] ;
    switch (__builtin_amdgcn_readfirstlane (t.tier))
    {
        case 0:  apply_block<2> (t.u.panel, ref.blk, smem) ; break ;
        case 1:  apply_block<4> (t.u.panel, ref.blk, smem) ; break ;
        default: apply_block<8> (t.u.panel, ref.blk, smem) ; break ;
    }
}

// every update tier in one launch: used when the 256-row launch of a round is too small to fill the
// chip, so that the small tiers ride along instead of waiting for it
__global__ void __launch_bounds__ (QR_THREADS, 2) k_apply_all (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrRef ref = refs [blockIdx.x] ;
    const QrTask &t = tasks [ref.task] ;
    switch (__builtin_amdgcn_readfirstlane (t.tier))
    {
        case 0:  apply_block<2>  (t.u.panel, ref.blk, smem) ; break ;
        case 1:  apply_block<4>  (t.u.panel, ref.blk, smem) ; break ;
        case 2:  apply_block<8>  (t.u.panel, ref.blk, smem) ; break ;
        default: apply_block<16> (t.u.panel, ref.blk, smem) ; break ;
    }
}

// every update tier and both assemblies in one launch: a round's handful of assembly workgroups ride
// with the round's updates instead of costing the stream a launch of their own
__global__ void __launch_bounds__ (QR_THREADS, 2) k_round_all (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrRef ref = refs [blockIdx.x] ;
    const QrTask &t = tasks [ref.task] ;
    const int type = __builtin_amdgcn_readfirstlane (t.type) ;
    if (type == QR_TASK_SASSEMBLE) { sassemble (t.u.sasm) ; return ; }
    if (type == QR_TASK_PACKASSEMBLE) { packassemble (t.u.pack, smem) ; return ; }
    switch (__builtin_amdgcn_readfirstlane (t.tier))
    {
        case 0:  apply_block<2>  (t.u.panel, ref.blk, smem) ; break ;
        case 1:  apply_block<4>  (t.u.panel, ref.blk, smem) ; break ;
        case 2:  apply_block<8>  (t.u.panel, ref.blk, smem) ; break ;
        default: apply_block<16> (t.u.panel, ref.blk, smem) ; break ;
    }
}

__global__ void __launch_bounds__ (QR_THREADS) k_assemble (const QrTask *tasks, const QrRef *refs)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    const QrTask &t = tasks [refs [blockIdx.x].task] ;
    if (t.type == QR_TASK_SASSEMBLE) sassemble (t.u.sasm) ;
    else                             packassemble (t.u.pack, smem) ;
}

// Kernel id of a task: 0..3 factorize tiers, 4..7 apply tiers, 8 assembly.
extern "C" int qr_task_kernel_id (int type, int tier)
{
    if (type == QR_TASK_FACTORIZE) return tier ;
    if (type == QR_TASK_APPLY)     return 4 + tier ;
    return 8 ;
}

static size_t kernel_lds_bytes (int kid)
{
    if (kid < 4 || kid == 9) return sizeof (FactorizeLds) ;
    if (kid == 10) kid = 6 ;
    if (kid == 11 || kid == 12) kid = 7 ;
    if (kid < 8) { size_t rows = (size_t) 32 << (kid - 4) ; return (rows + 32) * VS_LD * 8 + rows * 4 ; }
    return 2 * QR_PACK_TILE * sizeof (int) ;
}

// -----------------------------------------------------------------------------
// persistent work-queue kernel
// -----------------------------------------------------------------------------
//
// One launch runs a whole engine call.  The queue holds the (task, block) pairs in
// a topological order of their dependencies.  A resident workgroup draws the
// next ticket, waits until the counter its task depends on has reached the
// value recorded in the entry, runs the task, makes its stores visible
// device-wide and adds one to the task's own counter.  Whoever holds ticket q
// only ever waits for entries with smaller tickets, all of which are finished
// or held by running workgroups, so the queue cannot deadlock whatever the
// number of resident workgroups.  Every spin is bounded: when one runs out the
// error word is set and every workgroup leaves at its next check.

#define QR_SPIN_LIMIT (1u << 22)
#define QR_QUEUE_LDS_TASK_BYTES ((256 + 32) * VS_LD * 8 + 256 * 4)     // apply_block<16>, the largest

// Out-of-line bodies for the queue kernel: inlined into one kernel, the eight task variants
// are register-allocated together and spill; as separate functions each keeps its own budget.
// Arguments of a device function arrive in vector registers and the compiler no longer knows
// that they are the same in every lane; readfirstlane puts the task fields back into scalar
// registers (the descriptor is then read with scalar loads), so that the bodies compile as
// they do inside a kernel.
__device__ __forceinline__ int uni (int x) { return __builtin_amdgcn_readfirstlane (x) ; }
template <class T> __device__ __forceinline__ T *uni_ptr (T *p)
{
    unsigned long long v = (unsigned long long) p ;
    unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane ((int) (unsigned) v) ;
    unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane ((int) (unsigned) (v >> 32)) ;
    return (T *) (((unsigned long long) hi << 32) | lo) ;
}
// The LDS array is named again inside each body rather than passed in: a pointer argument would
// be a generic pointer and every LDS access through it a flat access.
template <int RPT>
__device__ __attribute__((noinline)) void pq_factorize (const QrPanelTask &t)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    factorize_panel<RPT, 8> (*uni_ptr (&t), smem) ;        // the queue kernel's workgroups have 256 threads
}
template <int MT>
__device__ __attribute__((noinline)) void pq_apply (const QrPanelTask &t, const int blk)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    apply_block<MT> (*uni_ptr (&t), uni (blk), smem) ;
}

__device__ __forceinline__ int queue_load (const int *p)
{
    return __hip_atomic_load (p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
}

__global__ void __launch_bounds__ (QR_THREADS, 2)
k_persistent (const QrTask *tasks, const QrQueueRef *refs, int nrefs, int *counters)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    // two queue words live behind the largest task's LDS needs (no static LDS: it would sit in
    // front of the dynamic array and shift its base)
    int &sh_ticket = *reinterpret_cast<int *>(smem + QR_QUEUE_LDS_TASK_BYTES) ;
    int &sh_ok = *reinterpret_cast<int *>(smem + QR_QUEUE_LDS_TASK_BYTES + 4) ;
    for ( ; ; )
    {
        __syncthreads ( ) ;         // everybody is done with smem and sh_* of the previous task
        if (threadIdx.x == 0)
        {
            int ok = 0 ;
            int q = nrefs ;
            if (queue_load (counters + QR_QUEUE_ERROR) == 0)
                q = __hip_atomic_fetch_add (counters + QR_QUEUE_HEAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
            if (q < nrefs)
            {
                const QrQueueRef r = refs [q] ;
                ok = 1 ;
                if (r.need > 0)
                {
                    unsigned spins = 0 ;
                    while (queue_load (counters + QR_QUEUE_FIRST + r.wait_ctr) < r.need)
                    {
                        __builtin_amdgcn_s_sleep (8) ;
                        if ((++spins & 1023u) == 0 && queue_load (counters + QR_QUEUE_ERROR) != 0) { ok = 0 ; break ; }
                        if (spins > QR_SPIN_LIMIT)
                        {
                            __hip_atomic_store (counters + QR_QUEUE_ERROR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
                            ok = 0 ;
                            break ;
                        }
                    }
                    // what the finished tasks wrote must not be served from this CU's stale L1 lines
                    __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "agent") ;
                    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ;
                }
            }
            sh_ticket = q ;
            sh_ok = ok ;
        }
        __syncthreads ( ) ;
        const int q = sh_ticket ;
        if (q >= nrefs || !sh_ok) return ;          // queue drained, or an error was raised
        const QrQueueRef r = refs [q] ;
        const QrTask &t = tasks [r.task] ;
        switch (t.type)
        {
            case QR_TASK_FACTORIZE:
                switch (t.tier)
                {
                    case 0:  pq_factorize<4>  (t.u.panel) ; break ;
                    case 1:  pq_factorize<8>  (t.u.panel) ; break ;
                    case 2:  pq_factorize<16> (t.u.panel) ; break ;
                    default: pq_factorize<32> (t.u.panel) ; break ;
                }
                break ;
            case QR_TASK_APPLY:
                switch (t.tier)
                {
                    case 0:  pq_apply<2>  (t.u.panel, r.blk) ; break ;
                    case 1:  pq_apply<4>  (t.u.panel, r.blk) ; break ;
                    case 2:  pq_apply<8>  (t.u.panel, r.blk) ; break ;
                    default: pq_apply<16> (t.u.panel, r.blk) ; break ;
                }
                break ;
            case QR_TASK_SASSEMBLE:
                sassemble (t.u.sasm) ;
                break ;
            default:
                packassemble (t.u.pack, smem) ;
                break ;
        }
        // publish: every wave drains its stores, then one lane releases and signals
        asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ;
        __syncthreads ( ) ;
        if (threadIdx.x == 0)
        {
            __builtin_amdgcn_fence (__ATOMIC_RELEASE, "agent") ;
            asm volatile ("s_waitcnt vmcnt(0)" ::: "memory") ;
            __hip_atomic_fetch_add (counters + QR_QUEUE_FIRST + r.signal_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ;
        }
    }
}

extern "C" hipError_t qr_launch_persistent (const QrTask *d_tasks, const QrQueueRef *d_refs, int nrefs,
    int *d_counters, int nworkgroups, hipStream_t stream)
{
    if (nrefs <= 0) return hipSuccess ;
    hipLaunchKernelGGL (k_persistent, dim3 (nworkgroups), dim3 (QR_THREADS), QR_QUEUE_LDS_TASK_BYTES + 16, stream,
        d_tasks, d_refs, nrefs, d_counters) ;
    return hipGetLastError ( ) ;
}

// Launch nrefs workgroups of kernel id kid: workgroup b runs (task, block) = d_refs [b].
// start / stop != NULL: the events are taken by the launch itself (the stop event is its completion
// signal; no separate marker packets)
extern "C" hipError_t qr_launch_tasks_timed (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream,
    hipEvent_t start, hipEvent_t stop)
{
    if (nrefs <= 0) return hipSuccess ;
    const dim3 grid (nrefs), block (QR_THREADS), fblock (QR_FACT_THREADS) ;
    const size_t lds = kernel_lds_bytes (kid) ;
    switch (kid)
    {
        case 0: hipExtLaunchKernelGGL ((k_factorize<4, 8>),   grid, block,  lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 1: hipExtLaunchKernelGGL ((k_factorize<4, 16>),  grid, fblock, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 2: hipExtLaunchKernelGGL ((k_factorize<8, 16>),  grid, fblock, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 3: hipExtLaunchKernelGGL ((k_factorize<16, 16>), grid, fblock, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 4: hipExtLaunchKernelGGL (k_apply<2>,      grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 5: hipExtLaunchKernelGGL (k_apply<4>,      grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 6: hipExtLaunchKernelGGL (k_apply<8>,      grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 7: hipExtLaunchKernelGGL (k_apply<16>,     grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 8: hipExtLaunchKernelGGL (k_assemble,      grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;
        case 9:  hipExtLaunchKernelGGL (k_factorize_all, grid, fblock, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;  // tiers 0..3
        case 10: hipExtLaunchKernelGGL (k_apply_small,   grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;   // tiers 0..2
        case 11: hipExtLaunchKernelGGL (k_apply_all,     grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;   // tiers 0..3
        case 12: hipExtLaunchKernelGGL (k_round_all,     grid, block, lds, stream, start, stop, 0, d_tasks, d_refs) ; break ;   // tiers 0..3 + assembly
        default: return hipErrorInvalidValue ;
    }
    return hipGetLastError ( ) ;
}

extern "C" hipError_t qr_launch_tasks_ev (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream,
    hipEvent_t stop)
{
    return qr_launch_tasks_timed (kid, d_tasks, d_refs, nrefs, stream, NULL, stop) ;
}

extern "C" hipError_t qr_launch_tasks (int kid, const QrTask *d_tasks, const QrRef *d_refs, int nrefs, hipStream_t stream)
{
    return qr_launch_tasks_timed (kid, d_tasks, d_refs, nrefs, stream, NULL, NULL) ;
}

// -----------------------------------------------------------------------------
// fixed workspace of the entry point that allocates nothing
// -----------------------------------------------------------------------------
//
// gpuqr_factor_sparse_noalloc keeps what the other entry points put in hipMalloc'd buffers (the
// task records, the workgroup list and the VT tiles) in this array.  Every GPU the library runs on
// holds its own copy, at its own address, until the process ends.  In host code the array's name
// is only a handle for hipGetSymbolAddress: the address kernels and copies must use is the one
// that call returns, looked up once per device with that device current.

static __device__ __align__(256) unsigned char scratch [QR_SCRATCH_BYTES] ;

#define QR_SCRATCH_MAX_DEVICES 64
static void *g_scratch_ptr [QR_SCRATCH_MAX_DEVICES] ;

// Device address of the current device's copy of the workspace, and its size.
extern "C" hipError_t qr_scratch_address (void **ptr, size_t *bytes)
{
    int dev = -1 ;
    hipError_t e = hipGetDevice (&dev) ;
    if (e != hipSuccess) return e ;
    if (dev < 0 || dev >= QR_SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice ;
    if (g_scratch_ptr [dev] == NULL)
    {
        void *p = NULL ;
        e = hipGetSymbolAddress ((void **) &p, HIP_SYMBOL (scratch)) ;
        if (e != hipSuccess) return e ;
        if (p == NULL) return hipErrorInvalidSymbol ;
        g_scratch_ptr [dev] = p ;
    }
    *ptr = g_scratch_ptr [dev] ;
    *bytes = QR_SCRATCH_BYTES ;
    return hipSuccess ;
}

extern "C" hipError_t qr_kernels_init (void)
{
    // dynamic LDS above 64 KB needs an explicit opt-in on HIP
    hipError_t e = hipFuncSetAttribute ((const void *) k_apply<16>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) kernel_lds_bytes (7)) ;
    if (e != hipSuccess) return e ;
    e = hipFuncSetAttribute ((const void *) k_apply_all,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) kernel_lds_bytes (7)) ;
    if (e != hipSuccess) return e ;
    e = hipFuncSetAttribute ((const void *) k_round_all,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) kernel_lds_bytes (7)) ;
    if (e != hipSuccess) return e ;
    return hipFuncSetAttribute ((const void *) k_persistent,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int) (QR_QUEUE_LDS_TASK_BYTES + 16)) ;
}
