// Host-side entry points of the library under AddressSanitizer and UndefinedBehaviorSanitizer: the
// symbolic analysis in both modes, the staging at several declared memories, the assembly maps, the
// planner's counts, the symbolic object made from arrays (and its validation), and R conversion,
// R / R' products and both triangular solves on packed blocks.  No GPU is touched: everything called
// here is host code.  Built by `make -C tests/cpp host_asan` with the sanitizers on the host side only
// (the reference has valgrind targets for the same purpose, SPQR/Tcov/Makefile:195-202).
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "spqrgpu_amd.h"

typedef int64_t Long ;
static int fails = 0 ;
#define CHECK(c) do { if (!(c)) { printf ("FAILED line %d: %s\n", __LINE__, #c) ; fails++ ; } } while (0)

// A = [L ; sqrt(0.1) I] for the 5-point Laplacian on a k-by-k grid, compressed columns
static void grid_ls (Long k, std::vector<Long> &Ap, std::vector<Long> &Ai, std::vector<double> &Ax)
{
    const Long n = k * k ;
    Ap.assign (1, 0) ; Ai.clear ( ) ; Ax.clear ( ) ;
    for (Long j = 0 ; j < n ; j++)
    {
        const Long x = j % k, y = j / k ;
        if (y > 0)     { Ai.push_back (j - k) ; Ax.push_back (-1) ; }
        if (x > 0)     { Ai.push_back (j - 1) ; Ax.push_back (-1) ; }
        Ai.push_back (j) ; Ax.push_back (4) ;
        if (x < k - 1) { Ai.push_back (j + 1) ; Ax.push_back (-1) ; }
        if (y < k - 1) { Ai.push_back (j + k) ; Ax.push_back (-1) ; }
        Ai.push_back (n + j) ; Ax.push_back (sqrt (0.1)) ;
        Ap.push_back ((Long) Ai.size ( )) ;
    }
}

static const Long *arr (const spqr_amd_symbolic *S, const char *name, Long *len = NULL)
{
    Long l = 0 ;
    const Long *p = spqr_amd_symbolic_array (S, name, &l) ;
    if (len) *len = l ;
    return p ;
}

int main ( )
{
    for (Long k : {1, 2, 7, 24})
    {
        std::vector<Long> Ap, Ai ; std::vector<double> Ax ;
        grid_ls (k, Ap, Ai, Ax) ;
        const Long n = k * k, m = 2 * n ;
        std::vector<Long> Q (n) ;
        for (Long j = 0 ; j < n ; j++) Q [j] = (j * 7 + 3) % n ;
        if (n % 7 == 0) for (Long j = 0 ; j < n ; j++) Q [j] = n - 1 - j ;
        for (int relax : {-1, 0, 20})
        for (int rd : {0, 1})
        for (size_t mem : {(size_t) 0, (size_t) 1, (size_t) 3000000, (size_t) 1 << 34})
        {
            int status = -1 ;
            spqr_amd_symbolic *S = spqr_amd_analyze_rd (m, n, Ap.data ( ), Ai.data ( ), Q.data ( ), relax, mem, rd, &status) ;
            CHECK (S != NULL) ;
            if (!S) continue ;
            CHECK (status == 0 || status == 2) ;
            const Long nf = (Long) spqr_amd_symbolic_scalar (S, "nf") ;
            const Long anz = (Long) spqr_amd_symbolic_scalar (S, "anz") ;
            CHECK (anz == Ap [n] && nf >= 1 && nf <= n) ;
            const Long *Super = arr (S, "Super"), *Rp = arr (S, "Rp"), *Rj = arr (S, "Rj"), *Fm = arr (S, "Fm"), *Cm = arr (S, "Cm") ;
            const Long *Parent = arr (S, "Parent"), *Childp = arr (S, "Childp"), *Child = arr (S, "Child"), *Post = arr (S, "Post") ;
            const Long *Sp = arr (S, "Sp"), *Sj = arr (S, "Sj"), *Sleft = arr (S, "Sleft") ;
            const Long *RimapOff = arr (S, "RimapOffsets"), *RjmapOff = arr (S, "RjmapOffsets") ;
            const Long *Stagingp = arr (S, "Stagingp"), *StageMap = arr (S, "StageMap") ;
            const Long RimapSize = (Long) spqr_amd_symbolic_scalar (S, "RimapSize"), RjmapSize = (Long) spqr_amd_symbolic_scalar (S, "RjmapSize") ;
            const Long numStages = (Long) spqr_amd_symbolic_scalar (S, "numStages") ;
            CHECK (Super [nf] == n && Rp [nf] == (Long) spqr_amd_symbolic_scalar (S, "rjsize")) ;
            // assembly maps
            std::vector<Long> Fmap (n + 1), Stair (Rp [nf] + 1) ;
            std::vector<int32_t> Rjmap (RjmapSize + 1), Rimap (RimapSize + 1) ;
            std::vector<gpuqr_sentry> cpuS (anz + 1) ;
            std::vector<double> Sx (anz + 1, 1.0) ;
            spqrgpu_amd_build_assembly_maps (nf, n, Fmap.data ( ), Post, Super, Rp, Rj, Sleft, Sp, Sj, Sx.data ( ), Fm, Cm,
                Childp, Child, Stair.data ( ), Rjmap.data ( ), RjmapOff, Rimap.data ( ), RimapOff, cpuS.data ( )) ;
            Long flops = 0, wsb = 0 ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const Long fn = Rp [f+1] - Rp [f] ;
                CHECK (Fm [f] >= 0 && Cm [f] <= Fm [f] && Parent [f] > f) ;
                flops += gpuqr_front_flops (Fm [f], fn, Stair.data ( ) + Rp [f]) ;
                wsb += gpuqr_front_workspace_bytes (Fm [f], fn) ;
            }
            CHECK (flops >= 0 && wsb >= 0) ;
            if (status == 0)
            {
                // the object an SPQR build makes from its own arrays, and what it must refuse
                int st2 = -1 ;
                spqr_amd_symbolic *S2 = spqr_amd_symbolic_from_arrays (m, n, anz, nf, Sp, Sj, Sleft, Super, Rp, Rj, Parent,
                    Childp, Child, Post, Fm, Cm, RimapOff, RimapSize, RjmapOff, RjmapSize, numStages, Stagingp, StageMap, mem, &st2) ;
                CHECK (S2 != NULL && st2 == 0) ;
                spqr_amd_symbolic_free (S2) ;
                std::vector<Long> badFm (Fm, Fm + nf) ; badFm [0] = -5 ;
                S2 = spqr_amd_symbolic_from_arrays (m, n, anz, nf, Sp, Sj, Sleft, Super, Rp, Rj, Parent,
                    Childp, Child, Post, badFm.data ( ), Cm, RimapOff, RimapSize, RjmapOff, RjmapSize, numStages, Stagingp, StageMap, mem, &st2) ;
                CHECK (S2 == NULL && st2 == 3) ;
            }
            if (!rd)
            {
                // packed R blocks with a dominant diagonal: conversion, products, both triangular solves
                std::vector<Long> Rblock (nf) ;
                std::vector<double> Rstack ;
                std::vector<char> Rdead (n + 1, 0) ;
                Long rank = 0, maxfrank = 0 ;
                for (Long f = 0 ; f < nf ; f++)
                {
                    const Long fp = Super [f+1] - Super [f], fn = Rp [f+1] - Rp [f] ;
                    const Long rm_all = (Fm [f] < fp) ? Fm [f] : fp ;
                    for (Long j = rm_all ; j < fp ; j++) Rdead [Super [f] + j] = 1 ;
                    Rblock [f] = (Long) Rstack.size ( ) ;
                    Long rm = 0 ;
                    for (Long j = 0 ; j < fn ; j++)
                    {
                        const bool live = (j < fp && !Rdead [Super [f] + j]) ;
                        if (live) rm++ ;
                        for (Long i = 0 ; i < rm ; i++)
                            Rstack.push_back ((live && i == rm - 1) ? 4.0 + 0.01 * (double) (j % 5) : 0.25 / (double) (1 + (i + j) % 7)) ;
                    }
                    rank += rm ; if (rm > maxfrank) maxfrank = rm ;
                }
                Rstack.push_back (0) ;
                spqr_amd_numeric num = spqr_amd_numeric ( ) ;
                num.nf = nf ; num.rsize = (Long) Rstack.size ( ) ; num.Rstack = Rstack.data ( ) ; num.Rblock = Rblock.data ( ) ;
                num.Rdead = Rdead.data ( ) ; num.sumfrank = rank ; num.maxfrank = maxfrank ;
                std::vector<Long> Rcp (n + 1) ;
                CHECK (spqr_amd_rconvert (S, &num, Rcp.data ( ), NULL, NULL) == 0) ;
                std::vector<Long> Ri (Rcp [n] + 1) ; std::vector<double> Rx (Rcp [n] + 1) ;
                CHECK (spqr_amd_rconvert (S, &num, Rcp.data ( ), Ri.data ( ), Rx.data ( )) == 0) ;
                std::vector<double> y (n, 1.0), t (rank + 1), z (n + 1), t2 (rank + 1, 0.0), z2 (n + 1, 0.0) ;
                CHECK (spqr_amd_rmult (S, &num, 0, y.data ( ), t.data ( )) == 0) ;
                CHECK (spqr_amd_rmult (S, &num, 1, t.data ( ), z.data ( )) == 0) ;
                for (Long j = 0 ; j < n ; j++) for (Long p = Rcp [j] ; p < Rcp [j+1] ; p++) t2 [Ri [p]] += Rx [p] * y [j] ;
                for (Long j = 0 ; j < n ; j++) for (Long p = Rcp [j] ; p < Rcp [j+1] ; p++) z2 [j] += Rx [p] * t2 [Ri [p]] ;
                double e1 = 0, e2 = 0 ;
                for (Long i = 0 ; i < rank ; i++) e1 = fmax (e1, fabs (t [i] - t2 [i])) ;
                for (Long j = 0 ; j < n ; j++) e2 = fmax (e2, fabs (z [j] - z2 [j])) ;
                CHECK (e1 <= 1e-12 && e2 <= 1e-10) ;
                if (rank == n)
                {
                    // R x = t gives back y; R' w = z gives back t
                    std::vector<double> x (n + 1), w (rank + 1) ;
                    CHECK (spqr_amd_rsolve (S, &num, NULL, n, 1, t.data ( ), rank, x.data ( )) == 0) ;
                    CHECK (spqr_amd_rtsolve (S, &num, 1, z.data ( ), n, w.data ( ), rank) == 0) ;
                    double e3 = 0, e4 = 0 ;
                    for (Long j = 0 ; j < n ; j++) e3 = fmax (e3, fabs (x [j] - y [j])) ;
                    for (Long i = 0 ; i < rank ; i++) e4 = fmax (e4, fabs (w [i] - t [i])) ;
                    CHECK (e3 <= 1e-9 && e4 <= 1e-9) ;
                }
            }
            spqr_amd_symbolic_free (S) ;
        }
        // invalid permutations are refused before anything is built
        if (n > 1)
        {
            std::vector<Long> bad (Q) ; bad [0] = bad [1] ;
            int status = -1 ;
            CHECK (spqr_amd_analyze (m, n, Ap.data ( ), Ai.data ( ), bad.data ( ), -1, 0, &status) == NULL && status == 3) ;
        }
    }
    for (size_t mem : {(size_t) 0, (size_t) 1, (size_t) 50000, (size_t) 1 << 40}) CHECK (gpuqr_max_queue_size (mem) >= 50000) ;
    printf (fails ? "host_asan: %d check(s) failed\n" : "host_asan: all checks passed\n", fails) ;
    return fails ? 1 : 0 ;
}
