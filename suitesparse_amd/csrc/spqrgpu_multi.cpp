// One process, several GPUs: the assembly tree is mapped onto the devices proportionally to the work,
// every device factorizes its fronts level by level through the subset driver, and the only data that
// crosses devices are contribution blocks, device to device (hipMemcpyPeer over xGMI).  The reference
// has no multi-GPU QR; the hook an SPQR build would use is cc->numGPU
// (CHOLMOD/Include/cholmod_core.h:988), and the loose precedent for "one host thread per device" is the
// subtree Cholesky of this fork (CHOLMOD/GPU/t_initialize_subtree.c:671-678).  It generalises the
// reference's stage / limbo hand-off (SPQR/SPQRGPU/spqrgpu_kernel.cpp:556-611,687-724) from "a later
// stage" to "another device".  suitesparse_amd/multigpu.py is the same orchestration with one process
// per GPU over torch.distributed; the mapping below is that module's map_tree, restated.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <thread>
#include <new>

#include "../../include/spqrgpu_amd.h"
#include "spqr_symbolic.h"

typedef int64_t Long ;

// dense Householder flop estimate of one front (fm-by-fn, fp pivots)
static double front_weight (const spqr_amd_symbolic *S, Long f)
{
    const double fm = (double) S->Fm [f], fn = (double) (S->Rp [f+1] - S->Rp [f]), fp = (double) (S->Super [f+1] - S->Super [f]) ;
    const double k = std::min (fm, fp) ;
    return 4.0 * k * fm * fn - 2.0 * k * k * (fm + fn) + (4.0 / 3.0) * k * k * k + 1.0 ;
}

// Proportional mapping of the assembly tree onto nparts devices.  A forest with a list of parts is split
// in two: the parts in halves, the subtrees into two bins whose weights match the halves (heaviest
// first, each to the bin that is further below its share).  A single subtree with several parts gives
// its root to the first of them and splits its children over all of them; a forest with one part goes to
// that part whole.  A subtree so heavy that it would unbalance any two bins keeps all the parts, and its
// light siblings are dealt to the least loaded parts afterwards.  owner [f] = part that factorizes front
// f; level [f] = number of times the owner changes on the longest path from a leaf up to f.
extern "C" int spqr_amd_map_tree
(
    const spqr_amd_symbolic *S, int nparts, int64_t *owner, int64_t *level, double *top_flops_share, double *speedup_bound
)
{
    if (!S || nparts < 1 || !owner || !level) return 3 ;
    const Long nf = S->nf ;
    try
    {
        std::vector<double> w ((size_t) nf + 1, 0.0), sub ((size_t) nf + 1, 0.0), load ((size_t) nparts, 0.0) ;
        for (Long f = 0 ; f < nf ; f++) w [f] = sub [f] = front_weight (S, f) ;
        for (Long f = 0 ; f < nf ; f++) { const Long p = S->Parent [f] ; if (p < nf) sub [p] += sub [f] ; }
        for (Long f = 0 ; f < nf ; f++) { owner [f] = 0 ; level [f] = 0 ; }
        auto kids = [&] (Long f, std::vector<Long> &out)
        {
            out.clear ( ) ;
            for (Long p = S->Childp [f] ; p < S->Childp [f+1] ; p++) out.push_back (S->Child [p]) ;
        } ;
        std::vector<Long> stack, tmp ;
        auto whole = [&] (Long f, int r)
        {
            stack.assign (1, f) ;
            while (!stack.empty ( ))
            {
                const Long g = stack.back ( ) ; stack.pop_back ( ) ;
                owner [g] = r ;
                for (Long p = S->Childp [g] ; p < S->Childp [g+1] ; p++) stack.push_back (S->Child [p]) ;
            }
            load [r] += sub [f] ;
        } ;
        struct Work { std::vector<Long> kids ; std::vector<int> ranks ; } ;
        std::vector<Work> work ;
        std::vector<std::pair<Long, std::vector<int> > > late ;
        {
            Work w0 ;
            kids (nf, w0.kids) ;
            for (int r = 0 ; r < nparts ; r++) w0.ranks.push_back (r) ;
            work.push_back (w0) ;
        }
        while (!work.empty ( ))
        {
            Work cur ;
            cur.kids.swap (work.back ( ).kids) ; cur.ranks.swap (work.back ( ).ranks) ;
            work.pop_back ( ) ;
            if (cur.kids.empty ( )) continue ;
            if (cur.ranks.size ( ) == 1)
            {
                for (Long k : cur.kids) whole (k, cur.ranks [0]) ;
                continue ;
            }
            std::stable_sort (cur.kids.begin ( ), cur.kids.end ( ), [&] (Long a, Long b) { return sub [a] > sub [b] ; }) ;
            double total = 0 ;
            for (Long k : cur.kids) total += sub [k] ;
            const size_t n1 = (cur.ranks.size ( ) + 1) / 2 ;
            const double share1 = total * (double) n1 / (double) cur.ranks.size ( ) ;
            if (cur.kids.size ( ) == 1 || sub [cur.kids [0]] > 1.25 * share1)
            {
                // one subtree carries the forest: its root goes to the first part, its children get all
                // the parts; light siblings are packed onto the least loaded parts later
                const Long head = cur.kids [0] ;
                owner [head] = cur.ranks [0] ;
                load [cur.ranks [0]] += w [head] ;
                Work nxt ;
                kids (head, nxt.kids) ;
                nxt.ranks = cur.ranks ;
                for (size_t i = 1 ; i < cur.kids.size ( ) ; i++) late.push_back (std::make_pair (cur.kids [i], cur.ranks)) ;
                work.push_back (nxt) ;
                continue ;
            }
            Work b0, b1 ;
            double fill [2] = {0, 0} ;
            const double want [2] = {share1, total - share1} ;
            for (Long k : cur.kids)
            {
                const int b = ((want [0] - fill [0]) >= (want [1] - fill [1])) ? 0 : 1 ;
                (b ? b1 : b0).kids.push_back (k) ;
                fill [b] += sub [k] ;
            }
            b0.ranks.assign (cur.ranks.begin ( ), cur.ranks.begin ( ) + n1) ;
            b1.ranks.assign (cur.ranks.begin ( ) + n1, cur.ranks.end ( )) ;
            work.push_back (b0) ;
            work.push_back (b1) ;
        }
        while (!late.empty ( ))
        {
            const Long k = late.back ( ).first ;
            std::vector<int> ranks = late.back ( ).second ;
            late.pop_back ( ) ;
            int best = ranks [0] ;
            for (int q : ranks) if (load [q] < load [best]) best = q ;
            whole (k, best) ;
        }
        for (Long f = 0 ; f < nf ; f++)
            for (Long p = S->Childp [f] ; p < S->Childp [f+1] ; p++)
            {
                const Long c = S->Child [p] ;
                level [f] = std::max (level [f], level [c] + (owner [c] != owner [f] ? 1 : 0)) ;
            }
        if (top_flops_share || speedup_bound)
        {
            // what the mapping promises from the weights alone: the share of the work in fronts whose
            // subtree spans several parts, and total / critical path when every part runs its calls in
            // level order and a call starts once the calls it takes blocks from have ended
            double total = 0, top = 0 ;
            std::vector<char> spans ((size_t) nf + 1, 0) ;
            Long nlev = 1 ;
            for (Long f = 0 ; f < nf ; f++)
            {
                total += w [f] ;
                nlev = std::max (nlev, level [f] + 1) ;
                for (Long p = S->Childp [f] ; p < S->Childp [f+1] ; p++)
                {
                    const Long c = S->Child [p] ;
                    if (spans [c] || owner [c] != owner [f]) spans [f] = 1 ;
                }
                if (spans [f]) top += w [f] ;
            }
            std::vector<double> cost ((size_t) (nparts * nlev), 0.0), end ((size_t) (nparts * nlev), 0.0) ;
            for (Long f = 0 ; f < nf ; f++) cost [owner [f] * nlev + level [f]] += w [f] ;
            std::vector<std::vector<Long> > need ((size_t) (nparts * nlev)) ;
            for (Long f = 0 ; f < nf ; f++)
            {
                const Long p = S->Parent [f] ;
                if (p < nf && (owner [p] != owner [f] || level [p] != level [f]))
                    need [owner [p] * nlev + level [p]].push_back (owner [f] * nlev + level [f]) ;
            }
            double crit = 0 ;
            for (Long L = 0 ; L < nlev ; L++)
                for (int r = 0 ; r < nparts ; r++)
                {
                    double start = (L > 0) ? end [r * nlev + L - 1] : 0.0 ;
                    for (Long src : need [r * nlev + L]) start = std::max (start, end [src]) ;
                    end [r * nlev + L] = start + cost [r * nlev + L] ;
                    crit = std::max (crit, end [r * nlev + L]) ;
                }
            if (top_flops_share) *top_flops_share = (total > 0) ? top / total : 0.0 ;
            if (speedup_bound) *speedup_bound = (crit > 0) ? total / crit : 1.0 ;
        }
    }
    catch (std::bad_alloc &) { return 1 ; }
    return 0 ;
}

// A copy of the symbolic object with a numeric cache of its own (the cache holds device arrays of ONE
// device, so every device that factorizes on the object needs its own copy).
static spqr_amd_symbolic *clone_symbolic (const spqr_amd_symbolic *S)
{
    spqr_amd_symbolic *C = new spqr_amd_symbolic ( ) ;
    C->m = S->m ; C->n = S->n ; C->anz = S->anz ; C->nf = S->nf ; C->maxfn = S->maxfn ; C->rjsize = S->rjsize ;
    C->RimapSize = S->RimapSize ; C->RjmapSize = S->RjmapSize ; C->numStages = S->numStages ; C->rnz = S->rnz ;
    C->flopcount = S->flopcount ; C->gpuMemorySize = S->gpuMemorySize ; C->do_rank_detection = S->do_rank_detection ;
    C->uid = S->uid ^ ((uint64_t) (uintptr_t) C << 8) ^ ((uint64_t) 1 << 62) ;     // an identity of its own for the schedule cache
    C->Qfill = S->Qfill ; C->PLinv = S->PLinv ; C->Sleft = S->Sleft ; C->Sp = S->Sp ; C->Sj = S->Sj ; C->Super = S->Super ;
    C->Rp = S->Rp ; C->Rj = S->Rj ; C->Parent = S->Parent ; C->Childp = S->Childp ; C->Child = S->Child ; C->Post = S->Post ;
    C->Fm = S->Fm ; C->Cm = S->Cm ; C->RjmapOffsets = S->RjmapOffsets ; C->RimapOffsets = S->RimapOffsets ;
    C->Stagingp = S->Stagingp ; C->StageMap = S->StageMap ; C->FSize = S->FSize ; C->RSize = S->RSize ; C->SSize = S->SSize ;
    C->FOffsets = S->FOffsets ; C->ROffsets = S->ROffsets ; C->SOffsets = S->SOffsets ; C->ColCount = S->ColCount ;
    return C ;
}

// The copies live with the object they were made from: a second factorization on the same symbolic
// object finds their assembly maps, S permutations and schedules where the first left them.
struct MultiCopies { std::vector<spqr_amd_symbolic *> of_device ; } ;
static void release_multi_copies (void *p)
{
    MultiCopies *mc = (MultiCopies *) p ;
    int cur = 0 ;
    (void) hipGetDevice (&cur) ;
    for (size_t d = 0 ; d < mc->of_device.size ( ) ; d++)
        if (mc->of_device [d]) { (void) hipSetDevice ((int) d) ; spqr_amd_symbolic_free (mc->of_device [d]) ; }
    (void) hipSetDevice (cur) ;
    delete mc ;
}

namespace
{
struct Call                     // one subset call: the fronts (part, level)
{
    int part = 0 ; Long level = 0 ;
    std::vector<char> mine ;
    double *cout = NULL ;       // device buffer of the leaving blocks (on the part's device)
    Long last_use = -1 ;        // level of the last parent that reads from cout
    spqr_amd_numeric *num = NULL ;
    int code = 0 ;
} ;
}

// Factorize on ndev devices from one process.  devices [k] = HIP device of part k (NULL: 0 .. ndev-1);
// two parts may name the same device (they then run one after the other there), which is how the path
// is rehearsed on a machine with fewer GPUs than parts.  The result is one numeric object with every R
// block on a pageable host stack, as spqrgpu_amd_factorize would leave it.
extern "C" int spqrgpu_amd_factorize_multi
(
    const spqr_amd_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
    int ndev, const int *devices, spqr_amd_numeric **num_out
)
{
    if (num_out) *num_out = NULL ;
    // (Ap == NULL: Ax holds the values of S = A(P,Q) in row form, SPQR's Sx, which is all an SPQR adapter has)
    if (!S || !Ax || !num_out || (Ap && !Ai && S->anz > 0) || ndev < 1 || ndev > 64) return 3 ;
    if (S->do_rank_detection || (Long) S->Qfill.size ( ) < S->n || (Long) S->PLinv.size ( ) < S->m) return 3 ;
    const Long nf = S->nf, n = S->n ;
    int ndevices_here = 0 ;
    if (hipGetDeviceCount (&ndevices_here) != hipSuccess || ndevices_here < 1) return 2 ;
    std::vector<int> dev ((size_t) ndev) ;
    for (int k = 0 ; k < ndev ; k++)
    {
        dev [k] = devices ? devices [k] : k ;
        if (dev [k] < 0 || dev [k] >= ndevices_here) return 3 ;
    }
    int cur_dev = 0 ;
    (void) hipGetDevice (&cur_dev) ;
    int code = 0 ;
    spqr_amd_numeric *num = NULL ;
    std::vector<spqr_amd_symbolic *> symdev ((size_t) ndevices_here, (spqr_amd_symbolic *) NULL) ;
    std::vector<Call> calls ;
    std::vector<double *> peer_blocks ;                 // blocks copied to another device: freed at the end of their level of use
    std::vector<Long> peer_until ;
    try
    {
        std::vector<Long> owner ((size_t) std::max ((Long) 1, nf)), level ((size_t) std::max ((Long) 1, nf)) ;
        if ((code = spqr_amd_map_tree (S, ndev, owner.data ( ), level.data ( ), NULL, NULL)) != 0) throw code ;
        Long nlev = 0 ;
        for (Long f = 0 ; f < nf ; f++) nlev = std::max (nlev, level [f] + 1) ;
        if (!S->multi_copies) { S->multi_copies = new MultiCopies ( ) ; S->multi_release = release_multi_copies ; }
        MultiCopies *mc = (MultiCopies *) S->multi_copies ;
        if (mc->of_device.size ( ) < (size_t) ndevices_here) mc->of_device.resize ((size_t) ndevices_here, (spqr_amd_symbolic *) NULL) ;
        for (int k = 0 ; k < ndev ; k++)
        {
            if (!mc->of_device [dev [k]]) mc->of_device [dev [k]] = clone_symbolic (S) ;
            symdev [dev [k]] = mc->of_device [dev [k]] ;
        }
        // peer access between the devices that will exchange blocks (an error here only means the copies
        // go through the host, or that access was enabled before)
        for (int a = 0 ; a < ndev ; a++)
            for (int b = 0 ; b < ndev ; b++)
                if (dev [a] != dev [b])
                {
                    int can = 0 ;
                    if (hipDeviceCanAccessPeer (&can, dev [a], dev [b]) == hipSuccess && can)
                    {
                        (void) hipSetDevice (dev [a]) ;
                        (void) hipDeviceEnablePeerAccess (dev [b], 0) ;
                        (void) hipGetLastError ( ) ;
                    }
                }

        // where the contribution block of front c waits for its parent: a device address on the
        // parent's device
        std::vector<const double *> held ((size_t) std::max ((Long) 1, nf), (const double *) NULL) ;
        std::vector<const double *> limbo ((size_t) std::max ((Long) 1, nf), (const double *) NULL) ;
        calls.reserve ((size_t) (nlev * ndev)) ;
        for (Long L = 0 ; L < nlev ; L++)
        {
            const size_t first = calls.size ( ) ;
            for (int k = 0 ; k < ndev ; k++)
            {
                Call c ;
                c.part = k ; c.level = L ;
                c.mine.assign ((size_t) std::max ((Long) 1, nf), 0) ;
                bool any = false ;
                for (Long f = 0 ; f < nf ; f++) if (owner [f] == k && level [f] == L) { c.mine [f] = 1 ; any = true ; }
                if (any) calls.push_back (std::move (c)) ;
            }
            // the calls of a level are independent; those on different devices run side by side, one host
            // thread per device (the library keeps per-device state for one calling thread at a time)
            std::vector<std::vector<size_t> > by_dev ((size_t) ndevices_here) ;
            for (size_t i = first ; i < calls.size ( ) ; i++) by_dev [dev [calls [i].part]].push_back (i) ;
            auto run_device = [&] (int d)
            {
                if (hipSetDevice (d) != hipSuccess) { for (size_t i : by_dev [d]) calls [i].code = 2 ; return ; }
                for (size_t i : by_dev [d])
                {
                    Call &c = calls [i] ;
                    const Long csize = spqrgpu_amd_subset_csize (symdev [d], c.mine.data ( )) ;
                    if (csize > 0 && hipMalloc ((void **) &c.cout, (size_t) csize * sizeof (double)) != hipSuccess)
                    {
                        (void) hipGetLastError ( ) ; c.cout = NULL ; c.code = 1 ; return ;
                    }
                    // (limbo holds, for the children of this call's fronts that another call made, the
                    // address where their block waits on this device; entries of other fronts are not read)
                    c.code = spqrgpu_amd_factorize_subset_dev (symdev [d], Ap, Ai, Ax, c.mine.data ( ), limbo.data ( ), c.cout, &c.num) ;
                    if (c.code != 0) return ;
                }
            } ;
            for (Long f = 0 ; f < nf ; f++) limbo [f] = held [f] ;
            {
                std::vector<std::thread> th ;
                std::vector<int> serial ;
                int main_dev = -1 ;
                for (int d = 0 ; d < ndevices_here ; d++)
                {
                    if (by_dev [d].empty ( )) continue ;
                    if (main_dev < 0) { main_dev = d ; continue ; }
                    try { th.emplace_back (run_device, d) ; } catch (...) { serial.push_back (d) ; }
                }
                if (main_dev >= 0) run_device (main_dev) ;
                for (auto &t : th) t.join ( ) ;
                for (int d : serial) run_device (d) ;
            }
            for (size_t i = first ; i < calls.size ( ) ; i++) if (calls [i].code != 0) throw calls [i].code ;

            // the blocks that left this level, to the device of whoever factorizes the parent
            for (size_t i = first ; i < calls.size ( ) ; i++)
            {
                Call &c = calls [i] ;
                if (!c.num || !c.num->Cblock) continue ;
                const int dsrc = dev [c.part] ;
                for (Long f = 0 ; f < nf ; f++)
                {
                    if (!c.mine [f] || c.num->Cblock [f] < 0) continue ;
                    const Long p = S->Parent [f] ;
                    if (p >= nf) continue ;
                    const Long csz = S->Cm [f] * ((S->Rp [f+1] - S->Rp [f]) - (S->Super [f+1] - S->Super [f])) ;
                    const double *src = c.cout + c.num->Cblock [f] ;
                    const int ddst = dev [owner [p]] ;
                    if (ddst == dsrc || csz == 0)
                    {
                        held [f] = src ;
                        c.last_use = std::max (c.last_use, level [p]) ;
                    }
                    else
                    {
                        double *dst = NULL ;
                        if (hipSetDevice (ddst) != hipSuccess || hipMalloc ((void **) &dst, (size_t) csz * sizeof (double)) != hipSuccess) throw 1 ;
                        peer_blocks.push_back (dst) ; peer_until.push_back (level [p]) ;
                        if (hipMemcpyPeer (dst, ddst, src, dsrc, (size_t) csz * sizeof (double)) != hipSuccess) throw 2 ;
                        held [f] = dst ;
                    }
                }
            }
            // device memory whose last reader was this level
            for (size_t i = 0 ; i < calls.size ( ) ; i++)
                if (calls [i].cout && calls [i].last_use <= L && calls [i].level <= L)
                {
                    (void) hipSetDevice (dev [calls [i].part]) ;
                    (void) hipFree (calls [i].cout) ; calls [i].cout = NULL ;
                }
            for (size_t i = 0 ; i < peer_blocks.size ( ) ; i++)
                if (peer_blocks [i] && peer_until [i] <= L) { (void) hipFree (peer_blocks [i]) ; peer_blocks [i] = NULL ; }
        }

        // ---- one numeric object: every R block on one host stack, in front order ----------------------
        num = (spqr_amd_numeric *) qr_host_calloc (1, sizeof (spqr_amd_numeric)) ;
        if (!num) throw 1 ;
        num->nf = nf ;
        num->rsize = S->rnz ;
        num->Rstack = (double *) qr_host_malloc ((size_t) std::max ((Long) 1, S->rnz) * sizeof (double)) ;
        num->Rblock = (Long *) qr_host_malloc ((size_t) std::max ((Long) 1, nf) * sizeof (Long)) ;
        num->Rdead = (char *) qr_host_calloc ((size_t) std::max ((Long) 1, n), 1) ;
        if (!num->Rstack || !num->Rblock || !num->Rdead) throw 1 ;
        num->RstackPinned = 0 ;
        std::vector<int> call_of ((size_t) std::max ((Long) 1, nf), -1) ;
        for (size_t i = 0 ; i < calls.size ( ) ; i++) for (Long f = 0 ; f < nf ; f++) if (calls [i].mine [f]) call_of [f] = (int) i ;
        Long head = 0 ;
        for (Long f = 0 ; f < nf ; f++)
        {
            num->Rblock [f] = -1 ;
            if (call_of [f] < 0) continue ;
            const spqr_amd_numeric *src = calls [call_of [f]].num ;
            if (!src || src->Rblock [f] < 0) continue ;
            const Long fn = S->Rp [f+1] - S->Rp [f] ;
            const Long rk = std::min (S->Fm [f], S->Super [f+1] - S->Super [f]) ;
            const Long sz = rk * (rk + 1) / 2 + rk * (fn - rk) ;
            if (head + sz > S->rnz) throw 2 ;
            memcpy (num->Rstack + head, src->Rstack + src->Rblock [f], (size_t) sz * sizeof (double)) ;
            num->Rblock [f] = head ;
            head += sz ;
        }
        for (size_t i = 0 ; i < calls.size ( ) ; i++)
        {
            const spqr_amd_numeric *src = calls [i].num ;
            if (!src) continue ;
            for (Long j = 0 ; j < n ; j++) if (src->Rdead [j]) num->Rdead [j] = 1 ;
            num->sumfrank += src->sumfrank ;
            num->maxfrank = std::max (num->maxfrank, src->maxfrank) ;
            num->gpuKernelTime += src->gpuKernelTime ;
            num->gpuFlops += src->gpuFlops ;
            num->gpuNumKernelLaunches += src->gpuNumKernelLaunches ;
            num->numStagesRun += src->numStagesRun ;
            num->assembleBytes += src->assembleBytes ;
        }
    }
    catch (std::bad_alloc &) { code = 1 ; }
    catch (int c) { code = c ; }
    for (size_t i = 0 ; i < calls.size ( ) ; i++)
    {
        if (calls [i].num) spqr_amd_numeric_free (calls [i].num) ;
        if (calls [i].cout) { (void) hipSetDevice (dev [calls [i].part]) ; (void) hipFree (calls [i].cout) ; }
    }
    for (size_t i = 0 ; i < peer_blocks.size ( ) ; i++) if (peer_blocks [i]) (void) hipFree (peer_blocks [i]) ;
    // (the per-device copies stay with S for its next factorization)
    (void) hipSetDevice (cur_dev) ;
    if (code != 0) { if (num) spqr_amd_numeric_free (num) ; return code ; }
    *num_out = num ;
    return 0 ;
}
