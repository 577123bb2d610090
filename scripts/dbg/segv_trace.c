/* Debug aid: loading this library installs a SIGSEGV handler that prints the native call stack
 * (module + offset per frame, for addr2line) to stderr and exits.  Load it with ctypes.CDLL
 * (tests/mgpu_worker.py does when MGPU_SEGV_TRACE names the built library).
 *   gcc -shared -fPIC -O1 -o scripts/dbg/libsegv_trace.so scripts/dbg/segv_trace.c */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_segv (int sig)
{
    void *frames [64] ;
    int n = backtrace (frames, 64) ;
    static const char msg [] = "\n== native stack at SIGSEGV ==\n" ;
    (void) !write (2, msg, sizeof (msg) - 1) ;
    backtrace_symbols_fd (frames, n, 2) ;
    _exit (139) ;
}
__attribute__((constructor)) static void install (void)
{
    struct sigaction sa ;
    sa.sa_handler = on_segv ;
    sigemptyset (&sa.sa_mask) ;
    sa.sa_flags = 0 ;
    sigaction (SIGSEGV, &sa, 0) ;
}
