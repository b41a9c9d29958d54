// Development aid: print a native backtrace on SIGSEGV (load with ctypes before the code under test).
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <unistd.h>
static void handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    fprintf(stderr, "== native backtrace (signal %d)\n", sig);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
__attribute__((constructor)) static void install(void) { signal(SIGSEGV, handler); }
