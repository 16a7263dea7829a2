/* LD_PRELOAD helper for debugging on the GPU box (no gdb there): prints a native backtrace on SIGSEGV.
 *   gcc -shared -fPIC -o /tmp/segv_trace.so tools/segv_trace.c && LD_PRELOAD=/tmp/segv_trace.so python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <string.h>
static int out_fd = 2;   /* a file of its own: pytest redirects fd 2 */
static void handler(int sig, siginfo_t* si, void* ctx) {
    void* frames[64];
    (void)ctx;
    char msg[128];
    int len = snprintf(msg, sizeof msg, "\n*** signal %d at address %p; native backtrace:\n", sig, si->si_addr);
    if (write(out_fd, msg, (size_t)len) < 0) {}
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, out_fd);
    _exit(139);
}
__attribute__((constructor)) static void install(void) {
    const char* path = getenv("SEGV_TRACE_FILE");
    if (path) { int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) out_fd = fd; }
    struct sigaction sa;
    sa.sa_sigaction = handler;
    sigemptyset(&sa.sa_mask);
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, NULL);
}
