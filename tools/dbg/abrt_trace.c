// abrt_trace.c -- debugging aid for crashes inside native code under pytest: on SIGABRT / SIGSEGV it writes the faulting
// context (RIP, RSP), a raw dump of the stack and /proc/self/maps to SADBM_ABRT_TRACE_FILE with async-signal-safe calls only
// (glibc's backtrace() allocates, which fails on a corrupted heap); tools/dbg/resolve_trace.py turns the dump into
// library+offset / symbol candidates offline.  Loaded by tests/conftest.py when SADBM_ABRT_TRACE=1.
//   gcc -O1 -g -shared -fPIC -o tools/dbg/libabrt_trace.so tools/dbg/abrt_trace.c
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

static struct sigaction prev_abrt, prev_segv;
static int out_fd = 2;
static volatile int once;
static int probe[2] = {-1, -1};

static void put(const char *s) { (void)!write(out_fd, s, strlen(s)); }
static void puthex(uint64_t v)
{
    char b[19]; b[0] = '0'; b[1] = 'x';
    for (int i = 0; i < 16; ++i) { int d = (v >> (60 - 4 * i)) & 15; b[2 + i] = d < 10 ? '0' + d : 'a' + d - 10; }
    b[18] = '\n';
    (void)!write(out_fd, b, 19);
}

static void on_fatal(int sig, siginfo_t *info, void *ucv)
{
    if (once) { signal(sig, SIG_DFL); raise(sig); return; }
    once = 1;
    ucontext_t *uc = (ucontext_t *)ucv;
    put(sig == SIGSEGV ? "SIGNAL SEGV\n" : "SIGNAL ABRT\n");
    put("ADDR\n"); puthex((uint64_t)(uintptr_t)info->si_addr);
    put("RIP\n"); puthex((uint64_t)uc->uc_mcontext.gregs[REG_RIP]);
    uint64_t rsp = (uint64_t)uc->uc_mcontext.gregs[REG_RSP];
    put("RSP\n"); puthex(rsp);
    put("STACK\n");
    // the stack of the faulting thread, upwards from RSP; stop at an unreadable page (write() returns EFAULT instead of faulting)
    const uint64_t *p = (const uint64_t *)(rsp & ~7ull);
    for (int i = 0; i < 6000; ++i) {
        if (i == 0 || ((uintptr_t)(p + i) & 4095) == 0) {       // a pipe write copies from user memory: EFAULT instead of a fault
            uint64_t sink;
            if (probe[1] < 0 || write(probe[1], p + i, 8) != 8) break;
            (void)!read(probe[0], &sink, 8);
        }
        puthex(p[i]);
    }
    put("MAPS\n");
    int m = open("/proc/self/maps", O_RDONLY);
    if (m >= 0) { char buf[4096]; ssize_t n; while ((n = read(m, buf, sizeof buf)) > 0) (void)!write(out_fd, buf, (size_t)n); close(m); }
    put("END\n");
    fsync(out_fd);
    struct sigaction *pv = sig == SIGSEGV ? &prev_segv : &prev_abrt;
    if (pv->sa_flags & SA_SIGINFO) { if (pv->sa_sigaction) pv->sa_sigaction(sig, info, ucv); }
    else if (pv->sa_handler && pv->sa_handler != SIG_DFL && pv->sa_handler != SIG_IGN) pv->sa_handler(sig);
    signal(sig, SIG_DFL);
    raise(sig);
}

void abrt_trace_install(void)
{
    struct sigaction sa;
    const char *path = getenv("SADBM_ABRT_TRACE_FILE");
    if (path) { int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) out_fd = fd; }
    if (pipe(probe) != 0) probe[0] = probe[1] = -1;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
    sigaction(SIGABRT, &sa, &prev_abrt);
    sigaction(SIGSEGV, &sa, &prev_segv);
}
