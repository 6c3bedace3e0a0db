/* exitprobe.c -- diagnostic helper for tools/exit_hang_probe.sh (NOT part of the product): loaded by a probe script via
 * ctypes, it (1) registers an atexit marker, (2) installs a SIGUSR1 handler that writes the native backtrace of the
 * thread that receives the signal to stderr.  A process that hangs inside exit() (after the Python interpreter is gone)
 * can thus still be asked where it is.   gcc -shared -fPIC -O1 -o exitprobe.so exitprobe.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void say(const char *s) { (void)!write(2, s, strlen(s)); }

static void on_usr1(int sig) {
    void *bt[64];
    (void)sig;
    say("[exitprobe] SIGUSR1: native backtrace of the receiving thread\n");
    int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    say("[exitprobe] end of backtrace\n");
}

static void at_exit_marker(void) { say("[exitprobe] atexit handler registered at load time runs now\n"); }

__attribute__((destructor)) static void dtor_marker(void) { say("[exitprobe] exitprobe.so destructor\n"); }

void exitprobe_install(void) {
    void *warm[4];
    backtrace(warm, 4); /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_usr1;
    sigaction(SIGUSR1, &sa, NULL);
    atexit(at_exit_marker);
    say("[exitprobe] installed\n");
}
