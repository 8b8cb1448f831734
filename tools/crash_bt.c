// Diagnosis aid for the soaks (not part of the library): a SIGSEGV / SIGABRT handler that writes the NATIVE stack of the faulting
// thread (python's faulthandler shows python frames only). Build: gcc -shared -fPIC -O1 -o gpurun_out/crash_bt.so tools/crash_bt.c
// and load it first thing (ctypes.CDLL(...).crash_bt_install()); frames inside libutopian_hip.so print as lib(+offset) and
// resolve with llvm-symbolizer against the same build.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static void on_crash(int sig) {
   void* frames[64];
   char head[96];
   int n = snprintf(head, sizeof head, "\n== crash_bt: signal %d in thread %ld (process %d)\n", sig, (long)syscall(SYS_gettid), (int)getpid());
   if (n > 0) (void)!write(2, head, (size_t)n);
   int depth = backtrace(frames, 64);
   backtrace_symbols_fd(frames, depth, 2);
   signal(sig, SIG_DFL);
   raise(sig);
}

void crash_bt_install(void) {
   void* warm[4];
   (void)backtrace(warm, 4);  // loads libgcc's unwinder now, not inside the handler
   struct sigaction sa;
   memset(&sa, 0, sizeof sa);
   sa.sa_handler = on_crash;
   sa.sa_flags = SA_NODEFER | SA_RESETHAND;
   sigaction(SIGSEGV, &sa, 0);
   sigaction(SIGABRT, &sa, 0);
   sigaction(SIGBUS, &sa, 0);
}
