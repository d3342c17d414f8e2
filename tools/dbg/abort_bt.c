// LD_PRELOAD shim for chasing a sporadic abort() in a process whose stderr is captured: on SIGABRT, write the native backtrace of
// the aborting thread to $ABORT_BT_FILE, then let the signal take its course.  (pytest's faulthandler installs its own handler
// later; it restores this one and re-raises once it has dumped the Python stacks.)
//   gcc -O1 -g -shared -fPIC -o tools/dbg/abort_bt.so tools/dbg/abort_bt.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_fd = -1;

static void on_abort(int sig)
{
    void* bt[96];
    const int n = backtrace(bt, 96);
    if (g_fd >= 0) {
        static const char head[] = "--- SIGABRT: native backtrace of the aborting thread ---\n";
        if (write(g_fd, head, sizeof head - 1) < 0) {}
        backtrace_symbols_fd(bt, n, g_fd);
        // the memory map, so that module offsets can be resolved
        const int m = open("/proc/self/maps", O_RDONLY);
        if (m >= 0) {
            static char buf[1 << 16];
            static const char mh[] = "--- /proc/self/maps (executable mappings) ---\n";
            if (write(g_fd, mh, sizeof mh - 1) < 0) {}
            ssize_t k;
            while ((k = read(m, buf, sizeof buf)) > 0)
                if (write(g_fd, buf, (size_t)k) < 0) break;
            close(m);
        }
        fsync(g_fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void abort_bt_init(void)
{
    const char* p = getenv("ABORT_BT_FILE");
    if (!p || !*p) return;
    g_fd = open(p, O_WRONLY | O_CREAT | O_APPEND, 0644);
    void* warm[4];
    (void)backtrace(warm, 4);       // loads libgcc's unwinder now, not inside the handler
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, NULL);
}
