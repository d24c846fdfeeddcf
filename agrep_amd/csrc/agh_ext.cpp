// agh_ext.cpp -- the second library.  libagrep_hip.so holds what the headline queries run: the sample filter (sweep,
// fused count-only kernels, verify kernels for k <= 4), record output, the one-pass -f kernels and the host side
// -- ~11 MB.  The engines behind the filter -- the automaton over every byte (two word widths: 11 MB), the table engine
// (6.6 MB), the multi-pattern sweep / verify kernels and the verify kernels for k = 5..8 -- live in
// libagrep_hip_engines.so next to it, opened with dlopen when a query first needs one of them: a `agrep-hip -2 -c`
// process maps and registers a third of the code objects it used to (profiles/r05_startup.log).
// The functions below are the launchers of agh_launch.h that moved: same names, forwarded through the loaded library.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "agh_internal.h"

namespace {
std::once_flag g_once;
void *g_handle = nullptr;
char g_err[384] = "";

void open_engines()
{
    // AGH_ENGINES_PATH, else the file next to this library that carries its name
    std::string path;
    const char *e = getenv("AGH_ENGINES_PATH");
    if (e && *e) {
        path = e;
    } else {
        Dl_info info;
        if (dladdr((const void *)&open_engines, &info) && info.dli_fname) {
            // this library's own file name with "_engines" in front of ".so": an A/B build loaded through AGH_LIB_PATH
            // (libagrep_hip_<variant>.so) opens the engines built with ITS flags, not the default build's
            path = info.dli_fname;
            const size_t dot = path.rfind(".so");
            if (dot != std::string::npos && dot + 3 == path.size()) path.insert(dot, "_engines");
            else path += "_engines.so";
        } else {
            path = "libagrep_hip_engines.so";
        }
    }
    agh_timeline("dlopen libagrep_hip_engines.so ...");
    g_handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    agh_timeline("... engines library loaded");
    if (!g_handle) snprintf(g_err, sizeof(g_err), "%s", dlerror());
}

void *engine_symbol(const char *mangled)
{
    std::call_once(g_once, open_engines);
    if (!g_handle) {
        // (agh_need_engines() in front of every such launch reports this as an ordinary error; a caller that got here
        // skipped it -- there is no fallback to fall back to)
        fprintf(stderr, "libagrep_hip: cannot load libagrep_hip_engines.so (%s)\n", g_err);
        abort();
    }
    void *f = dlsym(g_handle, mangled);
    if (!f) {
        fprintf(stderr, "libagrep_hip: libagrep_hip_engines.so lacks %s\n", mangled);
        abort();
    }
    return f;
}
}   // namespace

// 0, or -1 / errno 123 with the loader's message: called by the scan paths before they launch an engine that lives in
// the second library (no CPU path and no other engine stands in for a missing one)
int agh_need_engines()
{
    std::call_once(g_once, open_engines);
    if (!g_handle) return fail("cannot load libagrep_hip_engines.so: %s", g_err);
    return 0;
}

#define AGH_FORWARD(name, mangled, params, args)                         \
    void name params                                                     \
    {                                                                    \
        typedef void (*fn_t) params;                                     \
        static fn_t f = (fn_t)engine_symbol(mangled);                    \
        f args;                                                          \
    }

AGH_FORWARD(agh_launch_fullscan, "_Z19agh_launch_fullscanRK13agh_scan_argsP12ihipStream_t",
            (const agh_scan_args &a, hipStream_t st), (a, st))
AGH_FORWARD(agh_launch_tablescan, "_Z20agh_launch_tablescanRK13agh_scan_argsP12ihipStream_t",
            (const agh_scan_args &a, hipStream_t st), (a, st))
AGH_FORWARD(agh_launch_unmatched, "_Z20agh_launch_unmatchedRK13agh_scan_argsP12ihipStream_t",
            (const agh_scan_args &a, hipStream_t st), (a, st))
AGH_FORWARD(agh_launch_sweep_multi, "_Z22agh_launch_sweep_multiRK14agh_sweep_argsP12ihipStream_t",
            (const agh_sweep_args &a, hipStream_t st), (a, st))
AGH_FORWARD(agh_launch_dense_multi, "_Z22agh_launch_dense_multiRK14agh_sweep_argsRK13agh_multi_devRK9agh_marksP12ihipStream_t",
            (const agh_sweep_args &a, const agh_multi_dev &m, const agh_marks &mk, hipStream_t st), (a, m, mk, st))
AGH_FORWARD(agh_launch_verify_multi, "_Z23agh_launch_verify_multiRK13agh_scan_argsRK13agh_multi_devbP12ihipStream_t",
            (const agh_scan_args &a, const agh_multi_dev &m, bool lean, hipStream_t st), (a, m, lean, st))
void agh_launch_verify_k5(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k6(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k7(const agh_scan_args &, int, hipStream_t);
void agh_launch_verify_k8(const agh_scan_args &, int, hipStream_t);
AGH_FORWARD(agh_launch_verify_k5, "_Z20agh_launch_verify_k5RK13agh_scan_argsiP12ihipStream_t",
            (const agh_scan_args &a, int what, hipStream_t st), (a, what, st))
AGH_FORWARD(agh_launch_verify_k6, "_Z20agh_launch_verify_k6RK13agh_scan_argsiP12ihipStream_t",
            (const agh_scan_args &a, int what, hipStream_t st), (a, what, st))
AGH_FORWARD(agh_launch_verify_k7, "_Z20agh_launch_verify_k7RK13agh_scan_argsiP12ihipStream_t",
            (const agh_scan_args &a, int what, hipStream_t st), (a, what, st))
AGH_FORWARD(agh_launch_verify_k8, "_Z20agh_launch_verify_k8RK13agh_scan_argsiP12ihipStream_t",
            (const agh_scan_args &a, int what, hipStream_t st), (a, what, st))
