// rccl_bind.hpp — the library's RCCL entry points, resolved at run time from the ONE RCCL the process has mapped.
//
// Why not a DT_NEEDED entry: libwindtunnel.so used to be linked with -lrccl and a run path of /opt/rocm/lib.  A process that loaded it before
// PyTorch then held two RCCL copies — ROCm's, pulled in by this library, and the one the torch wheel brings and loads by path — and which of them a
// wt_comm_* call reached depended on the import order.  Two RCCL instances in one process are two sets of bootstrap threads,
// proxy state and IPC handles; the first contact with an N-GPU box should not be where that is found out.  So:
//   * no link-time dependency on RCCL: loading this library maps no librccl at all;
//   * the first wt_comm_* call binds, in this order: the ncclXxx symbols already in the global scope (PyTorch's copy when torch is imported —
//     it is loaded RTLD_GLOBAL — or an LD_PRELOADed stand-in, tests/_rccl_stub); else the single librccl.so* found in /proc/self/maps; else
//     dlopen("librccl.so.1") through this library's run path (/opt/rocm/lib) for processes without torch;
//   * TWO different librccl.so* files in /proc/self/maps at that moment is an error (WT_ERR_RCCL naming both), not a choice;
//   * what was bound — version and path — is part of wt_version() and of every wt_comm_* error message.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace wt {

struct RcclApi {
    decltype(&::ncclGetVersion) GetVersion = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommCount) CommCount = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    bool bound = false;
    int version = 0;
    std::string path;            // the shared object the bound ncclSend lives in (dladdr)
    std::string how;             // "global scope" / "mapped copy" / "dlopen"
    std::string error;           // why binding failed (empty when bound)
};

// distinct librccl.so* files mapped into this process (an LD_PRELOADed stand-in is called something else on purpose: librccl_stub.so)
static inline std::vector<std::string> rccl_mapped_copies()
{
    std::vector<std::string> out;
    FILE *fp = fopen("/proc/self/maps", "r");
    if (!fp) return out;
    char line[4608];
    while (fgets(line, sizeof(line), fp)) {
        char *p = strchr(line, '/');
        if (!p) continue;
        size_t n = strlen(p);
        while (n && (p[n - 1] == '\n' || p[n - 1] == ' ')) p[--n] = 0;
        const char *base = strrchr(p, '/');
        base = base ? base + 1 : p;
        if (strncmp(base, "librccl.so", 10) != 0) continue;
        const char *q = base + 10;                                   // "", ".1", ".1.0.70200", ...
        bool ok = true;
        for (; *q; q++) ok = ok && (*q == '.' || (*q >= '0' && *q <= '9'));
        if (!ok) continue;
        char real[4096];
        const std::string s = realpath(p, real) ? std::string(real) : std::string(p);
        bool seen = false;
        for (const std::string &o : out) seen = seen || o == s;
        if (!seen) out.push_back(s);
    }
    fclose(fp);
    return out;
}

static inline RcclApi &rccl_api_storage() { static RcclApi a; return a; }

// Binds once; later calls return the same verdict.  Returns the API with `bound` set, or with `error` filled.
static inline const RcclApi &rccl_api()
{
    static std::once_flag once;
    RcclApi &a = rccl_api_storage();
    std::call_once(once, [&a]() {
        const std::vector<std::string> copies = rccl_mapped_copies();
        if (copies.size() > 1) {
            a.error = "two RCCL libraries are mapped into this process (" + copies[0] + " and " + copies[1] +
                      "): refusing to pick one — load libwindtunnel.so AFTER importing torch (airfoil_cfd_tool_amd does), or keep a second librccl out of the process";
            return;
        }
        void *handle = nullptr;
        if (dlsym(RTLD_DEFAULT, "ncclSend") && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
            handle = RTLD_DEFAULT;
            a.how = "symbols already in the global scope";
        } else if (copies.size() == 1) {
            handle = dlopen(copies[0].c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            a.how = "the copy already mapped";
            if (!handle) { a.error = "librccl is mapped (" + copies[0] + ") but dlopen(RTLD_NOLOAD) failed: " + std::string(dlerror() ? dlerror() : "?"); return; }
        } else {
            handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            a.how = "dlopen(\"librccl.so.1\") through this library's run path";
            if (!handle) { a.error = "no RCCL in this process and dlopen(\"librccl.so.1\") failed: " + std::string(dlerror() ? dlerror() : "?"); return; }
        }
        const char *missing = nullptr;
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(handle, name);
            if (!p && !missing) missing = name;
            return p;
        };
        a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(sym("ncclGetVersion"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        if (missing) { a.error = std::string("the RCCL found (") + a.how + ") lacks " + missing; return; }
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(a.Send), &info) && info.dli_fname) {
            char real[4096];
            a.path = realpath(info.dli_fname, real) ? real : info.dli_fname;
        } else a.path = "?";
        int v = 0;
        if (a.GetVersion(&v) == ncclSuccess) a.version = v;
        // the copy bound must be the copy mapped (when one is): a stand-in in front of it (LD_PRELOAD) is the caller's explicit choice and is named as such
        a.bound = true;
    });
    return a;
}

// "RCCL 2.26.6 at /path/librccl.so (symbols already in the global scope)" / "RCCL not bound yet" / "RCCL unavailable: ..."
static inline std::string rccl_describe(bool bind_now)
{
    RcclApi &s = rccl_api_storage();
    if (!bind_now && !s.bound && s.error.empty()) return "RCCL: not bound yet (resolved at the first wt_comm_* call)";
    const RcclApi &a = rccl_api();
    if (!a.bound) return "RCCL unavailable: " + a.error;
    char v[64];
    snprintf(v, sizeof(v), "%d.%d.%d", a.version / 10000, (a.version / 100) % 100, a.version % 100);
    return std::string("RCCL ") + v + " at " + a.path + " (" + a.how + ")";
}

}  // namespace wt

// the call sites keep RCCL's own names
#define ncclGetErrorString (wt::rccl_api().GetErrorString)
#define ncclGetUniqueId (wt::rccl_api().GetUniqueId)
#define ncclCommInitRank (wt::rccl_api().CommInitRank)
#define ncclCommCount (wt::rccl_api().CommCount)
#define ncclCommDestroy (wt::rccl_api().CommDestroy)
#define ncclGroupStart (wt::rccl_api().GroupStart)
#define ncclGroupEnd (wt::rccl_api().GroupEnd)
#define ncclSend (wt::rccl_api().Send)
#define ncclRecv (wt::rccl_api().Recv)
#define ncclAllReduce (wt::rccl_api().AllReduce)
