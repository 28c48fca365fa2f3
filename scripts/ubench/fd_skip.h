// Timing experiments only (scripts/build_skip_ablation.sh): with -DFD_SKIP_ABLATION every launch site of libfdhip asks whether
// "<file>:<line>:<kernel expression>" contains one of the comma-separated substrings of $FD_SKIP and drops the launch if so - the
// what-if "this kernel class costs nothing" for the four-stream step (results are garbage by design).  Lives outside csrc/ because
// the product library must not reference the process environment (tests/test_abi.py).
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstdio>
inline bool fd_skip_match(const char* file, int line, const char* kern) {
    static const char* pat = getenv("FD_SKIP");
    if (!pat || !*pat) return false;
    const char* base = strrchr(file, '/');
    char site[512];
    snprintf(site, sizeof site, "%s:%d:%s", base ? base + 1 : file, line, kern);
    const char* p = pat;
    while (*p) {
        const char* e = strchr(p, ',');
        const size_t n = e ? (size_t)(e - p) : strlen(p);
        if (n > 0 && n < 256) {
            char one[256];
            memcpy(one, p, n); one[n] = 0;
            if (strstr(site, one)) {
                static const char* verbose = getenv("FD_SKIP_LOG");
                if (verbose) fprintf(stderr, "[fd_skip] %s\n", site);
                return true;
            }
        }
        p += n + (e ? 1 : 0);
    }
    return false;
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, ...) \
    do { static const bool fd_skip_site_ = fd_skip_match(__FILE__, __LINE__, #kern);   /* matched once per site: no host cost per launch */ \
         if (!fd_skip_site_) hipLaunchKernelGGLInternal((kern), __VA_ARGS__); } while (0)
