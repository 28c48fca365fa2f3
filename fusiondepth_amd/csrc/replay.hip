// fd_replay: issue a recorded sequence of libfdhip entry-point calls from ONE host call (include/fdhip.h, "replay").
//
// The Python layer launches ~830 kernels per training step at 10 - 18 us of interpreter / ctypes / autograd time each; for the parts of
// a step that are the same calls on the same shapes every time and need no autograd graph - the Refiner's frozen stage-1 networks
// (refiner.py:299-330 under no_grad), validation forwards - the host records the calls once (fusiondepth_amd/replay.py) and this
// function replays them: a loop over records, each a typed call through the table generated from the ABI's signature list
// (replay_table.inc).  Pointers are recorded as (arena offset | input slot + offset | literal): the caller allocates one arena per
// replay and the records' intermediates live in it, so nothing here allocates or synchronises; every launch goes to the stream given.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include <string.h>

extern "C" {
// every recordable entry point is declared in fdhip.h; the table needs their addresses only
}

namespace {
struct ReplayName { const char* name; const char* sig; };
#define FD_REPLAY_NAMES
const ReplayName g_names[] = {
#include "replay_table.inc"
};
#undef FD_REPLAY_NAMES

inline int replay_one(const fd_call_rec& r, const long long* a) {
#define FD_RP_P(i) (reinterpret_cast<void*>(a[i]))
#define FD_RP_I(i) ((int)a[i])
#define FD_RP_L(i) ((long)a[i])
#define FD_RP_D(i) (__builtin_bit_cast(double, a[i]))
#define FD_RP_F(i) ((float)__builtin_bit_cast(double, a[i]))
    switch (r.fn) {
#define FD_REPLAY_CASES
#include "replay_table.inc"
#undef FD_REPLAY_CASES
        default: break;
    }
    fd_set_error("fd_replay: unknown function index %d", r.fn);
    return -1;
}
}  // namespace

extern "C" int fd_replay_function_count(void) { return FD_REPLAY_NFUNCS; }
extern "C" const char* fd_replay_function_name(int i) { return (i >= 0 && i < FD_REPLAY_NFUNCS) ? g_names[i].name : nullptr; }
extern "C" const char* fd_replay_function_signature(int i) { return (i >= 0 && i < FD_REPLAY_NFUNCS) ? g_names[i].sig : nullptr; }

extern "C" int fd_replay(const fd_call_rec* recs, int n, void* arena, const void* const* inputs, int n_inputs, void* stream) {
    FD_REQUIRE(recs && n >= 0, "fd_replay: bad record list");
    for (int k = 0; k < n; ++k) {
        const fd_call_rec& r = recs[k];
        FD_REQUIRE(r.nargs >= 1 && r.nargs <= FD_REPLAY_MAX_ARGS, "fd_replay: record %d has %d arguments", k, r.nargs);
        long long a[FD_REPLAY_MAX_ARGS];
        for (int i = 0; i < r.nargs; ++i) {
            const int kind = r.kind[i] & 15, slot = r.kind[i] >> 4;
            switch (kind) {
                case 0: a[i] = r.arg[i]; break;                                                             // literal
                case 1: a[i] = (long long)(reinterpret_cast<char*>(arena) + r.arg[i]); break;               // arena + offset
                case 2:
                    FD_REQUIRE(inputs && slot < n_inputs, "fd_replay: record %d refers to input %d of %d", k, slot, n_inputs);
                    a[i] = (long long)(reinterpret_cast<const char*>(inputs[slot]) + r.arg[i]);
                    break;
                case 3: a[i] = (long long)stream; break;
                default: fd_set_error("fd_replay: record %d, argument %d: bad kind %d", k, i, kind); return -1;
            }
        }
        if (int rc = replay_one(r, a)) return rc;
    }
    return 0;
}
