// Process-wide kernel-selection thresholds (include/fdhip.h: fd_tuning).  The ONLY mutable state of the library besides the
// last-error string; the library itself never reads the environment - `fusiondepth_amd/tuning.py` (host side) maps the FD_*
// variables of the A/B scripts onto fd_set_tuning once at import.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include <string.h>

namespace {
fd_tuning make_defaults() {
    fd_tuning t;
    memset(&t, 0, sizeof(t));
    t.size = (int)sizeof(fd_tuning);
    t.wino_fwd = 1; t.wino_wgrad = 1;
    t.wino_fwd_2d_min = 65536; t.wino_fwd_2dp_min_wgs = 160; t.wino_fwd_2dp_dma = 1; t.wino_fwd_2dp_deep = 0; t.wino_wgrad_2d = 2;
    t.wino_target = 384; t.wino_wgrad_target = 256;      // alone on the GPU 768 is best; inside the step 256 - 384 (less slab traffic; round 4: profiles/round4_targets.log)
    t.conv_target = 768; t.wgrad_target = 768;
    t.conv_c1 = 1; t.conv_n16_min_pixels = 16384;
    t.reflect_ring = 1; t.reflect_wino = 1; t.reflect_wino_min_pixels = 1; t.reflect_wino_padded_max = 4096;
    t.force_cfg = -1; t.force_splits = 1;
    t.stem7 = 1;
    t.log = 0;
    t.wino_fwd_2d_m128 = 1;
    t.grp_tile64_below = 0;
    t.wino_wgrad_xcd_few = 1; t.wino_fwd_halfm = 1; t.wino_wgrad_halfm = 1;
    t.limb_1x1 = 1; t.limb_depth = 2; t.limb_target = 256; t.limb_split_max_out = 4194304; t.limb_wgrad_target = 256;
    t.limb_conv = 1; t.wino_wgrad_limb = 2; t.wino_fwd_limb = 0;
    t.wino_min_cout = 32; t.wino_wgrad_min_cout = 32;     // the decoder's 32-channel blocks with half a tile idle: profiles/round5_decoder_m32_time.log
    return t;
}
fd_tuning g_tuning = make_defaults();
long g_generation = 0;
}  // namespace

const fd_tuning& fd_tun() { return g_tuning; }

extern "C" void fd_tuning_defaults(fd_tuning* t) {
    if (t) *t = make_defaults();
}

extern "C" int fd_set_tuning(const fd_tuning* t) {
    FD_REQUIRE(t, "fd_set_tuning: NULL");
    FD_REQUIRE(t->size >= (int)(2 * sizeof(int)) && t->size <= 4096, "fd_set_tuning: bad size field %d", t->size);
    fd_tuning n = make_defaults();
    const size_t k = (size_t)t->size < sizeof(fd_tuning) ? (size_t)t->size : sizeof(fd_tuning);
    memcpy(&n, t, k);
    n.size = (int)sizeof(fd_tuning);
    FD_REQUIRE(n.wino_target >= 1 && n.wino_wgrad_target >= 1 && n.conv_target >= 1 && n.wgrad_target >= 1,
               "fd_set_tuning: workgroup targets must be >= 1");
    FD_REQUIRE(n.limb_depth >= 2 && n.limb_depth <= 4 && n.limb_target >= 1 && n.limb_wgrad_target >= 1, "fd_set_tuning: limb_depth must be 2..4, limb_target >= 1");
    FD_REQUIRE(n.wino_min_cout >= 1 && n.wino_wgrad_min_cout >= 1, "fd_set_tuning: wino_min_cout / wino_wgrad_min_cout must be >= 1");
    FD_REQUIRE(n.force_cfg >= -1 && n.force_cfg <= 2 && n.force_splits >= 1, "fd_set_tuning: force_cfg must be -1..2, force_splits >= 1");
    g_tuning = n;
    ++g_generation;
    return 0;
}

extern "C" void fd_get_tuning(fd_tuning* t) {
    if (t) *t = g_tuning;
}

extern "C" long fd_tuning_generation(void) { return g_generation; }
