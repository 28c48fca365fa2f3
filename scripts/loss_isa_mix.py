"""Static instruction mix of k_photo_ms<true, true>'s row loop (csrc/photometric_ms.hip), from hipcc's own assembly, priced with the
measured gfx950 issue rates of scripts/ubench/valu_rate{2,3}.hip -> the vector-issue floor of the kernel as it is and of a PACKED
two-columns-per-lane formulation of the same arithmetic (profiles/round5_pmc_loss.md).  Runs in the build container (no GPU):

    python scripts/loss_isa_mix.py [dynamic VALU wave-instructions per launch, default 102.4e6 (profiles/round3_pmc_loss.md)]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fusiondepth_amd", "csrc", "photometric_ms.hip")
# cycles per wave-instruction and SIMD with two waves resident (the kernel's occupancy), measured: profiles/round5_valu_rate3.log and
# profiles/round3_pmc_loss.md (scripts/ubench/valu_rate2.hip)
RATE = {"plain": 2.95, "sgpr": 4.5, "other": 4.5, "dpp": 4.6, "trans": 8.4}
PK = {"fma": (3.33, 5.67), "mul": (2.65, 4.98), "add": (2.77, 5.06), "mov": (2.7, 5.14)}      # (plain, packed = two lanes' worth) cycles


def main():
    dyn = float(sys.argv[1]) if len(sys.argv) > 1 else 102.4e6
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "photo_ms.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-comment", "-Wno-pass-failed",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(SRC), "-S", "--cuda-device-only", "-o", asm, SRC],
                       check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_ZN\S*k_photo_msILb1ELb1E\S*:", l)][0]
    end = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end") and i > start][0]
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    loop = body[best[1]:best[2] + 1]
    cls, kinds = collections.Counter(), collections.Counter()
    fast = {"v_add_f32": "add", "v_sub_f32": "add", "v_subrev_f32": "add", "v_mul_f32": "mul", "v_fma_f32": "fma", "v_fmac_f32": "fma",
            "v_fmaak_f32": "fma", "v_fmamk_f32": "fma", "v_mov_b32": "mov"}
    for l in loop:
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "")
            dpp = "dpp" in op or "row_" in l or "wave_sh" in l
            sgpr = bool(re.search(r"\bs\d+\b|\bs\[", l)) or " vcc" in l
            if dpp:
                cls["dpp"] += 1
            elif base in fast and not sgpr:
                cls["plain"] += 1; kinds[fast[base]] += 1
            elif base in fast:
                cls["sgpr"] += 1
            elif base.startswith(("v_rcp", "v_log", "v_exp", "v_sqrt", "v_rsq")):
                cls["trans"] += 1
            else:
                cls["other"] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            cls["vmem"] += 1
    valu = sum(cls[k] for k in RATE)
    print("row loop of k_photo_ms<true,true> (two image rows of one frame per trip): %d assembly lines" % len(loop))
    names = {"plain": "fp32 add / mul / fma / mov, VGPR or constant operands", "sgpr": "the same with an SGPR operand", "dpp": "DPP (wave shifts of the window sums)",
             "other": "min / max / med3 / cmp / cndmask / cvt / fract / integer", "trans": "v_rcp_f32"}
    print("| class | static count | share | cycles per wave-instruction (2 waves / SIMD, measured) |\n|---|---:|---:|---:|")
    for k in ("plain", "other", "dpp", "trans", "sgpr"):
        print("| %s | %d | %.1f %% | %.2f |" % (names[k], cls[k], 100.0 * cls[k] / valu, RATE[k]))
    print("| (scalar / LDS / memory instructions of the same trip) | %d / %d / %d | | |" % (cls["salu"], cls["lds"], cls["vmem"]))
    cyc = sum(cls[k] * RATE[k] for k in RATE)
    trips = dyn / valu
    t_now = trips * cyc / 1024 / 2.4e9 * 1e6
    print("\nVALU instructions per trip %d; issue cycles per trip %.0f; %.3g dynamic VALU wave-instructions per launch = %.0f wave-trips" % (valu, cyc, dyn, trips))
    print("vector-issue floor of the kernel AS IT IS: %.1f us (1 024 SIMDs, 2.4 GHz)" % t_now)
    # packed: two columns per lane.  Per PAIR of pixels: the plain arithmetic as packed instructions, the window shifts 3 per pair instead
    # of 2 per pixel, everything else (no packed form exists) twice.
    plain_pk = sum(n * PK[k][1] for k, n in kinds.items())
    plain_2x = 2 * sum(n * PK[k][0] for k, n in kinds.items())
    pair_now = 2 * cyc
    pair_pk = plain_pk + 0.75 * 2 * cls["dpp"] * RATE["dpp"] + 2 * (cls["other"] * RATE["other"] + cls["trans"] * RATE["trans"] + cls["sgpr"] * RATE["sgpr"])
    print("plain arithmetic of a pixel PAIR: %.0f cycles as 2 x plain, %.0f as packed (v_pk_fma / v_pk_mul / v_pk_add / v_pk_mov at their measured rates)" % (plain_2x, plain_pk))
    print("vector-issue floor of the PACKED formulation: %.1f us = %.2f x the present floor" % (t_now * pair_pk / pair_now, pair_pk / pair_now))


if __name__ == "__main__":
    main()
