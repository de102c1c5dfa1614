"""Count the fp64 instructions of one half-stage of k_ket<14> in the compiled ISA (the flop
accounting behind bench.py's roofline): python tools/count_isa.py [mode] > profiles/r03_kket_isa.md"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# round 5: the library is four translation units (k_split_reg_inst.hpp) - k_split_reg<14, ..> lives in the part unit
# rydemu_splitreg.hip (-DRYD_SPLITR_N=14), everything else in rydemu.hip (-DRYD_SPLIT_TUS)
splitreg = len(sys.argv) > 1 and sys.argv[1] == "splitreg"
sr_n = int(sys.argv[2]) if splitreg and len(sys.argv) > 2 else 14   # count_isa.py splitreg [N [NR]]
sr_nr = int(sys.argv[3]) if splitreg and len(sys.argv) > 3 else 5
src = os.path.join(ROOT, "pulser_amd", "csrc", "rydemu_splitreg.hip" if splitreg else "rydemu.hip")
defs = [f"-DRYD_SPLITR_N={sr_n}"] if splitreg else ["-DRYD_SPLIT_TUS"]
if os.environ.get("RYD_ISA_TEXT") and not splitreg:  # (a listing of rydemu.hip made earlier: the tests compile it once)
    text = open(os.environ["RYD_ISA_TEXT"]).read()
else:
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "r.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Wno-unused-value", "-Wno-unused-function", *defs, src, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(asm).read()
    if os.environ.get("RYD_ISA_KEEP") and not splitreg:
        open(os.environ["RYD_ISA_KEEP"], "w").write(text)
if len(sys.argv) > 1 and sys.argv[1] == "splitreg":
    # k_split_reg<14, 5, false>: fp64 work of the stage loop (ONE stage body per iteration, 32 amplitudes per lane)
    m = re.search(r"^(_Z\d+k_split_regILi%dELi%dELb0ELb0ELb0ELb0E\w*):(.*?)s_endpgm" % (sr_n, sr_nr), text, re.S | re.M)
    body = m.group(2).split("\n")
    best = None
    for hdr in [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]:
        lab = body[hdr].split(":")[0]
        backs = [i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+%s\b" % re.escape(lab), l)]
        if backs and (best is None or backs[-1] - hdr > best[1] - best[0]):
            best = (hdr, backs[-1])
    reg = body[best[0]:best[1] + 1]
    c = lambda pat: sum(1 for l in reg if re.search(pat, l))
    fma, mul, add, rnd = c(r"\sv_fmac?_f64"), c(r"\sv_mul_f64"), c(r"\sv_add_f64"), c(r"v_rndne_f64")
    valu = c(r"^\s+v_")
    na = 1 << sr_nr
    print(f"# instruction count of the stage loop of `k_split_reg<{sr_n}, {sr_nr}>` (hipcc 7.2, gfx950, -O3)\n")
    print(f"The loop body holds ONE stage (positions, not parities: the same code runs even and odd stages), {na} amplitudes per lane.\n")
    print("| v_fma/v_fmac_f64 | v_mul_f64 | v_add_f64 | v_rndne_f64 | v_mov_b32_dpp | v_permlane*_swap | other VALU | ds ops | scratch ops | barriers | flops / amplitude / stage |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    dpp, swp, n_ds = c(r"v_mov_b32_dpp"), c(r"v_permlane\d+_swap"), c(r"^\s+ds_")
    print(f"| {fma} | {mul} | {add} | {rnd} | {dpp} | {swp} | {valu - fma - mul - add - rnd - dpp - swp} | {n_ds} | {c('scratch_')} | {c('s_barrier')} | {(2 * fma + mul + add + rnd) / na:.2f} |")
    print("\nPer stage and amplitude: 14 tan-form rotations x 2 FMAs + the phase factor as a product (tree over the register bits, the")
    print("uniform factor, the amplitude: three complex multiplications) + 7 table-and-series sin / cos per lane.  On paper a stage")
    print("needs 14 x 2 FMAs + one complex multiplication = 62 flops per amplitude (bench.py: SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE).")
    print("bench.py uses KSPLITREG_FLOPS_PER_AMP_STAGE = the last column; re-run after changing the kernel.")
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "split14":
    # k_split14_loop<true>: fp64 work of the stage loop (two unrolled stage bodies, 32 amplitudes per lane)
    m = re.search(r"^(_Z\d+k_split14_loopILb1E\w*):(.*?)s_endpgm", text, re.S | re.M)
    body = m.group(2).split("\n")
    hdr = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l][0]
    lab = body[hdr].split(":")[0]
    back = [i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+%s\b" % re.escape(lab), l)][-1]
    reg = body[hdr:back + 1]
    c = lambda pat: sum(1 for l in reg if re.search(pat, l))
    fma, mul, add, rnd = c(r"\sv_fmac?_f64"), c(r"\sv_mul_f64"), c(r"\sv_add_f64"), c(r"v_rndne_f64")
    valu = c(r"^\s+v_")
    print("# r03: fp64 instruction count of the stage loop of `k_split14_loop<true>` (hipcc 7.2, gfx950, -O3)\n")
    print("The loop body holds TWO stages (even: layouts LA -> LB -> LC, odd: back), 32 amplitudes per lane.\n")
    print("| v_fma/v_fmac_f64 | v_mul_f64 | v_add_f64 | v_rndne_f64 | other VALU | ds ops | scratch ops | barriers | flops / amplitude / stage |")
    print("|---|---|---|---|---|---|---|---|---|")
    n_ds = c(r"^\s+ds_")
    print(f"| {fma} | {mul} | {add} | {rnd} | {valu - fma - mul - add - rnd} | {n_ds} | {c('scratch_')} | {c('s_barrier')} | {(2 * fma + mul + add + rnd) / 64:.2f} |")
    print("\nPer stage and amplitude: 14 rotations x 2 FMAs (tan form) + the phase factor (range reduction, table entry, degree-5 series,")
    print("two complex multiplications).  bench.py uses KSPLIT14_FLOPS_PER_AMP_STAGE = the last column; re-run after changing the kernel.")
    sys.exit(0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 0 plain, 1 rows, 2 gauge
m = re.search(r"^_Z5k_ketILi14ELi%dE\w*EEv7KetArgs:(.*?)s_endpgm" % mode, text, re.S | re.M)
body = m.group(1).split("\n")
bars = [i for i, l in enumerate(body) if "s_barrier" in l]
# half-stages = the longest barrier-to-barrier regions that contain ds_read_b128 partner reads
regions = sorted(((b - a, a, b) for a, b in zip(bars, bars[1:])), reverse=True)
# since the LDS publishes ride along, a half-stage is two barrier-to-barrier regions of 8 pairs each
hs = sorted([r for r in regions if sum("ds_read_b128" in l for l in body[r[1]:r[2]]) >= 40][:4], key=lambda r: r[1])
print(f"# r03: fp64 instruction count of one half-stage of `k_ket<14, MODE {mode}>` (hipcc 7.2, gfx950, -O3)\n")
print("One half-stage = `dst += coef (H~ - shift) src` for the 32 amplitudes of a lane (16 pairs), in two")
print("barrier-to-barrier halves of 8 pairs (16 amplitudes) each; the two hot copies are q <- p and p <- q.\n")
print("| region (ISA lines) | v_fma/v_fmac_f64 | v_mul_f64 | v_add_f64 | ds_read_b128 | v_mov_dpp | other VALU | scratch ops | flops / amplitude |")
print("|---|---|---|---|---|---|---|---|---|")
for _, a, b in hs:
    reg = body[a:b]
    c = lambda pat: sum(1 for l in reg if re.search(pat, l))
    fma, mul, add = c(r"v_fmac?_f64"), c(r"v_mul_f64"), c(r"v_add_f64")
    valu = c(r"^\s+v_")
    print(f"| {a}..{b} | {fma} | {mul} | {add} | {c('ds_read_b128')} | {c('_dpp')} | {valu - fma - mul - add - c('_dpp')} | {c('scratch_')} | {(2 * fma + mul + add) / 16:.2f} |")
print("\nTwo half-stages per stage (q += a x p; p -= b x q) -> flops per amplitude per stage = 2 x the last column.")
print("bench.py uses KKET_FLOPS_PER_AMP_STAGE = 2 x the mean of the last column; re-run this script after changing the kernel.")
