"""profiles/r<N>_roofline_table.md from the committed per-kernel durations (tools/kernel_trace.py --sum) and PMC byte
counts (tools/pmc_hbm.py) of each workload.  usage: python tools/roofline_table.py [round] > profiles/r3_roofline_table.md"""
import json, os, re, sys

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "3"
WORK = [("cfg3 (256^3 p=3)", "r%s_cfg3_kernel_stats.txt" % RND, "r%s_cfg3_pmc_hbm.json" % RND),
        ("cfg3 size, mapped geometry", "r%s_cfg3_mapped_kernel_stats.txt" % RND, "r%s_cfg3_mapped_pmc_hbm.json" % RND),
        ("general PtAP, element split (64^3 p=3)", "r%s_elemsplit_kernel_stats.txt" % RND, "r%s_elemsplit_pmc_hbm.json" % RND),
        ("cfg3 size, nothing assumed about M or A (element chunks)", "r%s_cfg3_general_kernel_stats.txt" % RND,
         "r%s_cfg3_general_pmc_hbm.json" % RND),
        ("mapped elasticity, one field block (64^3 p=3 elements; element-per-wave kernel, before k_asf3_quad)",
         "r%s_asm_elast_p3_kernel_stats.txt" % RND, "r%s_asm_elast_p3_pmc_hbm.json" % RND),
        ("mapped stiffness matrix (64^3 p=3 elements; k_asf3_quad)", "r%s_asm_quad_p3_kernel_stats.txt" % RND,
         "r%s_asm_quad_p3_pmc_hbm.json" % RND),
        ("half-storage product, stencil radius 4 (160^3 p=4)", "r%s_symgrid_p4_kernel_stats.txt" % RND, "r%s_symgrid_p4_pmc_hbm.json" % RND),
        ("half-storage product, three fields (elasticity 96^3 p=3)", "r%s_symgrid_fields_kernel_stats.txt" % RND,
         "r%s_symgrid_fields_pmc_hbm.json" % RND),
        ("cfg4, banded Cholesky (256^2 p=4)", "r%s_cfg4_cholesky_kernel_stats.txt" % RND, "r%s_cfg4_cholesky_pmc_hbm.json" % RND),
        ("cfg2 (128^3 p=2)", "r%s_cfg2_kernel_stats.txt" % RND, "r%s_cfg2_pmc_hbm.json" % RND),
        ("cfg4 (256^2 p=4, CG)", "r%s_cfg4_kernel_stats.txt" % RND, "r%s_cfg4_pmc_hbm.json" % RND),
        ("cfg5 (128^2 p=3, 3 fields)", "r%s_cfg5_kernel_stats.txt" % RND, "r%s_cfg5_pmc_hbm.json" % RND),
        # the general extraction kernels (count / scan / fill: active filter, explicit points), forced with TIGAR_EXTRACT_KRON=0
        ("cfg2, general M build", "r%s_cfg2_general_extraction_kernel_stats.txt" % RND,
         "r%s_cfg2_general_extraction_pmc_hbm.json" % RND),
        # the same with the separable-filter shortcut off too (TIGAR_EXTRACT_SEPARABLE=0): the count / scan / fill kernels
        ("cfg2, general M build, count / fill kernels", "r%s_cfg2_general_extraction_count_fill_kernel_stats.txt" % RND,
         "r%s_cfg2_general_extraction_count_fill_pmc_hbm.json" % RND)]
WORK = [w for w in WORK if os.path.exists(os.path.join(HERE, w[1])) and os.path.exists(os.path.join(HERE, w[2]))]


def short(name):
    return re.sub(r"\s+", " ", name.strip())[:52]


def main():
    print("# Achieved HBM rates per kernel, round %s (one MI355X)\n" % RND)
    print("Per launch: average duration from `rocprofv3 --kernel-trace` (`rN_cfg*_kernel_stats.txt`, summarised from the trace\n"
          "database by `tools/kernel_trace.py --sum`; `rN_cfg*_rocprofv3_kernel_stats.csv` is rocprofv3's own `--stats` output of\n"
          "the same command, `--output-format csv`), HBM bytes from the PMC passes (`rN_cfg*_pmc_hbm.json`: FETCH_SIZE and\n"
          "WRITE_SIZE in separate runs, reads x2 = the gfx950 correction of the guide, calibrated for 8 B/lane streams by\n"
          "`k_cg1_dot` / `k_cg1_update`, DESIGN.md section 5).  Rate = (read + written) / duration, as a fraction of the 8 TB/s\n"
          "HBM3E peak and of the ceiling a plain streaming kernel of the same read:write mix reaches on these boxes\n"
          "(`r2_hbm_ceilings.txt`, `tools/mb/write_bw.hip`: pure read 6.2-6.4, pure write 5.9-6.6, mixed 5.1-5.3 TB/s; the\n"
          "rate of a buffer depends on where it was placed, 5.6-6.7 for pure writes).  Durations and counters come from\n"
          "different runs of the same command, so rows of kernels whose launches vary in size (sub-slabs) are averages.\n"
          "(rN = r%s in the file names.)  `k_cg_persistent` / `k_gmres_persistent` / `k_bicgstab_persistent` / `k_pcg_cheb_persistent`\n"
          "(round 4) are WHOLE Krylov solves in one kernel each: K sits in registers, what they read from HBM is the vector the\n"
          "workgroups exchange (agent-scope loads, once per product) -- they are bound by their device-wide barriers, not by HBM,\n"
          "and their rows are here for the byte counts only.\n" % RND)
    print("| workload | kernel | avg ms | read GB | written GB | TB/s | of 8 TB/s | of the streaming ceiling |")
    print("|---|---|---|---|---|---|---|---|")
    for label, fstats, fpmc in WORK:
        dur = {}
        for line in open(os.path.join(HERE, fstats)):
            m = re.match(r"\s*(\S+)\s+(\d+) x\s+([\d.]+) ms total\s+([\d.]+) ms avg", line)
            if m:
                dur[m.group(1).replace(".kd", "")] = (int(m.group(2)), float(m.group(4)))
        pmc = json.load(open(os.path.join(HERE, fpmc)))["kernels"]
        for k in pmc:
            name = k["kernel"]
            key = None
            for d in dur:
                # mangled name in the stats file, demangled in the counter file: match on the bare function name
                bare = re.match(r"(?:void )?([A-Za-z_0-9]+)", name).group(1)
                if "%d%s" % (len(bare), bare) in d or bare == d:       # (_Z<len><name>...: exact function name)
                    key = d
                    break
            if key is None or k["launches"] == 0:
                continue
            n, ms = dur[key]
            r, w = k["hbm_read_GB_total_corrected"] / k["launches"], k["hbm_write_GB_total"] / k["launches"]
            if (r + w < 0.05 or ms < 0.03) and "k_chol_fused" not in name:       # (that one: 2 111 short launches are the solve)
                continue
            if "general M build" in label and "k_extract" not in name and not re.match(r"k_kron3_(fill|rowptr)\(", name):
                continue                                   # (k_kron3_fill_sum_rows is the FE matrix of the bench, not M)
            rate = (r + w) / ms
            frac_w = w / (r + w)
            ceiling = 6.3 if frac_w < 0.05 else (6.2 if frac_w > 0.95 else 5.2)
            print("| %s | `%s` | %.3f | %.3f | %.3f | %.2f | %.0f %% | %.0f %% |" % (label, short(name), ms, r, w, rate, 100 * rate / 8.0,
                                                                                  100 * rate / ceiling))
        print("")


if __name__ == "__main__":
    main()
