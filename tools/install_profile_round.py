#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/<round>_*, stamping every counter table with
the hash of the library they were collected on (bench.py compares it with the library it runs: config.counter_tables_match_this_build).

    python tools/install_profile_round.py r06f r06        # the tree must be the one the round ran on"""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    sha = hashlib.sha256(open(os.path.join(ROOT, "ansel_amd", "libansel_hip.so"), "rb").read()).hexdigest()[:16]
    tables = {"pmc_hbm_bytes.json": "%s_pmc_hbm_bytes_100MP_full.json", "pmc_sq.json": "%s_pmc_sq_100MP_full.json",
              "sclk_per_kernel.json": "%s_sclk_per_kernel_100MP_full.json", "isa_mix.json": "%s_isa_mix.json"}
    for a, b in tables.items():
        j = json.load(open(os.path.join(src, a)))
        j["lib_sha16"] = sha
        json.dump(j, open(os.path.join(dst, b % rnd), "w"), indent=1)
    for a, b in {"pmc_sq.txt": "%s_pmc_sq_100MP_full.txt", "kernel_stats.csv": "%s_kernel_stats_100MP_full.csv",
                 "bench.log": "%s_bench_100MP_full_profile_round.log"}.items():
        shutil.copy(os.path.join(src, a), os.path.join(dst, b % rnd))
    print("installed the tables of library", sha)


if __name__ == "__main__":
    main()
