#!/usr/bin/env python3
"""Copy what an evidence session (tools/r5_evidence.sh <tag>) left under gpurun_out/<tag>/ into profiles/ under the round's names, and recompute the
static HBM-traffic records (profiles/spmv_pmc_traffic.json, spgemm_pmc_traffic.json) from the PMC summaries.  usage: tools/copy_evidence.py <tag> [prefix]"""
import json, os, re, shutil, sys
tag = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else "r05"
out = os.path.join("gpurun_out", tag); P = "profiles"
def cp(a, b):
    src = os.path.join(out, a)
    if os.path.exists(src): shutil.copy(src, os.path.join(P, b))
    else: print("missing", src)
cp("bench_line.json", f"{pre}_bench_line.json"); cp("bench_line_under_rocprof.json", f"{pre}_bench_line_under_rocprof.json"); cp("bench_kernel_stats.csv", f"{pre}_bench_kernel_stats.csv")
cp("pmc_spmv/summary.txt", f"{pre}_spmv_pmc_summary.txt"); cp("pmc_tc/pmc_summary.txt", f"{pre}_spgemm_pmc_summary.txt"); cp("pmc_tc/kernel_stats.csv", f"{pre}_spgemm_kernel_stats.csv")
cp("bfs_per_call_timeline.txt", f"{pre}_bfs_per_call_timeline.txt"); cp("sssp_per_call_timeline.txt", f"{pre}_sssp_per_call_timeline.txt")
cp("aa_kernel_stats.txt", f"{pre}_aa_kernel_stats.txt"); cp("workloads_scale22.jsonl", f"{pre}_workloads_scale22.jsonl")
cp("reftests/pytest_reference.log", f"{pre}_reference_tests_through_shim_on_mi355x.log"); cp("refdoctests/doctests.log", f"{pre}_reference_doctests_through_shim_on_mi355x.log")
cp("refnotebooks/notebooks.log", f"{pre}_reference_notebooks_through_shim_on_mi355x.log")
open(os.path.join(P, f"{pre}_pytest_gpu_tail.txt"), "w").write("".join(open(os.path.join(out, "pytest_gpu.log")).readlines()[-18:]))
f = os.path.join(out, "pmc_spmv/summary.txt")
if os.path.exists(f):
    txt = open(f).read()
    def grab(kernel):
        m = re.search(re.escape(kernel) + r".*?\n((?:   .*\n)+)", txt); d = {}
        for l in m.group(1).splitlines(): d[l.split()[0]] = float(l.split("mean=")[1])
        return d
    per = {}; tot = 0
    for name, key in (("k_xp_hot_gather", "k_xp_hot_gather<double>"), ("k_spmv_tiles", "k_spmv_tiles<double"), ("k_xp_merge", "k_xp_merge<double")):
        d = grab(key); r = int(d["TCC_EA0_RDREQ_sum"] * 128); w = int(d["WRITE_SIZE"] * 1024); per[name] = [r, w]; tot += r + w
    j = json.load(open(os.path.join(P, "spmv_pmc_traffic.json")))
    j["hbm_bytes_per_launch"] = tot; j["per_kernel_read_write_bytes"] = per
    j["history"][f"{pre}: same kernels, counters re-collected with the round's final library"] = tot
    j["ratio_to_algorithmic"] = round(tot / j["algorithmic_bytes"], 4)
    json.dump(j, open(os.path.join(P, "spmv_pmc_traffic.json"), "w"), indent=1)
    print("spmv traffic", tot, j["ratio_to_algorithmic"])
f = os.path.join(out, "pmc_tc/pmc_summary.txt")
if os.path.exists(f):
    t2 = open(f).read()
    rd = float(re.search(r"HBM read bytes .*= ([0-9.e+]+)", t2).group(1)); wr = float(re.search(r"HBM write bytes .*= ([0-9.e+]+)", t2).group(1))
    k = json.load(open(os.path.join(P, "spgemm_pmc_traffic.json")))
    k["read_bytes"] = int(rd); k["write_bytes"] = int(wr); k["hbm_bytes_per_launch"] = int(rd + wr); k["ratio_to_algorithmic"] = round((rd + wr) / k["algorithmic_bytes"], 4)
    json.dump(k, open(os.path.join(P, "spgemm_pmc_traffic.json"), "w"), indent=1)
    print("spgemm traffic", k["hbm_bytes_per_launch"], k["ratio_to_algorithmic"])
