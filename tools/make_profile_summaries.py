#!/usr/bin/env python3
"""profiles/rNN_{traffic,issue,steady_state}.json from the raw outputs of tools/profile.sh and the bench runs.
usage: make_profile_summaries.py ROUND KERNEL_TAG   (e.g. r02 v19): reads profiles/r02_v19_* , writes profiles/r02_*.json"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_tag  # noqa: E402  (the summaries carry the tag of the build they were measured on)
BUILD = os.environ.get("BISBM_BUILD_TAG") or build_tag()
P = os.path.join(ROOT, "profiles")
rnd, tag = sys.argv[1], sys.argv[2]
pre = "%s_%s" % (rnd, tag)
if len(sys.argv) > 3 and sys.argv[3] == "--config5":
    # the K > 32 variant on the config-5 shape (N = 4e6, E = 5e7, Ka = Kb = 64, 1024 chains): instructions and bytes per update
    rows = list(csv.DictReader(open(os.path.join(P, pre + "_pmc_counters.csv"))))
    c = {r["counter"]: float(r["value_sum_over_launches"]) / int(r["launches"]) for r in rows}
    ks = max((r for r in csv.DictReader(open(os.path.join(P, pre + "_kernel_stats.csv"))) if "sweep_fast" in r["Name"]), key=lambda r: int(r["Calls"]))
    avg_s, upd = float(ks["AverageNs"]) / 1e9, 4.096e9
    out = {"_file": "profiles/%s_config5_issue.json" % rnd, "build": BUILD, "kernel": ks["Name"],
           "_what": "sweep_fast_kernel<*,*,false,...> (more than 32 blocks of a type: two steps per pass, two blocks per lane, a window of eta "
                    "in LDS) at N_a = N_b = 2e6, E = 5e7, Ka = Kb = 64, 1024 chains: one sweep = 4.096e9 node updates per launch; rocprofv3 "
                    "--pmc in separate passes, both waves of a chain",
           "source": ["profiles/%s_pmc_counters.csv" % pre, "profiles/%s_kernel_stats.csv" % pre],
           "instructions_per_update": sum(c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")) / upd,
           "valu_per_update": c["SQ_INSTS_VALU"] / upd, "salu_per_update": c["SQ_INSTS_SALU"] / upd, "lds_per_update": c["SQ_INSTS_LDS"] / upd,
           "vmem_per_update": c["SQ_INSTS_VMEM"] / upd, "branch_per_update": c["SQ_INSTS_BRANCH"] / upd,
           "fetch_bytes_per_update_as_counted": c["FETCH_SIZE"] * 1024 / upd, "write_bytes_per_update": c["WRITE_SIZE"] * 1024 / upd,
           "algorithmic_bytes_per_update": 8 + 5 * 25.0 + 2, "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
           "clock_ghz": c["GRBM_GUI_ACTIVE"] / 8 / avg_s / 1e9, "kernel_avg_s_under_rocprof": avg_s, "updates_per_s": upd / avg_s,
           "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "active_inst_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
           "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]}
    json.dump(out, open(os.path.join(P, rnd + "_config5_issue.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
rows = list(csv.DictReader(open(os.path.join(P, pre + "_pmc_counters.csv"))))
# (the counter passes pin the pass depth of the bench line's timed launches; should another variant of the kernel show up
# anyway -- the depth is chosen per launch -- the one with the most launches is the one summarised)
by_kernel = {}
for r in rows:
    by_kernel.setdefault(r["kernel"], []).append(r)
rows = max(by_kernel.values(), key=lambda rs: sum(int(r["launches"]) for r in rs))
c = {r["counter"]: float(r["value_sum_over_launches"]) / int(r["launches"]) for r in rows}
upd = 1.024e9
fetch, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
ids_alg = 8 + 4 * 20.0
traffic = {
    "_file": "profiles/%s_traffic.json" % rnd,
    "build": BUILD,
    "_what": "HBM traffic of sweep_fast_kernel<true,true,true> (two steps per pass) at BASELINE configs[2], 1024 chains, one "
             "sweep = 1.024e9 node updates per launch; rocprofv3 --pmc in separate passes (3 launches each: 2 spin-up + 1 "
             "timed, counters divided by 3); counter unit KB",
    "source": ["profiles/%s_pmc_counters.csv" % pre, "profiles/r02_fetch_calibration.json"],
    "updates_per_launch": upd,
    "fetch_bytes_per_update_as_counted": fetch / upd, "write_bytes_per_update": write / upd,
    "calibration": "known-bytes runs (tools/probe/fetch_calib.hip, 2 GiB touched once): 16-B and 4-B coalesced streams and "
                   "64-B rows read by four 16-B loads per lane are counted at exactly 1/2 of their bytes; single-byte "
                   "gathers at 64 B per touched sector (exact for 64-B requests); 16-B stores exact",
    "id_stream_algorithmic_bytes_per_update": ids_alg,
    "bytes_per_update": (fetch + write) / upd + ids_alg / 2,
    "bytes_per_update_uncorrected": (fetch + write) / upd,
    "tcc_hit": c["TCC_HIT_sum"], "tcc_miss": c["TCC_MISS_sum"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "note": "bytes_per_update = FETCH + WRITE as counted + half of the id / row-offset stream (88 B per update, wide per-lane "
            "loads counted at 1/2). The label gathers (one useful byte per 64-B sector on a miss) separate it from the 110 "
            "algorithmic bytes: 64 consecutive ids of this graph gather from ~486 distinct label sectors (7.6 per update).",
}
json.dump(traffic, open(os.path.join(P, rnd + "_traffic.json"), "w"), indent=1)
ks = max((r for r in csv.DictReader(open(os.path.join(P, pre + "_kernel_stats.csv"))) if "sweep_fast" in r["Name"]),
         key=lambda r: int(r["Calls"]))
avg_s = float(ks["AverageNs"]) / 1e9
ipu = sum(c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")) / upd
issue = {
    "_file": "profiles/%s_issue.json" % rnd,
    "build": BUILD,
    "_what": "instructions per node update of sweep_fast_kernel<true,true,true>, both waves of a chain (stepping + feeder), "
             "rocprofv3 --pmc SQ_INSTS_* (profiles/%s_pmc_counters.csv), and the clock from GRBM_GUI_ACTIVE / 8 XCDs / kernel time" % pre,
    "instructions_per_update": ipu, "valu_per_update": c["SQ_INSTS_VALU"] / upd, "salu_per_update": c["SQ_INSTS_SALU"] / upd,
    "lds_per_update": c["SQ_INSTS_LDS"] / upd, "vmem_per_update": c["SQ_INSTS_VMEM"] / upd,
    "branch_per_update": c["SQ_INSTS_BRANCH"] / upd,
    "clock_ghz": c["GRBM_GUI_ACTIVE"] / 8 / avg_s / 1e9, "kernel_avg_s_under_rocprof": avg_s,
    "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "active_inst_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    "round1": {"instructions_per_update": 241, "valu": 177, "salu": 36, "lds": 10.3, "vmem": 7.5, "branch": 10},
    "round2": {"instructions_per_update": 214.4, "valu": 125.8, "salu": 69.9, "lds": 8.7, "vmem": 4.7, "branch": 5.2},
}
json.dump(issue, open(os.path.join(P, rnd + "_issue.json"), "w"), indent=1)


def bench(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else None


steady = {"_file": "profiles/%s_steady_state.json" % rnd, "build": BUILD,
          "_what": "the bench workload further along and with scrambled ids, one MI355X, kernel %s: --spinup 150 (156 sweeps before "
                   "the warm-up); --shuffle-ids without and with the ingest-time renumbering" % tag}
for key, name, what in (("after_150_sweeps", pre + "_bench_after_150_sweeps.json", "same command with --spinup 150"),
                        ("shuffled_ids", pre + "_bench_shuffled_ids.json", "--shuffle-ids --no-reorder"),
                        ("shuffled_ids_reordered_at_ingest", pre + "_bench_shuffled_ids_reordered.json", "--shuffle-ids (bisbm_io_locality_order)")):
    d = bench(name)
    if d:
        steady[key] = {"updates_per_s": d["value"], "avg_launch_ms": d["roofline"]["avg_launch_ms"],
                       "accepted_fraction": d["config"]["accepted_fraction_last_timed_sweep"],
                       "sweeps_before": d["config"]["spinup_sweeps_before_warmup"], "what": what, "source": "profiles/" + name}
json.dump(steady, open(os.path.join(P, rnd + "_steady_state.json"), "w"), indent=1)
print(json.dumps({"traffic_B_per_update": traffic["bytes_per_update"], "instructions_per_update": ipu, "clock": issue["clock_ghz"],
                  "steady": {k: v["updates_per_s"] for k, v in steady.items() if isinstance(v, dict)}}, indent=1))
