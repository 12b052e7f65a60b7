"""Dev tool: (re)generate profiles/roofline_traffic.json -- the table bench.py reads for `roofline.traffic` -- from the
FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, and STAMP it: the commit it was measured at, the time, and a
hash of the kernel sources each entry belongs to (code only: comments and whitespace are stripped, bench.source_hash).  bench.py
recomputes those hashes and drops an entry whose kernel has changed since (a stale counter is not a measurement of the run that prints it).

  python3 tools/traffic_json.py <prof dir of profile_round.sh> <commit> [--out profiles/roofline_traffic.json]
  python3 tools/traffic_json.py --restamp <commit>      re-hash the sources for the EXISTING values (only when the
                                                        kernels are byte-identical to the ones that were measured)
Counters are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE reports half
the bytes of wide (16-byte-per-lane) coalesced loads -> reads = 2 * FETCH_SIZE * 1024 for kernels that load that way
(attention forward / backward, GroupNorm statistics; calibrated on gn_stats_kernel, which reads its tensor exactly once);
the 3x3 convolution stages its input with 4-byte loads, for which the raw counter already equals the expected bytes.
WRITE_SIZE is exact."""
import collections, csv, datetime, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench

# key in the table            kernel-name filter         pass dirs prefix   wide loads   kernel sources
ENTRIES = (
    ("mha_flash_fwd_L65536_B16_bf16x3", "mha_flash_fwd_h2_kernel", "x3", True, ["attention_h2.hip", "attention_x3p.hip", "common.h"]),
    ("mha_flash_fwd_L65536_B16", "fast_kernel<16", "attn", True, ["attention.hip", "common.h"]),
    ("conv3x3_128_256_B16", "igemm_kernel<2, 8, 12, 5, 1", "conv", False, ["conv_igemm.hip", "common.h"]),
    ("gn_stats_128_256_B16", "gn_stats_kernel", "gn", True, ["groupnorm.hip", "common.h"]),
    ("conv3x3_128_256_B16_pairs", "conv3x3_x3_kernel", "convh2", False, ["conv3x3_x3.hip", "conv_igemm.hip", "common.h"]),
    ("mha_flash_bwd_L65536_B4_bf16x3", "mha_bwd_h2p_kernel", "bwd", True, ["attention_bwd_h2.hip", "attention_bwd.hip", "common.h"]),
    ("mha_flash_bwd_L65536_B4", "bwd_fused", "bwdf32", True, ["attention_bwd.hip", "common.h"]),
)
ALGORITHMIC = {
    "mha_flash_fwd_L65536_B16_bf16x3": 16 * 128 * 65536 * (4 + 8 + 4 + 4),   # main kernel (round 5): q as two fp16 pieces (4 B per element), k as four (8 B), v as an fp16 pair (4 B) read, o written
    "conv3x3_128_256_B16_pairs": 1074331648,
    "mha_flash_fwd_L65536_B16": 2147483648, "conv3x3_128_256_B16": 1074331648, "gn_stats_128_256_B16": 536870912,
    "mha_flash_bwd_L65536_B4": 4 * 65536 * 128 * 4 * 8,      # q, k, v, o, dO read; dq, dk, dv written: 8 tensors of B*C*L floats
    "mha_flash_bwd_L65536_B4_bf16x3": 4 * 65536 * 128 * (4 + 4 + 8 + 4 + 4 + 4 + 4 + 3 * 4),   # main kernel (round 5), 2-byte pieces read: q (2), Q c_q (2), k (4), k^T (2), V' (2), dO per query (2), dO per head (2); dq / dk / dv written
}


def mean_counter(path, flt):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if flt in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {c: sum(v) / len(v) for c, v in acc.items()}


def stamp(commit, how):
    return {"commit": commit, "utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%SZ"), "how": how,
            "kernels": {key: {"sources": src, "sha256_16": bench.source_hash(src)} for key, _, _, _, src in ENTRIES}}


def main():
    out = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    if sys.argv[1] == "--restamp":
        tab = json.load(open(out))
        tab["_stamp"] = stamp(sys.argv[2], "values kept; sources re-hashed (kernels byte-identical to the measured ones)")
        json.dump(tab, open(out, "w"), indent=1)
        return
    prof, commit = sys.argv[1], sys.argv[2]
    tab = {"_how": __doc__.split("Counters are", 1)[1].strip().replace("\n", " "), "_raw_KiB": {}, "_algorithmic_bytes": ALGORITHMIC}
    for key, flt, pre, wide, _ in ENTRIES:
        f = mean_counter(os.path.join(prof, f"{pre}_FETCH_SIZE", "pmc_counter_collection.csv"), flt).get("FETCH_SIZE")
        w = mean_counter(os.path.join(prof, f"{pre}_WRITE_SIZE", "pmc_counter_collection.csv"), flt).get("WRITE_SIZE")
        if f is None or w is None:
            print("missing counters for", key, file=sys.stderr)
            continue
        tab["_raw_KiB"][key] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "fetch_doubled": wide}
        tab[key] = int(((2 if wide else 1) * f + w) * 1024)
    # instruction counts per dispatch (the SQ passes <prefix>_sq1 / _sq2 of profile_round.sh): what bench.py's issue model is made of
    tab["_issue_counts"] = {}
    for key, flt, pre, _, _ in ENTRIES:
        try:
            c1 = mean_counter(os.path.join(prof, f"{pre}_sq1", "pmc_counter_collection.csv"), flt)
            c2 = mean_counter(os.path.join(prof, f"{pre}_sq2", "pmc_counter_collection.csv"), flt)
        except OSError:
            continue
        if "SQ_INSTS_VALU" in c1 and "SQ_INSTS_MFMA" in c1 and "SQ_INSTS_VALU_TRANS_F32" in c2:
            tab["_issue_counts"][key] = {"SQ_INSTS_VALU": c1["SQ_INSTS_VALU"], "SQ_INSTS_MFMA": c1["SQ_INSTS_MFMA"],
                                         "SQ_INSTS_VALU_TRANS_F32": c2["SQ_INSTS_VALU_TRANS_F32"],
                                         "SQ_VALU_MFMA_BUSY_CYCLES": c1.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                                         "SQ_VALU_MFMA_COEXEC_CYCLES": c2.get("SQ_VALU_MFMA_COEXEC_CYCLES")}
    tab["_stamp"] = stamp(commit, f"tools/profile_round.sh -> {prof} (separate --pmc FETCH_SIZE / WRITE_SIZE passes with --kernel-trace only)")
    tab["_note"] = ("FETCH_SIZE counts the L2's fabric-side read requests, Infinity-Cache hits included: an upper bound on HBM bytes.")
    json.dump(tab, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in tab.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
