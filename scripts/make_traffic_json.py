"""profiles/r0N_traffic.json from the FETCH_SIZE / WRITE_SIZE summary (scripts/pmc_summary.py).

    APN_COLLECTED_AT=<commit> python scripts/make_traffic_json.py profiles/r05_pmc_fetch_write_summary_block.csv profiles/r05_traffic.json

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): MI355X_MICROARCH.md, section
HBM -- on gfx950 FETCH_SIZE reports half of the bytes of a coalesced read, WRITE_SIZE is exact.
The file is stamped with the commit it was collected at (handed in: the GPU box has no .git) and with a hash of the
kernel sources, which bench.py re-computes at run time (`roofline.traffic_sources_match`): a kernel change after the
collection shows on the bench line instead of silently keeping stale bytes.
"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources whose kernels the file's bytes belong to
KERNEL_SOURCES = ("sa_fused.hip", "sa_glue.hip", "sa_geo.hip", "sa_seq.hip", "sa_wide_glue.hip", "fps.hip", "ball_query.hip",
                  "ball_query_body.h", "apn_common.h", "apn_mfma.h", "sa_chain.h")

ALIAS = {"fps_atomic_kernel": "fps", "fps_reg_kernel": "fps", "ball_query_kernel": "ball_query", "sa_prep_stats_kernel": "sa_prep_stats",
         "sa_geo_kernel": "sa_point_geo",
         "sa_prep_features_kernel": "sa_prep_features", "sa_fwd_stats1_kernel": "sa_fwd_stats1",
         "sa_fwd_main_kernel": "sa_fwd_main", "fwd_out_kernel": "sa_fwd_out",
         "bwd_prep_kernel": "sa_bwd_prep", "sa_bwd_kernel": "sa_bwd_main",
         "bwd_point_grads_kernel": "sa_bwd_point_grads", "bwd_finalize_kernel": "sa_bwd_finalize",
         "bn_fold_kernel": "sa_bn_fold", "bwd_consts1_kernel": "sa_bwd_consts1",
         "bwd_consts2_kernel": "sa_bwd_consts2"}
ALIAS.update({k: k.replace("_kernel", "") for k in (
    "wide_stats1_kernel", "wide_fwd_main_kernel", "wide_bwd_main_kernel", "wide_wgrad_kernel", "wide_bwd_prep_kernel",
    "wide_point_terms_kernel", "wide_geo_kernel", "wide_colsum_kernel", "wide_image_kernel")})
# one bench-line entry, two kernels: their bytes add (both run once per launch of the entry)
ALIAS.update({"tilemap_pack_kernel": "sa_wide_tilemap_many", "tilemap_fill_kernel": "sa_wide_tilemap_many"})
ALIAS["csr_cloud_kernel"] = "sa_rowmap_many"      # the row map of a replay's tile maps (one launch, apn_sa_rowmap_many)


def sources_sha16(root=ROOT):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(root, "adaptpoint_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


def main():
    out = {}
    for r in csv.DictReader(open(sys.argv[1])):
        k = ALIAS.get(r["kernel"])
        if k and r.get("FETCH_SIZE") and r.get("WRITE_SIZE"):
            out[k] = out.get(k, 0) + int(round((2.0 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024))
    json.dump({"structure": "default",
               "collected_at_commit": os.environ.get("APN_COLLECTED_AT", "unknown"),
               "kernel_sources_sha16": sources_sha16(),
               "_note": "HBM bytes per launch (B=32 per MLP-stream launch; the index-stream launches cover 20 batches) from "
                        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py's DEFAULT launch structure "
                        "(hipGraph replay, index stages on the second stream, tile map; bf16x3; "
                        "scripts/collect_profiles.sh pmc), averaged over all dispatches of the pass; "
                        "FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes for gfx950, WRITE_SIZE "
                        "taken as is. Summary: profiles/" + os.path.basename(sys.argv[1]),
               "bytes_per_launch": out}, open(sys.argv[2], "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
