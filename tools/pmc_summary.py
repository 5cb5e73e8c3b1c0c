#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes) into per-kernel HBM bytes per launch.

Corrections applied (gfx950): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts 128-byte
requests at 64 bytes, i.e. reports HALF the bytes of a wide coalesced (16 B/lane) read stream -> doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.

usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

KEYS = {  # json key -> substring of the kernel name (+ optional grid filter)
    "gemm_edge_l1": "gemm_kernel<4, 1, 5, 0>",
    "edge_chain": "edge_chain_kernel",
    "edge_chain_bf16": "edge_chain_bf16_kernel",
    # the 256-d variant's SAVE (training) and inference instantiations: <..., NW=4, CTI=1, SAVE, DEPTH>
    "edge_chain_bf16_save": "1, true, 4>(mpnhip::EdgeChainBf16Args",
    "edge_chain_bf16_infer": "1, false, 4>(mpnhip::EdgeChainBf16Args",
    "edge_chain_bf16_bwd": "edge_chain_bf16_bwd_kernel",
    "segment_reduce3_b16": "k_segment_reduce3_b16",
    "sum_blocks_bf16": "k_sum_blocks_bf16",
    "node_chain": "node_chain_kernel",
    "node_chain_bwd": "node_chain_bwd_kernel",
    "edge_chain_bwd": "edge_chain_bwd_kernel",
    "k_aggregate": "k_aggregate",
    "gemm_tn": "gemm_tn_kernel",
    "wgrad_panel": "wgrad_panel_kernel",
    "wgrad_rows16": "wgrad_rows16_kernel",
    "gemm_bf16_ring": "gemm_bf16_ring_kernel",
    "gemm_bf16_tiled": "gemm_bf16_kernel<",
    "wgrad_reduce": "wgrad_reduce_kernel",
    "segment_reduce3": "k_segment_reduce3",
}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key, sub in KEYS.items():
        f = [v for k, vs in fetch.items() if sub in k for v in vs]
        w = [v for k, vs in write.items() if sub in k for v in vs]
        if not f or not w:
            continue
        rd = sum(f) / len(f) * 1024.0 * 2.0
        wr = sum(w) / len(w) * 1024.0
        out[key] = {"kernel": sub, "launches_fetch_pass": len(f), "launches_write_pass": len(w),
                    "fetch_size_kib_avg": sum(f) / len(f), "write_size_kib_avg": sum(w) / len(w),
                    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "hbm_bytes_per_launch": rd + wr}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
