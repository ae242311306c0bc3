#!/usr/bin/env python3
"""HBM traffic per launch and kernel class from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE need separate passes: together they exceed the TCC counter slots).

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [command] [workload]

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced
reads at 64 bytes, so the read side is doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import sys

CLASSES = [  # (substring of the kernel name, kernel class as bench.py reports it)
    ('gemm_f32_kernel<false, false', 'gemm_forward'), ('gemm_f32_kernel<false, true', 'gemm_input_grad'),
    ('gemm_f32_kernel<true,', 'gemm_weight_grad'), ('gemm_dw_split_kernel', 'gemm_weight_grad'), ('gemm_dw_wide_kernel', 'gemm_weight_grad'), ('gemm_reduce_kernel', 'gemm_slab_reduce'),
    ('gine_agg_fwd_kernel', 'gine_aggregate_fwd'), ('gine_agg_bwd_kernel', 'gine_aggregate_bwd'),
    ('gine_etab_reduce_kernel', 'gine_edge_grad'), ('embed3_gather_kernel', 'embed_gather'), ('rows_gather_kernel', 'embed_gather'),
    ('table_grad_', 'table_grad'), ('segment_pool_kernel', 'segment_pool'), ('pair_gather_kernel', 'pair_gather'),
    ('pair_scatter_kernel', 'pair_scatter'), ('gru_fwd', 'gru_forward'), ('gru_bwd', 'gru_backward'),
    ('softmax_ce_', 'softmax_ce'), ('sqnorm_partial_kernel', 'clip_adam'), ('clip_adam_kernel', 'clip_adam'),
    ('gine_layer_fwd_', 'gine_layer_fused'), ('gine_layer_bwd_', 'gine_layer_fused_bwd'),
    ('scorer_fwd_', 'scorer_fused'), ('scorer_bwd_', 'scorer_fused'), ('pack_weights_', 'weight_pack'),
    ('normal_rows_kernel', 'epsilon_draw'),
]


def per_class(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        for sub, cls in CLASSES:
            if sub in r['Kernel_Name']:
                tot[cls] += float(r['Counter_Value'])
                cnt[cls] += 1
                break
    return tot, cnt


def main():
    fetch, nf = per_class(sys.argv[1], 'FETCH_SIZE')
    write, nw = per_class(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for cls in sorted(set(fetch) | set(write)):
        f = fetch[cls] / max(nf[cls], 1)
        w = write[cls] / max(nw[cls], 1)
        out[cls] = {'launches_sampled': nf[cls], 'FETCH_SIZE_KiB_per_launch': f, 'WRITE_SIZE_KiB_per_launch': w,
                    'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0}
    json.dump({'formula': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read-side correction)',
               'command': 'rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 ' + (sys.argv[4] if len(sys.argv) > 4 else
                          'bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bulk --profile-steps 1 --config3-steps 0'),
               'workload': sys.argv[5] if len(sys.argv) > 5 else 'zinc, vocab 300, batch 128', 'classes': out},
              open(sys.argv[3], 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:20s} {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch  ({v['launches_sampled']} launches)")


if __name__ == '__main__':
    main()
