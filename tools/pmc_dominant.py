#!/usr/bin/env python3
"""Launches the dominant kernel of the step (last generator upBlock conv: nearest x2 + conv3x3 64->64 at
256x256, B=20, BatchNorm statistics epilogue) a few times, for rocprofv3 PMC passes:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 tools/pmc_dominant.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 tools/pmc_dominant.py
    python3 tools/pmc_dominant.py --collect out_f out_w > profiles/rNN_pmc_dominant_kernel.json
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = 'conv3x3_halo_kernel'


def collect(dirs):
    out = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if KERNEL in r['Kernel_Name']:
                    out.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    res = {'kernel': KERNEL + '<64,64,ups> upBlock conv3x3 64->64 @256px B=20 (+BN statistics epilogue)'}
    for k, v in out.items():
        v = v[2:] if len(v) > 4 else v           # drop warm-up launches
        res[k + '_per_launch_raw'] = sum(v) / len(v)
        res[k + '_launches'] = len(v)
    # units and gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
    # FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE is exact for 16-B-per-lane stores
    rd = res.get('FETCH_SIZE_per_launch_raw', 0.0) * 1024 * 2
    wr = res.get('WRITE_SIZE_per_launch_raw', 0.0) * 1024
    res['read_bytes_per_launch'] = rd
    res['write_bytes_per_launch'] = wr
    res['traffic_bytes_per_launch'] = rd + wr
    B, S, C = 20, 256, 64
    res['algorithmic_bytes_per_launch'] = B * (S // 2) ** 2 * C * 2 + B * S * S * C * 2 + 64 * 9 * C * 2
    print(json.dumps(res, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--collect':
        return collect(sys.argv[2:])
    sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
    import torch
    from sbagan import ops
    dev = torch.device('cuda:0')
    x = torch.randn((20, 64, 128, 128), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter((torch.randn((64, 64, 3, 3), device=dev) / 24).contiguous(memory_format=torch.channels_last))
    pw = ops.PackedWeight(w)
    ops.ARENA.begin(dev)
    for _ in range(10):
        ops.conv_forward(x, pw, '3x3up', want_stats=True)
    torch.cuda.synchronize()
    ops.ARENA.end()


if __name__ == '__main__':
    main()
