#!/usr/bin/env python3
"""HBM traffic of the step's time-dominant kernel (igemm_dma2_kernel<64, 64, 32, 32, 4>) from the PMC counters.

bench.py --dump-shapes writes the shapes this kernel handles in one step (geometry + launches per step); this script
launches every one of them that many times (after one warm-up launch each) for rocprofv3 PMC passes, one counter per pass
(MI355X_MICROARCH.md, HBM section):

    python bench.py --child --steps 2 --warmup 2 --no-cpu-baseline --no-also --dump-shapes gpurun_out/dominant_shapes.json
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 tools/pmc_dominant.py gpurun_out/dominant_shapes.json
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 tools/pmc_dominant.py gpurun_out/dominant_shapes.json
    python3 tools/pmc_dominant.py --collect gpurun_out/dominant_shapes.json out_f out_w > profiles/rNN_pmc_dominant_kernel.json
(tools/pmc_dominant.sh runs the four steps.)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = 'igemm_dma2_kernel<64, 64, 32, 32, 4>'


def collect(shapes_path, dirs):
    spec = json.load(open(shapes_path))
    per = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if KERNEL.replace(' ', '') in r['Kernel_Name'].replace(' ', '')]
            for r in rows:
                per.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    launches = sum(s['count'] for s in spec['shapes'])
    warm = len(spec['shapes'])
    res = {'kernel': KERNEL, 'launches_per_step': launches, 'distinct_shapes': len(spec['shapes'])}
    for k, v in per.items():
        # dispatch order of main(): per shape one warm-up launch, then `count` launches -- drop the warm-ups
        keep, i = [], 0
        for s in spec['shapes']:
            keep += v[i + 1:i + 1 + s['count']]
            i += 1 + s['count']
        res[k + '_dispatches_seen'] = len(v)
        res[k + '_per_launch_raw'] = sum(keep) / max(1, len(keep))
        assert len(v) == launches + warm, (k, len(v), launches + warm)
    # units and gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE tallies
    # 128-B requests at 64 B -> x2; WRITE_SIZE is exact for 16-B-per-lane stores
    rd = res.get('FETCH_SIZE_per_launch_raw', 0.0) * 1024 * 2
    wr = res.get('WRITE_SIZE_per_launch_raw', 0.0) * 1024
    res['read_bytes_per_launch'], res['write_bytes_per_launch'] = rd, wr
    res['traffic_bytes_per_launch'] = rd + wr
    alg = 0.0
    for s in spec['shapes']:
        g = s['geom']
        M = g['N'] * g['OHs'] * g['OWs']
        alg += s['count'] * 2.0 * (g['N'] * g['IH'] * g['IW'] * g['Cin'] + g['Cout'] * g['ntaps'] * g['Cin'] + M * g['Cout'])
    res['algorithmic_bytes_per_launch'] = alg / launches
    print(json.dumps(res, indent=1))


def main():
    if sys.argv[1] == '--collect':
        return collect(sys.argv[2], sys.argv[3:])
    sys.path.insert(0, os.path.join(ROOT, 'sba-gan_amd'))
    import ctypes

    import torch

    import bench
    from sbagan import ops
    from sbagan._lib import call
    spec = json.load(open(sys.argv[1]))
    dev = torch.device('cuda:0')
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    for s in spec['shapes']:
        g = bench.geom_from_dict(s['geom'])
        xcs, ycs = g.x_cstride or g.Cin, g.y_cstride or g.Cout
        x = torch.randn(g.N, g.IH, g.IW, xcs, device=dev).bfloat16()
        w = (torch.randn(g.Cout, g.ntaps, g.Cin, device=dev) / (g.Cin * g.ntaps) ** 0.5).bfloat16()
        y = torch.empty(g.N, g.OH, g.OW, ycs, device=dev, dtype=torch.bfloat16)
        for _ in range(1 + s['count']):
            call('sba_conv_igemm', 1, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, ctypes.byref(g),
                 ws.data_ptr(), ops.WORKSPACE_BYTES, st)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
