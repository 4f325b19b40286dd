#!/usr/bin/env python3
"""Benchmark of the SBA-GAN adversarial training step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full G+D update (trainer.py:245-299 of the reference: G forward, three
discriminator updates, generator update incl. DAMSM loss and image encoder, Adam, EMA) on
one synthetic CUB-shaped batch of B=20 per GPU (cfg/bird_style.yml), 3 stages 64/128/256 px,
inputs resident in HBM.  Rank 0 prints ONE JSON line (see README / DESIGN.md section 6).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'sba-gan_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic GFLOP per image of the step (BASELINE.md section 2, 2*MAC, dead D weight-grads excluded)
GFLOP_PER_IMG = {1: 6.39, 2: 26.59, 3: 113.95}
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}        # MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=20, help='per-GPU batch (cfg/bird_style.yml: 20)')
    ap.add_argument('--branch', type=int, default=3, help='TREE.BRANCH_NUM: 1/2/3 = 64/128/256 px')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--variant', default='model', choices=['model', 'bert', 'mix'])
    ap.add_argument('--image-encoder', default='inception', choices=['inception', 'inception-miopen', 'standin'],
                    help='CNN_ENCODER inside the G step: the Inception-v3 trunk on the HIP kernels '
                         '(sbagan.inception_hip), the same module through PyTorch-ROCm/MIOpen, or the light '
                         'stand-in used by the parity fixtures')
    ap.add_argument('--text-encoder', default='rnn', choices=['rnn', 'bert', 'none'],
                    help="rnn: the frozen RNN_ENCODER forward (trainer.py:248-252) runs inside every timed step "
                         "(hand-written bi-LSTM, caption lengths read on the device); bert: the frozen BertEncoder of "
                         "the bert / mix variants (BASELINE config 3, sbagan.bert_hip); none: embeddings are inputs")
    ap.add_argument('--graph', type=int, default=2,
                    help='0: eager launches; 1: replay the step from captured hipGraphs (hipGraphLaunch); 3: the '
                         'native multi-stream launch replayer (csrc/replay.hip) over the captured step; 2: build '
                         'all of them, time a few untimed probe steps of each during warmup and keep the fastest')
    ap.add_argument('--phases', action='store_true', help='also print per-phase times of an eager step (stderr)')
    ap.add_argument('--attn-fp8', action='store_true',
                    help='BASELINE config 5: FP8 (e4m3) operands for the attention key projection GEMM')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='skip the extra 64 px / 128 px / f32 lines')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dump-shapes', default=None, help='write the shapes of the time-dominant kernel (for '
                                                        'tools/pmc_dominant.py) to this JSON file')
    ap.add_argument('--child', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


def build(args, dev):
    from miscc.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(ROOT, 'sba-gan_amd', 'cfg', 'bird_style.yml'))
    cfg.TREE.BRANCH_NUM = args.branch
    cfg.TRAIN.BATCH_SIZE = args.batch
    cfg.TRAIN.NET_G = ''
    import model
    import model_bert
    from miscc.utils import weights_init
    from sbagan import ops
    from sbagan.trainer import GANStep
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    ops.set_attention_fp8(args.attn_fp8)
    netG = {'model': model.G_NET, 'bert': model_bert.G_NET, 'mix': model_bert.G_NET_MIX}[args.variant]()
    netsD = [model.D_NET64(), model.D_NET128(), model.D_NET256()][:args.branch]
    torch.manual_seed(100)
    netG.apply(weights_init)
    for d in netsD:
        d.apply(weights_init)
    netG.to(dev).train()
    for d in netsD:
        d.to(dev).train()
    netG.set_return_attention(False)        # unused in training (trainer.py:262)
    if args.image_encoder == 'inception':
        from sbagan.inception_hip import InceptionHIP
        torch.manual_seed(101)
        enc_mod = model.CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM).to(dev).eval()
        enc = InceptionHIP(enc_mod)
    elif args.image_encoder == 'inception-miopen':
        enc_mod = model.CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM).to(dev).eval()
        for p in enc_mod.parameters():
            p.requires_grad = False
        enc_mod = enc_mod.to(memory_format=torch.channels_last)
        amp = args.dtype == 'bf16'

        def enc(x):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
                f, c = enc_mod(x)
            return f.float(), c.float()
    else:       # diagnostic only: the step without the Inception trunk (a 1x1 conv on a 17x17 average pool)
        enc = _LightEncoder(cfg.TEXT.EMBEDDING_DIM, dev)
    step = GANStep(netG, netsD, enc, args.batch,
                   distributed=(args.gpus > 1 or os.environ.get('SBA_BENCH_FORCE_DIST', '0') == '1'))
    step.overlap_wgrad_d = os.environ.get('SBA_OVERLAP_WGRAD_D', '0') == '1'
    step.overlap_wgrad = os.environ.get('SBA_OVERLAP_WGRAD', '1') == '1'
    step.concurrent_d = os.environ.get('SBA_CONCURRENT_D', '1') == '1'
    return step


def measure_dominant_kernel(args, dev):
    """Roofline of the dominant kernel: the conv of the last generator upBlock (nearest x2 + conv3x3
    64->64 at 256x256, B images) exactly as the step launches it -- BatchNorm statistics in the
    epilogue included -- timed with events on the launch stream.  Algorithmic FLOP = 2*9*Cin*Cout*H*W*B."""
    from sbagan import ops
    dt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    B, C, S = args.batch, 64, 64 * 2 ** (args.branch - 1)
    if args.branch == 1:
        B, C, S = args.batch, 128, 64      # h_net1.upsample4: 128 -> 64 at 32 -> 64 px
    x = torch.randn((B, C, S // 2, S // 2), device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter((torch.randn((64, C, 3, 3), device=dev) / 24).contiguous(memory_format=torch.channels_last))
    pw = ops.PackedWeight(w)
    for _ in range(3):
        ops.conv_forward(x, pw, '3x3up')
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ops.ARENA.begin(dev)                   # statistics accumulators come from the step's pre-zeroed arena
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ops.conv_forward(x, pw, '3x3up', want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    ops.ARENA.end()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * 9 * C * 64 * S * S * B
    achieved = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.dtype]
    kname = 'conv3x3_halo_kernel<64,64,ups>' if (args.dtype == 'bf16' and C == 64) else 'igemm_kernel<%s>' % args.dtype
    # HBM traffic per launch: rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> bytes) of
    # tools/pmc_dominant.py, committed under profiles/ (cannot be collected from inside this process)
    traffic = None      # (PMC traffic is reported for the time-dominant kernel: measure_igemm_kernels)
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': traffic,
            'kernel': '%s upBlock conv3x3 %d->64 @%dpx B=%d (+BN statistics epilogue)' % (kname, C, S, B),
            'kernel_ms': round(ms, 4), 'algorithmic_gflop_per_launch': round(flops / 1e9, 2)}


DOMINANT = (1, 1)      # (family, tile) of sba_conv_igemm_plan: igemm_dma2_kernel<64, 64, 32, 32, 4> -- the kernel with the
#                        largest share of the step's kernel time (profiles/r03_step_*_summary.txt: ~120 launches per step)
DOMINANT_NAME = 'igemm_dma2_kernel<64, 64, 32, 32, 4>'
GEOM_FIELDS = ('N', 'IH', 'IW', 'Cin', 'OH', 'OW', 'Cout', 'OHs', 'OWs', 'sy', 'sx', 'osy', 'osx', 'ooy', 'oox', 'ups',
               'ntaps', 'x_cstride', 'x_coff', 'y_cstride', 'y_coff', 'relu', 'tile', 'ksplit')


def geom_to_dict(g):
    d = {k: int(getattr(g, k)) for k in GEOM_FIELDS}
    d['ty'], d['tx'] = [int(g.ty[t]) for t in range(g.ntaps)], [int(g.tx[t]) for t in range(g.ntaps)]
    return d


def geom_from_dict(d):
    from sbagan._lib import ConvGeom
    g = ConvGeom()
    for k in GEOM_FIELDS:
        setattr(g, k, d[k])
    for t in range(d['ntaps']):
        g.ty[t], g.tx[t] = d['ty'][t], d['tx'][t]
    return g


def time_shape(g, dev, ws, st, n=10):
    """one implicit-GEMM geometry launched alone, n times back to back, with events on the launch stream: us per launch"""
    import ctypes
    from sbagan import ops
    from sbagan._lib import call
    xcs, ycs = g.x_cstride or g.Cin, g.y_cstride or g.Cout
    x = torch.randn(g.N, g.IH, g.IW, xcs, device=dev).bfloat16()
    w = (torch.randn(g.Cout, g.ntaps, g.Cin, device=dev) / (g.Cin * g.ntaps) ** 0.5).bfloat16()
    y = torch.empty(g.N, g.OH, g.OW, ycs, device=dev, dtype=torch.bfloat16)

    def run():
        call('sba_conv_igemm', 1, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, ctypes.byref(g),
             ws.data_ptr(), ops.WORKSPACE_BYTES, st)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def measure_igemm_kernels(args, dev, step, one_step, dump=None):
    """The implicit-GEMM launches of one step (forward / data-gradient convs of G, the discriminators and the Inception
    trunk through sba_conv_igemm, all tile configurations; the grouped Inception launches are listed, not timed here).
    The geometries of one eager step are recorded, each distinct one is timed ALONE with events on the launch stream
    (10 launches), and attributed to the kernel sba_conv_igemm_plan names for it.  Returns (roofline of the
    time-dominant KERNEL, roofline of the whole family)."""
    import ctypes
    from sbagan import ops
    from sbagan._lib import ConvGeom, call
    if args.dtype != 'bf16':
        return None, None
    ops.IGEMM_LOG = []
    one_step()
    torch.cuda.synchronize()
    log, ops.IGEMM_LOG = ops.IGEMM_LOG, None
    uniq, grouped = {}, 0
    for g in log:
        if isinstance(g, tuple):
            grouped += 1
            continue
        k = (ops.geom_key(g), g.x_cstride, g.y_cstride, g.tile, g.ksplit)
        if k not in uniq:
            uniq[k] = [g, 0]
        uniq[k][1] += 1
    ws = ops.workspace(dev)
    st = torch.cuda.current_stream().cuda_stream
    fam = {'t': 0.0, 'fl': 0.0, 'n': 0}
    dom = {'t': 0.0, 'fl': 0.0, 'by': 0.0, 'n': 0, 'shapes': []}
    plan = (ctypes.c_int * 3)()
    for g0, count in uniq.values():
        g = ConvGeom()
        ctypes.memmove(ctypes.byref(g), ctypes.byref(g0), ctypes.sizeof(ConvGeom))
        us = time_shape(g, dev, ws, st)
        M = g.N * g.OHs * g.OWs
        fl = 2.0 * M * g.Cout * g.ntaps * g.Cin
        fam['t'] += us * count
        fam['fl'] += fl * count
        fam['n'] += count
        call('sba_conv_igemm_plan', 1, ctypes.byref(g), ops.WORKSPACE_BYTES, plan)
        if (plan[0], plan[1]) == DOMINANT:
            # algorithmic bytes: every input pixel / weight / output element once (bf16)
            by = 2.0 * (g.N * g.IH * g.IW * g.Cin + g.Cout * g.ntaps * g.Cin + M * g.Cout)
            dom['t'] += us * count
            dom['fl'] += fl * count
            dom['by'] += by * count
            dom['n'] += count
            dom['shapes'].append({'geom': geom_to_dict(g), 'count': count, 'us': round(us, 2), 'ksplit': int(plan[2])})
    if dump:
        with open(dump, 'w') as f:
            json.dump({'kernel': DOMINANT_NAME, 'shapes': dom['shapes']}, f)
    peak = PEAK_TFLOPS['bf16']
    family = {'bound': 'mfma', 'achieved': round(fam['fl'] / (fam['t'] * 1e-6) / 1e12, 2), 'peak': peak, 'unit': 'TFLOP/s',
              'frac': round(fam['fl'] / (fam['t'] * 1e-6) / 1e12 / peak, 4), 'traffic': None,
              'kernel': 'sba_conv_igemm single launches, all kernels (igemm_dma2 / igemm_dma / conv3x3_halo / igemm): %d '
                        'launches, %d distinct shapes per step (+ %d grouped Inception launches, not in this sum)'
                        % (fam['n'], len(uniq), grouped),
              'kernel_ms_per_step': round(fam['t'] / 1e3, 3), 'algorithmic_gflop_per_step': round(fam['fl'] / 1e9, 1)}
    if not dom['n']:
        return None, family
    traffic = None
    for tag in ('r04', 'r03'):       # the newest committed PMC record whose launch set is the one measured here
        pmc = os.path.join(ROOT, 'profiles', '%s_pmc_dominant_kernel.json' % tag)
        if traffic is None and args.branch == 3 and args.batch == 20 and args.variant == 'model' and os.path.exists(pmc):
            try:    # rocprofv3 PMC passes over exactly these shapes (tools/pmc_dominant.py; not collectable in-process)
                rec = json.load(open(pmc))
                if rec.get('launches_per_step') == dom['n']:
                    traffic = int(rec['traffic_bytes_per_launch'])
            except (KeyError, ValueError):
                traffic = None
    ach = dom['fl'] / (dom['t'] * 1e-6) / 1e12
    dominant = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': traffic,
                'kernel': '%s: the time-dominant kernel of the step (%d launches per step over %d distinct shapes: '
                          'Inception 17x17 / 8x8 convs, discriminator tails, small generator layers); achieved = '
                          'sum(2*M*Cout*K) / sum(duration), each shape timed alone' % (DOMINANT_NAME, dom['n'],
                                                                                       len(dom['shapes'])),
                'launches_per_step': dom['n'], 'kernel_avg_us': round(dom['t'] / dom['n'], 2),
                'kernel_ms_per_step': round(dom['t'] / 1e3, 3),
                'algorithmic_gflop_per_launch': round(dom['fl'] / dom['n'] / 1e9, 3),
                'algorithmic_bytes_per_launch': int(dom['by'] / dom['n'])}
    return dominant, family


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference step, fp32, `kind: port`) on the host cores, with the SAME work as
    the GPU line: the generator-loss term runs the Inception-v3 CNN_ENCODER (the repo's nn.Module definition evaluated
    by PyTorch on the CPU, frozen, eval mode: forward + backward to the fake image) exactly as the GPU step runs it on
    the HIP kernels.  SURVEY.md 8d's three configurations on a bounded sample (~30 s of CPU work): 3-stage B=4: 1 warm-up
    + 4 timed steps, median (= `value`); stage 1 only B=4: 2 warm-up + 6 timed; 3-stage B=20: one step."""
    import statistics
    from oracle import fill
    from oracle import sbagan_oracle as O
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import FULL, SMOOTH, d_shapes, g_shapes, make_inputs
    # the GPU box exposes many host cores but grants a CPU share of ~16 per GPU: oversubscribing
    # torch's intra-op pool makes the baseline (and the bench) crawl
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    from sbagan.encoders import CNN_ENCODER
    torch.manual_seed(101)
    enc_mod = CNN_ENCODER(256).eval()
    for p in enc_mod.parameters():
        p.requires_grad_(False)

    def enc(x):
        return enc_mod(x)

    def run(branch, B, warm, timed):
        x = make_inputs(FULL, B, 18, branch=branch, lmax=18, tag=500)
        PG = fill.fill_state_dict(g_shapes(FULL, branch, 'model'))
        PDs = [fill.fill_state_dict(d_shapes(FULL, i), salt=i) for i in range(branch)]
        st = O.OracleState(PG, PDs)
        ts = []
        for s in range(warm + timed):
            t0 = time.time()
            O.train_step(st, x['imgs'], x['sent'], x['words'], x['mask'], x['cap_lens'], x['class_ids'],
                         fill.unit((B, 100), 550 + s), fill.unit((B, 100), 560 + s), enc, SMOOTH)
            if s >= warm:
                ts.append(time.time() - t0)
        return B / statistics.median(ts)
    main = run(args.branch, 4, 1, 4)
    also = [{'config': 'stage 1 only (64 px), B=4: 2 warm-up + 6 timed steps, median', 'value': round(run(1, 4, 2, 6), 3)}]
    if args.branch == 3:
        also.append({'config': '3-stage, B=20: one step', 'value': round(run(3, 20, 0, 1), 3)})
    return {'value': round(main, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d-stage step at B=4 (bird_style dims, fp32, Inception-v3 image encoder INCLUDED: forward + '
                      'backward to the fake image, as in the GPU line), 1 warm-up + 4 timed steps, median; torch CPU '
                      'threads=%d' % (args.branch, cores), 'also': also}


class _LightEncoder(object):
    """--image-encoder standin: regions = conv1x1(avgpool -> 17x17), code = Linear(global mean).  Isolates the
    cost of the hand-written Inception trunk; never the reported configuration."""

    def __init__(self, nef, dev):
        g = torch.Generator().manual_seed(7)
        self.wr = (torch.rand((nef, 3, 1, 1), generator=g) * 1.6 - 0.8).to(dev)
        self.wc = (torch.rand((nef, 3), generator=g) * 1.6 - 0.8).to(dev)
        self.bc = (torch.rand((nef,), generator=g) * 0.2 - 0.1).to(dev)

    def __call__(self, x):
        import torch.nn.functional as F
        return F.conv2d(F.adaptive_avg_pool2d(x, 17), self.wr), F.linear(x.mean((2, 3)), self.wc, self.bc)


def supervise(args):
    """Single-GPU runs execute in a child process so that a failure of the optional hipGraph
    capture (a ROCm runtime crash cannot be caught in-process) degrades to eager launches instead
    of losing the measurement.  The parent never touches the GPU."""
    import subprocess
    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:]] + ['--child']
    ladder = [[], ['--graph', '1'], ['--graph', '0']]     # all launch modes -> hipGraph only -> eager

    def child(extra):
        p = subprocess.run(base + extra, stdout=subprocess.PIPE, text=True)
        lines = [l for l in p.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
        return (json.loads(lines[-1]) if p.returncode == 0 and lines else None), p.returncode
    res = None
    for n, extra in enumerate(ladder):
        res, rc = child(extra)
        if res is not None:
            break
        sys.stderr.write('bench child failed (rc=%d)%s\n' % (rc, '; retrying with %s' % ' '.join(ladder[n + 1])
                                                               if n + 1 < len(ladder) else ''))
    if res is None:
        return 1
    if not args.no_also and args.branch == 3 and args.dtype == 'bf16':
        # the other points the metric names (64 / 128 px) and the fp32 step (the parity-compliant number: bf16
        # storage costs ~2e-3 on a discriminator loss at B=4, tests/test_step_gpu.py), shorter runs
        also = []
        short = ['--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-roofline', '--no-also']
        for label, extra in (('256px f32', ['--dtype', 'f32']), ('128px bf16', ['--branch', '2']),
                             ('64px bf16', ['--branch', '1'])):
            r, _ = child(short + extra)
            if r is not None:
                also.append({'what': label, 'metric': r['metric'], 'value': r['value'], 'ms_per_step': r['ms_per_step'],
                             'dtype': r['dtype'], 'launch': r['config']['launch']})
        res['also'] = also
    print(json.dumps(res), flush=True)
    return 0


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1 and args.gpus <= 1 and args.graph and not args.child:
        sys.exit(supervise(args))
    # SBA_BENCH_FORCE_DIST=1: run the multi-rank code path (RCCL communicator, eager all-reduce between the
    # per-network graphs) with ONE rank -- the only way to exercise RCCL on a 1-GPU development box
    force_dist = os.environ.get('SBA_BENCH_FORCE_DIST', '0') == '1'
    if args.gpus > 1 or world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # SBA_BENCH_BACKEND=gloo rehearses the multi-rank control flow with several ranks on ONE card
        # (development aid; RCCL refuses duplicate devices).  The driver's runs use nccl = RCCL.
        backend = os.environ.get('SBA_BENCH_BACKEND', 'nccl')
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local)
    multi = dist.is_initialized()
    from sbagan.synth import synthetic_batch
    step = build(args, dev)
    b = synthetic_batch(args.batch, branch_num=args.branch, device=dev, seed=100 + rank)
    noise_shape = (2, args.batch, 100) if args.variant == 'mix' else (args.batch, 100)
    noise = torch.empty(noise_shape, device=dev)
    torch.manual_seed(100 + rank)

    txt = None
    if args.text_encoder == 'rnn':
        import model
        torch.manual_seed(102)                    # the frozen encoder is identical on every rank
        txt = model.RNN_ENCODER(5450, nhidden=b['sent_emb'].size(1)).to(dev).eval()   # CUB vocabulary size
        torch.manual_seed(100 + rank)
        hid = txt.init_hidden(args.batch)

    bert = None
    if args.text_encoder == 'bert':
        import model_bert
        torch.manual_seed(102)
        bert = model_bert.BertEncoder(b['sent_emb'].size(1)).to(dev).eval()
        torch.manual_seed(100 + rank)
        bert_caps = torch.randint(1000, 30522, (args.batch, 20), device=dev)
        b['words_embs'] = torch.zeros((args.batch, b['sent_emb'].size(1), 20), device=dev)   # BERT path: L = 20 always
        b['mask'] = torch.zeros((args.batch, 20), dtype=torch.bool, device=dev)
        b['cap_lens'] = torch.full((args.batch,), 20, dtype=torch.int64, device=dev)

    def encode():
        # words_embs, sent_emb = text_encoder(captions, cap_lens, hidden) of trainer.py:248-252 (no_grad, eval)
        if bert is not None:
            with torch.no_grad():
                w, s = bert(bert_caps)
            b['words_embs'].copy_(w)
            b['sent_emb'].copy_(s)
        if txt is not None:
            with torch.no_grad():       # written straight into the step's static input tensors
                txt(b['captions'], b['cap_lens'], hid, max_len=b['words_embs'].size(2),
                    out=(b['words_embs'], b['sent_emb']))

    def one_step():
        noise.normal_(0, 1)
        encode()
        return step.step(b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'],
                         noise)

    if args.phases:
        for _ in range(3):
            one_step()
        step.phase_events = []
        one_step()
        torch.cuda.synchronize()
        ev = step.phase_events
        step.phase_events = None
        sys.stderr.write('phases (eager, ms): ' + ', '.join(
            '%s %.2f' % (ev[i + 1][0], ev[i][1].elapsed_time(ev[i + 1][1])) for i in range(len(ev) - 1)) + '\n')
    graph = None
    mode = 'eager'
    n_eager = max(args.warmup, 3) if args.graph else args.warmup
    for _ in range(n_eager):
        out = one_step()
    torch.cuda.synchronize()

    def probe(fn, n=4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    # Data-parallel launch mode (N > 1), SBA_DP_REPLAY:
    #   4 (default) the WHOLE data-parallel step as ONE recording through the native replayer, the gradient exchanges and the
    #     deferred generator update as host-call nodes (sbagan.trainer.ReplayedStep + ExchangeRecorder): every collective
    #     where the eager data-parallel step has it -- D_NET128 / D_NET256 in two buckets under their backward passes, the
    #     generator's exchange behind the next step's text encoder + real-image forwards -- and the phases overlapping as on
    #     one GPU;
    #   3 / 2 / 1 sbagan.trainer.ReplayedStepDP: several recordings with the exchange between them (3: deferred generator
    #     update + image encoder beside the discriminators' exchange, 2: no deferral, 1: the exchange fully exposed);
    #   0 the per-phase hipGraphs (GraphedStep).
    # Every rank reads the same environment; the collective-bearing warm-up step runs OUTSIDE the try block (a rank that
    # threw inside it would leave its peers blocked in the step's all-reduces); the captures issue no collective, and a rank
    # whose capture fails makes ALL ranks fall back to the per-phase graphs.
    dp_mode = os.environ.get('SBA_DP_REPLAY', '4')
    dp_graph = None
    if args.graph and multi and dp_mode in ('1', '2', '3', '4'):
        from sbagan.trainer import ReplayedStep, ReplayedStepDP
        if os.environ.get('SBA_DP_OVERLAP_G', '1') == '0':
            # the alternative measured in profiles/r04_dp_overlap_g.txt: the generator's 30 MB exchange waited for where it is
            # issued (exposed), the discriminator loss as ONE grouped real|fake pass instead of two passes
            step.overlap_g = False
        flags = (step.overlap_g, step.bucket_d)
        a = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
        nstreams = int(os.environ.get('SBA_REPLAY_STREAMS', '4'))
        if dp_mode == '4':
            warm = ReplayedStep.warm_up(step, *a, recorded_prologue=encode)
        else:
            warm = ReplayedStepDP.warm_up(step, *a, recorded_prologue=encode, defer_g=dp_mode == '3')
        try:
            if dp_mode == '4':
                dp_graph = ReplayedStep(step, *a, recorded_prologue=encode, max_streams=nstreams, verbose=True, warm=warm)
            else:
                dp_graph = ReplayedStepDP(step, *a, recorded_prologue=encode, max_streams=nstreams,
                                          e_beside_exchange=dp_mode in ('2', '3'), defer_g=dp_mode == '3', warm=warm)
        except Exception as e:
            sys.stderr.write('data-parallel launch replayer unavailable (%s: %s)\n' % (type(e).__name__, e))
            dp_graph = None
            torch.cuda.synchronize()
        ok = torch.tensor([1 if dp_graph is not None else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) == 0:
            dp_graph = None
            step.finish()
            step.overlap_g, step.bucket_d = flags
    if dp_graph is not None:
        graph = dp_graph
        if dp_mode == '4':      # (a step with its collectives: after the ranks have agreed on the launch mode)
            graph.prioritize(verbose=int(os.environ.get("SBA_REPLAY_PRIO_VERBOSE", "1")))
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        mode, out = 'replayer-dp%s' % dp_mode, graph.out
        sys.stderr.write('launch probe (ms per step): %s %.2f\n' % (mode, probe(graph.replay) * 1e3))
    elif args.graph:
        cands = {}
        a = (b['imgs'], b['sent_emb'], b['words_embs'], b['mask'], b['cap_lens'], b['class_ids'], noise)
        if args.graph in (1, 2):
            try:
                from sbagan.trainer import GraphedStep
                g1 = GraphedStep(step, *a, prologue=lambda: (noise.normal_(0, 1), encode()),
                                 single=os.environ.get('SBA_GRAPH_SINGLE', '0') == '1')
                for _ in range(2):
                    g1.replay()
                torch.cuda.synchronize()
                cands['hipgraph'] = g1
            except Exception as e:      # capture is an optimisation, never a requirement
                sys.stderr.write('graph capture failed (%s: %s)\n' % (type(e).__name__, e))
                torch.cuda.synchronize()
        if args.graph in (1, 2) and os.environ.get('SBA_PHASE_REPLAY', '0') == '1':
            # the same per-phase captures re-issued by the native replayer (GraphedStep(native=True): the gradient exchange
            # stays between the phases).  Bit-identical to the eager step (tests/test_determinism_gpu.py) but measured
            # SLOWER than hipGraphLaunch of the same phases (14.7 against 14.0 ms single-GPU, 16.1 against 15.5 with the
            # data-parallel decomposition): ten replayers' fork streams on four hardware queues.  Off by default.
            try:
                from sbagan.trainer import GraphedStep
                g2 = GraphedStep(step, *a, prologue=lambda: noise.normal_(0, 1), recorded_prologue=encode, native=True)
                for _ in range(2):
                    g2.replay()
                torch.cuda.synchronize()
                cands['replayed-phases'] = g2
            except Exception as e:
                import traceback
                sys.stderr.write('per-phase launch replayer unavailable (%s: %s)\n' % (type(e).__name__, e))
                if os.environ.get('SBA_BENCH_TRACEBACK') == '1':
                    traceback.print_exc()
                torch.cuda.synchronize()
        if args.graph in (2, 3) and not multi:
            try:
                from sbagan.trainer import ReplayedStep
                g3 = ReplayedStep(step, *a, recorded_prologue=encode,
                                  max_streams=int(os.environ.get('SBA_REPLAY_STREAMS', '4')), verbose=True)
                g3.prioritize(verbose=int(os.environ.get('SBA_REPLAY_PRIO_VERBOSE', '1')))
                for _ in range(2):
                    g3.replay()
                torch.cuda.synchronize()
                cands['replayer'] = g3
            except Exception as e:
                sys.stderr.write('launch replayer unavailable (%s: %s)\n' % (type(e).__name__, e))
                torch.cuda.synchronize()
        if cands:
            times = {k: probe(v.replay) for k, v in cands.items()}
            if world == 1 and args.graph == 2:
                times['eager'] = probe(one_step)
            sys.stderr.write('launch probe (ms per step): ' + ', '.join('%s %.2f' % (k, v * 1e3)
                                                                       for k, v in times.items()) + '\n')
            mode = min(times, key=times.get)
            graph = cands.get(mode)
            if graph is not None:
                out = graph.out
        if multi:           # every rank must issue the same sequence of collectives: graphs only if ALL captured
            ok = torch.tensor([1 if graph is not None else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok) == 0:
                graph, mode = None, 'eager'

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            out = one_step()
    # data-parallel: the last step's generator update is deferred (its all-reduce overlaps the next step's real-image
    # forwards); it belongs to the timed steps -- the first timed step applied the update of the last warm-up step
    (graph if (graph is not None and hasattr(graph, 'finish')) else step).finish()
    sync_all()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    finite = all(bool(torch.isfinite(v).all()) for v in out.values() if torch.is_tensor(v))
    if not finite and rank == 0:
        sys.stderr.write('WARNING: non-finite losses after the timed steps -- the measurement is not a valid '
                         'training step (config.losses_finite = false)\n')
    ips = world * args.batch * args.steps / dt
    res = {
        'metric': 'images/sec (G+D step) at %dpx' % (64 * 2 ** (args.branch - 1)),
        'value': round(ips, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'bird_style.yml %d-stage G+D step (64..%dpx), B=%d per GPU, G_NET variant=%s, '
                               'image_encoder=%s' % (args.branch, 64 * 2 ** (args.branch - 1), args.batch,
                                                     args.variant, args.image_encoder)
                               + (', text_encoder=RNN_ENCODER in the step' if args.text_encoder == 'rnn' else '')
                               + (', text_encoder=BertEncoder in the step' if args.text_encoder == 'bert' else '')
                               + (', attention key projection in fp8' if args.attn_fp8 else ''),
                   'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'launch': mode,
                   'losses_finite': finite},
        'step_tflops': round(GFLOP_PER_IMG[args.branch] * ips / 1e3, 2),
        'step_mfma_frac': round(GFLOP_PER_IMG[args.branch] * ips / 1e3 / world / PEAK_TFLOPS[args.dtype], 4),
    }
    if rank == 0:
        if not args.no_roofline:
            big = measure_dominant_kernel(args, dev)
            dominant, fam = measure_igemm_kernels(args, dev, step, one_step, dump=args.dump_shapes)
            # `roofline` = the TIME-dominant kernel; the largest-FLOP launch and the whole family ride along
            res['roofline'] = dominant if dominant is not None else big
            res['roofline_largest_flop_kernel'] = big
            if fam is not None:
                res['roofline_igemm_family'] = fam
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(res), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
