"""The adversarial training step of condGANTrainer.train (AttnGAN2/code/trainer.py:238-299)
re-driven over the HIP modules: same call order, Adam / EMA semantics and label tensors,
with the host-side inefficiencies of the reference removed (no .item() syncs, one fused
Adam+EMA launch per network instead of per-tensor loops, flat gradient buffers that are
zeroed with one memset and all-reduced with one RCCL call per network).

Ordering facts preserved: (a) G forward once, D steps use fake.detach();
(b) each D optimizer steps BEFORE generator_loss, so the G loss sees the updated D nets and
their BN running stats move once more in that forward; (c) RNG draw order: noise, then
CA_NET eps.  The discriminator weight-gradients the reference computes in the G step and
then discards (trainer.py:270,287) are not computed: D parameters are frozen for that
backward, which changes no result.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from miscc.config import cfg
from miscc.losses import (KL_loss, backward_with_image_grad, backward_with_image_grads, damsm_image_terms,
                          discriminator_fake_term, discriminator_loss, discriminator_real_term, generator_d_term,
                          generator_loss)

from . import ops
from ._lib import call


class FlatParams(object):
    """Re-homes every parameter of `net` into one contiguous f32 buffer (each keeps its own
    strides, e.g. channels_last conv weights) with matching flat buffers for gradients and
    Adam moments.  p.grad are persistent views, so kernels accumulate straight into the
    buffer that is all-reduced and consumed by the fused Adam."""

    def __init__(self, net, with_ema=False):
        params = [p for p in net.parameters()]
        self.params = params
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4          # 16-byte aligned starts
        dev = params[0].device
        self.n = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            k = p.numel()
            view = self.data[o:o + k].as_strided(p.shape, p.stride())
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + k].as_strided(p.shape, p.stride())
        self.offsets = offs
        self.offset_of = {id(p): o for p, o in zip(params, offs)}
        self.avg = self.data.clone() if with_ema else None
        self.epoch = [0]
        ops.register_epoch(params, self.epoch)
        self.gepoch = [1]                      # bumped by zero_grad: ops.conv_wgrad's first-write bookkeeping
        for p in params:
            p._sba_gepoch = self.gepoch
        ops.weights_changed()
        from . import nets
        self.packs = nets.pack_group(net)      # all packed conv weights of the network: one launch per step

    def zero_grad(self):
        self.grad.zero_()
        self.gepoch[0] += 1

    def ema_params(self):
        """EMA shadow as a list of tensors shaped like the parameters (copy_G_params order)."""
        return [self.avg[o:o + p.numel()].as_strided(p.shape, p.stride())
                for p, o in zip(self.params, self.offsets)]


class FusedAdam(object):
    """torch.optim.Adam(lr, betas=(0.5, 0.999)) of trainer.py:136-143 as one launch over a
    FlatParams buffer (+ the EMA of trainer.py:298-299 when the buffer has a shadow).  The
    step counter and bias corrections live on the device (graph replayable)."""

    def __init__(self, flat, lr, betas=(0.5, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, float(lr), betas, eps
        self.state = torch.zeros(4, dtype=torch.int32, device=flat.data.device)

    def step(self, grad_scale=1.0):
        self.prepare()
        self.step_range(0, self.flat.n, grad_scale)
        self.done()

    # the same update in pieces: prepare() once (step counter, bias corrections on the device), step_range over disjoint
    # element ranges of the flat buffer in any order / on any streams ordered behind prepare(), done() once
    def prepare(self):
        call('sba_adam_prepare', self.state.data_ptr(), self.lr, self.betas[0], self.betas[1],
             torch.cuda.current_stream().cuda_stream)

    def step_range(self, lo, hi, grad_scale=1.0):
        f = self.flat
        if hi <= lo:
            return
        b = 4 * lo
        call('sba_adam_step', f.data.data_ptr() + b, f.grad.data_ptr() + b, f.m.data_ptr() + b, f.v.data_ptr() + b,
             None if f.avg is None else f.avg.data_ptr() + b, None, self.state.data_ptr(), hi - lo,
             self.betas[0], self.betas[1], self.eps, float(grad_scale), torch.cuda.current_stream().cuda_stream)

    def done(self):
        self.flat.epoch[0] += 1          # this network's packed weights are stale; the others are not


def prepare_labels(batch_size, device):
    """trainer.py:147-157."""
    real = torch.ones(batch_size, dtype=torch.float32, device=device)
    fake = torch.zeros(batch_size, dtype=torch.float32, device=device)
    match = torch.arange(batch_size, dtype=torch.int64, device=device)
    return real, fake, match


def build_mask(captions, num_words):
    """trainer.py:253-256: mask = (captions == 0)[:, :Lmax]  (integer work, bit-exact)."""
    mask = (captions == 0)
    if mask.size(1) > num_words:
        mask = mask[:, :num_words]
    return mask


def sort_by_caption_length(captions_lens):
    """datasets.py:32-33: descending sort of the caption lengths, returns (lens, permutation)."""
    return torch.sort(captions_lens, 0, True)


class _RecordedHandle(object):
    """what GradExchange.start returns while a step is being RECORDED: the tag of the host-call node that will start
    the exchange at that point of every replay"""
    __slots__ = ('tag',)

    def __init__(self, tag):
        self.tag = tag


class ExchangeRecorder(object):
    """Capture-time stand-in for everything of the data-parallel step that cannot be recorded into the launch replayer's
    graph -- torch.distributed calls (RCCL collectives are host API calls on their own stream) and launches that depend
    on host state (the deferred generator update).  While it is installed (GradExchange.recorder), start() / wait() /
    host() emit a HOST-CALL node (include/sbagan_hip.h: sba_replay_marker) on the current stream and remember what to
    do there; dispatch() is the replay-time callback: it runs the remembered action on the stream the replayer gave the
    node, so the exchange sits between the recorded launches exactly where the eager step has it."""

    def __init__(self, exchange):
        self.exchange = exchange
        self.calls = []                 # tag -> ('start', tensor) | ('wait', start tag) | ('host', callable)
        self.live = {}                  # start tag -> handle of the exchange started by the current replay
        self.recorded_update = False    # the recording holds the deferred generator update as launches (GANStep.finish)

    def _mark(self, entry):
        tag = len(self.calls)
        self.calls.append(entry)
        call('sba_replay_marker', tag, torch.cuda.current_stream().cuda_stream)
        return tag

    def start(self, flat_grad):
        return _RecordedHandle(self._mark(('start', flat_grad)))

    def wait(self, handle):
        self._mark(('wait', handle.tag))

    def host(self, fn):
        self._mark(('host', fn))

    def dispatch(self, tag, stream_ptr, device):
        kind, arg = self.calls[tag]
        with torch.cuda.stream(torch.cuda.ExternalStream(int(stream_ptr or 0), device=device)):
            if kind == 'start':
                self.live[tag] = self.exchange.start(arg)
            elif kind == 'wait':
                self.exchange.wait(self.live.pop(arg, None))
            else:
                arg()


class GradExchange(object):
    """Data-parallel gradient exchange (new functionality; the reference is single-GPU, SURVEY.md
    8e): ONE sum all-reduce per network over its flat f32 gradient buffer, issued on a side
    stream so that it overlaps the next network's forward/backward; the 1/world mean is folded
    into the fused Adam (grad_scale).  Backend = torch.distributed: "nccl" is RCCL over xGMI on
    ROCm; "gloo" on CPU tensors is used by the world_size-2 tests."""

    def __init__(self, device, enabled=True):
        self.enabled = bool(enabled) and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.enabled else 1
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if (self.enabled and self.device.type == 'cuda') else None
        self.recorder = None            # an ExchangeRecorder while a step is being recorded (ReplayedStep)

    def start(self, flat_grad):
        if not self.enabled:
            return None
        if self.recorder is not None:
            return self.recorder.start(flat_grad)
        if self.stream is None:
            return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True)
        if dist.get_backend() == 'gloo':
            # development / test path (several ranks on one card): gloo reduces host memory
            host = flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat_grad.copy_(host)
            return None
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return self.stream

    def wait(self, handle):
        if handle is None:
            return
        if isinstance(handle, _RecordedHandle):
            self.recorder.wait(handle)
            return
        if isinstance(handle, torch.cuda.Stream):
            torch.cuda.current_stream().wait_stream(handle)
        else:
            handle.wait()


class GANStep(object):
    """One G+D update.  `netsD` is a list (D_NET64[, D_NET128[, D_NET256]]); `image_encoder` maps
    the last fake image to (region features B x nef x 17 x 17, global code B x nef)."""

    def __init__(self, netG, netsD, image_encoder, batch_size, lr_g=None, lr_d=None, distributed=False):
        self.netG, self.netsD, self.image_encoder = netG, netsD, image_encoder
        dev = next(netG.parameters()).device
        self.device = dev
        self.batch_size = batch_size
        self.flatG = FlatParams(netG, with_ema=True)
        self.flatD = [FlatParams(d) for d in netsD]
        self.optG = FusedAdam(self.flatG, cfg.TRAIN.GENERATOR_LR if lr_g is None else lr_g)
        self.optD = [FusedAdam(f, cfg.TRAIN.DISCRIMINATOR_LR if lr_d is None else lr_d) for f in self.flatD]
        self.real_labels, self.fake_labels, self.match_labels = prepare_labels(batch_size, dev)
        self.exchange = GradExchange(dev, enabled=distributed)
        self.distributed, self.world = self.exchange.enabled, self.exchange.world
        self._d_params = [p for d in netsD for p in d.parameters()]
        # data-parallel: where each discriminator's second gradient bucket starts in its flat buffer (None: one bucket)
        self._bucket_off = []
        for d, f in zip(netsD, self.flatD):
            name = getattr(d, 'bucket_from', None)
            first = next(getattr(d, name).parameters()) if name else None
            self._bucket_off.append(None if first is None else f.offset_of[id(first)])
        self._real_feats = [None] * len(netsD)
        self._real_terms = [None] * len(netsD)
        self._begun = False
        self._g_terms = [None] * len(netsD)
        self._adam_early = [None] * len(netsD)
        self._d_zeroed = [False] * len(netsD)
        self._d_frozen = [False] * len(netsD)
        self._forked_d = False
        self._g_pending = None
        self._d_buckets = {}

    def _allreduce_start(self, flat):
        return self.exchange.start(flat.grad)

    # ---- data-parallel overlap (new functionality) --------------------------------------------------
    # (1) generator: its all-reduce is issued right behind its backward pass and NOT waited for; the Adam + EMA update
    #     is applied at the start of the next step (or by finish()), after everything of that step that does not
    #     depend on the generator has been issued: the frozen text encoder (the caller's prologue, trainer.py:248-252)
    #     and the three discriminators' forward passes on the REAL images (losses.py:139) -- which is why the
    #     data-parallel discriminator loss runs the real and the fake half as two passes (the reference's own shape)
    #     instead of one grouped pass.
    # (2) discriminators: D_NET128 / D_NET256 exchange their gradients in two buckets.  The backward pass reaches the
    #     tail of the network first (D_NET256: img_code_s64 .. heads = 242 of 287 MB): its all-reduce starts there and
    #     runs under the rest of the backward pass (the large-map layers, most of the time) instead of behind it.
    bucket_d = True
    overlap_g = True
    force_overlap_layout = False    # tests: a SINGLE-process step with the data-parallel launch decomposition (real-image
    #                                 forwards ahead as their own passes, split backward passes) -- what a rank of the
    #                                 data-parallel step must reproduce bit for bit in the deterministic mode

    # Single GPU: the discriminators' forward passes on the REAL images need nothing of the generator (losses.py:139): as
    # their own passes (the reference's shape: netD(real), netD(fake) are separate calls) they run at the start of the step,
    # beside the generator's forward pass -- a serial chain that leaves most of the chip idle -- instead of inside the
    # crowded stretch between the generator's forward and backward passes (SBA_REAL_FIRST).
    real_first = os.environ.get('SBA_REAL_FIRST', '0') == '1'

    def _two_pass(self):
        return (self.distributed and self.overlap_g) or self.force_overlap_layout or self.real_first or self.real_bwd_early

    # Experiment (last hours of round 4): with the two-pass layout, ALSO the real half's loss terms and their backward pass
    # at the start of the step (they need neither the generator nor the fake images), so that only the fake half's
    # forward + backward pass is left between the generator's passes.  SBA_REAL_BWD_EARLY=1; see losses.discriminator_real_term
    # for the one deviation (order of the conditional head's BatchNorm running-statistic updates).
    real_bwd_early = os.environ.get('SBA_REAL_BWD_EARLY', '0') == '1'

    def phase_pre(self, imgs, streams=None, sent_emb=None):
        """netD_i(real_i) for every discriminator, ahead of the generator's (pending) update; a no-op outside the
        overlapped data-parallel mode.  streams[i]: the stream discriminator i's update will run on -- autograd
        replays a node's backward on the stream of its forward, and the two passes of one network add into the same
        weight gradients, so both must live on ONE stream."""
        self._real_feats = [None] * len(self.netsD)
        if not self._two_pass():
            return
        ops.SIDE_WGRAD = False
        main = torch.cuda.current_stream()
        for i, d in enumerate(self.netsD):
            st = streams[i] if streams is not None else main
            if st is not main:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                if self.real_bwd_early and sent_emb is not None and imgs[i].is_cuda:
                    if self._d_frozen[i]:
                        for p in d.parameters():
                            p.requires_grad_(True)
                        self._d_frozen[i] = False
                    d.clear_cuts(record=False)
                    if not self._d_zeroed[i]:
                        self.flatD[i].zero_grad()
                        self._d_zeroed[i] = True
                    term = discriminator_real_term(d, d(imgs[i]), sent_emb)
                    term.backward()
                    self._real_terms[i] = term.detach()
                    continue
                d.clear_cuts()
                self._real_feats[i] = d(imgs[i])

    def finish(self):
        """Apply a pending generator update (overlapped data-parallel mode: the last step's all-reduce may still be in
        flight and its Adam + EMA step not applied).  Call before reading the generator: checkpoints, sampling,
        snapshot(); the next step() does it by itself.  While the step is being RECORDED (ReplayedStep) this point becomes
        a host-call node: whether an update is pending is host state, so every replay applies it from the host, here."""
        rec = self.exchange.recorder
        if rec is not None:
            if self.record_g_update and self.distributed and self.overlap_g:
                # Only the WAIT for the exchange in flight is a host call (which handle to wait for is host state); the
                # Adam + EMA launch and the repack of the bf16 weight copies are recorded like any other launch -- as
                # eager launches issued from the callback they cost 0.38 ms at the head of the generator's forward pass
                # (profiles/r04_dp_recorded_update.txt).  A replay therefore REQUIRES a pending update: ReplayedStep
                # runs an eager step instead when there is none (after finish() / restore()).
                rec.host(self._wait_pending_replayed)
                self.optG.step(1.0 / self.world)
                if self.flatG.packs is not None:
                    self.flatG.packs.refresh(ops.compute_dtype())
                rec.recorded_update = True
            else:
                rec.host(self._finish_replayed)
            return
        self._finish_now()

    record_g_update = os.environ.get('SBA_DP_RECORD_UPDATE', '1') == '1'

    def _wait_pending_replayed(self):
        """the host-call node in front of the RECORDED generator update: order the node's stream behind the exchange the
        previous step left in flight"""
        if self._g_pending is None:
            raise RuntimeError('the recorded generator update was replayed with no update pending')
        h, self._g_pending = self._g_pending, None
        self._allreduce_wait(h[0])

    def _finish_now(self):
        if self._g_pending is not None:
            h, self._g_pending = self._g_pending, None
            self._allreduce_wait(h[0])
            self.optG.step(1.0 / self.world)
            return True
        return False

    def _finish_replayed(self):
        """the host-call node of a recorded step: apply the pending update AND repack the generator's bf16 weight copies --
        the recorded generator forward holds no repack of its own (at capture time the copies were fresh)"""
        if self._finish_now() and self.flatG.packs is not None:
            self.flatG.packs.refresh(ops.compute_dtype())

    def _start_g_exchange(self):
        """start the generator's all-reduce and leave its update pending (host state: under recording a host-call node)"""
        rec = self.exchange.recorder
        if rec is not None:
            rec.host(self._start_g_exchange_now)
            return
        self._start_g_exchange_now()

    def _start_g_exchange_now(self):
        self._g_pending = (self._allreduce_start(self.flatG),)

    def drop_pending(self):
        """Wait for a pending generator exchange and DISCARD it without applying the update: the parameters are about to
        be overwritten (restore(), a checkpoint load), and the gradient in flight belongs to the weights being replaced --
        applying it later would run Adam on the new weights with a stale gradient and bump its step counter."""
        if self._g_pending is not None:
            h, self._g_pending = self._g_pending, None
            self._allreduce_wait(h[0])

    def _allreduce_wait(self, h):
        self.exchange.wait(h)

    # The step is written as three phases so that a host can replay each from its own hipGraph
    # (GraphedStep below): A = generator forward, D(i) = update of discriminator i, B = generator
    # loss, backward, Adam + EMA.  step() composes them with the eager stream forks.
    def phase_a(self, sent_emb, words_embs, mask, noise, eps=None):
        ops.SIDE_WGRAD = self.overlap_wgrad
        if self._begun:                       # (step() did both ahead of the early real-half backward pass)
            self._begun = False
        else:
            ops.det_reset()                   # deterministic mode: the step's partial sums start at the ring's base
            ops.ARENA.begin(self.device)      # one memset for all per-layer accumulators of the step
        self.netG.ca_net.eps = eps
        fake_imgs, _, mu, logvar = self.netG(noise, sent_emb, words_embs, mask)
        self._ctx = (fake_imgs, mu, logvar)
        self._out = {}

    def phase_e(self, sent_emb, words_embs, cap_lens, class_ids):
        """The DAMSM ranking terms of the generator loss and their gradient w.r.t. the last fake image (image
        encoder forward, words / sentence loss, backward through the frozen encoder: losses.py:187-204).  They
        depend on the generator's output only, so this phase runs BESIDE the discriminator updates instead of
        behind them (the reference evaluates them inside generator_loss, after the updates: ~4.5 ms of the
        critical path at B=20)."""
        fake_imgs = self._ctx[0]
        self._damsm = damsm_image_terms(self.image_encoder, fake_imgs[-1], words_embs, sent_emb, self.match_labels,
                                        cap_lens, class_ids)

    def phase_d_bwd_tail(self, i, imgs, sent_emb):
        """loss of discriminator i and the backward pass down to its bucket boundary (the whole backward pass when the
        network exchanges one bucket); returns True when phase_d_bwd_rest has work left"""
        fake_imgs = self._ctx[0]
        netD = self.netsD[i]
        if self._d_frozen[i]:           # left frozen by a phase_g_term whose phase_b_bwd never ran
            for p in netD.parameters():
                p.requires_grad_(True)
            self._d_frozen[i] = False
        ops.SIDE_WGRAD = self.overlap_wgrad and self.overlap_wgrad_d
        if self._d_zeroed[i]:           # cleared at the start of the step, beside the generator's forward pass (step())
            self._d_zeroed[i] = False
        else:
            self.flatD[i].zero_grad()
        rf = self._real_feats[i]
        self._real_feats[i] = None
        split = ((self.distributed and self.bucket_d) or self.force_overlap_layout or
                 (self.bucket_adam and self._forked_d)) and \
            self._bucket_off[i] is not None and not ops.SIDE_WGRAD
        rt = self._real_terms[i]
        self._real_terms[i] = None
        if rf is None:
            netD.clear_cuts(record=split)
        if rt is not None:          # the real half, backward pass included, ran at the start of the step (phase_pre)
            errD = discriminator_fake_term(netD, fake_imgs[i], sent_emb)
            self._out['errD%d' % i] = rt + errD.detach()
        else:
            errD = discriminator_loss(netD, imgs[i], fake_imgs[i], sent_emb, self.real_labels, self.fake_labels,
                                      real_features=rf)
            self._out['errD%d' % i] = errD.detach()
        cuts = list(netD._cuts)
        netD.clear_cuts(record=False)
        if split and cuts:
            gcuts = torch.autograd.grad(errD, cuts)         # heads + tail: their parameter gradients are complete now
            self._d_buckets[i] = (cuts, gcuts)
            return True
        errD.backward()
        return False

    def phase_d_bwd_rest(self, i):
        cuts, gcuts = self._d_buckets.pop(i)
        torch.autograd.backward(cuts, gcuts)

    def _d_exchange(self, i, split_done):
        """all-reduce handles of discriminator i's gradient: the whole flat buffer, or (bucketed) its head part --
        the tail part was started by the caller between the two halves of the backward pass"""
        off = self._bucket_off[i]
        g = self.flatD[i].grad
        return self.exchange.start(g[:off] if split_done else g)

    def phase_d_bwd(self, i, imgs, sent_emb, forked):
        """loss + backward of discriminator i on the CURRENT stream (data-parallel: with the gradient exchange started
        bucket by bucket); returns (the stream on which the update must continue -- the weight-gradient companion
        when `forked`, see ops.wgrad_tail_stream --, the exchange handles to wait for)"""
        handles = []
        self._adam_early[i] = None
        if self.phase_d_bwd_tail(i, imgs, sent_emb):
            if self.distributed:
                handles.append(self.exchange.start(self.flatD[i].grad[self._bucket_off[i]:]))
            elif forked and self.bucket_adam:
                # The backward pass has reached the bucket boundary: the gradients of the tail + heads (D_NET256: 60.6 M of
                # 71.9 M parameters) are complete.  Their Adam update -- 280 of the 334 us of an HBM-bound launch that
                # otherwise sits on this discriminator's critical chain -- starts now on a side stream, beside the
                # large-map trunk's backward pass; the rest of the chain continues on that stream (a one-way edge: ROCm
                # 7.2's stream capture does not survive a fork that is joined back into a forked stream).
                side = self._adam_stream(i)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.optD[i].prepare()
                    self.optD[i].step_range(self._bucket_off[i], self.flatD[i].n, 1.0 / self.world)
                self._adam_early[i] = side
            self.phase_d_bwd_rest(i)
            split = True
        else:
            split = False
        if forked:
            tail = ops.wgrad_tail_stream()
            if self._adam_early[i] is not None:
                self._adam_early[i].wait_stream(tail)
                tail = self._adam_early[i]
        else:
            ops.join_wgrads()
            tail = torch.cuda.current_stream()
        ops.SIDE_WGRAD = self.overlap_wgrad
        if self.distributed:
            with torch.cuda.stream(tail):
                handles.append(self._d_exchange(i, split))
        return tail, handles

    def phase_d_bwd_join(self):
        """end of a discriminator's backward pass on the current stream (per-phase graph capture)"""
        ops.join_wgrads()
        ops.SIDE_WGRAD = self.overlap_wgrad

    def phase_d_opt(self, i):
        """Adam step of discriminator i (its averaged gradient must be in place)."""
        if self._adam_early[i] is not None:      # the tail bucket was updated beside the backward pass (phase_d_bwd)
            self.optD[i].step_range(0, self._bucket_off[i], 1.0 / self.world)
            self.optD[i].done()
            self._adam_early[i] = None
            return
        self.optD[i].step(1.0 / self.world)

    def _adam_stream(self, i):
        if getattr(self, '_adam_streams', None) is None:
            self._adam_streams = [torch.cuda.Stream(device=self.device) for _ in self.netsD]
        return self._adam_streams[i]

    def phase_d(self, i, imgs, sent_emb, forked):
        self._forked_d = bool(forked) and not self.distributed
        tail, handles = self.phase_d_bwd(i, imgs, sent_emb, forked)
        self._forked_d = False
        with torch.cuda.stream(tail):
            for h in handles:
                self._allreduce_wait(h)
            self.phase_d_opt(i)
        return tail

    def phase_g_term(self, i, sent_emb):
        """Discriminator i's term of the generator loss and its gradient w.r.t. fake image i, on the CURRENT stream --
        the stream discriminator i has just been updated on.  The term depends on that update and on the image only, so
        it runs behind discriminator i's optimizer step while the larger discriminators are still updating, instead of
        behind the slowest of them (losses.py:168-186 evaluates the terms inside generator_loss, after all updates)."""
        netD = self.netsD[i]
        for p in netD.parameters():
            p.requires_grad_(False)         # (phase_b_bwd restores them after the generator's backward pass; if that never
        self._d_frozen[i] = True            #  runs -- an exception, a caller driving the phases by hand -- the next
        #                                      phase_d_bwd_tail of this discriminator does)
        value, grad = generator_d_term(netD, self._ctx[0][i], sent_emb)
        self._g_terms[i] = (value, grad)

    def phase_b_bwd(self, sent_emb, words_embs, cap_lens, class_ids):
        fake_imgs, mu, logvar = self._ctx
        mark = self._mark
        ops.SIDE_WGRAD = self.overlap_wgrad
        ops.HOME_STREAM = torch.cuda.current_stream().cuda_stream
        for p in self._d_params:
            p.requires_grad_(False)
        self.flatG.zero_grad()
        damsm = getattr(self, '_damsm', None) if self.early_damsm else None
        d_terms = self._g_terms if all(t is not None for t in self._g_terms) else None
        if d_terms is not None and damsm is not None:
            # every term but the KL one was evaluated ahead of time, gradients and all: the backward pass starts from the
            # image gradients and the KL term at once; errG_total -- a VALUE now, summed in the reference's order from the
            # same numbers -- is put together behind it (seven scalar launches off the head of the critical chain)
            kl = KL_loss(mu, logvar)
            grads = [t[1] for t in d_terms]
            if not torch.cuda.is_current_stream_capturing():
                for g in grads:             # produced on the discriminators' streams, consumed on this one
                    g.record_stream(torch.cuda.current_stream())
            grads[-1] = grads[-1] + damsm[2]
            mark('g_loss_forward')
            backward_with_image_grads(kl, fake_imgs, grads)
            errG_total, logs = generator_loss(self.netsD, self.image_encoder, fake_imgs, self.real_labels, words_embs,
                                              sent_emb, self.match_labels, cap_lens, class_ids, damsm=damsm,
                                              d_terms=[t[0] for t in d_terms])
            errG_total = errG_total + kl.detach()
            self._damsm = None
            self._g_terms = [None] * len(self.netsD)
            return self._phase_b_bwd_end(errG_total, kl, logs, fake_imgs)
        errG_total, logs = generator_loss(self.netsD, self.image_encoder, fake_imgs, self.real_labels, words_embs,
                                          sent_emb, self.match_labels, cap_lens, class_ids,
                                          streams=self._d_streams() if self.concurrent_d else None, damsm=damsm,
                                          d_terms=None if d_terms is None else [t[0] for t in d_terms])
        kl = KL_loss(mu, logvar)
        errG_total = errG_total + kl
        mark('g_loss_forward')
        if d_terms is not None:
            grads = [t[1] for t in d_terms]
            if not torch.cuda.is_current_stream_capturing():
                for g in grads:             # produced on the discriminators' streams, consumed on this one
                    g.record_stream(torch.cuda.current_stream())
            if damsm is not None:
                grads[-1] = grads[-1] + damsm[2]
            # (damsm None: errG_total carries the graph of the ranking terms, evaluated inside generator_loss)
            backward_with_image_grads(errG_total, fake_imgs, grads)
            self._damsm = None
        elif damsm is not None:
            backward_with_image_grad(errG_total, fake_imgs[-1], damsm[2])
            self._damsm = None
        else:
            errG_total.backward()
        self._g_terms = [None] * len(self.netsD)
        return self._phase_b_bwd_end(errG_total, kl, logs, fake_imgs)

    def _phase_b_bwd_end(self, errG_total, kl, logs, fake_imgs):
        ops.join_wgrads()
        self._mark('g_backward')
        for p in self._d_params:
            p.requires_grad_(True)
        self._d_frozen = [False] * len(self.netsD)
        out = self._out
        out['errG_total'] = errG_total.detach()
        out['kl_loss'] = kl.detach()
        out.update(logs)
        self.fake_imgs = [f.detach() for f in fake_imgs]
        self._ctx = None

    def phase_b_opt(self):
        self.optG.step(1.0 / self.world)
        self._mark('g_adam')
        ops.ARENA.end()
        ops.SIDE_WGRAD = False
        return self._out

    def phase_b(self, sent_emb, words_embs, cap_lens, class_ids):
        self.phase_b_bwd(sent_emb, words_embs, cap_lens, class_ids)
        if self.distributed and self.overlap_g:
            # the exchange runs behind this step; the update is applied by the next step (after its text encoder and
            # real-image forwards have been issued) or by finish()
            self._start_g_exchange()
            self._mark('g_adam')
            ops.ARENA.end()
            ops.SIDE_WGRAD = False
            return self._out
        self._allreduce_wait(self._allreduce_start(self.flatG))
        return self.phase_b_opt()

    def step(self, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, eps=None):
        """Returns a dict of DEVICE scalars (errD0.., errG_total, kl_loss, g_loss*, w_loss, s_loss)."""
        mark = self._mark
        mark('start')
        main = torch.cuda.current_stream()
        nD = len(self.netsD)
        streams = self._d_streams()[:nD] if self.concurrent_d else [main] * nD
        if self.early_zero and self.concurrent_d:
            # the discriminators' gradient buffers (D_NET256: 287 MB, a 57 us fill at the head of the longest chain of the
            # step) are cleared on their update streams NOW, beside the generator's forward pass
            for i in range(nD):
                if streams[i] is not main:
                    streams[i].wait_stream(main)
                    with torch.cuda.stream(streams[i]):
                        self.flatD[i].zero_grad()
                    self._d_zeroed[i] = True
        if self.real_bwd_early and self._two_pass():
            ops.det_reset()
            ops.ARENA.begin(self.device)
            self._begun = True
        self.phase_pre(imgs, streams, sent_emb)         # (data-parallel: beside the generator's pending gradient exchange)
        self.finish()
        # discriminator i reads fake image i only: its update forks from the point where that image has been issued
        # (64 px: after the first stage, 128 px: after the second), not from the end of the generator's forward pass
        img_ready = [None] * nD
        if self.early_d and self.concurrent_d:
            def on_image(i):
                if i < nD - 1:
                    img_ready[i] = torch.cuda.Event()
                    img_ready[i].record()
            self.netG.on_image = on_image
            if self.fork_heads:
                self.netG.image_stream = lambda i: streams[i] if (i < nD - 1 and streams[i] is not main) else None
        self.phase_a(sent_emb, words_embs, mask, noise, eps)
        self.netG.on_image = None
        self.netG.image_stream = None
        mark('g_forward')
        # The three discriminator updates are independent of each other (different networks, the
        # same detached fakes): each runs on its own HIP stream so that the small launches of the
        # 4x4 / 8x8 layers of one network fill CUs the others leave idle.
        tails = []
        for i in range(nD):
            st = streams[i]
            if st is not main:
                if img_ready[i] is not None:
                    st.wait_event(img_ready[i])
                else:
                    st.wait_stream(main)
            with torch.cuda.stream(st):
                tails.append(self.phase_d(i, imgs, sent_emb, forked=st is not main))
            if self.early_g_terms and st is not main:
                with torch.cuda.stream(tails[-1]):
                    self.phase_g_term(i, sent_emb)
        if self.early_damsm:
            # on the ORIGIN stream (the discriminator updates are the forks): the image encoder forks streams of
            # its own, and a fork inside a forked branch crashes hipStreamEndCapture on ROCm 7.2
            es = getattr(self, '_warm_e_stream', None)
            if es is None:
                self.phase_e(sent_emb, words_embs, cap_lens, class_ids)
            else:       # (eager warm-up before a per-phase capture that gives this phase its own stream: per-stream
                #          workspaces and the encoder's buffers must exist before the capture)
                es.wait_stream(main)
                with torch.cuda.stream(es):
                    self.phase_e(sent_emb, words_embs, cap_lens, class_ids)
                main.wait_stream(es)
        for st in streams + tails:
            if st is not main:
                main.wait_stream(st)
        mark('d_steps')
        return self.phase_b(sent_emb, words_embs, cap_lens, class_ids)

    phase_events = None      # set to [] to record (name, cuda event) pairs per step (bench.py --phases)

    def _mark(self, name):
        if self.phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.phase_events.append((name, e))

    concurrent_d = True
    early_d = os.environ.get('SBA_EARLY_D', '1') == '1'      # fork the 64 / 128 px discriminator updates inside the G forward
    fork_heads = os.environ.get('SBA_FORK_HEADS', '0') == '1'      # ... and evaluate the 64 / 128 px image heads on those
    #                                                              discriminators' streams (nets._GBase.image_stream).  Bit-equal
    #                                                              and audit-clean, but no measurable gain: 9.42 / 9.53 against
    #                                                              9.52 / 9.39 ms (profiles/r04_ab_fork_heads.txt).  Off.
    bucket_adam = os.environ.get('SBA_BUCKET_ADAM', '0') == '1'      # D_NET128 / D_NET256: Adam of the tail + heads beside the
    #                                                                  trunk's backward pass (phase_d_bwd).  Correct (the GPU
    #                                                                  suite passes with it on) but measured SLOWER: 11.50
    #                                                                  against 11.29 ms -- the HBM-bound update competes with
    #                                                                  the trunk's weight gradients, and the split backward
    #                                                                  pass adds a graph boundary.  Off.
    early_zero = os.environ.get('SBA_EARLY_ZERO', '0') == '1'      # clear the discriminators' gradients at the start of the step,
    #                                                              beside the generator's forward pass.  Measured SLOWER (11.7
    #                                                              against 11.0 ms): the earlier forks change the replayer's
    #                                                              stream assignment.  Off.
    early_g_terms = os.environ.get('SBA_EARLY_G_TERMS', '1') == '1'      # each discriminator's generator-loss term right
    #                                                                  behind its own update (phase_g_term)
    early_damsm = True           # DAMSM terms + their image gradient beside the discriminator updates (phase_e)
    overlap_wgrad = True
    overlap_wgrad_d = os.environ.get('SBA_OVERLAP_WGRAD_D', '0') == '1'      # companion streams inside the (already concurrent) discriminator updates cost
                                 # 1.5 ms under hipGraph replay: ROCm 7.2 runs graph branches nearly serially

    def _e_stream(self):
        if getattr(self, '_estream', None) is None:
            self._estream = torch.cuda.Stream(device=self.device)
        return self._estream

    def _d_streams(self):
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(len(self.netsD) + 1)]
            # The two small discriminators' updates (and their generator-loss terms) share ONE stream: D_NET64 then D_NET128,
            # 3.1 + 4.3 ms of slack against the longest path (sba_replay_prioritize's report, profiles/r04_ab_stream_priorities.txt).
            # Three chains beside the image encoder's instead of four: the encoder's ~160 short launches, which gate the
            # generator's backward pass, get a third of the dispatch slots instead of a quarter -- 10.46 -> 10.19..10.39 ms
            # (profiles/r04_ab_d_merge.txt).  SBA_D_MERGE=0: one stream per discriminator.
            if os.environ.get('SBA_D_MERGE', '1') == '1' and len(self.netsD) >= 3:
                self._streams[0] = self._streams[1]
        return self._streams

    def grad_norm(self, flat):
        return flat.grad.double().norm()

    # ---- training state as a whole (checkpoint / resume, trainer.py:159-170 saves netG, the EMA copy and netsD;
    # the reference restarts Adam from zero moments on resume, this keeps them as well)
    def _trained(self):
        return [(self.netG, self.flatG, self.optG)] + list(zip(self.netsD, self.flatD, self.optD))

    def snapshot(self):
        """Device copies of everything a step mutates: parameters, Adam moments and step counters, the EMA
        shadow, BatchNorm running statistics / batch counters."""
        self.finish()
        snap = []
        for net, flat, opt in self._trained():
            snap.append({'data': flat.data.clone(), 'm': flat.m.clone(), 'v': flat.v.clone(),
                         'avg': None if flat.avg is None else flat.avg.clone(), 'state': opt.state.clone(),
                         'buffers': {n: b.detach().clone() for n, b in net.named_buffers()}})
        return snap

    def restore(self, snap):
        """Write a snapshot() back in place (pointers are unchanged, so captured graphs stay valid; call
        GraphedStep.resync() before the next replay: the graphs read packed bf16 copies of the weights).  A generator
        update still pending from the last data-parallel step is dropped, not applied (drop_pending); a launch-mode wrapper
        that defers the update itself (GraphedStep) drops its own in resync()."""
        self.drop_pending()
        for (net, flat, opt), s in zip(self._trained(), snap):
            flat.data.copy_(s['data'])
            flat.m.copy_(s['m'])
            flat.v.copy_(s['v'])
            if flat.avg is not None:
                flat.avg.copy_(s['avg'])
            opt.state.copy_(s['state'])
            bufs = dict(net.named_buffers())
            for n, b in s['buffers'].items():
                bufs[n].copy_(b)
            flat.epoch[0] += 1
        ops.weights_changed()


# "thread_local": only the capturing thread is policed -- the process group's watchdog thread polls events
# concurrently and would otherwise invalidate a capture in multi-rank runs
_CAPTURE_MODE = 'thread_local'


class _NativeGraph(object):
    """One captured phase re-issued by the native replayer (csrc/replay.hip) instead of hipGraphLaunch: its first chain
    runs on the CALLER's stream (phase graphs are single chains apart from the weight-gradient companion), so phases
    replayed on different streams really overlap -- hipGraph launches on different streams do not, on ROCm 7.2."""

    def __init__(self, graph, max_streams=2, flags=2):
        self.graph = graph                       # keeps the hipGraph (and the kernel arguments in its nodes) alive
        self.handle = ctypes.c_void_p()
        from ._lib import lib
        rc = lib.sba_replay_create(ctypes.c_void_p(int(graph.raw_cuda_graph())), int(max_streams), int(flags),
                                   ctypes.byref(self.handle))      # flags = 2: the first chain on the caller's stream
        if rc != 0:
            raise RuntimeError('sba_replay_create failed (%d)' % rc)

    def replay(self):
        call('sba_replay_launch', self.handle, torch.cuda.current_stream().cuda_stream)

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            try:
                call('sba_replay_destroy', h)
            except Exception:
                pass
            self.handle = None


class GraphedStep(object):
    """GANStep replayed from captured hipGraphs: one graph for the generator forward, one PER
    DISCRIMINATOR update -- replayed concurrently, each on its own stream -- and one for the generator
    loss / backward / Adam.  (A single captured graph of the whole step runs its discriminator branches
    almost back to back on ROCm 7.2: measured 8.5 ms for the three updates against ~4.5 ms here.)
    Inputs are static tensors; `prologue` (e.g. drawing the noise in place) is captured with phase A.
    Call after a few eager steps on the same GANStep (per-stream workspaces and packed buffers exist).
    The graphs repack the bf16 weight copies exactly where the capture did (after each network's Adam step), so
    between replays parameters must only change through the replayed optimizer steps; after loading a checkpoint
    or editing parameters run one eager `gan.step` (or rebuild the GraphedStep) before replaying again."""

    def __init__(self, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue=None,
                 single=False, native=False, recorded_prologue=None):
        """native: re-issue every phase with the native replayer (_NativeGraph) instead of hipGraphLaunch.  Random draws
        cannot be recorded then (a captured Philox kernel reads an offset only torch's own graph replay advances):
        `prologue` runs EAGERLY at the start of replay(), the conditioning noise eps is drawn eagerly into a static
        tensor, and deterministic launches that belong in front of the step go to `recorded_prologue`."""
        self.native = bool(native)
        self._eager_prologue = None
        self.eps = None
        if self.native:
            self._eager_prologue, prologue = prologue, recorded_prologue
            self.eps = torch.empty((noise.shape[-2], cfg.GAN.CONDITION_DIM), dtype=torch.float32, device=gan.device)
            self.eps.normal_(0, 1)
        elif recorded_prologue is not None:
            user = prologue
            prologue = (lambda: (user() if user is not None else None, recorded_prologue()))
        self._capture(gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue, single)
        if self.native:
            self._go_native()
        # the weight repacks issued during capture were recorded, not executed, yet the host-side change counters
        # now call the packed copies fresh: invalidate them so that an eager step after the capture repacks
        ops.weights_changed()

    def _capture(self, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue, single):
        # hipGraph launches on different streams do not overlap each other on ROCm 7.2, so a separate graph for
        # the DAMSM phase would only lengthen the chain of phase graphs (measured 16.9 against 15.4 ms): the
        # per-phase graphs keep the ranking terms inside the generator-loss phase; the whole-step captures
        # (single=True, ReplayedStep) record the early branch
        early = gan.early_damsm
        # (native: phases replayed on different streams DO overlap -- the ranking terms get their own phase beside the
        # discriminator updates, as in the whole-step captures, also in the data-parallel path)
        gan.early_damsm = early and ((single and not gan.distributed) or self.native)
        try:
            self._capture_phases(gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue, single)
        finally:
            gan.early_damsm = early

    def _capture_phases(self, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue, single):
        self.gan = gan
        self.single = single and not gan.distributed
        dev = gan.device
        nD = len(gan.netsD)
        self.cap = torch.cuda.Stream(device=dev)
        self.dstreams = gan._d_streams()[:nD]

        def eager_on(stream):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                if prologue is not None:
                    prologue()
                gan.step(imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise)
            torch.cuda.current_stream().wait_stream(stream)
        if self.native and gan.early_damsm and not single:
            gan._warm_e_stream = gan._e_stream()
        try:
            eager_on(self.cap)                   # warm the capture stream (its split-K workspace)
        finally:
            gan._warm_e_stream = None
        torch.cuda.synchronize()
        gan.finish()                             # (data-parallel: the warm-up step's generator update may be pending)
        mk = self._mk
        self.gA, self.gB = mk(), mk()
        self.gD = [mk() for _ in range(nD)]
        self.gPre, self._pending = None, None
        if gan.distributed and gan.overlap_g:
            # the discriminators' forward passes on the real images: replayed BEFORE the generator's pending update
            # (and the prologue -- noise draw, frozen text encoder, trainer.py:248-252 -- which does not depend on it either)
            self.gPre = mk()
            with torch.cuda.graph(self.gPre, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                if prologue is not None:
                    prologue()
                gan.phase_pre(imgs)
            prologue = None
        if single and not gan.distributed:       # the whole step as ONE graph, discriminator updates as forked branches
            with torch.cuda.graph(self.gA, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                if prologue is not None:
                    prologue()
                self.out = gan.step(imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, eps=self.eps)
            torch.cuda.synchronize()
            return
        with torch.cuda.graph(self.gA, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
            if prologue is not None:
                prologue()
            gan.phase_a(sent_emb, words_embs, mask, noise, self.eps)
        self.gE = None
        if gan.early_damsm and not self.single:     # (kept for experiments: see _capture)
            self.gE = mk()
            self.estream = gan._e_stream()
            with torch.cuda.graph(self.gE, stream=self.estream, capture_error_mode=_CAPTURE_MODE):
                gan.phase_e(sent_emb, words_embs, cap_lens, class_ids)
        if gan.distributed:
            # the gradient exchange (RCCL) stays OUTSIDE the graphs: per network one graph for loss +
            # backward and one for the Adam step, the all-reduce issued eagerly between them
            # (a bucketed discriminator: one graph down to the bucket boundary, one for the rest of its backward pass,
            # the tail bucket's all-reduce issued between them)
            self.gDo = [mk() for _ in range(nD)]
            self.gD2 = [None] * nD
            self.gBo = mk()
            for i in range(nD):
                with torch.cuda.graph(self.gD[i], stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                    split = gan.phase_d_bwd_tail(i, imgs, sent_emb)
                    if not split:
                        gan.phase_d_bwd_join()
                if split:
                    self.gD2[i] = mk()
                    with torch.cuda.graph(self.gD2[i], stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                        gan.phase_d_bwd_rest(i)
                        gan.phase_d_bwd_join()
                with torch.cuda.graph(self.gDo[i], stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                    gan.phase_d_opt(i)
                    if gan.early_g_terms:       # its generator-loss term right behind the update (GANStep.phase_g_term)
                        gan.phase_g_term(i, sent_emb)
            with torch.cuda.graph(self.gB, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                gan.phase_b_bwd(sent_emb, words_embs, cap_lens, class_ids)
            with torch.cuda.graph(self.gBo, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                self.out = gan.phase_b_opt()
            torch.cuda.synchronize()
            return
        for i in range(nD):
            with torch.cuda.graph(self.gD[i], stream=self.dstreams[i], capture_error_mode=_CAPTURE_MODE):
                gan.phase_d(i, imgs, sent_emb, forked=False)
                if gan.early_g_terms:
                    gan.phase_g_term(i, sent_emb)
        with torch.cuda.graph(self.gB, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
            self.out = gan.phase_b(sent_emb, words_embs, cap_lens, class_ids)
        torch.cuda.synchronize()

    def _mk(self):
        return torch.cuda.CUDAGraph(keep_graph=True) if self.native else torch.cuda.CUDAGraph()

    def _go_native(self):
        """wrap every captured phase in a native replayer (same .replay() surface)"""
        wrap = lambda g: None if g is None else _NativeGraph(g)
        for name in ('gPre', 'gA', 'gB', 'gE', 'gBo'):
            if getattr(self, name, None) is not None:
                setattr(self, name, wrap(getattr(self, name)))
        for name in ('gD', 'gD2', 'gDo'):
            if getattr(self, name, None) is not None:
                setattr(self, name, [wrap(g) for g in getattr(self, name)])

    def resync(self):
        """Bring the packed weight copies the graphs read in line with the f32 masters, eagerly.  Needed after
        parameters changed behind the graphs' back (GANStep.restore, load_state_dict, a checkpoint): the graphs
        repack a network only where the capture did, right after its own Adam step.  Contract: a generator update this
        wrapper still holds back (data-parallel, overlapped exchange) is waited for and DROPPED here -- it was computed
        for the weights that have just been replaced; call finish() BEFORE changing parameters to apply it instead."""
        pend = getattr(self, '_pending', None)
        if pend is not None:
            self._pending = None
            self.gan._allreduce_wait(pend[0])
        self.gan.drop_pending()
        dt = ops.compute_dtype()
        for flat in [self.gan.flatG] + list(self.gan.flatD):
            if flat.packs is not None:
                flat.packs.refresh(dt)
        ops.weights_changed()

    def finish(self):
        """data-parallel, overlapped generator exchange: apply the pending generator update (see GANStep.finish)"""
        if self._pending is not None:
            h, self._pending = self._pending, None
            self.gan._allreduce_wait(h[0])
            if self.native and not getattr(self, '_in_native', False):
                cur = torch.cuda.current_stream()
                self.cap.wait_stream(cur)
                with torch.cuda.stream(self.cap):
                    self.gBo.replay()
                cur.wait_stream(self.cap)
            else:
                self.gBo.replay()

    def replay(self):
        if self.native and not getattr(self, '_in_native', False):
            # the phases' first chains run on the stream they are launched from: keep that off the NULL stream
            cur = torch.cuda.current_stream()
            self.cap.wait_stream(cur)
            self._in_native = True
            try:
                with torch.cuda.stream(self.cap):
                    out = self.replay()
            finally:
                self._in_native = False
            cur.wait_stream(self.cap)
            return out
        main = torch.cuda.current_stream()
        # the graphs repack the bf16 weight copies at the points where the capture did, without consulting the
        # host-side change counters; invalidate those so that an EAGER call after a replay repacks as well
        ops.weights_changed()
        if self.native:
            self.eps.normal_(0, 1)
            if self._eager_prologue is not None:
                self._eager_prologue()
        if self.gPre is not None:
            self.gPre.replay()      # netD_i(real_i): beside the generator's all-reduce of the previous replay
            self.finish()
        self.gA.replay()
        if self.single:
            return self.out
        gan = self.gan
        if self.gE is not None:
            self.estream.wait_stream(main)
            with torch.cuda.stream(self.estream):
                self.gE.replay()
        if gan.distributed:
            # every discriminator update on its own stream (as in the single-GPU path): loss + backward graph, the
            # gradient exchange (eager RCCL call; the exchange stream orders the collectives identically on every
            # rank: largest network first, so D256's 287 MB all-reduce runs under the other updates), Adam graph
            order = sorted(range(len(self.gD)), key=lambda i: -gan.flatD[i].n)
            for i in order:
                st = self.dstreams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self.gD[i].replay()
                    handles = []
                    if self.gD2[i] is not None:
                        handles.append(gan.exchange.start(gan.flatD[i].grad[gan._bucket_off[i]:]))
                        self.gD2[i].replay()
                    handles.append(gan._d_exchange(i, self.gD2[i] is not None))
                    for h in handles:
                        gan._allreduce_wait(h)
                    self.gDo[i].replay()
            for st in self.dstreams:
                main.wait_stream(st)
            if self.gE is not None:
                main.wait_stream(self.estream)
            self.gB.replay()
            if self.gPre is not None:
                self._pending = (gan._allreduce_start(gan.flatG),)  # applied by the next replay / finish()
                return self.out
            gan._allreduce_wait(gan._allreduce_start(gan.flatG))
            self.gBo.replay()
            return self.out
        for i, st in enumerate(self.dstreams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self.gD[i].replay()
        for st in self.dstreams:
            main.wait_stream(st)
        if self.gE is not None:
            main.wait_stream(self.estream)
        self.gB.replay()
        return self.out



# ReplayedStep.prioritize: 'mode:streams:n_high:slack' unless SBA_REPLAY_PRIO says otherwise ('0' = no priorities)
REPLAY_PRIO_DEFAULT = '0'


class ReplayedStep(object):
    """GANStep re-issued by the native multi-stream launch replayer (csrc/replay.hip): the whole step is stream-
    captured ONCE into a hipGraph -- with every fork GANStep.step makes (the three discriminator updates, the
    branches of the generator loss and of the Inception blocks, weight gradients beside data gradients) -- and the
    graph is never launched: libsbagan_hip.so walks its nodes and edges and re-issues them as ordinary launches
    over up to `max_streams` HIP streams.  hipGraphLaunch (GraphedStep) runs those branches nearly back to back
    on ROCm 7.2; launched this way they overlap, at ~3 us of host time per launch instead of ~20 us from Python.

    Inputs are static tensors.  Random draws stay OUTSIDE the recorded launches (a captured Philox kernel reads an
    offset that only torch's own graph replay advances): `noise` and `eps` are drawn eagerly by replay().
    Same contract as GraphedStep for parameters changed behind its back (resync())."""

    @staticmethod
    def warm_up(gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, recorded_prologue=None):
        """One eager step on a fresh capture stream (per-stream workspaces, packed weights).  Data-parallel: this is the
        COLLECTIVE-BEARING part of the construction -- callers that fall back when a capture fails run it outside
        their try block (a rank that threw in here would leave its peers blocked in the step's all-reduces) and pass
        the result as `warm`."""
        dev = gan.device
        eps = torch.empty((noise.shape[-2], cfg.GAN.CONDITION_DIM), dtype=torch.float32, device=dev)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            noise.normal_(0, 1)
            eps.normal_(0, 1)
            if recorded_prologue is not None:
                recorded_prologue()
            gan.step(imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, eps=eps)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        return {'cap': cap, 'eps': eps}

    def __init__(self, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue=None,
                 recorded_prologue=None, max_streams=None, verbose=False, warm=None):
        """prologue: called eagerly before every replay (after the random draws); recorded_prologue: deterministic
        launches recorded in front of the step (e.g. the frozen text encoder's forward, trainer.py:248-252).
        max_streams: replay streams; None = SBA_REPLAY_STREAMS from the environment, else 4.  warm: the result of
        warm_up() when the caller has run it already."""
        # Data-parallel (gan.distributed): the SAME single recording -- the gradient exchanges and the deferred generator
        # update become host-call nodes (ExchangeRecorder): every collective sits between the recorded launches where the
        # eager data-parallel step has it (buckets under the backward passes, the generator's exchange behind the next
        # step's real-image forwards), and the phases overlap as on one GPU.  The warm-up step below issues collectives:
        # every rank must construct the ReplayedStep at the same point.
        self.gan, self.noise, self.prologue = gan, noise, prologue
        dev = gan.device
        args = (imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise)
        self._args, self._recorded_prologue = args, recorded_prologue
        if warm is None:
            warm = self.warm_up(gan, *args, recorded_prologue=recorded_prologue)
        self.eps, self.cap = warm['eps'], warm['cap']
        self.draw = True
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        self._rec = ExchangeRecorder(gan.exchange) if gan.distributed else None
        gan.exchange.recorder = self._rec
        try:
            with torch.cuda.graph(self.graph, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                if recorded_prologue is not None:
                    recorded_prologue()
                self.out = gan.step(*args, eps=self.eps)
        finally:
            gan.exchange.recorder = None
        torch.cuda.synchronize()
        raw = self.graph.raw_cuda_graph()
        self.handle = ctypes.c_void_p()
        from ._lib import lib
        # 4 = the number of hardware queues a process gets by default (GPU_MAX_HW_QUEUES): one replay stream per queue.
        # More streams share queues in an order the runtime picks (two independent chains on one queue run back to
        # back): 8 streams 11.31 ms, 6: 11.18, 5: 11.15, **4: 11.05**, 3: 11.7, 2: 12.5; raising GPU_MAX_HW_QUEUES instead
        # is far worse (5: 13.2 ms, 6: 19.8, 8: 20.7) -- profiles/r03_ab_replay_streams.txt
        if max_streams is None:
            max_streams = int(os.environ.get('SBA_REPLAY_STREAMS', '4'))
        rc = lib.sba_replay_create(ctypes.c_void_p(int(raw)), int(max_streams), 1 if verbose else 0,
                                   ctypes.byref(self.handle))
        if rc != 0:
            raise RuntimeError('sba_replay_create failed (%d): the captured step holds a node the replayer '
                               'cannot re-issue' % rc)
        self._refresh_info()
        self._cb = self._cb_error = None
        if self._rec is not None:
            assert self.info['host_calls'] == len(self._rec.calls), (self.info, len(self._rec.calls))
            rec, dev_ = self._rec, dev

            def on_marker(tag, stream_ptr, _user):
                try:
                    rec.dispatch(tag, stream_ptr, dev_)
                except BaseException as e:      # (an exception cannot cross the C frame: re-raised by replay())
                    if self._cb_error is None:
                        self._cb_error = e
            self._cb = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)(on_marker)
            call('sba_replay_set_callback', self.handle, ctypes.cast(self._cb, ctypes.c_void_p), None)
        ops.weights_changed()

    def _draw(self):
        if self.draw:
            self.noise.normal_(0, 1)
            self.eps.normal_(0, 1)
        if self.prologue is not None:
            self.prologue()

    def resync(self):
        GraphedStep.resync(self)

    def finish(self):
        """data-parallel: apply the generator update the last replay left pending (GANStep.finish)"""
        self.gan.finish()

    def _refresh_info(self):
        info = (ctypes.c_int * 8)()
        call('sba_replay_info', self.handle, info)
        self.info = dict(zip(('nodes', 'kernels', 'copies', 'memsets', 'streams', 'waits', 'events', 'host_calls'),
                             list(info)))

    def prioritize(self, spec=None, verbose=0):
        """Stream priorities for the recording's longest dependency path (include/sbagan_hip.h: sba_replay_prioritize).
        THIS IS ONE TRAINING STEP -- the recording is executed once, every launch alone, to time it -- and is to be counted
        as a replay() by the caller; afterwards replay() issues the nodes within `slack` of the longest path on `n_high`
        streams of the higher priority and the rest on `streams - n_high` others.
        spec: 'mode[:streams[:n_high[:slack]]]' (None = SBA_REPLAY_PRIO from the environment, default below); mode 0 = off
        (returns None without running anything).  Data-parallel: every rank calls it at the same point (the step's
        collectives run)."""
        spec = os.environ.get('SBA_REPLAY_PRIO', REPLAY_PRIO_DEFAULT) if spec is None else str(spec)
        f = spec.split(':')
        if f[0] == 'c':         # calibrate only (the durations / the longest path on stderr), one pool, no priorities
            mode = 0
        else:
            mode = int(f[0])
            if mode == 0:
                return None
        streams = int(f[1]) if len(f) > 1 else 6
        n_high = int(f[2]) if len(f) > 2 else 2
        slack = float(f[3]) if len(f) > 3 else 0.08
        if self._no_update_pending():
            self._eager_step()          # (now one is)
        ops.weights_changed()
        self._draw()
        call('sba_replay_prioritize', self.handle, torch.cuda.current_stream().cuda_stream, mode, streams, n_high,
             ctypes.c_float(slack), int(verbose))
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        self._refresh_info()
        return self.out

    def _no_update_pending(self):
        """data-parallel, the deferred generator update recorded as launches: the recording applies an update whenever
        it runs, so it must not run when none is pending (after finish(), restore(), resync())"""
        return self._rec is not None and self._rec.recorded_update and self.gan._g_pending is None

    def _eager_step(self):
        """the same step, launched eagerly (bit-identical in the deterministic mode); leaves an update pending"""
        ops.weights_changed()
        self._draw()
        if self._recorded_prologue is not None:
            self._recorded_prologue()
        out = self.gan.step(*self._args, eps=self.eps)
        for k, v in out.items():
            if k in self.out:
                self.out[k].copy_(v)
        return self.out

    def replay(self):
        if self._no_update_pending():
            return self._eager_step()
        ops.weights_changed()
        self._draw()
        call('sba_replay_launch', self.handle, torch.cuda.current_stream().cuda_stream)
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        return self.out

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            try:
                call('sba_replay_destroy', h)
            except Exception:
                pass
            self.handle = None


class ReplayedStepDP(object):
    """The DATA-PARALLEL step from recordings re-issued by the native replayer, the gradient exchange between them:

        R0  (defer_g) the frozen text encoder + the three discriminators' forward passes on the REAL images, each on its
            discriminator's stream -- nothing here depends on the generator: it runs while the generator's all-reduce of
            the PREVIOUS step is still in flight (trainer.py:248-252, losses.py:139)
        --  (defer_g) wait for that all-reduce; R3 of the previous step: the generator's Adam + EMA step
        R1  generator forward | the three discriminators' loss + backward passes, each on its own stream, forked where
            its fake image is issued | (unless e_beside_exchange) the image encoder + DAMSM terms beside them
        --  all-reduce of the three discriminators' flat gradient buffers (RCCL, eager, largest first)
        RE  (e_beside_exchange) image encoder + DAMSM terms, launched while that exchange is in flight
        R2  every discriminator's Adam step and its generator-loss term on its own stream | the generator's backward pass
        --  all-reduce of the generator's gradients: started, and (defer_g) NOT waited for -- the next replay() or
            finish() applies the update
        R3  the generator's Adam + EMA step

    GraphedStep's per-phase hipGraphs keep every RCCL call exactly where the eager data-parallel step has it (buckets
    under the backward passes) but hipGraph launches do not overlap each other on ROCm 7.2, so its phases run back to
    back: 15.5 ms with one rank against 11.1 ms for the single-GPU replayer.  Here the phases overlap inside a recording
    as they do on one GPU; the discriminators exchange one bucket each (`bucket_d` off), hidden behind the image encoder
    with e_beside_exchange.  defer_g = the generator exchange overlapped with the next step's forward work, as the
    eager data-parallel step (GANStep.overlap_g) and the per-phase graphs do it: the discriminator loss then runs its real
    and fake halves as two passes.  No multi-GPU node was available in rounds 1-4: RCCL has only ever run with one rank;
    the two-rank tests (tests/dist_worker.py: gloo, one card) hold every variant bit-identical to the eager step."""

    @classmethod
    def warm_up(cls, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, recorded_prologue=None,
                defer_g=False):
        """The COLLECTIVE-BEARING part of the construction: set the step's decomposition, run one eager data-parallel step
        on the capture stream (per-stream workspaces, packed weights).  Every rank must call it; callers that want to fall
        back when a capture fails call it OUTSIDE their try block and pass the result as `warm` -- an exception in here
        would leave the peers blocked in the step's all-reduces (the captures themselves issue no collective)."""
        if not gan.distributed:
            raise RuntimeError('ReplayedStepDP is the data-parallel launch mode; use ReplayedStep on one GPU')
        gan.finish()
        gan.overlap_g = bool(defer_g)
        gan.bucket_d = False
        dev = gan.device
        eps = torch.empty((noise.shape[-2], cfg.GAN.CONDITION_DIM), dtype=torch.float32, device=dev)
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            noise.normal_(0, 1)
            eps.normal_(0, 1)
            if recorded_prologue is not None:
                recorded_prologue()
            gan.step(imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, eps=eps)
            gan.finish()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        return {'cap': cap, 'eps': eps, 'defer_g': bool(defer_g)}

    def __init__(self, gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise, prologue=None,
                 recorded_prologue=None, max_streams=4, e_beside_exchange=False, defer_g=False, warm=None):
        """e_beside_exchange: the image encoder + DAMSM terms as a recording of their own, launched AFTER the
        discriminators' all-reduces have been started -- they run while the exchange is in flight (hides up to their
        3.3 ms of it) instead of beside the discriminators' backward passes (one rank: 13.5 against 12.1 ms; pays
        once the exposed exchange exceeds ~1.4 ms).  defer_g: see the class docstring.  warm: the result of warm_up()."""
        if warm is None:
            warm = self.warm_up(gan, imgs, sent_emb, words_embs, mask, cap_lens, class_ids, noise,
                                recorded_prologue=recorded_prologue, defer_g=defer_g)
        self.defer_g = warm['defer_g']
        self.e_beside_exchange = bool(e_beside_exchange) and gan.early_damsm
        self.gan, self.noise, self.prologue = gan, noise, prologue
        self.eps, self.cap = warm['eps'], warm['cap']
        self.draw = True
        self._pending = None
        nD = len(gan.netsD)
        streams = gan._d_streams()[:nD]

        def seg0():
            if recorded_prologue is not None:
                recorded_prologue()
            main = torch.cuda.current_stream()
            gan.phase_pre(imgs, streams)
            for st in streams:
                main.wait_stream(st)

        def seg1():
            if recorded_prologue is not None and not self.defer_g:
                recorded_prologue()
            main = torch.cuda.current_stream()
            ready = [None] * nD
            if gan.early_d:
                def on_image(i):
                    if i < nD - 1:
                        ready[i] = torch.cuda.Event()
                        ready[i].record()
                gan.netG.on_image = on_image
            try:
                gan.phase_a(sent_emb, words_embs, mask, noise, self.eps)
            finally:
                gan.netG.on_image = None
            for i in range(nD):
                st = streams[i]
                if ready[i] is not None:
                    st.wait_event(ready[i])
                else:
                    st.wait_stream(main)
                with torch.cuda.stream(st):
                    if gan.phase_d_bwd_tail(i, imgs, sent_emb):
                        gan.phase_d_bwd_rest(i)
                    gan.phase_d_bwd_join()
            if gan.early_damsm and not self.e_beside_exchange:
                gan.phase_e(sent_emb, words_embs, cap_lens, class_ids)
            for st in streams:
                main.wait_stream(st)

        def seg_e():
            gan.phase_e(sent_emb, words_embs, cap_lens, class_ids)

        def seg2():
            main = torch.cuda.current_stream()
            for i in range(nD):
                st = streams[i]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    gan.phase_d_opt(i)
                    if gan.early_g_terms:
                        gan.phase_g_term(i, sent_emb)
            for st in streams:
                main.wait_stream(st)
            gan.phase_b_bwd(sent_emb, words_embs, cap_lens, class_ids)

        def seg3():
            self.out = gan.phase_b_opt()

        self.graphs, self.replayers = [], []
        segs = [(seg1, max_streams), (seg2, max_streams), (seg3, 1)]
        if self.e_beside_exchange:
            segs.insert(1, (seg_e, 2))
        if self.defer_g:
            segs.insert(0, (seg0, max_streams))
        for fn, ns in segs:
            g = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g, stream=self.cap, capture_error_mode=_CAPTURE_MODE):
                fn()
            torch.cuda.synchronize()
            self.graphs.append(g)
            self.replayers.append(_NativeGraph(g, max_streams=ns, flags=0))
        ops.weights_changed()

    def _draw(self):
        if self.draw:
            self.noise.normal_(0, 1)
            self.eps.normal_(0, 1)
        if self.prologue is not None:
            self.prologue()

    def resync(self):
        GraphedStep.resync(self)        # (drops a pending generator update: see its contract)

    def _apply_pending(self):
        """on the CURRENT stream: wait for the generator's exchange of the previous replay, then its Adam + EMA step (R3)"""
        if self._pending is not None:
            h, self._pending = self._pending, None
            self.gan._allreduce_wait(h[0])
            self.replayers[-1].replay()

    def finish(self):
        """apply a pending generator update (defer_g); a no-op otherwise"""
        if self._pending is not None:
            cur = torch.cuda.current_stream()
            self.cap.wait_stream(cur)
            with torch.cuda.stream(self.cap):
                self._apply_pending()
            cur.wait_stream(self.cap)

    def replay(self):
        gan = self.gan
        ops.weights_changed()
        self._draw()
        cur = torch.cuda.current_stream()
        self.cap.wait_stream(cur)
        with torch.cuda.stream(self.cap):
            reps = list(self.replayers[:-1])
            if self.defer_g:
                reps.pop(0).replay()        # text encoder + netD_i(real_i): beside the generator's exchange in flight
                self._apply_pending()
            reps.pop(0).replay()
            order = sorted(range(len(gan.flatD)), key=lambda i: -gan.flatD[i].n)       # same order on every rank
            handles = [gan.exchange.start(gan.flatD[i].grad) for i in order]
            if self.e_beside_exchange:
                reps.pop(0).replay()        # image encoder + DAMSM terms, while the exchange is in flight
            for h in handles:
                gan._allreduce_wait(h)
            reps.pop(0).replay()
            h = gan._allreduce_start(gan.flatG)
            if self.defer_g:
                self._pending = (h,)        # applied by the next replay() / finish()
            else:
                gan._allreduce_wait(h)
                self.replayers[-1].replay()
        cur.wait_stream(self.cap)
        return self.out
