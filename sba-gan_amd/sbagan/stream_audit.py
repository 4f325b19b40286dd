"""SBA_STREAM_AUDIT=1: a happens-before checker for ONE hazard class of the multi-stream step -- the caching allocator
handing a block back out while a kernel on ANOTHER stream may still be reading or writing it.

The pattern (the MAPPING_NET fork of round 3): a tensor is allocated on stream A (its block belongs to
A's pool), a kernel on stream B uses it, Python drops the last reference while B's kernel is still queued, the allocator
returns the block to A's pool at once, the next allocation on A gets the same bytes and A's next kernel overwrites them.
It is safe only if (a) `record_stream(B)` was called on the tensor (the allocator then defers the reuse until B's work
has finished), or (b) A had already waited for B's use when the block was handed out again (a join before the free, a
keep-alive list until the join, a step-level join).  Nothing in the losses notices when it is not: round 3 found one
instance by a wrong weight gradient in 4 of 30 runs.

What is recorded (all on the host, in issue order -- no clocks):
  * every launch through the C ABI (sbagan._lib.call: entry point, stream, device-pointer arguments; the items of a
    grouped conv launch are unpacked) and every ATen op of the calling thread (TorchDispatchMode: tensor arguments and
    outputs on the current stream; pure view ops are skipped);
  * every torch.cuda.Event.record / Event.wait (Stream.wait_stream / wait_event / record_event are built from these),
    and torch.cuda.synchronize / Stream.synchronize;
  * every Tensor.record_stream call (any thread), and the allocator's own trace
    (torch.cuda.memory._record_memory_history): alloc / free_requested / free_completed with address, size and pool
    stream.  A block whose free completes LATER than it was requested had stream uses recorded: the allocator protects it.
The two logs are interleaved exactly by MARKER allocations: after every recorded host event a 1-byte tensor is allocated
and dropped on a private stream; its trace entries separate the allocator events that happened before the host event
from those after it.

The replay keeps a vector clock per stream (what it has waited for) and per live block the last use on every stream.
When a block is freed for immediate reuse while a use on a foreign stream S has not been joined into the pool's stream,
the range is TAINTED; the next allocation that overlaps it is a HAZARD unless the allocating stream has waited for that
use by then.  Limits: ATen ops issued from the autograd engine's worker thread are not seen (the library's own launches
are); graph capture is out of scope (captured steps are covered by the bit-equality tests)."""
import bisect
import ctypes

import torch

from . import _lib

_MARK_BYTES = 1


class _Block(object):
    __slots__ = ('addr', 'size', 'stream', 'uses', 'recorded')

    def __init__(self, addr, size, stream):
        self.addr, self.size, self.stream, self.uses, self.recorded = addr, size, stream, {}, set()


class StreamAudit(object):
    def __init__(self, device=None, torch_ops=True):
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.torch_ops = torch_ops
        self.log = []               # host events: ('launch', name, stream, ptrs) | ('record', ev_id, stream) |
        #                             ('wait', ev_id, stream) | ('sync',)
        self._patched = []
        self._mode = None
        self._marker = None
        self._active = False

    # ------------------------------------------------------------------ recording
    def _mark(self):
        """a marker allocation on the private stream: ties this host event to the allocator's trace"""
        torch._C._cuda_setStream(stream_id=self._marker.stream_id, device_index=self._marker.device_index,
                                 device_type=self._marker.device_type)
        try:
            torch.empty(_MARK_BYTES, dtype=torch.uint8, device=self.device)
        finally:
            s = self._cur
            torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)

    def _event(self, ev):
        if not self._active or self._busy:
            return
        self._busy = True
        try:
            self._cur = torch.cuda.current_stream(self.device)
            self.log.append(ev)
            self._mark()
        finally:
            self._busy = False

    def _on_call(self, name, args):
        sig = _lib.SIGNATURES.get(name)
        if not sig or sig[-1] is not ctypes.c_void_p or name.startswith(('sba_replay', 'sba_set_', 'sba_det_')):
            return
        stream = args[-1] or 0
        stream = stream.value if hasattr(stream, 'value') else int(stream or 0)
        ptrs = []
        for a, t in zip(args[:-1], sig[:-1]):
            if t is ctypes.c_void_p and a:
                ptrs.append(a.value if hasattr(a, 'value') else int(a))
            elif name.startswith('sba_conv_igemm_group') and isinstance(a, ctypes.Array):
                for it in a:
                    ptrs += [p for p in (it.x, it.w, it.y, it.addend, it.bias, it.relu_mask) if p]
        self._event(('launch', name, stream, ptrs))

    def start(self):
        assert not self._active
        torch.cuda.synchronize()
        self._marker = torch.cuda.Stream(device=self.device)
        self._busy = False
        # live blocks at the start (everything allocated before the audit): address ranges and their pool's stream
        self._initial = []
        for seg in torch.cuda.memory_snapshot():
            if seg.get('device', 0) != (self.device.index or 0):
                continue
            a = seg['address']
            for b in seg['blocks']:
                if b['state'].startswith('active'):
                    self._initial.append((a, b['size'], seg.get('stream', 0)))
                a += b['size']
        torch.cuda.memory._record_memory_history(enabled='all', context=None, stacks='python', max_entries=8000000)
        audit = self
        E, S = torch.cuda.Event, torch.cuda.Stream
        o_rec, o_wait, o_sync, o_ssync = E.record, E.wait, torch.cuda.synchronize, S.synchronize

        def record(ev, stream=None):
            st = torch.cuda.current_stream() if stream is None else stream
            r = o_rec(ev, st)
            audit._event(('record', id(ev), st.cuda_stream))
            return r

        def wait(ev, stream=None):
            st = torch.cuda.current_stream() if stream is None else stream
            r = o_wait(ev, st)
            audit._event(('wait', id(ev), st.cuda_stream))
            return r

        def sync(*a, **k):
            r = o_sync(*a, **k)
            audit._event(('sync',))
            return r

        def ssync(st):
            r = o_ssync(st)
            audit._event(('ssync', st.cuda_stream))
            return r
        o_rs = torch.Tensor.record_stream

        def record_stream(t, stream):
            r = o_rs(t, stream)
            if t.is_cuda and t.numel() > 0:
                audit._event(('record_stream', t.data_ptr(), stream.cuda_stream))
            return r
        E.record, E.wait, torch.cuda.synchronize, S.synchronize = record, wait, sync, ssync
        torch.Tensor.record_stream = record_stream
        self._patched = [(E, 'record', o_rec), (E, 'wait', o_wait), (torch.cuda, 'synchronize', o_sync),
                         (S, 'synchronize', o_ssync), (torch.Tensor, 'record_stream', o_rs)]
        _lib.AUDIT_HOOK = self._on_call
        if self.torch_ops:
            from torch.utils._python_dispatch import TorchDispatchMode
            from torch.utils._pytree import tree_leaves

            class Mode(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    out = func(*args, **(kwargs or {}))
                    if audit._active and not audit._busy:
                        try:
                            view = any(r.alias_info is not None and not r.alias_info.is_write for r in func._schema.returns)
                        except Exception:
                            view = False
                        if not view and 'record_stream' not in str(func):
                            ptrs = [t.data_ptr() for t in tree_leaves((args, kwargs, out))
                                    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0]
                            if ptrs:
                                audit._event(('launch', str(func), torch.cuda.current_stream().cuda_stream, ptrs))
                    return out
            self._mode = Mode()
            self._mode.__enter__()
        self._active = True
        return self

    def stop(self):
        """stop recording and replay the two logs; returns the list of hazards (dicts)"""
        assert self._active
        self._active = False
        if self._mode is not None:
            self._mode.__exit__(None, None, None)
            self._mode = None
        _lib.AUDIT_HOOK = None
        for obj, name, orig in self._patched:
            setattr(obj, name, orig)
        self._patched = []
        torch.cuda.synchronize()
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
        traces = snap['device_traces'][self.device.index or 0]
        return self._replay(traces)

    # ------------------------------------------------------------------ replay
    def _replay(self, traces):
        marker = self._marker.cuda_stream
        live, starts = {}, []           # addr -> _Block; sorted start addresses

        def add(b):
            live[b.addr] = b
            bisect.insort(starts, b.addr)

        def drop(addr):
            b = live.pop(addr, None)
            if b is not None:
                i = bisect.bisect_left(starts, addr)
                if i < len(starts) and starts[i] == addr:
                    starts.pop(i)
            return b

        def find(p):
            i = bisect.bisect_right(starts, p) - 1
            if i >= 0:
                b = live[starts[i]]
                if p < b.addr + b.size:
                    return b
            return None
        for a, sz, st in self._initial:
            add(_Block(a, sz, st))
        count, vc, events = {}, {}, {}      # launches issued per stream; vector clocks; event id -> clock snapshot
        tainted = []                        # (addr, end, pool stream, [(stream, t, kernel)], info)
        pending_free = {}
        hazards = []
        stats = {'launches': 0, 'foreign_uses': 0, 'frees_with_unjoined_foreign_use': 0, 'deferred_frees': 0,
                 'allocator_events': 0, 'host_events': len(self.log)}

        def clock(s):
            return vc.setdefault(s, {})

        def merge(dst, src):
            for k, v in src.items():
                if dst.get(k, 0) < v:
                    dst[k] = v

        def host(ev):
            kind = ev[0]
            if kind == 'launch':
                _, name, s, ptrs = ev
                count[s] = count.get(s, 0) + 1
                clock(s)[s] = count[s]
                stats['launches'] += 1
                for p in ptrs:
                    b = find(p)
                    if b is not None:
                        b.uses[s] = (count[s], name)
                        if s != b.stream:
                            stats['foreign_uses'] += 1
            elif kind == 'record_stream':
                b = find(ev[1])
                if b is not None:
                    b.recorded.add(ev[2])
            elif kind == 'record':
                c = dict(clock(ev[2]))
                c[ev[2]] = count.get(ev[2], 0)
                events[ev[1]] = c
            elif kind == 'wait':
                c = events.get(ev[1])
                if c is not None:
                    merge(clock(ev[2]), c)
            elif kind == 'ssync':           # the host waited for ONE stream: everything issued to it so far is done
                for s in list(count) + [ev[1]]:
                    clock(s)[ev[1]] = count.get(ev[1], 0)
            elif kind == 'sync':
                for s in list(count):
                    for s2, n in count.items():
                        clock(s)[s2] = n
                del tainted[:]              # everything issued so far has completed

        def alloc(e):
            addr, size, s = e['addr'], e['size'], e.get('stream', 0)
            end = addr + size
            keep = []
            for t in tainted:
                if t[0] < end and addr < t[1]:
                    for (fs, ft, fk) in t[3]:
                        if clock(s).get(fs, 0) < ft:
                            hazards.append({'addr': hex(t[0]), 'bytes': t[1] - t[0], 'pool_stream': t[2],
                                            'reallocated_on': s, 'unjoined_stream': fs, 'last_use': fk,
                                            'freed_after_host_event': t[4]})
                else:
                    keep.append(t)
            tainted[:] = keep
            add(_Block(addr, size, s))

        hi = 0                              # next host event
        prev = None
        n = len(traces)
        k = 0
        while k < n:
            e = traces[k]
            k += 1
            act = e['action']
            if e.get('stream', None) == marker and e.get('size', 0) <= 512 and act in ('alloc', 'free_requested',
                                                                                         'free_completed'):
                if act == 'alloc' and hi < len(self.log):
                    host(self.log[hi])
                    hi += 1
                prev = None
                continue
            stats['allocator_events'] += 1
            if act == 'alloc':
                alloc(e)
            elif act == 'free_requested':
                b = drop(e['addr'])
                if b is not None:
                    pending_free[e['addr']] = b
            elif act == 'free_completed':
                b = pending_free.pop(e['addr'], None)
                if b is not None:
                    immediate = prev is not None and prev['action'] == 'free_requested' and prev['addr'] == e['addr']
                    if not immediate:
                        stats['deferred_frees'] += 1
                    else:
                        own = clock(b.stream)
                        foreign = [(s, t, kn) for s, (t, kn) in b.uses.items()
                                   if s != b.stream and s not in b.recorded and own.get(s, 0) < t]
                        if foreign:
                            stats['frees_with_unjoined_foreign_use'] += 1
                            tainted.append((b.addr, b.addr + b.size, b.stream, foreign, hi))
            prev = e
        while hi < len(self.log):
            host(self.log[hi])
            hi += 1
        self.stats = stats
        return hazards


def audited(fn, device=None):
    """run fn() under the audit; returns (fn's result, hazards, stats)"""
    a = StreamAudit(device).start()
    try:
        out = fn()
    finally:
        hz = a.stop()
    return out, hz, a.stats
