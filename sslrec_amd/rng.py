"""Device-side augmentation RNG (perf mode `model.device_rng`).

The reference draws EdgeDrop masks and EmbedPerturb noise with `t.rand` on the CPU generator and copies them to the
device (models/aug_utils.py:28,130); parity mode does exactly that.  In perf mode nothing is drawn or stored: the
kernels compute the uniform they need with Philox4x32-10 (sslrec_amd/csrc/philox.h) from

    seed, step  -- two uint64 in DEVICE memory (`PhiloxState.state`); `advance()` bumps `step` with a kernel, once per
                   training step, so a hipGraph-captured step draws fresh numbers on every replay;
    stream      -- a host constant, distinct for every augmentation call inside one step (`next_stream()`);
    element     -- the COO entry id (EdgeDrop) or the float index / 4 of the output row (EmbedPerturb).
"""
import numpy as np
import torch

from . import _lib


class PhiloxState:
    def __init__(self, device, seed=None):
        if seed is None:      # from the CPU generator: reproducible under torch.manual_seed, like everything else
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        self._stream = 0

    def advance(self):
        """step += 1 on the device (a kernel on the current stream: capturable) and restart the per-step stream ids"""
        rc = _lib.load().sslrec_philox_advance(self.state.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'sslrec_philox_advance')
        self._stream = 0

    def next_stream(self):
        self._stream += 1
        return self._stream


class PhiloxNoise:
    """stands for a uniform [N, d] noise tensor that is never materialized: the SpMM epilogue computes its rows"""

    def __init__(self, state, shape):
        self.state, self.stream, self.shape = state, state.next_stream(), tuple(shape)

    def materialize(self):
        """the same numbers as a tensor (tests / the dense EmbedPerturb.forward): u[r, 4g..4g+3] = uniform4(r*d/4 + g)"""
        n, d = self.shape
        lib = _lib.load()
        out = torch.empty((n, d), dtype=torch.float32, device=self.state.state.device)
        rc = lib.sslrec_philox_fill_f32(self.state.state.data_ptr(), self.stream, out.data_ptr(), n * d,
                                        torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'sslrec_philox_fill_f32')
        return out


def philox_uniforms(state, stream, n):
    """the first n uniforms of call `stream` at the state's current step, written out: uniform k is what the kernels compute
    for element k of that call (EdgeDrop: COO entry k; philox.h: philox_uniform1) -- for tests that hand the SAME draws to
    the oracle"""
    n4 = (int(n) + 3) // 4 * 4
    out = torch.empty(n4, dtype=torch.float32, device=state.state.device)
    rc = _lib.load().sslrec_philox_fill_f32(state.state.data_ptr(), int(stream), out.data_ptr(), n4,
                                            torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'sslrec_philox_fill_f32')
    return out[:int(n)]


# ---------------------------------------------------------------------------------------------------------------------
# Parity mode without the host stall: the reference's CPU generator, replayed on the device
# ---------------------------------------------------------------------------------------------------------------------
class HostGeneratorReplay:
    """The stream of `t.rand` calls the reference makes on the global CPU generator (models/aug_utils.py:28,130), produced
    on the device bit for bit (sslrec_amd/csrc/mt19937.hip): the generator's MT19937 state is taken from
    `torch.get_rng_state()`, advanced by the kernels, and put back with `torch.set_rng_state()` by `flush()` before any
    HOST code draws again (the Trainer flushes at every epoch boundary: DataLoader shuffling, model construction and
    evaluation then see exactly the state the reference's run would have).  While the device is ahead the host generator
    is stale; a host draw in that window would silently break the sequence, so the next device draw checks the host state
    against the snapshot taken at upload time and raises if somebody used it."""

    _OFF_LEFT, _OFF_NEXT, _OFF_STATE, _N = 8, 16, 24, 624      # THGeneratorState: seed u64, left i32, seeded i32, next u64, state u64[624]
    _CUR_CAP = 256               # draws of one step at most (SimGCL: 2 L; SGL: 2): a longer list means nobody calls begin_step()
    # a large draw: every stretch of 128 blocks (79,872 numbers) has its own workgroup, whose start state is two polynomial jumps
    # from the current one (csrc/mt19937.hip, mt_jump.py): 8 x 16 stretches = 10.2 M numbers per pass
    STRETCH_BLOCKS, FAN1, FAN2 = 128, 8, 16

    def __init__(self, device):
        self.device = torch.device(device)
        self.mt = torch.zeros(self._N + 1, dtype=torch.int32, device=self.device)
        self._snapshot = None        # host generator state the device state was taken from / last written back
        self.ahead = False           # the device has produced numbers the host generator does not know about
        self._jump = None            # the jump polynomials (55 KB), computed on the first large draw (~0.7 s of host arithmetic)
        self._jump_ws = {}
        # Draw-ahead: a training step asks for the same sequence of draws every time (EdgeDrop: nnz numbers; SimGCL: 2 L tables
        # of N x d), they do not depend on anything the step computes, and the generator is sequential (55 M numbers: 3.5 ms).
        # Once a step's requests are known (`begin_step` marks the boundaries), the NEXT step's numbers are generated on a side
        # stream while this step computes, into the other half of a double buffer; a request that does not match the plan puts
        # the generator back to the state after the last matching draw and is served the ordinary way.
        self.draw_ahead = True
        self._plan = None            # requests of the last completed step: [(kind, shape, keep_rate)]
        self._cur = []               # requests of the running step so far
        self._ready = None           # the draws generated ahead: reqs, bufs, event, saved state, state after every draw, served
        self._pool = [{}, {}]
        self._pool_idx = 0
        self._side = None

    def _generate(self, out, n, keep_rate=None):
        """the next n numbers of the stream into `out`: one workgroup for short draws, one per STRETCH_BLOCKS blocks for long ones"""
        lib = _lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        stretch = self.STRETCH_BLOCKS
        if n < 2 * stretch * self._N:
            if keep_rate is None:
                rc = lib.sslrec_mt19937_uniform_f32(self.mt.data_ptr(), out.data_ptr(), n, st)
            else:
                rc = lib.sslrec_mt19937_keep_mask(self.mt.data_ptr(), float(keep_rate), out.data_ptr(), n, st)
            _lib.check(rc, 'sslrec_mt19937')
            return
        if self._jump is None:
            from . import mt_jump
            table = mt_jump.two_level_table(stretch, self.FAN1, self.FAN2)
            self._jump = torch.from_numpy(table.view(np.int32)).to(self.device)
        st_key = int(st)
        if st_key not in self._jump_ws:           # per stream: the draw-ahead runs on its own
            self._jump_ws[st_key] = torch.empty(lib.sslrec_mt19937_par_ws_bytes(self.FAN1, self.FAN2) // 4 + 1, dtype=torch.int32, device=self.device)
        ws = self._jump_ws[st_key]
        if keep_rate is None:
            rc = lib.sslrec_mt19937_uniform_par_f32(self.mt.data_ptr(), self._jump.data_ptr(), self.FAN1, self.FAN2, stretch, ws.data_ptr(),
                                                    out.data_ptr(), n, st)
        else:
            rc = lib.sslrec_mt19937_keep_mask_par(self.mt.data_ptr(), self._jump.data_ptr(), self.FAN1, self.FAN2, stretch, ws.data_ptr(),
                                                  float(keep_rate), out.data_ptr(), n, st)
        _lib.check(rc, 'sslrec_mt19937_par')

    # -- the two directions -------------------------------------------------------------------------------------------
    @classmethod
    def parse(cls, state_bytes):
        """(words uint32[624], pos) of a torch CPU generator state; pos = index of the next output, 624 = block used up"""
        raw = np.ascontiguousarray(state_bytes, dtype=np.uint8)
        left = int(raw[cls._OFF_LEFT:cls._OFF_LEFT + 4].view(np.int32)[0])
        nxt = int(raw[cls._OFF_NEXT:cls._OFF_NEXT + 8].view(np.uint64)[0])
        words = raw[cls._OFF_STATE:cls._OFF_STATE + 8 * cls._N].view(np.uint64).astype(np.uint32)
        return words, (cls._N if left == 1 else nxt)

    @classmethod
    def compose(cls, template_bytes, words, pos):
        """the generator state `template_bytes` with its MT19937 part replaced (left = 625 - next, see at::mt19937)"""
        raw = np.array(template_bytes, dtype=np.uint8, copy=True)
        raw[cls._OFF_LEFT:cls._OFF_LEFT + 4] = np.array([cls._N + 1 - int(pos)], dtype=np.int32).view(np.uint8)
        raw[cls._OFF_NEXT:cls._OFF_NEXT + 8] = np.array([int(pos)], dtype=np.uint64).view(np.uint8)
        raw[cls._OFF_STATE:cls._OFF_STATE + 8 * cls._N] = np.asarray(words, dtype=np.uint32).astype(np.uint64).view(np.uint8)
        return raw

    def _upload(self, host_state):
        words, pos = self.parse(host_state.numpy())
        packed = np.concatenate([words.view(np.int32), np.array([pos], dtype=np.int32)])
        self.mt.copy_(torch.from_numpy(packed))
        self._snapshot = host_state.clone()
        self.ahead = False

    def attach(self):
        host_state = torch.get_rng_state()
        if self._ready is not None and self._snapshot is not None and not torch.equal(host_state, self._snapshot):
            self._drop_ahead()
        if self._snapshot is None:
            self._upload(host_state)
        elif not torch.equal(host_state, self._snapshot):
            if self.ahead:
                raise RuntimeError('the CPU generator was used while its device replay was ahead of it (the draw returned stale '
                                   'numbers): call sslrec_amd.rng.flush_host_replay() before host code draws random numbers, '
                                   'or disable train.host_rng_replay')
            self._upload(host_state)      # the host moved on while nothing was pending: follow it

    # -- draw-ahead ---------------------------------------------------------------------------------------------------
    def begin_step(self):
        """a training step starts (GraphCF._begin_step): the requests since the previous call were one step's plan"""
        if self._cur:
            self._plan = list(self._cur)
        self._cur = []

    def _buffer(self, kind, shape):
        pool = self._pool[self._pool_idx]
        key = (kind, shape, sum(1 for r in self._building if r == (kind, shape)))      # the i-th buffer of that kind and shape
        self._building.append((kind, shape))
        if key not in pool:
            pool[key] = torch.empty(shape, dtype=torch.float32 if kind == 'rand' else torch.uint8, device=self.device)
        return pool[key]

    def _draw_ahead(self):
        """generate the plan's draws for the NEXT step on the side stream (called when this step has drawn its last planned number)"""
        if not self.draw_ahead or not self._plan or self._ready is not None or torch.cuda.is_current_stream_capturing():
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        ev_main = torch.cuda.Event()
        ev_main.record(main)                      # behind this step's own generation AND behind the previous users of the buffers
        self._side.wait_event(ev_main)
        self._building = []
        # the pool's buffers are allocated HERE, under the consumers' stream (they are written on the side stream, ordered by the
        # two events, and read on the main one: the caching allocator must tie them to the stream that reads them)
        bufs = [self._buffer(kind, shape) for kind, shape, _ in self._plan]
        states = []
        with torch.cuda.stream(self._side):
            saved = self.mt.clone()
            saved.record_stream(main)             # (the small state copies are made on the side stream and consumed on the main one)
            for (kind, shape, keep_rate), buf in zip(self._plan, bufs):
                if buf.numel():
                    self._generate(buf, buf.numel(), None if kind == 'rand' else keep_rate)
                states.append(self.mt.clone())
                states[-1].record_stream(main)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._ready = {'reqs': list(self._plan), 'bufs': bufs, 'event': ev, 'saved': saved, 'states': states, 'served': 0}
        self._pool_idx ^= 1

    def _drop_ahead(self):
        """forget what was generated ahead and not consumed: the generator goes back to the state after the last consumed draw"""
        r, self._ready = self._ready, None
        if r is None:
            return
        main = torch.cuda.current_stream(self.device)
        main.wait_event(r['event'])
        self.mt.copy_(r['saved'] if r['served'] == 0 else r['states'][r['served'] - 1])

    def _request(self, kind, shape, keep_rate):
        req = (kind, tuple(int(x) for x in shape), None if keep_rate is None else float(keep_rate))
        if len(self._cur) >= self._CUR_CAP:      # nobody marks step boundaries (a user outside GraphCF._begin_step): no plan to draw
            self._drop_ahead()                     # ahead for, and no list growing by one tuple per draw
            self._cur, self._plan = [], None
        idx = len(self._cur)
        self._cur.append(req)
        r = self._ready
        if r is not None:
            if idx == r['served'] and idx < len(r['reqs']) and r['reqs'][idx] == req:
                self._check_host()
                if idx == 0:
                    torch.cuda.current_stream(self.device).wait_event(r['event'])
                r['served'] = idx + 1
                out = r['bufs'][idx]
                if r['served'] == len(r['reqs']):      # this step is supplied: start on the next one
                    self._ready = None
                    self._draw_ahead()
                self.ahead = True
                return out
            self._drop_ahead()                          # not the planned sequence (a different model, an extra draw, ...)
        self.attach()
        out = torch.empty(req[1], dtype=torch.float32 if kind == 'rand' else torch.uint8, device=self.device)
        if out.numel():
            self._generate(out, out.numel(), req[2])
        self.ahead = True
        if self._plan is not None and self._cur == self._plan:      # the step's last planned draw, served the ordinary way
            self._draw_ahead()
        return out

    def _check_host(self):
        if self._snapshot is not None and not torch.equal(torch.get_rng_state(), self._snapshot):
            raise RuntimeError('the CPU generator was used while its device replay was ahead of it (the draw returned stale '
                               'numbers): call sslrec_amd.rng.flush_host_replay() before host code draws random numbers, '
                               'or disable train.host_rng_replay')

    def flush(self):
        """write the device's generator state back into the CPU generator (one small device-to-host copy)"""
        self._drop_ahead()           # numbers generated ahead of a step that never came do not count
        if not self.ahead:
            return
        mt = self.mt.cpu().numpy()
        new = torch.from_numpy(self.compose(self._snapshot.numpy(), mt[:self._N].view(np.uint32), int(mt[self._N])))
        torch.set_rng_state(new)
        self._snapshot = torch.get_rng_state()
        self.ahead = False

    # -- draws --------------------------------------------------------------------------------------------------------
    def rand(self, shape):
        """`t.rand(shape)` of the reference, as a device tensor.  LIFETIME (the same whichever way the draw was produced): the
        tensor is the caller's for the step that drew it and the next one; with draw-ahead it is a buffer of a double-buffered
        pool that the step after the next overwrites, so a tensor that must live longer has to be cloned."""
        return self._request('rand', tuple(shape), None)

    def keep_mask(self, n, keep_rate):
        """`(t.rand(n) + keep_rate).floor().type(t.bool)` of EdgeDrop (aug_utils.py:28-29), as a device bool tensor"""
        return self._request('mask', (int(n),), keep_rate).view(torch.bool)


_replays = {}


def enable_host_replay(device):
    device = torch.device(device)
    if device.type != 'cuda':
        raise ValueError('the generator replay runs on a GPU')
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _replays:
        _replays[key] = HostGeneratorReplay(torch.device('cuda', key))
    return _replays[key]


def active_host_replay(device):
    """the replay object of `device`, or None (parity draws then come from the CPU generator itself)"""
    device = torch.device(device)
    if device.type != 'cuda' or not _replays:
        return None
    return _replays.get(device.index if device.index is not None else torch.cuda.current_device())


def any_host_replay():
    return bool(_replays)


def flush_host_replay():
    for rep in _replays.values():
        rep.flush()


def disable_host_replay(device=None):
    """hand the generator(s) back to the host and drop the replay of `device` (None: of every device)"""
    flush_host_replay()
    if device is None:
        _replays.clear()
        return
    device = torch.device(device)
    _replays.pop(device.index if device.index is not None else torch.cuda.current_device(), None)
