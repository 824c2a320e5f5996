"""Device-side augmentation RNG (perf mode `model.device_rng`).

The reference draws EdgeDrop masks and EmbedPerturb noise with `t.rand` on the CPU generator and copies them to the
device (models/aug_utils.py:28,130); parity mode does exactly that.  In perf mode nothing is drawn or stored: the
kernels compute the uniform they need with Philox4x32-10 (sslrec_amd/csrc/philox.h) from

    seed, step  -- two uint64 in DEVICE memory (`PhiloxState.state`); `advance()` bumps `step` with a kernel, once per
                   training step, so a hipGraph-captured step draws fresh numbers on every replay;
    stream      -- a host constant, distinct for every augmentation call inside one step (`next_stream()`);
    element     -- the COO entry id (EdgeDrop) or the float index / 4 of the output row (EmbedPerturb).
"""
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
