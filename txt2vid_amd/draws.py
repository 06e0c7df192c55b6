"""Random draws of one training iteration, made on the HOST generators in the reference's order (RNG parity); see
`txt2vid_amd.functional.draws` / `set_draws` for the instance the ops read."""
import torch


# ------------------------------------------------------------------------------------------------
# random draws of one training iteration (SURVEY §7 "RNG parity"): always made on the HOST generators in
# the reference's order. `HostDraws` hands them out as they are made (eager mode); `StaticDraws` makes
# all of an iteration's draws up front and keeps them in fixed device buffers, so that a captured HIP
# graph reads this iteration's phases / z / alphas from the same addresses on every replay.
# ------------------------------------------------------------------------------------------------

import os as _os

_UPLOAD_BY_MEMCPY = _os.environ.get('T2V_DRAWS_MEMCPY') is not None


class HostDraws(object):
    def multiscale_t0(self, n):
        """n Subsample draws of trainer.multiscale_data; returns per level (t0 host int, None)."""
        out, t0, st = [], 0, 1
        for _ in range(n):
            out.append((t0, None))
            bt = int(torch.randint(2, (1,)))
            t0, st = t0 + bt * st, st * 2
        return out

    def phase(self):
        return int(torch.randint(2, (1,))), None

    def z(self, batch, latent, device):
        return torch.randn(batch, latent).to(device, non_blocking=True)

    def alpha(self, b, ndim, device):
        return torch.rand(b, *([1] * (ndim - 1))).reshape(b).to(device=device, dtype=torch.float32)

    def perm(self, n, device):
        """A non-identity permutation of range(n) from numpy's global RNG (util/misc.gen_perm), as a device int32 tensor."""
        import numpy as _np
        from .util.misc import gen_perm
        return torch.from_numpy(_np.ascontiguousarray(gen_perm(n), dtype=_np.int32)).to(device)


class StaticDraws(object):
    def __init__(self, device, batch, latent, n_levels, n_gen_phases=3, gp=True, subsample_input=True, n_perms=0):
        self.n_perms = n_perms                      # caption permutations per iteration (conditional path: D step, G step)
        self._p = 0
        self.device, self.batch, self.latent = device, batch, latent
        self.n_levels, self.n_gen, self.gp, self.sub = n_levels, n_gen_phases, gp, subsample_input
        self.bs = [batch]
        for _ in range(n_levels - 1):
            self.bs.append((self.bs[-1] + 1) // 2)
        # one pinned staging buffer and one device buffer of 32-bit words for every draw of an iteration (int32 phases |
        # float32 z | float32 alphas | int32 caption permutations): ONE H2D copy node per replayed iteration instead of four
        # (a copy node costs ~20 us of idle time in the replayed graph)
        n_int, n_z, n_a, n_p = n_levels + n_gen_phases, batch * latent, sum(self.bs), max(1, n_perms) * batch
        # The staging side is a RING of pinned buffers: the upload is asynchronous and the training loop runs the host ahead of
        # the GPU (losses are read one iteration late), so a single staging buffer would be rewritten with iteration i + 1's draws
        # before the copy of iteration i has executed. A slot is reused only after the event behind its last copy has completed.
        self.RING = 4
        self.h_ring = [torch.zeros(n_int + n_z + n_a + n_p, dtype=torch.int32).pin_memory() for _ in range(self.RING)]
        self.h_events = [None] * self.RING
        self._slot = 0
        self.h_all = self.h_ring[0]
        self.d_all = self.h_all.to(device)               # (zeros by copy: a memcpy, no fill kernel)

        def carve(buf):
            o = 0
            i_ = buf[o:o + n_int]; o += n_int
            z_ = buf[o:o + n_z].view(torch.float32).view(batch, latent); o += n_z
            a_ = buf[o:o + n_a].view(torch.float32); o += n_a
            p_ = buf[o:o + n_p].view(max(1, n_perms), batch)
            return i_, z_, a_, p_
        self._carve = carve
        self.h_int, self.h_z, self.h_a, self.h_perm = carve(self.h_all)
        self.d_int, self.d_z, self.d_a, self.d_perm = carve(self.d_all)
        self._i = self._a = 0

    def begin_step(self):
        """All host draws of one iteration, reference order: n_levels Subsample phases, z, the generator's
        phases, the GP alphas per level; then one small H2D copy on the current stream."""
        slot = self._slot
        self._slot = (slot + 1) % self.RING
        if self.h_events[slot] is not None:
            self.h_events[slot].synchronize()            # (the copy that last read this slot is done; normally long ago)
        self.h_all = self.h_ring[slot]
        self.h_int, self.h_z, self.h_a, self.h_perm = self._carve(self.h_all)
        t0, st = 0, 1
        for l in range(self.n_levels):
            self.h_int[l] = t0
            if self.sub:
                bt = int(torch.randint(2, (1,)))
                t0, st = t0 + bt * st, st * 2
        self.h_z.copy_(torch.randn(self.batch, self.latent))
        for k in range(self.n_gen):
            self.h_int[self.n_levels + k] = int(torch.randint(2, (1,)))
        if self.gp:
            off = 0
            for b in self.bs:
                self.h_a[off:off + b] = torch.rand(b, 1, 1, 1, 1).reshape(b)
                off += b
        if self.n_perms:
            from .util.misc import gen_perm
            for k in range(self.n_perms):
                self.h_perm[k].copy_(torch.from_numpy(gen_perm(self.batch).astype('int32')))
        if self.d_all.is_cuda and not _UPLOAD_BY_MEMCPY:
            # a copy KERNEL reading the pinned buffer in place (pinned host memory is mapped into the device's address space): it
            # stays on the compute queue, where the copy engine's memcpy costs a queue hand-over each way in front of the graph
            # that reads the draws (~100 us of idle time per iteration in the text-conditioned loop, ~20 us in the unconditional one)
            from . import functional as TF
            TF._copy2d(self.h_all, 0, self.h_all.numel(), self.d_all, 0, self.h_all.numel(), 1, self.h_all.numel())
        else:
            self.d_all.copy_(self.h_all, non_blocking=True)
        if self.d_all.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.h_events[slot] = ev
        self._i = self._a = self._p = 0

    def multiscale_t0(self, n):
        assert n == self.n_levels
        return [(0, self.d_int[l:l + 1]) for l in range(n)]

    def phase(self):
        k = self._i
        self._i += 1
        return 0, self.d_int[self.n_levels + k:self.n_levels + k + 1]

    def z(self, batch, latent, device):
        return self.d_z

    def alpha(self, b, ndim, device):
        off = self._a
        self._a += b
        return self.d_a[off:off + b]

    def perm(self, n, device):
        assert n == self.batch and self._p < self.n_perms
        k = self._p
        self._p += 1
        return self.d_perm[k]

    def rewind(self):
        self._i = self._a = self._p = 0
