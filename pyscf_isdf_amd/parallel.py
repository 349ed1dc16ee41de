"""Process-group plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).  Only thin wrappers; every collective the ISDF path issues is
listed here so the communication pattern can be read in one place (multi-GPU section)."""
import os
import torch
import torch.distributed as dist


class Comm:
    def __init__(self, rank=0, size=1, local_rank=0, group=None, always=False):
        self.rank, self.size, self.local_rank, self.group = rank, size, local_rank, group
        # always: issue every collective even with one rank (a process group must be initialised).  Lets a ONE-GPU box
        # execute the RCCL code paths (tests/nccl_one_rank.py); with more than one rank it changes nothing.
        self.always = bool(always)
        # what went over the fabric, per collective kind: calls, payload bytes this rank contributed (all_reduce / broadcast:
        # the tensor; all_to_all: what this rank sent to OTHER ranks; all_gather: its own block), and - when timing is on -
        # device time between event pairs recorded around the call on the stream it was issued on (no host synchronisation
        # until the figures are read: the side-stream overlap of the sharded build is not disturbed)
        self.stats = {}
        self.timing = bool(os.environ.get('ISDF_COMM_TIMING'))
        self._events = []

    def _note(self, kind, nbytes, t=None):
        st = self.stats.setdefault(kind, {'calls': 0, 'bytes': 0, 'seconds': 0.0})
        st['calls'] += 1
        st['bytes'] += int(nbytes)
        ev = None
        if self.timing and t is not None and t.is_cuda:
            ev = (kind, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[1].record(torch.cuda.current_stream(t.device))
            self._events.append(ev)
        return ev

    @staticmethod
    def _done(ev, t):
        if ev is not None:
            ev[2].record(torch.cuda.current_stream(t.device))

    def reset_stats(self):
        self.stats = {}
        self._events = []

    def collect_stats(self):
        """{kind: {calls, bytes, seconds}} since the last reset (synchronises the device once to read the event pairs)."""
        if self._events:
            torch.cuda.synchronize()
            for kind, e0, e1 in self._events:
                self.stats[kind]['seconds'] += e0.elapsed_time(e1) * 1e-3
            self._events = []
        return {k: dict(v) for k, v in self.stats.items()}

    @property
    def _live(self):
        return self.size > 1 or self.always

    @classmethod
    def from_env(cls):
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size(), int(os.environ.get('LOCAL_RANK', '0')),
                       always=bool(os.environ.get('ISDF_COMM_ALWAYS')))
        return cls()

    def split_range(self, n, r=None):
        """Contiguous share [lo, hi) of range(n) for rank r (default: this rank)."""
        r = self.rank if r is None else r
        base, rem = divmod(n, self.size)
        lo = r * base + min(r, rem)
        return lo, lo + base + (1 if r < rem else 0)

    def _staged(self, t):
        """gloo moves host memory: device tensors are staged through the host (rehearsals of the N > 1 path with
        several ranks on ONE GPU, tools/rehearse_ranks_one_gpu.sh; the production backend is nccl = RCCL)."""
        return self._live and t.is_cuda and dist.get_backend(self.group) == 'gloo'

    def all_reduce_sum(self, t):
        if self._live:
            ev = self._note('all_reduce', t.numel() * t.element_size(), t)
            if self._staged(t):
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self._done(ev, t)
        return t

    def broadcast(self, t, src=0):
        """Every rank ends with rank src's tensor."""
        if self._live:
            ev = self._note('broadcast', t.numel() * t.element_size(), t)
            if self._staged(t):
                h = t.cpu()
                dist.broadcast(h, src, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src, group=self.group)
            self._done(ev, t)
        return t

    def agree_max(self, x):
        """max over the ranks of a host float: decisions taken on replicated data must not diverge in the last bit."""
        if not self._live:
            return float(x)
        h = torch.tensor([float(x)], dtype=torch.float64)
        if dist.get_backend(self.group) != 'gloo':
            h = h.to(torch.device('cuda', self.local_rank))
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
        return float(h.item())

    def run_on_root(self, fn):
        """fn() on rank 0 only; a failure there is raised on EVERY rank (nobody is left waiting in a collective).
        Returns fn()'s value on rank 0, None elsewhere."""
        out, err = None, None
        if self.rank == 0:
            try:
                out = fn()
            except Exception as e:                      # noqa: BLE001 - re-raised below on every rank
                err = e
        if self.agree_max(0.0 if err is None else 1.0) > 0.0:
            if err is not None:
                raise err
            raise RuntimeError('rank 0 failed in a root-only stage (see its traceback)')
        return out

    def all_gather_rows(self, t_local, counts):
        """Concatenate row blocks of unequal height (counts[r] rows from rank r)."""
        if not self._live:
            return t_local
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t_local.shape[1:]), dtype=t_local.dtype, device=t_local.device)
        pad[:t_local.shape[0]] = t_local
        self._note('all_gather', t_local.numel() * t_local.element_size())
        if self._staged(pad):
            hp = pad.cpu()
            hb = [torch.empty_like(hp) for _ in range(self.size)]
            dist.all_gather(hb, hp, group=self.group)
            bufs = [b.to(pad.device) for b in hb]
        else:
            bufs = [torch.empty_like(pad) for _ in range(self.size)]
            dist.all_gather(bufs, pad, group=self.group)
        return torch.cat([bufs[r][:counts[r]] for r in range(self.size)], dim=0)

    def all_gather_object(self, obj):
        if not self._live:
            return [obj]
        out = [None] * self.size
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def all_to_all(self, recv, send):
        """recv[q] <- what rank q put in its send[self.rank]; lists of contiguous tensors."""
        if not self._live:
            recv[0].copy_(send[0])
            return
        send = [s.contiguous() for s in send]
        ev = self._note('all_to_all', sum(x.numel() * x.element_size() for q, x in enumerate(send) if q != self.rank), send[0])
        if dist.get_backend(self.group) == 'gloo':
            # gloo has no all_to_all: pairwise exchange (CPU tests only)
            reqs = []
            staged = {}
            for q in range(self.size):
                if q == self.rank:
                    recv[q].copy_(send[q])
                else:
                    sq, rq = send[q], recv[q]
                    if rq.is_cuda:
                        sq = sq.cpu()
                        staged[q] = torch.empty(rq.shape, dtype=rq.dtype)
                        rq = staged[q]
                    reqs.append(dist.isend(sq, q, group=self.group))
                    reqs.append(dist.irecv(rq, q, group=self.group))
            for r in reqs:
                r.wait()
            for q, h in staged.items():
                recv[q].copy_(h)
        else:
            dist.all_to_all(recv, send, group=self.group)
        self._done(ev, send[0])

    def barrier(self):
        if self._live:
            dist.barrier(group=self.group)
